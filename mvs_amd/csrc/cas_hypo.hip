// Per-pixel depth hypotheses of a cascade stage (CasMVSNet/models/cas_mvsnet.py:129-152,
// module.py:485-502), one kernel instead of the reference's four full-resolution passes:
//   cur   = bilinear(prev_depth -> [H,W], align_corners=False)             (cas_mvsnet.py:134-136)
//   lo/hi = cur -/+ half;  step = (hi - lo) / (D - 1);  v_d = lo + d * step  (module.py:489-500)
//   out   = trilinear(v -> [D,Hs,Ws], align_corners=False)                 (cas_mvsnet.py:150-151)
// The reference materialises v at full resolution ([B,D,H,W]: 242 MB at stage 2) and resamples
// it; here a thread owns one output pixel, evaluates the up to four full-resolution pixels its
// trilinear footprint touches (the depth axis keeps its size: weights 1 and 0) and writes its D
// values.  Same operation order as ATen's nested linear interpolation (x, then y), no FMA
// contraction; the quotient by D - 1 is a true division as on the CPU.
#include "mvs_common.h"

namespace mvs {

struct HypoArgs {
    const float *prev;   // [B,hp,wp]
    float *out;          // [B,D,Hs,Ws]
    int B, hp, wp, H, W, Hs, Ws, D;
    float half;          // ndepth / 2 * interval
};

// ATen area_pixel_compute_source_index (align_corners = false, linear): clamped at 0
__device__ __forceinline__ void linear_src(int dst, float scale, int in_size, int &i0, int &i1, float &l0, float &l1) {
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    if (src < 0.0f) src = 0.0f;
    i0 = (int)src;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
    l1 = src - (float)i0;
    l0 = 1.0f - l1;
}

__global__ __launch_bounds__(256) void cas_hypotheses_kernel(HypoArgs a) {
    const int64_t n = (int64_t)a.B * a.Hs * a.Ws;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const int xs = (int)(idx % a.Ws), ys = (int)((idx / a.Ws) % a.Hs), b = (int)(idx / ((int64_t)a.Ws * a.Hs));
    // footprint of the output pixel on the full-resolution grid
    int fy[2], fx[2];
    float wy[2], wx[2];
    linear_src(ys, (float)a.H / (float)a.Hs, a.H, fy[0], fy[1], wy[0], wy[1]);
    linear_src(xs, (float)a.W / (float)a.Ws, a.W, fx[0], fx[1], wx[0], wx[1]);
    const float *pv = a.prev + (int64_t)b * a.hp * a.wp;
    const float sh = (float)a.hp / (float)a.H, sw = (float)a.wp / (float)a.W;
    float lo[2][2], step[2][2];
    const float dm1 = (float)(a.D - 1);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        int py0, py1; float ly0, ly1;
        linear_src(fy[j], sh, a.hp, py0, py1, ly0, ly1);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int px0, px1; float lx0, lx1;
            linear_src(fx[i], sw, a.wp, px0, px1, lx0, lx1);
            const float top = lx0 * pv[py0 * a.wp + px0] + lx1 * pv[py0 * a.wp + px1];
            const float bot = lx0 * pv[py1 * a.wp + px0] + lx1 * pv[py1 * a.wp + px1];
            const float cur = ly0 * top + ly1 * bot;
            const float l = cur - a.half, h = cur + a.half;
            lo[j][i] = l;
            step[j][i] = (h - l) / dm1;
        }
    }
    float *o = a.out + (int64_t)b * a.D * a.Hs * a.Ws + (int64_t)ys * a.Ws + xs;
    const int64_t dstride = (int64_t)a.Hs * a.Ws;
    for (int d = 0; d < a.D; ++d) {
        const float fd = (float)d;
        const float v00 = lo[0][0] + fd * step[0][0], v01 = lo[0][1] + fd * step[0][1];
        const float v10 = lo[1][0] + fd * step[1][0], v11 = lo[1][1] + fd * step[1][1];
        const float r0 = wx[0] * v00 + wx[1] * v01;
        const float r1 = wx[0] * v10 + wx[1] * v11;
        o[d * dstride] = wy[0] * r0 + wy[1] * r1;
    }
}

}  // namespace mvs

using namespace mvs;

extern "C" int mvs_cas_depth_hypotheses_f32(const float *prev_depth, int B, int hp, int wp, int H, int W, int Hs,
                                            int Ws, int D, float half_range, float *out, void *stream) {
    if (!prev_depth || !out || B <= 0 || hp <= 0 || wp <= 0 || H <= 0 || W <= 0 || Hs <= 0 || Ws <= 0 || D < 2) {
        set_error("mvs_cas_depth_hypotheses_f32: invalid argument");
        return MVS_EINVAL;
    }
    HypoArgs a{prev_depth, out, B, hp, wp, H, W, Hs, Ws, D, half_range};
    const int64_t n = (int64_t)B * Hs * Ws;
    if ((n + 255) / 256 > 0x7fffffffLL) return bare_error(MVS_EINVAL, __func__, __LINE__);
    hipLaunchKernelGGL(cas_hypotheses_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), a);
    return check_launch("mvs_cas_depth_hypotheses_f32");
}
