// Device side of the input pipeline: what the reference's loader does to
// a decoded image and a camera file after parsing, MVSNet/datasets/dtu_yao_eval.py:60-67,
// 93-95, 102 -- np.array(img, float32) / 255., crop of the bottom rows, HWC -> CHW, and
// proj[:3,:4] = (K with rows 0-1 / 4) @ E[:3,:4] -- so that the host ships uint8 pixels (a
// quarter of the bytes) and never touches them again.
#include "mvs_common.h"

namespace mvs {

// lane = 4 consecutive pixels of one row: 12 bytes in (three dwords), one float4 per plane out
__global__ __launch_bounds__(256) void u8_hwc_to_planar_f32_kernel(
    const unsigned char *__restrict__ in, int Hs, int Ws, int H, int W, float *__restrict__ out) {
    const int n = blockIdx.z, y = blockIdx.y;
    const int x4 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (x4 >= W) return;
    const unsigned char *row = in + ((size_t)n * Hs + y) * (size_t)Ws * 3;
    float *o = out + ((size_t)n * 3 * H + y) * (size_t)W;
    const size_t plane = (size_t)H * W;
    if (x4 + 4 <= W && ((uintptr_t)(row + (size_t)x4 * 3) & 3) == 0 && (W & 3) == 0) {
        const unsigned *p = reinterpret_cast<const unsigned *>(row + (size_t)x4 * 3);
        const unsigned a = p[0], b = p[1], c = p[2];   // r0 g0 b0 r1 | g1 b1 r2 g2 | b2 r3 g3 b3
        const float r[4] = {(float)(a & 255u), (float)(a >> 24), (float)((b >> 16) & 255u), (float)((c >> 8) & 255u)};
        const float g[4] = {(float)((a >> 8) & 255u), (float)(b & 255u), (float)(b >> 24), (float)((c >> 16) & 255u)};
        const float bl[4] = {(float)((a >> 16) & 255u), (float)((b >> 8) & 255u), (float)(c & 255u), (float)(c >> 24)};
        // IEEE division, as numpy's float32 array / 255.
        *reinterpret_cast<float4 *>(o + x4) = make_float4(r[0] / 255.0f, r[1] / 255.0f, r[2] / 255.0f, r[3] / 255.0f);
        *reinterpret_cast<float4 *>(o + plane + x4) = make_float4(g[0] / 255.0f, g[1] / 255.0f, g[2] / 255.0f, g[3] / 255.0f);
        *reinterpret_cast<float4 *>(o + 2 * plane + x4) = make_float4(bl[0] / 255.0f, bl[1] / 255.0f, bl[2] / 255.0f, bl[3] / 255.0f);
    } else {
        for (int x = x4; x < min(x4 + 4, W); ++x)
#pragma unroll
            for (int c = 0; c < 3; ++c) o[c * plane + x] = (float)row[(size_t)x * 3 + c] / 255.0f;
    }
}

// one thread per view.  The 3-term dot products are the FMA chain of the BLAS sgemm kernel numpy's
// float32 matmul runs (first product rounded, then two FMAs): bit-identical to the reference
// loader's matrices (tests/golden/g10_io.npz).
__global__ void proj_matrices_kernel(const float *__restrict__ K, const float *__restrict__ E, float div,
                                     int N, float *__restrict__ out) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float *k = K + n * 9, *e = E + n * 16;
    float *o = out + n * 16;
    for (int i = 0; i < 3; ++i) {
        const float s = i < 2 ? div : 1.0f;
        const float k0 = k[i * 3] / s, k1 = k[i * 3 + 1] / s, k2 = k[i * 3 + 2] / s;
        for (int j = 0; j < 4; ++j)
            o[i * 4 + j] = __fmaf_rn(k2, e[8 + j], __fmaf_rn(k1, e[4 + j], k0 * e[j]));
    }
    for (int j = 0; j < 4; ++j) o[12 + j] = e[12 + j];
}

}  // namespace mvs

using namespace mvs;

extern "C" int mvs_images_u8_to_planar_f32(const unsigned char *in, int N, int Hs, int Ws, int H, int W,
                                           float *out, void *stream) {
    if (!in || !out || N <= 0 || H <= 0 || W <= 0 || H > Hs || W > Ws || N > 65535 || H > 65535) {
        set_error("mvs_images_u8_to_planar_f32: invalid argument (N=%d, source %dx%d, crop %dx%d)", N, Hs, Ws, H, W);
        return MVS_EINVAL;
    }
    const dim3 grid((unsigned)((W + 1023) / 1024), (unsigned)H, (unsigned)N);
    hipLaunchKernelGGL(u8_hwc_to_planar_f32_kernel, grid, dim3(256), 0, as_stream(stream), in, Hs, Ws, H, W, out);
    return check_launch("mvs_images_u8_to_planar_f32");
}

extern "C" int mvs_proj_matrices_f32(const float *K, const float *E, float intrinsics_div, int N, float *out,
                                     void *stream) {
    if (!K || !E || !out || N <= 0 || !(intrinsics_div > 0.0f)) {
        set_error("mvs_proj_matrices_f32: invalid argument");
        return MVS_EINVAL;
    }
    hipLaunchKernelGGL(proj_matrices_kernel, dim3((N + 63) / 64), dim3(64), 0, as_stream(stream), K, E,
                       intrinsics_div, N, out);
    return check_launch("mvs_proj_matrices_f32");
}
