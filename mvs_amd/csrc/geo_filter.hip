// Geometric-consistency check of the depth-map filter that follows the path (SURVEY.md 8f rank 2;
// MVSNet/eval.py:136-214 reproject_with_depth + check_geometric_consistency, and the per-reference
// sums of filter_depth, eval.py:239-262).  The reference does this in numpy + cv2.remap on the
// CPU, once per (reference, source) pair -- 49 x 10 pairs per DTU scan.  Here one thread owns a
// reference pixel and walks the source views: project with the reference depth, sample the
// source depth map, project back, test |p' - p| < 1 px and |d' - d| / d < 1 %, and accumulate
// the number of consistent views and the sum of their reprojected depths in the reference's
// order.  Per-pixel arithmetic is fp64 with fp32 matrices and maps, as numpy's type promotion
// makes it; the sampling restates cv2.remap(INTER_LINEAR): coordinates rounded to 1/32 pixel
// (half to even), fp32 bilinear weights from the rounded fraction, constant border 0.
#include "mvs_common.h"

namespace mvs {

struct GeoArgs {
    const float *depth_ref;    // [H,W]
    const float *depth_src;    // [S,H,W]
    const float *mats;         // [18 + 50 S]: Kref^-1 (9), Kref (9); per view: K (9), K^-1 (9), Esrc Eref^-1 (16), Eref Esrc^-1 (16)
    unsigned char *mask;       // [S,H,W] or null
    float *depth_rep;          // [S,H,W] or null (0 outside the mask)
    float *xy_src;             // [S,2,H,W] or null
    int *geo_sum;              // [H,W]
    double *depth_avg;         // [H,W]
    int H, W, S;
};

__device__ __forceinline__ void mat3(const float *m, double a, double b, double c, double &x, double &y, double &z) {
    x = (double)m[0] * a + (double)m[1] * b + (double)m[2] * c;
    y = (double)m[3] * a + (double)m[4] * b + (double)m[5] * c;
    z = (double)m[6] * a + (double)m[7] * b + (double)m[8] * c;
}
__device__ __forceinline__ void mat4x3(const float *m, double a, double b, double c, double &x, double &y, double &z) {
    x = (double)m[0] * a + (double)m[1] * b + (double)m[2] * c + (double)m[3];
    y = (double)m[4] * a + (double)m[5] * b + (double)m[6] * c + (double)m[7];
    z = (double)m[8] * a + (double)m[9] * b + (double)m[10] * c + (double)m[11];
}

// cv2.remap(src, x, y, INTER_LINEAR), float image, BORDER_CONSTANT 0
__device__ __forceinline__ float remap_linear(const float *__restrict__ src, int H, int W, float x, float y) {
    const double fx = (double)x * 32.0, fy = (double)y * 32.0;
    if (!(fabs(fx) <= 1073741824.0) || !(fabs(fy) <= 1073741824.0)) return 0.0f;   // NaN / inf / far outside
    const long long sx = (long long)rint(fx), sy = (long long)rint(fy);            // cvRound: half to even
    const int x0 = (int)(sx >> 5), y0 = (int)(sy >> 5);
    const float ax = (float)(sx & 31) / 32.0f, ay = (float)(sy & 31) / 32.0f;
    const float w00 = (1.0f - ay) * (1.0f - ax), w01 = (1.0f - ay) * ax, w10 = ay * (1.0f - ax), w11 = ay * ax;
    auto tap = [&](int yy, int xx) {
        return (xx >= 0 && xx < W && yy >= 0 && yy < H) ? src[(int64_t)yy * W + xx] : 0.0f;
    };
    return tap(y0, x0) * w00 + tap(y0, x0 + 1) * w01 + tap(y0 + 1, x0) * w10 + tap(y0 + 1, x0 + 1) * w11;
}

__global__ __launch_bounds__(256) void geo_consistency_kernel(GeoArgs a) {
    const int64_t plane = (int64_t)a.H * a.W;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= plane) return;
    const int px = (int)(idx % a.W), py = (int)(idx / a.W);
    const float dref = a.depth_ref[idx];
    const double d = (double)dref;
    double rx, ry, rz;
    mat3(a.mats, (double)px * d, (double)py * d, d, rx, ry, rz);             // Kref^-1 (x d, y d, d)
    int cnt = 0;
    float dsum = 0.0f;   // sum(all_srcview_depth_ests): fp32, source views in order (eval.py:256)
    for (int s = 0; s < a.S; ++s) {
        const float *m = a.mats + 18 + 50 * s;
        double sx3, sy3, sz3, kx, ky, kz;
        mat4x3(m + 18, rx, ry, rz, sx3, sy3, sz3);                           // into the source camera
        mat3(m, sx3, sy3, sz3, kx, ky, kz);
        const double xs = kx / kz, ys = ky / kz;
        const float xsf = (float)xs, ysf = (float)ys;
        const float sampled = remap_linear(a.depth_src + s * plane, a.H, a.W, xsf, ysf);
        const double sd = (double)sampled;
        double bx, by, bz, qx, qy, qz, ux, uy, uz;
        mat3(m + 9, xs * sd, ys * sd, sd, bx, by, bz);                       // Ksrc^-1 (xy_src, 1) * sampled depth
        mat4x3(m + 34, bx, by, bz, qx, qy, qz);                              // back into the reference camera
        const float drep = (float)qz;
        mat3(a.mats + 9, qx, qy, qz, ux, uy, uz);
        const float xr = (float)(ux / uz), yr = (float)(uy / uz);
        const double ddx = (double)xr - (double)px, ddy = (double)yr - (double)py;
        const double dist = sqrt(ddx * ddx + ddy * ddy);
        const float rel = fabsf(drep - dref) / dref;
        const bool ok = (dist < 1.0) && (rel < 0.01f);
        const float dout = ok ? drep : 0.0f;
        cnt += ok ? 1 : 0;
        dsum = s == 0 ? (0.0f + dout) : (dsum + dout);
        if (a.mask) a.mask[s * plane + idx] = ok ? 1 : 0;
        if (a.depth_rep) a.depth_rep[s * plane + idx] = dout;
        if (a.xy_src) { a.xy_src[(2 * s) * plane + idx] = xsf; a.xy_src[(2 * s + 1) * plane + idx] = ysf; }
    }
    a.geo_sum[idx] = cnt;
    a.depth_avg[idx] = (double)(dsum + dref) / (double)(cnt + 1);          // fp32 sum / int -> fp64 (eval.py:256)
}

}  // namespace mvs

using namespace mvs;

extern "C" int mvs_geo_consistency_f32(const float *depth_ref, const float *depth_src, const float *mats, int S,
                                       int H, int W, unsigned char *mask, float *depth_reprojected, float *xy_src,
                                       int *geo_mask_sum, double *depth_averaged, void *stream) {
    if (!depth_ref || !depth_src || !mats || !geo_mask_sum || !depth_averaged || S <= 0 || H <= 0 || W <= 0) {
        set_error("mvs_geo_consistency_f32: invalid argument");
        return MVS_EINVAL;
    }
    GeoArgs a{depth_ref, depth_src, mats, mask, depth_reprojected, xy_src, geo_mask_sum, depth_averaged, H, W, S};
    const int64_t n = (int64_t)H * W;
    if ((n + 255) / 256 > 0x7fffffffLL) return bare_error(MVS_EINVAL, __func__, __LINE__);
    hipLaunchKernelGGL(geo_consistency_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), a);
    return check_launch("mvs_geo_consistency_f32");
}
