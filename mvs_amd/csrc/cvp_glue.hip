// Glue between the levels of the CVP-MVSNet pyramid, on the device:
//   image pyramid      net.py:45   F.interpolate(img, scale_factor=0.5, 'bilinear')
//   depth upsample     net.py:171  F.interpolate(depth[None], scale_factor=2, 'bicubic')
//   matrix algebra     modules.py:149-152,205-219 (fp64 inverses / products of the 3x3 and 4x4 camera matrices)
//   hypotheses         modules.py:213-219  depth_up + k * mean interval, k = -4..3
#include "mvs_common.h"

namespace mvs {

// x0.5 bilinear, align_corners=False: the source coordinate of output pixel o is 2 o + 0.5, i.e. the
// four pixels of its 2x2 block with weights 0.25 -- summed in ATen's order (row 0 left, right, row 1
// left, right): bit-identical to the CPU kernel.
__global__ __launch_bounds__(256) void downsample_half_kernel(const float *__restrict__ in, int H, int W, int Ho,
                                                              int Wo, int64_t planes, float *__restrict__ out) {
    const int64_t total = planes * Ho * Wo;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int x = (int)(idx % Wo), y = (int)((idx / Wo) % Ho);
    const int64_t pl = idx / ((int64_t)Wo * Ho);
    const float *r0 = in + (pl * H + 2 * y) * (int64_t)W + 2 * x;
    const float *r1 = r0 + (2 * y + 1 < H ? W : 0);
    const int dx = 2 * x + 1 < W ? 1 : 0;
    out[idx] = ((r0[0] * 0.25f + r0[dx] * 0.25f) + r1[0] * 0.25f) + r1[dx] * 0.25f;
}

// ATen's cubic convolution coefficients, A = -0.75
__device__ __forceinline__ void cubic_coeffs(float t, float (&w)[4]) {
    const float A = -0.75f;
    auto c1 = [&](float x) { return ((A + 2.0f) * x - (A + 3.0f)) * x * x + 1.0f; };
    auto c2 = [&](float x) { return ((A * x - 5.0f * A) * x + 8.0f * A) * x - 4.0f * A; };
    w[0] = c2(t + 1.0f); w[1] = c1(t); w[2] = c1(1.0f - t); w[3] = c2(2.0f - t);
}

// x2 bicubic, align_corners=False: source coordinate (o + 0.5) / 2 - 0.5, 4x4 taps with indices clamped
// to the image (upsample_bicubic2d); rows interpolated along x, then along y, FMA chains.
__global__ __launch_bounds__(256) void upsample_bicubic2x_kernel(const float *__restrict__ in, int H, int W,
                                                                 int64_t planes, float *__restrict__ out) {
    const int Ho = 2 * H, Wo = 2 * W;
    const int64_t total = planes * Ho * Wo;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int ox = (int)(idx % Wo), oy = (int)((idx / Wo) % Ho);
    const float *src = in + (idx / ((int64_t)Wo * Ho)) * (int64_t)H * W;
    const float rx = ((float)ox + 0.5f) * 0.5f - 0.5f, ry = ((float)oy + 0.5f) * 0.5f - 0.5f;
    const float fx = floorf(rx), fy = floorf(ry);
    float wx[4], wy[4];
    cubic_coeffs(rx - fx, wx);
    cubic_coeffs(ry - fy, wy);
    const int ix = (int)fx, iy = (int)fy;
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float *row = src + (int64_t)min(max(iy - 1 + i, 0), H - 1) * W;
        float t = row[min(max(ix - 1, 0), W - 1)] * wx[0];
#pragma unroll
        for (int j = 1; j < 4; ++j) t = __fmaf_rn(row[min(max(ix - 1 + j, 0), W - 1)], wx[j], t);
        acc = i == 0 ? t * wy[0] : __fmaf_rn(t, wy[i], acc);
    }
    out[idx] = acc;
}

// ---- fp64 camera algebra of calDepthHypo, one thread ------------------------------------------
__device__ void inv3(const double *m, double *o) {
    const double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
    const double det = m[0] * c00 + m[1] * c01 + m[2] * c02, r = 1.0 / det;
    o[0] = c00 * r; o[1] = (m[2] * m[7] - m[1] * m[8]) * r; o[2] = (m[1] * m[5] - m[2] * m[4]) * r;
    o[3] = c01 * r; o[4] = (m[0] * m[8] - m[2] * m[6]) * r; o[5] = (m[2] * m[3] - m[0] * m[5]) * r;
    o[6] = c02 * r; o[7] = (m[1] * m[6] - m[0] * m[7]) * r; o[8] = (m[0] * m[4] - m[1] * m[3]) * r;
}
// Gauss-Jordan with partial pivoting (the matrices are rigid transforms: well conditioned)
__device__ void inv4(const double *m, double *o) {
    double a[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) { a[i][j] = m[i * 4 + j]; a[i][4 + j] = i == j ? 1.0 : 0.0; }
    for (int c = 0; c < 4; ++c) {
        int p = c;
        for (int r = c + 1; r < 4; ++r) if (fabs(a[r][c]) > fabs(a[p][c])) p = r;
        for (int j = 0; j < 8; ++j) { const double t = a[c][j]; a[c][j] = a[p][j]; a[p][j] = t; }
        const double inv = 1.0 / a[c][c];
        for (int j = 0; j < 8; ++j) a[c][j] *= inv;
        for (int r = 0; r < 4; ++r) {
            if (r == c) continue;
            const double f = a[r][c];
            for (int j = 0; j < 8; ++j) a[r][j] -= f * a[c][j];
        }
    }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) o[i * 4 + j] = a[i][4 + j];
}
__device__ void mul3(const double *a, const double *b, double *o) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) o[i * 3 + j] = a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j] + a[i * 3 + 2] * b[6 + j];
}

// mats[59] = inverse(K_ref) (9), inverse(E_ref) (16), K_src (9), E_src (16), K_ref R_ref inverse(K_src R_src) (9)
__global__ void cvp_hypothesis_mats_kernel(const float *__restrict__ Kr, const float *__restrict__ Ks,
                                           const float *__restrict__ Er, const float *__restrict__ Es,
                                           double *__restrict__ mats) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double kr[9], ks[9], er[16], es[16], rr[9], rs[9], t0[9], t1[9], t2[9];
    for (int i = 0; i < 9; ++i) { kr[i] = (double)Kr[i]; ks[i] = (double)Ks[i]; }
    for (int i = 0; i < 16; ++i) { er[i] = (double)Er[i]; es[i] = (double)Es[i]; }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) { rr[i * 3 + j] = er[i * 4 + j]; rs[i * 3 + j] = es[i * 4 + j]; }
    inv3(kr, mats);
    inv4(er, mats + 9);
    for (int i = 0; i < 9; ++i) mats[25 + i] = ks[i];
    for (int i = 0; i < 16; ++i) mats[34 + i] = es[i];
    mul3(kr, rr, t0);
    mul3(ks, rs, t1);
    inv3(t1, t2);
    mul3(t0, t2, mats + 50);
}

// out[k] = depth_up + (k - d) * interval, interval = float(sum_abs / (H W))   (modules.py:213-219)
__global__ __launch_bounds__(256) void cvp_hypotheses_kernel(const float *__restrict__ depth_up,
                                                             const double *__restrict__ sum_abs, int64_t n, int d,
                                                             float *__restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const float interval = (float)(sum_abs[0] / (double)n);
    const float v = depth_up[idx];
    for (int k = 0; k < 2 * d; ++k) out[(int64_t)k * n + idx] = v + (float)(k - d) * interval;
}

}  // namespace mvs

using namespace mvs;

extern "C" int mvs_downsample_bilinear_half_f32(const float *in, int64_t planes, int H, int W, float *out,
                                                void *stream) {
    if (!in || !out || planes <= 0 || H < 2 || W < 2) {
        set_error("mvs_downsample_bilinear_half_f32: invalid argument");
        return MVS_EINVAL;
    }
    const int Ho = H / 2, Wo = W / 2;
    const int64_t total = planes * Ho * Wo;
    if ((total + 255) / 256 > 0x7fffffffLL) return bare_error(MVS_EINVAL, __func__, __LINE__);
    hipLaunchKernelGGL(downsample_half_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), in,
                       H, W, Ho, Wo, planes, out);
    return check_launch("mvs_downsample_bilinear_half_f32");
}

extern "C" int mvs_upsample_bicubic2x_f32(const float *in, int64_t planes, int H, int W, float *out, void *stream) {
    if (!in || !out || planes <= 0 || H < 1 || W < 1) {
        set_error("mvs_upsample_bicubic2x_f32: invalid argument");
        return MVS_EINVAL;
    }
    const int64_t total = planes * 4 * (int64_t)H * W;
    if ((total + 255) / 256 > 0x7fffffffLL) return bare_error(MVS_EINVAL, __func__, __LINE__);
    hipLaunchKernelGGL(upsample_bicubic2x_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream),
                       in, H, W, planes, out);
    return check_launch("mvs_upsample_bicubic2x_f32");
}

extern "C" int mvs_cvp_hypothesis_mats_f64(const float *K_ref, const float *K_src, const float *E_ref,
                                           const float *E_src, double *mats, void *stream) {
    if (!K_ref || !K_src || !E_ref || !E_src || !mats) {
        set_error("mvs_cvp_hypothesis_mats_f64: invalid argument");
        return MVS_EINVAL;
    }
    hipLaunchKernelGGL(cvp_hypothesis_mats_kernel, dim3(1), dim3(64), 0, as_stream(stream), K_ref, K_src, E_ref, E_src, mats);
    return check_launch("mvs_cvp_hypothesis_mats_f64");
}

extern "C" int mvs_cvp_hypotheses_f32(const float *depth_up, const double *sum_abs, int H, int W, int d, float *out,
                                      void *stream) {
    if (!depth_up || !sum_abs || !out || H <= 0 || W <= 0 || d <= 0) {
        set_error("mvs_cvp_hypotheses_f32: invalid argument");
        return MVS_EINVAL;
    }
    const int64_t n = (int64_t)H * W;
    hipLaunchKernelGGL(cvp_hypotheses_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), depth_up,
                       sum_abs, n, d, out);
    return check_launch("mvs_cvp_hypotheses_f32");
}
