// K4+K5: softmax over the depth axis, depth expectation and photometric
// confidence in one kernel (replaces mvsnet.py:183-191 and module.py:91-103:
// the reference makes ~7 passes over the probability volume; this makes three
// reads of the cost volume -- max, normaliser, expectation -- that stay in
// L2 / Infinity Cache, and never writes the probability volume unless asked).
//
// One thread per pixel, x fastest: every read of cost[b,d,y,:] is a coalesced
// 256-B wavefront row.  Arithmetic follows ATen's softmax: p = exp(c-max)/sum.
#include "mvs_common.h"

#include <cmath>

namespace mvs {

// Block = 64 pixels x 4 depth quarters (wave w owns depths [w*D/4, (w+1)*D/4)):
// 4x the memory-level parallelism of one thread per pixel at the 118k-pixel sizes of
// this path; the three partial reductions go through a few hundred bytes of LDS.
// REG = 1 (quarters of up to 64 planes, i.e. D <= 256): a thread's slice of the cost column is
// read ONCE into registers and the three passes run from there (the memory version re-read it
// per pass: 276 MB of HBM-side traffic for a 92 MB volume, rocprofv3 FETCH_SIZE); same
// arithmetic, same results.
template <int REG>
__global__ __launch_bounds__(256) void softmax_regress_conf_kernel(
    const float *__restrict__ cost, const float *__restrict__ depth, int depth_mode,
    int clamp_idx, int B, int D, int64_t plane, float *__restrict__ out_depth,
    float *__restrict__ out_conf, float *__restrict__ out_prob) {
    constexpr int NR = 64;
    __shared__ float s_f[4][64];
    __shared__ double s_d[2][4][64];
    const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + lane;
    const bool live = i < (int64_t)B * plane;
    const int64_t ic = live ? i : (int64_t)B * plane - 1;
    const int b = (int)(ic / plane);
    const int64_t pix = ic % plane;
    const float *c = cost + (int64_t)b * D * plane + pix;
    const int d0 = (int)((int64_t)D * part / 4), d1 = (int)((int64_t)D * (part + 1) / 4);
    float v[REG ? NR : 1];
    if constexpr (REG) {
#pragma unroll
        for (int k = 0; k < NR; ++k) v[k] = d0 + k < d1 ? c[(int64_t)(d0 + k) * plane] : -INFINITY;
    }
    // max
    float m = -INFINITY;
    if constexpr (REG) {
#pragma unroll
        for (int k = 0; k < NR; ++k) m = fmaxf(m, v[k]);
    } else {
        for (int d = d0; d < d1; ++d) m = fmaxf(m, c[(int64_t)d * plane]);
    }
    s_f[part][lane] = m;
    __syncthreads();
    m = fmaxf(fmaxf(s_f[0][lane], s_f[1][lane]), fmaxf(s_f[2][lane], s_f[3][lane]));
    __syncthreads();
    // normaliser: the fp32 exponentials are summed in fp64 and rounded once, i.e. the
    // correctly rounded sum -- whatever order ATen's vectorised fp32 reduction uses, this
    // is within its rounding error, and it does not depend on the quarter split
    double psum = 0.0;
    if constexpr (REG) {
#pragma unroll
        for (int k = 0; k < NR; ++k)
            if (d0 + k < d1) psum += (double)expf(v[k] - m);
    } else {
        for (int d = d0; d < d1; ++d) psum += (double)expf(c[(int64_t)d * plane] - m);
    }
    s_d[0][part][lane] = psum;
    __syncthreads();
    const float sum =
        (float)(((s_d[0][0][lane] + s_d[0][1][lane]) + s_d[0][2][lane]) + s_d[0][3][lane]);
    __syncthreads();
    // The fp32 products p_d * dv_d are the reference's (module.py:102); their SUM is
    // carried in fp64: at D=192 and depths ~900 mm a naive fp32 running sum alone
    // costs up to 5e-4 mm against ATen's cascade summation, half the parity budget.
    double dep = 0.0, fidx = 0.0;
    const float *dv = depth_mode == 0 ? depth + (int64_t)b * D : depth + (int64_t)b * D * plane + pix;
    const int64_t dstride = depth_mode == 0 ? 1 : plane;
    float *pp = (out_prob && live) ? out_prob + (int64_t)b * D * plane + pix : nullptr;
    auto step = [&](int d, float cv) {
        const float pr = expf(cv - m) / sum;
        dep += (double)(pr * dv[(int64_t)d * dstride]);   // module.py:102
        fidx += (double)(pr * (float)d);                  // mvsnet.py:189
        if (pp) pp[(int64_t)d * plane] = pr;
    };
    if constexpr (REG) {
#pragma unroll
        for (int k = 0; k < NR; ++k)
            if (d0 + k < d1) step(d0 + k, v[k]);
    } else {
        for (int d = d0; d < d1; ++d) step(d, c[(int64_t)d * plane]);
    }
    s_d[0][part][lane] = dep;
    s_d[1][part][lane] = fidx;
    __syncthreads();
    if (part != 0 || !live) return;
    dep = ((s_d[0][0][lane] + s_d[0][1][lane]) + s_d[0][2][lane]) + s_d[0][3][lane];
    fidx = ((s_d[1][0][lane] + s_d[1][1][lane]) + s_d[1][2][lane]) + s_d[1][3][lane];
    // .long() truncates toward zero (mvsnet.py:189); Cas clamps (cas_mvsnet.py:63)
    int idx = (int)(float)fidx;
    if (clamp_idx) idx = min(max(idx, 0), D - 1);
    // 4 * avg_pool3d over the (1,2)-padded window == p[idx-1] + ... + p[idx+2]
    float s4 = 0.0f;
#pragma unroll
    for (int k = -1; k <= 2; ++k) {
        int dd = idx + k;
        if (dd >= 0 && dd < D) s4 += expf(c[(int64_t)dd * plane] - m) / sum;
    }
    out_depth[i] = (float)dep;
    out_conf[i] = s4;
}

// d depth / d cost_k = p_k (dv_k - depth); grad_cost = grad_depth * that.
__global__ __launch_bounds__(256) void softmax_regress_bwd_kernel(
    const float *__restrict__ cost, const float *__restrict__ depth, int depth_mode,
    const float *__restrict__ gdepth, int B, int D, int64_t plane,
    float *__restrict__ gcost) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * plane) return;
    const int b = (int)(i / plane);
    const int64_t pix = i % plane;
    const float *c = cost + (int64_t)b * D * plane + pix;
    float m = c[0];
    for (int d = 1; d < D; ++d) m = fmaxf(m, c[(int64_t)d * plane]);
    float sum = 0.0f;
    for (int d = 0; d < D; ++d) sum += expf(c[(int64_t)d * plane] - m);
    const float *dv = depth_mode == 0 ? depth + (int64_t)b * D : depth + (int64_t)b * D * plane + pix;
    const int64_t dstride = depth_mode == 0 ? 1 : plane;
    float dep = 0.0f;
    for (int d = 0; d < D; ++d) dep += expf(c[(int64_t)d * plane] - m) / sum * dv[(int64_t)d * dstride];
    const float g = gdepth[i];
    float *gc = gcost + (int64_t)b * D * plane + pix;
    for (int d = 0; d < D; ++d) {
        float pr = expf(c[(int64_t)d * plane] - m) / sum;
        gc[(int64_t)d * plane] = g * pr * (dv[(int64_t)d * dstride] - dep);
    }
}

}  // namespace mvs

using namespace mvs;

extern "C" int mvs_softmax_regress_conf_f32(const float *cost, const float *depth_values,
                                            int depth_mode, int clamp_idx, int B, int D, int H,
                                            int W, float *out_depth, float *out_conf,
                                            float *out_prob, void *stream) {
    if (!cost || !depth_values || !out_depth || !out_conf || B <= 0 || D <= 0 || H <= 0 ||
        W <= 0 || depth_mode < 0 || depth_mode > 1) {
        set_error("mvs_softmax_regress_conf_f32: invalid argument");
        return MVS_EINVAL;
    }
    const int64_t plane = (int64_t)H * W;
    const int64_t n = (int64_t)B * plane;
    unsigned grid = (unsigned)((n + 63) / 64);
    if (D <= 256)   // a quarter of the column (<= 64 planes) fits a thread's registers
        hipLaunchKernelGGL(softmax_regress_conf_kernel<1>, dim3(grid), dim3(256), 0, as_stream(stream),
                           cost, depth_values, depth_mode, clamp_idx, B, D, plane, out_depth, out_conf,
                           out_prob);
    else
        hipLaunchKernelGGL(softmax_regress_conf_kernel<0>, dim3(grid), dim3(256), 0, as_stream(stream),
                           cost, depth_values, depth_mode, clamp_idx, B, D, plane, out_depth, out_conf,
                           out_prob);
    return check_launch("mvs_softmax_regress_conf_f32");
}

extern "C" int mvs_softmax_regress_bwd_f32(const float *cost, const float *depth_values,
                                           int depth_mode, const float *grad_depth, int B, int D,
                                           int H, int W, float *grad_cost, void *stream) {
    if (!cost || !depth_values || !grad_depth || !grad_cost || B <= 0 || D <= 0 || H <= 0 ||
        W <= 0 || depth_mode < 0 || depth_mode > 1) {
        set_error("mvs_softmax_regress_bwd_f32: invalid argument");
        return MVS_EINVAL;
    }
    const int64_t plane = (int64_t)H * W;
    const int64_t n = (int64_t)B * plane;
    unsigned grid = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(softmax_regress_bwd_kernel, dim3(grid), dim3(256), 0, as_stream(stream),
                       cost, depth_values, depth_mode, grad_depth, B, D, plane, grad_cost);
    return check_launch("mvs_softmax_regress_bwd_f32");
}
