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

// Block = 64 pixels x 4 (REG: 8) depth slices (wave w owns depths [w*D/4, (w+1)*D/4)):
// 4x the memory-level parallelism of one thread per pixel at the 118k-pixel sizes of
// this path; the three partial reductions go through a few hundred bytes of LDS.
// REG = 1 (slices of up to 32 planes, i.e. D <= 256): a thread's slice of the cost column is
// read ONCE into registers and the three passes run from there (the memory version re-read it
// per pass: 276 MB of HBM-side traffic for a 92 MB volume, rocprofv3 FETCH_SIZE); same
// arithmetic, same results.
template <int REG, int NPART = (REG ? 8 : 4)>
__global__ __launch_bounds__(NPART * 64) void softmax_regress_conf_kernel(
    const float *__restrict__ cost, const float *__restrict__ depth, int depth_mode,
    int clamp_idx, int B, int D, int64_t plane, float *__restrict__ out_depth,
    float *__restrict__ out_conf, float *__restrict__ out_prob) {
    // REG > 0: eight depth slices of up to REG planes each, held in registers (REG = 8, 16, 24, 32: the launcher takes
    // the smallest that covers D / 8 -- a slice shorter than its registers would still run every unrolled step)
    // (NPART = 1: up to 32 planes -- the cascade's later stages at full resolution -- a thread takes its whole column)
    constexpr int NP = NPART, NR = REG ? REG : 1;
    __shared__ float s_f[NP][64];
    __shared__ double s_d[2][NP][64];
    const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + lane;
    const bool live = i < (int64_t)B * plane;
    const int64_t ic = live ? i : (int64_t)B * plane - 1;
    const int b = (int)(ic / plane);
    const int64_t pix = ic % plane;
    const float *c = cost + (int64_t)b * D * plane + pix;
    const int d0 = (int)((int64_t)D * part / NP), d1 = (int)((int64_t)D * (part + 1) / NP);
    const int n = d1 - d0;               // wave-uniform
    float v[NR];
    if constexpr (REG) {
        const float *cp = c + (int64_t)d0 * plane;
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            v[k] = k < n ? *cp : -INFINITY;
            cp += plane;
        }
    }
    // max
    float m = -INFINITY;
    if constexpr (REG) {
        // (four chains each for the max and the three sums below: a dependent chain advances every ~50 cycles)
        float m4[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int k = 0; k < NR; ++k) m4[k & 3] = fmaxf(m4[k & 3], v[k]);
        m = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
    } else {
        for (int d = d0; d < d1; ++d) m = fmaxf(m, c[(int64_t)d * plane]);
    }
    s_f[part][lane] = m;
    __syncthreads();
    if constexpr (NP >= 4) m = fmaxf(fmaxf(s_f[0][lane], s_f[1][lane]), fmaxf(s_f[2][lane], s_f[3][lane]));
    if constexpr (NP == 8) m = fmaxf(m, fmaxf(fmaxf(s_f[4][lane], s_f[5][lane]), fmaxf(s_f[6][lane], s_f[7][lane])));
    __syncthreads();
    // normaliser: the fp32 exponentials are summed in fp64 and rounded once, i.e. the
    // correctly rounded sum -- whatever order ATen's vectorised fp32 reduction uses, this
    // is within its rounding error, and it does not depend on the slice split
    double psum = 0.0;
    if constexpr (REG) {
        double p4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            v[k] = expf(v[k] - m);       // kept for the expectation pass (padding planes: exp(-inf) = 0)
            p4[k & 3] += (double)v[k];
        }
        psum = (p4[0] + p4[1]) + (p4[2] + p4[3]);
    } else {
        for (int d = d0; d < d1; ++d) psum += (double)expf(c[(int64_t)d * plane] - m);
    }
    s_d[0][part][lane] = psum;
    __syncthreads();
    double sum_d = psum;
    if constexpr (NP >= 4) sum_d = ((s_d[0][0][lane] + s_d[0][1][lane]) + s_d[0][2][lane]) + s_d[0][3][lane];
    if constexpr (NP == 8) sum_d += ((s_d[0][4][lane] + s_d[0][5][lane]) + s_d[0][6][lane]) + s_d[0][7][lane];
    const float sum = (float)sum_d;
    __syncthreads();
    // The fp32 products p_d * dv_d are the reference's (module.py:102); their SUM is
    // carried in fp64: at D=192 and depths ~900 mm a naive fp32 running sum alone
    // costs up to 5e-4 mm against ATen's cascade summation, half the parity budget.
    double dep = 0.0, fidx = 0.0;
    const float *dv = depth_mode == 0 ? depth + (int64_t)b * D : depth + (int64_t)b * D * plane + pix;
    const int64_t dstride = depth_mode == 0 ? 1 : plane;
    float *pp = (out_prob && live) ? out_prob + (int64_t)b * D * plane + pix : nullptr;
    if constexpr (REG) {
        double dep4[4] = {0.0, 0.0, 0.0, 0.0}, fidx4[4] = {0.0, 0.0, 0.0, 0.0};
        const float *dp = dv + (int64_t)d0 * dstride;
        float *ppk = pp ? pp + (int64_t)d0 * plane : nullptr;
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            if (k < n) {
                const float pr = v[k] / sum;
                dep4[k & 3] += (double)(pr * *dp);                  // module.py:102
                fidx4[k & 3] += (double)(pr * (float)(d0 + k));     // mvsnet.py:189
                if (ppk) *ppk = pr;
            }
            dp += dstride;
            if (ppk) ppk += plane;
        }
        dep = (dep4[0] + dep4[1]) + (dep4[2] + dep4[3]);
        fidx = (fidx4[0] + fidx4[1]) + (fidx4[2] + fidx4[3]);
    } else {
        for (int d = d0; d < d1; ++d) {
            const float pr = expf(c[(int64_t)d * plane] - m) / sum;
            dep += (double)(pr * dv[(int64_t)d * dstride]);   // module.py:102
            fidx += (double)(pr * (float)d);                  // mvsnet.py:189
            if (pp) pp[(int64_t)d * plane] = pr;
        }
    }
    s_d[0][part][lane] = dep;
    s_d[1][part][lane] = fidx;
    __syncthreads();
    if (part != 0 || !live) return;
    if constexpr (NP >= 4) {
        dep = ((s_d[0][0][lane] + s_d[0][1][lane]) + s_d[0][2][lane]) + s_d[0][3][lane];
        fidx = ((s_d[1][0][lane] + s_d[1][1][lane]) + s_d[1][2][lane]) + s_d[1][3][lane];
    }
    if constexpr (NP == 8) {
        dep += ((s_d[0][4][lane] + s_d[0][5][lane]) + s_d[0][6][lane]) + s_d[0][7][lane];
        fidx += ((s_d[1][4][lane] + s_d[1][5][lane]) + s_d[1][6][lane]) + s_d[1][7][lane];
    }
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
#define MVS_SM_LAUNCH(R, T)                                                                                     \
    hipLaunchKernelGGL((softmax_regress_conf_kernel<R, (T) / 64>), dim3(grid), dim3(T), 0, as_stream(stream), cost, depth_values,   \
                       depth_mode, clamp_idx, B, D, plane, out_depth, out_conf, out_prob)
    const int per = (D + 7) / 8;   // planes of the longest of eight slices
    if (D <= 8) MVS_SM_LAUNCH(8, 64);           // a thread per pixel takes the whole column
    else if (D <= 16) MVS_SM_LAUNCH(16, 64);
    else if (D <= 32) MVS_SM_LAUNCH(32, 64);
    else if (per <= 8) MVS_SM_LAUNCH(8, 512);
    else if (per <= 16) MVS_SM_LAUNCH(16, 512);
    else if (per <= 24) MVS_SM_LAUNCH(24, 512);
    else if (per <= 32) MVS_SM_LAUNCH(32, 512);
    else hipLaunchKernelGGL((softmax_regress_conf_kernel<0, 4>), dim3(grid), dim3(256), 0, as_stream(stream), cost, depth_values,
                            depth_mode, clamp_idx, B, D, plane, out_depth, out_conf, out_prob);
#undef MVS_SM_LAUNCH
    return check_launch("mvs_softmax_regress_conf_f32");
}

// The same gradient with the depth axis split over the block: 64 pixels x 4 depth slices (d = s, s + 4, ...) per block, the
// slice's costs held in registers (one pass over the volume instead of four), maximum / sum / expected depth joined through
// LDS.  One thread per pixel was 20480 threads walking 4 x 192 dependent loads for the training step's 128 x 160 map: 0.15 ms
// for 32 MB of traffic.  D <= 4 * kBwdRegs; the sums differ from the one-thread form's in summation order only.
constexpr int kBwdRegs = 64;
__global__ __launch_bounds__(256) void softmax_regress_bwd_sliced_kernel(
    const float *__restrict__ cost, const float *__restrict__ depth, int depth_mode,
    const float *__restrict__ gdepth, int B, int D, int64_t plane, float *__restrict__ gcost) {
    __shared__ float red[3][4][64];
    const int px = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + px;
    const bool live = i < (int64_t)B * plane;
    const int b = live ? (int)(i / plane) : 0;
    const int64_t pix = live ? i % plane : 0;
    const float *c = cost + (int64_t)b * D * plane + pix;
    const float *dv = depth_mode == 0 ? depth + (int64_t)b * D : depth + (int64_t)b * D * plane + pix;
    const int64_t dstride = depth_mode == 0 ? 1 : plane;
    float v[kBwdRegs];
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < kBwdRegs; ++k) {
        const int d = sl + 4 * k;
        v[k] = (live && d < D) ? c[(int64_t)d * plane] : -INFINITY;
        m = fmaxf(m, v[k]);
    }
    red[0][sl][px] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0][0][px], red[0][1][px]), fmaxf(red[0][2][px], red[0][3][px]));
    float sum = 0.0f, wd = 0.0f;
#pragma unroll
    for (int k = 0; k < kBwdRegs; ++k) {
        const int d = sl + 4 * k;
        if (live && d < D) {
            v[k] = expf(v[k] - m);
            sum += v[k];
            wd += v[k] * dv[(int64_t)d * dstride];
        }
    }
    red[1][sl][px] = sum; red[2][sl][px] = wd;
    __syncthreads();
    sum = (red[1][0][px] + red[1][1][px]) + (red[1][2][px] + red[1][3][px]);
    wd = (red[2][0][px] + red[2][1][px]) + (red[2][2][px] + red[2][3][px]);
    if (!live) return;
    const float dep = wd / sum, g = gdepth[i];
    float *gc = gcost + (int64_t)b * D * plane + pix;
#pragma unroll
    for (int k = 0; k < kBwdRegs; ++k) {
        const int d = sl + 4 * k;
        if (d < D) gc[(int64_t)d * plane] = g * (v[k] / sum) * (dv[(int64_t)d * dstride] - dep);
    }
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
    if (D <= 4 * kBwdRegs) {
        hipLaunchKernelGGL(softmax_regress_bwd_sliced_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, as_stream(stream),
                           cost, depth_values, depth_mode, grad_depth, B, D, plane, grad_cost);
        return check_launch("mvs_softmax_regress_bwd_f32");
    }
    unsigned grid = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(softmax_regress_bwd_kernel, dim3(grid), dim3(256), 0, as_stream(stream),
                       cost, depth_values, depth_mode, grad_depth, B, D, plane, grad_cost);
    return check_launch("mvs_softmax_regress_bwd_f32");
}
