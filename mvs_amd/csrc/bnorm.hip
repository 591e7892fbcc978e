// Training-mode BatchNorm (batch statistics) fused with the ReLU and the skip add that follow
// it in the reference's blocks: ConvBnReLU3D = relu(bn(conv(x))) (module.py:26-33) and the
// CostRegNet decoder's `skip + relu(bn(deconv(x)))` (mvsnet.py:89-91).  Channels-last rows
// [N][C] (N = B*D*H*W voxels, C in {8,16,32,64}).
//
// As torch ops these are 4 (forward) + 5 (backward) full passes over the activations per layer
// plus the BatchNorm kernels' own -- 3.4 of the 13.3 ms of a training step.  Here:
//   forward   stats (read x) -> finalize [C] -> apply (read x [+ skip], write y)
//   backward  reduce (read x, dy) -> finalize [C] -> elementwise (read x, dy, write dx)
// The ReLU mask is recomputed from x (z = bn(x) > 0), so neither z nor the mask is stored.
// Per-channel sums are carried in fp64 (per thread, per block and across blocks): var =
// E[x^2] - mean^2 is then safe at N ~ 4e6, and the result does not depend on the grid.
#include "mvs_common.h"

namespace mvs {

constexpr int kBnBlocks = 512;   // 2 per CU; every block writes one [2][C] fp64 partial

struct BnArgs {
    const float *x, *dy, *skip, *weight, *bias;
    const float *mean, *invstd;      // [C]  (apply / backward)
    const float *coef;               // [2][C] backward: k1 = sum(dz (x-mean)) invstd^2 / N, k2 = sum(dz) / N
    float *out;                      // y or dx
    double *partial;                 // [kBnBlocks][2][C]
    int64_t nq;                      // float4 elements of ONE group: N * C / 4
    int C, relu;
    // groups (round 4): G consecutive blocks of N rows, each with its OWN statistics -- the reference calls FeatureNet once per
    // view (mvsnet.py:146), so a batch of views [V*B, h, w, C] is V groups; blockIdx.y = group; mean / invstd / coef are
    // [G][C] / [G][2][C], partial is [G][gridDim.x][2][C]
    int G;
};

__device__ __forceinline__ float4 ld4(const float *p, int64_t e) { return reinterpret_cast<const float4 *>(p)[e]; }

// MODE 0: sum x, sum x^2.  MODE 1: sum dz, sum dz (x - mean), dz = dy masked by z > 0.
template <int MODE>
__global__ __launch_bounds__(256) void bn_reduce_kernel(BnArgs a) {
    __shared__ double red[8][256];
    const int Q = a.C >> 2, tid = threadIdx.x, q = tid % Q, grp = blockIdx.y;
    const int64_t g0 = (int64_t)grp * a.nq;        // first float4 of this group
    float mean[4] = {0, 0, 0, 0}, istd[4] = {0, 0, 0, 0}, w[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
    if (MODE == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            mean[j] = a.mean[grp * a.C + q * 4 + j]; istd[j] = a.invstd[grp * a.C + q * 4 + j];
            w[j] = a.weight[q * 4 + j]; b[j] = a.bias[q * 4 + j];
        }
    }
    double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
    const int64_t step = (int64_t)gridDim.x * 256;
    // Four elements of the thread's stride per round, their loads issued together: with two blocks per CU and one 16-byte
    // load in flight per thread the pass ran at 2.5 TB/s (8 KB in flight per CU against ~1 us of latency).  The sums take
    // the elements in the same order as before.
    constexpr int U = 4;
    for (int64_t el = (int64_t)blockIdx.x * 256 + tid; el < a.nq; el += step * U) {
        float4 xv[U], gv[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t eu = el + u * step;
            ok[u] = eu < a.nq;
            const int64_t e = g0 + (ok[u] ? eu : el);
            xv[u] = ld4(a.x, e);
            if (MODE == 1) gv[u] = ld4(a.dy, e);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!ok[u]) continue;
            const float xs[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};
            if (MODE == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { s0[j] += (double)xs[j]; s1[j] += (double)xs[j] * (double)xs[j]; }
            } else {
                const float gs[4] = {gv[u].x, gv[u].y, gv[u].z, gv[u].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float xm = xs[j] - mean[j];
                    const float z = xm * istd[j] * w[j] + b[j];
                    const float dz = (a.relu && !(z > 0.0f)) ? 0.0f : gs[j];
                    s0[j] += (double)dz; s1[j] += (double)dz * (double)xm;
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[j][tid] = s0[j]; red[4 + j][tid] = s1[j]; }
    __syncthreads();
    if (tid < 2 * a.C) {
        const int stat = tid / a.C, c = tid % a.C, k = stat * 4 + (c & 3);
        double s = 0.0;
        for (int t = c >> 2; t < 256; t += Q) s += red[k][t];
        a.partial[(((int64_t)grp * gridDim.x + blockIdx.x) * 2 + stat) * a.C + c] = s;
    }
}

// One block: sums the per-block partials of every group, then the per-channel results.
// MODE 0: mean, invstd per group; the running statistics take the groups IN ORDER, one momentum update each, as the reference's
// per-view calls do.  MODE 1: coef per group; grad_weight, grad_bias summed over the groups.
template <int MODE>
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const double *__restrict__ partial, int nblocks, int C, int G,
                                                          double n, float eps, float momentum,
                                                          const float *__restrict__ invstd_in,
                                                          float *__restrict__ o0, float *__restrict__ o1,
                                                          float *__restrict__ r0, float *__restrict__ r1,
                                                          long long *__restrict__ counter) {
    __shared__ double red[1024];
    const int tid = threadIdx.x, u = tid % (2 * C), part = tid / (2 * C), nparts = 1024 / (2 * C);
    float gw = 0.0f, gb = 0.0f;
    for (int grp = 0; grp < G; ++grp) {
        const double *pg = partial + (int64_t)grp * nblocks * 2 * C;
        double s = 0.0;
        int g = part;
        for (; g + 7 * nparts < nblocks; g += 8 * nparts) {   // 8 loads in flight: a lone block is latency-bound
            double v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = pg[(int64_t)(g + k * nparts) * 2 * C + u];
            s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
        for (; g < nblocks; g += nparts) s += pg[(int64_t)g * 2 * C + u];
        __syncthreads();      // (the previous group's sums have been read)
        red[tid] = s;
        __syncthreads();
        if (tid < C) {
            double a0 = 0.0, a1 = 0.0;
            for (int p = 0; p < nparts; ++p) { a0 += red[p * 2 * C + tid]; a1 += red[p * 2 * C + C + tid]; }
            if (MODE == 0) {
                const double mean = a0 / n;
                double var = a1 / n - mean * mean;
                if (var < 0.0) var = 0.0;
                o0[grp * C + tid] = (float)mean;
                o1[grp * C + tid] = 1.0f / sqrtf((float)var + eps);
                if (r0) r0[tid] = (1.0f - momentum) * r0[tid] + momentum * (float)mean;
                if (r1) r1[tid] = (1.0f - momentum) * r1[tid] + momentum * (float)(n > 1.0 ? var * n / (n - 1.0) : var);
            } else {
                const float is = invstd_in[grp * C + tid];
                gw += (float)a1 * is;                                      // grad_weight = sum(dz (x - mean)) invstd
                gb += (float)a0;                                           // grad_bias   = sum(dz)
                r0[(grp * 2 + 0) * C + tid] = (float)(a1 / n) * is * is;   // k1
                r0[(grp * 2 + 1) * C + tid] = (float)(a0 / n);             // k2
            }
        }
    }
    if (MODE == 1 && tid < C) { o0[tid] = gw; o1[tid] = gb; }
    if (MODE == 0 && tid == 0 && counter) *counter += G;
}

// MODE 0: y = relu((x - mean) invstd w + b) [+ skip].
// MODE 1: dx = (dz - k2 - (x - mean) k1) invstd w.
template <int MODE>
__global__ __launch_bounds__(256) void bn_apply_kernel(BnArgs a) {
    const int Q = a.C >> 2, tid = threadIdx.x, q = tid % Q, grp = blockIdx.y;
    const int64_t g0 = (int64_t)grp * a.nq;
    float mean[4], istd[4], w[4], b[4], k1[4], k2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = q * 4 + j;
        mean[j] = a.mean[grp * a.C + c]; istd[j] = a.invstd[grp * a.C + c]; w[j] = a.weight[c]; b[j] = a.bias[c];
        k1[j] = MODE == 1 ? a.coef[(grp * 2 + 0) * a.C + c] : 0.f;
        k2[j] = MODE == 1 ? a.coef[(grp * 2 + 1) * a.C + c] : 0.f;
    }
    const int64_t step = (int64_t)gridDim.x * 256;
    for (int64_t el = (int64_t)blockIdx.x * 256 + tid; el < a.nq; el += step) {
        const int64_t e = g0 + el;
        const float4 xv = ld4(a.x, e);
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
        float o[4];
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float z = (xs[j] - mean[j]) * istd[j] * w[j] + b[j];
                o[j] = (a.relu && z <= 0.0f) ? 0.0f : z;       // (a NaN stays: torch's relu)
            }
            if (a.skip) {
                const float4 sv = ld4(a.skip, e);
                o[0] += sv.x; o[1] += sv.y; o[2] += sv.z; o[3] += sv.w;
            }
        } else {
            const float4 gv = ld4(a.dy, e);
            const float gs[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float xm = xs[j] - mean[j];
                const float z = xm * istd[j] * w[j] + b[j];
                const float dz = (a.relu && !(z > 0.0f)) ? 0.0f : gs[j];
                o[j] = (dz - k2[j] - xm * k1[j]) * istd[j] * w[j];
            }
        }
        reinterpret_cast<float4 *>(a.out)[e] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

static bool bn_channels_ok(int C) { return C == 8 || C == 16 || C == 32 || C == 64; }
constexpr int kBnMaxGroups = 16;
static unsigned bn_grid(int64_t nq, int G) {
    const int64_t cap = kBnBlocks / G < 1 ? 1 : kBnBlocks / G;
    return (unsigned)min(cap, (nq + 255) / 256);
}

}  // namespace mvs

using namespace mvs;

// [kBnBlocks][2][C] fp64 partials (shared by the groups: kBnBlocks / G blocks each) + [kBnMaxGroups][2][C] backward coefficients
extern "C" size_t mvs_bn_train_workspace_bytes(int C) {
    return bn_channels_ok(C) ? ((size_t)kBnBlocks * 2 * C) * sizeof(double) + (size_t)kBnMaxGroups * 2 * C * sizeof(float) : 0;
}

extern "C" int mvs_bn_train_fwd_groups_f32(const float *x, const float *weight, const float *bias, const float *skip, int G,
                                           int64_t N, int C, float eps, float momentum, int relu, float *running_mean,
                                           float *running_var, long long *num_batches_tracked, float *save_mean,
                                           float *save_invstd, float *y, void *workspace, size_t workspace_bytes,
                                           void *stream) {
    if (!x || !weight || !bias || !save_mean || !save_invstd || !y || !workspace || N <= 0 || G < 1 || G > kBnMaxGroups) {
        set_error("mvs_bn_train_fwd_f32: invalid argument (1 <= groups <= %d)", kBnMaxGroups);
        return MVS_EINVAL;
    }
    if (!bn_channels_ok(C)) {
        set_error("mvs_bn_train_fwd_f32: C=%d (channels-last rows of 8, 16, 32 or 64)", C);
        return MVS_EUNSUPPORTED;
    }
    if (workspace_bytes < mvs_bn_train_workspace_bytes(C)) {
        set_error("mvs_bn_train_fwd_f32: workspace too small");
        return MVS_EINVAL;
    }
    hipStream_t st = as_stream(stream);
    BnArgs a = {};
    a.x = x; a.skip = skip; a.weight = weight; a.bias = bias; a.mean = save_mean; a.invstd = save_invstd;
    a.out = y; a.partial = static_cast<double *>(workspace); a.nq = N * (C / 4); a.C = C; a.relu = relu; a.G = G;
    const unsigned g = bn_grid(a.nq, G);
    hipLaunchKernelGGL((bn_reduce_kernel<0>), dim3(g, G), dim3(256), 0, st, a);
    hipLaunchKernelGGL((bn_finalize_kernel<0>), dim3(1), dim3(1024), 0, st, a.partial, (int)g, C, G, (double)N, eps,
                       momentum, (const float *)nullptr, save_mean, save_invstd, running_mean, running_var,
                       num_batches_tracked);
    hipLaunchKernelGGL((bn_apply_kernel<0>), dim3(4 * g, G), dim3(256), 0, st, a);
    return check_launch("mvs_bn_train_fwd_f32");
}

extern "C" int mvs_bn_train_fwd_f32(const float *x, const float *weight, const float *bias, const float *skip,
                                    int64_t N, int C, float eps, float momentum, int relu, float *running_mean,
                                    float *running_var, long long *num_batches_tracked, float *save_mean,
                                    float *save_invstd, float *y, void *workspace, size_t workspace_bytes,
                                    void *stream) {
    return mvs_bn_train_fwd_groups_f32(x, weight, bias, skip, 1, N, C, eps, momentum, relu, running_mean, running_var,
                                       num_batches_tracked, save_mean, save_invstd, y, workspace, workspace_bytes, stream);
}

extern "C" int mvs_bn_train_bwd_groups_f32(const float *grad_y, const float *x, const float *weight, const float *bias,
                                           const float *save_mean, const float *save_invstd, int G, int64_t N, int C, int relu,
                                           float *grad_x, float *grad_weight, float *grad_bias, void *workspace,
                                           size_t workspace_bytes, void *stream) {
    if (!grad_y || !x || !weight || !bias || !save_mean || !save_invstd || !grad_x || !grad_weight || !grad_bias ||
        !workspace || N <= 0 || G < 1 || G > kBnMaxGroups) {
        set_error("mvs_bn_train_bwd_f32: invalid argument (1 <= groups <= %d)", kBnMaxGroups);
        return MVS_EINVAL;
    }
    if (!bn_channels_ok(C)) {
        set_error("mvs_bn_train_bwd_f32: C=%d (channels-last rows of 8, 16, 32 or 64)", C);
        return MVS_EUNSUPPORTED;
    }
    if (workspace_bytes < mvs_bn_train_workspace_bytes(C)) {
        set_error("mvs_bn_train_bwd_f32: workspace too small");
        return MVS_EINVAL;
    }
    hipStream_t st = as_stream(stream);
    BnArgs a = {};
    a.x = x; a.dy = grad_y; a.weight = weight; a.bias = bias; a.mean = save_mean; a.invstd = save_invstd;
    a.out = grad_x; a.partial = static_cast<double *>(workspace); a.nq = N * (C / 4); a.C = C; a.relu = relu; a.G = G;
    float *coef = reinterpret_cast<float *>(a.partial + (size_t)kBnBlocks * 2 * C);
    a.coef = coef;
    const unsigned g = bn_grid(a.nq, G);
    hipLaunchKernelGGL((bn_reduce_kernel<1>), dim3(g, G), dim3(256), 0, st, a);
    hipLaunchKernelGGL((bn_finalize_kernel<1>), dim3(1), dim3(1024), 0, st, a.partial, (int)g, C, G, (double)N, 0.f, 0.f,
                       save_invstd, grad_weight, grad_bias, coef, (float *)nullptr, (long long *)nullptr);
    hipLaunchKernelGGL((bn_apply_kernel<1>), dim3(4 * g, G), dim3(256), 0, st, a);
    return check_launch("mvs_bn_train_bwd_f32");
}

extern "C" int mvs_bn_train_bwd_f32(const float *grad_y, const float *x, const float *weight, const float *bias,
                                    const float *save_mean, const float *save_invstd, int64_t N, int C, int relu,
                                    float *grad_x, float *grad_weight, float *grad_bias, void *workspace,
                                    size_t workspace_bytes, void *stream) {
    return mvs_bn_train_bwd_groups_f32(grad_y, x, weight, bias, save_mean, save_invstd, 1, N, C, relu, grad_x, grad_weight,
                                       grad_bias, workspace, workspace_bytes, stream);
}
