// rot_trans of every source view against the reference view on the device (MVSNet/models/module.py:63-65:
// proj = src_proj @ inverse(ref_proj); rot = proj[:3,:3]; trans = proj[:3,3:4]) without a host hop and without a
// library call that synchronises -- usable inside a captured HIP graph (the training step of configs[4] as one graph
// launch).  One thread per (source view, batch item): 4x4 Gauss-Jordan with partial pivoting and the product in
// float64, rounded once to float32.  The eval default stays the reference's own float32 LAPACK inverse on the host
// (ops.HostRotTrans: its rounding moves the depth by up to 1.8e-4 mm); this is proj_where="device".
#include "mvs_common.h"

namespace mvs {

__global__ __launch_bounds__(64) void rot_trans_kernel(const float *__restrict__ proj, int B, int V, float *__restrict__ out) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= (V - 1) * B) return;
    const int v = i / B + 1, b = i % B;
    const float *ref = proj + ((int64_t)b * V) * 16, *src = proj + ((int64_t)b * V + v) * 16;
    double a[4][8];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) { a[r][c] = (double)ref[r * 4 + c]; a[r][4 + c] = r == c ? 1.0 : 0.0; }
#pragma unroll
    for (int col = 0; col < 4; ++col) {
        int piv = col;
        double best = fabs(a[col][col]);
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r > col && fabs(a[r][col]) > best) { best = fabs(a[r][col]); piv = r; }
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r == piv && piv != col) {
#pragma unroll
                for (int c = 0; c < 8; ++c) { const double t = a[col][c]; a[col][c] = a[r][c]; a[r][c] = t; }
            }
        const double inv = 1.0 / a[col][col];
#pragma unroll
        for (int c = 0; c < 8; ++c) a[col][c] *= inv;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r != col) {
                const double f = a[r][col];
#pragma unroll
                for (int c = 0; c < 8; ++c) a[r][c] -= f * a[col][c];
            }
    }
    float *o = out + ((int64_t)(v - 1) * B + b) * 12;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) s += (double)src[r * 4 + k] * a[k][4 + c];
            o[r * 4 + c] = (float)s;
        }
}

}  // namespace mvs

using namespace mvs;

extern "C" int mvs_rot_trans_f32(const float *proj_matrices, int B, int V, float *rot_trans, void *stream) {
    if (!proj_matrices || !rot_trans || B <= 0 || V < 2) {
        set_error("mvs_rot_trans_f32: invalid argument");
        return MVS_EINVAL;
    }
    const int n = (V - 1) * B;
    hipLaunchKernelGGL(rot_trans_kernel, dim3((n + 63) / 64), dim3(64), 0, as_stream(stream), proj_matrices, B, V, rot_trans);
    return check_launch("mvs_rot_trans_f32");
}
