// Depth-hypothesis interval of a CVP-MVSNet refinement level (CVP-MVSNet/models/modules.py:147-219,
// calDepthHypo in test mode): for every reference pixel, the depth step that moves its projection
// into the first source view by `pixel_interval` along the epipolar line -- project at depth d and
// d + 1, take the direction, step one pixel, triangulate back (first unknown of a 2x2 system) --
// and the MEAN of |step| over the image.  The reference (and the torch mirror) runs ~25 fp64
// tensor ops over [3, H W] arrays per level; here one pass computes the per-pixel step in fp64
// registers and reduces it: block sums meet in one fp64 atomic.
#include "mvs_common.h"

namespace mvs {

struct CvpArgs {
    const float *depth;    // [H,W] upsampled depth of the previous level
    const double *mats;    // Kref^-1 (9), Eref^-1 (16), Ksrc (9), Esrc (16), A = Kref Rref (Ksrc Rsrc)^-1 (9)
    double *sum_abs;       // [1], zero on entry
    int H, W;
    double pixel_interval;
};

__device__ __forceinline__ void m3(const double *m, double a, double b, double c, double &x, double &y, double &z) {
    x = m[0] * a + m[1] * b + m[2] * c;
    y = m[3] * a + m[4] * b + m[5] * c;
    z = m[6] * a + m[7] * b + m[8] * c;
}
__device__ __forceinline__ void m43(const double *m, double a, double b, double c, double &x, double &y, double &z) {
    x = m[0] * a + m[1] * b + m[2] * c + m[3];
    y = m[4] * a + m[5] * b + m[6] * c + m[7];
    z = m[8] * a + m[9] * b + m[10] * c + m[11];
}

__global__ __launch_bounds__(256) void cvp_interval_kernel(CvpArgs a) {
    __shared__ double red[256];
    const int64_t n = (int64_t)a.H * a.W;
    const double *Kri = a.mats, *Eri = a.mats + 9, *Ks = a.mats + 25, *Es = a.mats + 34, *A = a.mats + 50;
    double acc = 0.0;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * 256) {
        const double px = (double)(idx % a.W), py = (double)(idx / a.W);
        const double d1 = (double)a.depth[idx];
        double x1[3], x2[2], z1 = 0.0;
#pragma unroll
        for (int k = 0; k < 2; ++k) {   // reference pixel at depth d1 (k = 0) and d1 + 1 -> source pixel
            const double d = d1 + (double)k;
            double cx, cy, cz, wx, wy, wz, sx, sy, sz, qx, qy, qz;
            m3(Kri, px * d, py * d, d, cx, cy, cz);
            m43(Eri, cx, cy, cz, wx, wy, wz);
            m43(Es, wx, wy, wz, sx, sy, sz);
            m3(Ks, sx, sy, sz, qx, qy, qz);
            if (k == 0) { x1[0] = qx / qz; x1[1] = qy / qz; x1[2] = qz / qz; z1 = qz; }
            else { x2[0] = qx / qz; x2[1] = qy / qz; }
        }
        const double theta = atan((x2[1] - x1[1]) / (x2[0] - x1[0]));
        const double x3x = x1[0] + cos(theta) * a.pixel_interval, x3y = x1[1] + sin(theta) * a.pixel_interval;
        double t1x, t1y, t1z, t2x, t2y, t2z;
        m3(A, x1[0], x1[1], x1[2], t1x, t1y, t1z);
        t1x *= z1; t1y *= z1; t1z *= z1;
        m3(A, x3x, x3y, x1[2], t2x, t2y, t2z);
        const double det = py * t2z - t2y * 1.0;
        const double ans0 = (t1y * t2z - t2y * t1z) / det;
        acc += fabs(ans0);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) unsafeAtomicAdd(a.sum_abs, red[0]);
}

}  // namespace mvs

using namespace mvs;

extern "C" int mvs_cvp_interval_sum_f64(const float *depth, const double *mats, int H, int W, double pixel_interval,
                                        double *sum_abs, void *stream) {
    if (!depth || !mats || !sum_abs || H <= 0 || W <= 0) {
        set_error("mvs_cvp_interval_sum_f64: invalid argument");
        return MVS_EINVAL;
    }
    hipStream_t st = as_stream(stream);
    if (launch_zero_words(sum_abs, 2, st) != MVS_OK) return MVS_ELAUNCH;
    CvpArgs a{depth, mats, sum_abs, H, W, pixel_interval};
    const int64_t n = (int64_t)H * W;
    const int64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(cvp_interval_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, st, a);
    return check_launch("mvs_cvp_interval_sum_f64");
}
