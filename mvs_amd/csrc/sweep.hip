// Plane-sweep kernels: K1 (homo_warping) and fused K1+K2 (warp + variance).
//
// Replaces MVSNet/models/module.py:46-87 and mvsnet.py:152-170.  The per-view
// warped volumes [B,C,D,H,W] of the reference are never written: each thread
// (group) transforms its voxel through every source view's homography,
// bilinear-samples the source feature maps and keeps the running sum / sum of
// squares in registers; only the variance is stored.
//
// Compiled with -ffp-contract=off: the coordinate arithmetic mirrors the
// reference op for op (see mvs_common.h) and is bit-exact with it.
#include "mvs_common.h"

namespace mvs {

constexpr int kMaxSrcViews = 8;

__device__ __forceinline__ float4 sel4(bool keep, float4 v) {
    return make_float4(keep ? v.x : 0.f, keep ? v.y : 0.f, keep ? v.z : 0.f, keep ? v.w : 0.f);
}
// value at idx if idx >= 0 else 0, without a branch around the load
__device__ __forceinline__ float ldz(const float *__restrict__ p, int idx) {
    float v = p[max(idx, 0)];
    return idx >= 0 ? v : 0.0f;
}

struct SweepParams {
    int B, C, D, H, W, V;     // V = total views (ref + sources)
    int depth_mode;           // 0: [B,D]   1: [B,D,H,W]
    int align_corners;
    int alias_quirk;
    float half_w, half_h;     // (W-1)/2, (H-1)/2
    float unn_w, unn_h;       // un-normalisation scale
    float fV;
};

__device__ __forceinline__ float depth_at(const float *__restrict__ depth, const SweepParams &p,
                                          int b, int d, int64_t pix) {
    return p.depth_mode == 0 ? depth[(int64_t)b * p.D + d]
                             : depth[((int64_t)b * p.D + d) * ((int64_t)p.H * p.W) + pix];
}

// ---------------------------------------------------------------------
// K1: warp, planar.  One thread per (b, d, y, x); loops over channels.
__global__ __launch_bounds__(256) void warp_fwd_planar_kernel(
    const float *__restrict__ src, const float *__restrict__ rt, const float *__restrict__ depth,
    SweepParams p, float *__restrict__ out) {
    const int64_t plane = (int64_t)p.H * p.W;
    const int64_t total = (int64_t)p.B * p.D * plane;
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    int64_t pix = idx % plane;
    int d = (int)((idx / plane) % p.D);
    int b = (int)(idx / (plane * p.D));
    int x = (int)(pix % p.W), y = (int)(pix / p.W);
    const float *r = rt + (int64_t)b * 12;
    float rx, ry, rz, ix, iy;
    sweep_ray(r, (float)x, (float)y, rx, ry, rz);
    sweep_coord(r, rx, ry, rz, depth_at(depth, p, b, d, pix), p.half_w, p.half_h, p.unn_w,
                p.unn_h, p.align_corners, ix, iy);
    Taps t = make_taps(ix, iy, p.H, p.W);
    const int o00 = t.y0 * p.W + t.x0, o01 = t.y0 * p.W + t.x1;
    const int o10 = t.y1 * p.W + t.x0, o11 = t.y1 * p.W + t.x1;
    const bool m00 = t.x0ok && t.y0ok, m01 = t.x1ok && t.y0ok;
    const bool m10 = t.x0ok && t.y1ok, m11 = t.x1ok && t.y1ok;
    const float *sp = src + (int64_t)b * p.C * plane;
    float *op = out + (((int64_t)b * p.C) * p.D + d) * plane + pix;
    for (int c = 0; c < p.C; ++c) {
        const float *pl = sp + (int64_t)c * plane;
        float v00 = pl[o00]; v00 = m00 ? v00 : 0.0f;
        float v01 = pl[o01]; v01 = m01 ? v01 : 0.0f;
        float v10 = pl[o10]; v10 = m10 ? v10 : 0.0f;
        float v11 = pl[o11]; v11 = m11 ? v11 : 0.0f;
        op[(int64_t)c * p.D * plane] = blend(t, v00, v01, v10, v11);
    }
}

// K1 backward: scatter-add grad_out * tap weight into grad_src (module.py:83-84
// differentiated w.r.t. src_fea only; the grid is built under no_grad).
__global__ __launch_bounds__(256) void warp_bwd_planar_kernel(
    const float *__restrict__ gout, const float *__restrict__ rt,
    const float *__restrict__ depth, SweepParams p, float *__restrict__ gsrc) {
    const int64_t plane = (int64_t)p.H * p.W;
    const int64_t total = (int64_t)p.B * p.D * plane;
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    int64_t pix = idx % plane;
    int d = (int)((idx / plane) % p.D);
    int b = (int)(idx / (plane * p.D));
    int x = (int)(pix % p.W), y = (int)(pix / p.W);
    const float *r = rt + (int64_t)b * 12;
    float rx, ry, rz, ix, iy;
    sweep_ray(r, (float)x, (float)y, rx, ry, rz);
    sweep_coord(r, rx, ry, rz, depth_at(depth, p, b, d, pix), p.half_w, p.half_h, p.unn_w,
                p.unn_h, p.align_corners, ix, iy);
    Taps t = make_taps(ix, iy, p.H, p.W);
    const int o00 = t.y0 * p.W + t.x0, o01 = t.y0 * p.W + t.x1;
    const int o10 = t.y1 * p.W + t.x0, o11 = t.y1 * p.W + t.x1;
    const bool m00 = t.x0ok && t.y0ok, m01 = t.x1ok && t.y0ok;
    const bool m10 = t.x0ok && t.y1ok, m11 = t.x1ok && t.y1ok;
    float *gp = gsrc + (int64_t)b * p.C * plane;
    const float *go = gout + (((int64_t)b * p.C) * p.D + d) * plane + pix;
    for (int c = 0; c < p.C; ++c) {
        float g = go[(int64_t)c * p.D * plane];
        float *pl = gp + (int64_t)c * plane;
        if (m00) atomicAdd(pl + o00, g * t.nw);
        if (m01) atomicAdd(pl + o01, g * t.ne);
        if (m10) atomicAdd(pl + o10, g * t.sw);
        if (m11) atomicAdd(pl + o11, g * t.se);
    }
}

// ---------------------------------------------------------------------
// K1+K2 fused, planar: features [B,C,H,W] -> variance [B,C,D,H,W].
// One thread per voxel; tap set of every source view lives in registers,
// channels are the inner loop so S and Q accumulate in the reference's view
// order (mvsnet.py:156-166).
template <int NV>
__global__ __launch_bounds__(256) void variance_fwd_planar_kernel(
    const float *__restrict__ ref, const float *__restrict__ srcs, const float *__restrict__ rt,
    const float *__restrict__ depth, SweepParams p, float *__restrict__ out) {
    const int64_t plane = (int64_t)p.H * p.W;
    const int64_t total = (int64_t)p.B * p.D * plane;
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    int64_t pix = idx % plane;
    int d = (int)((idx / plane) % p.D);
    int b = (int)(idx / (plane * p.D));
    int x = (int)(pix % p.W), y = (int)(pix / p.W);
    const float dv = depth_at(depth, p, b, d, pix);

    float wnw[NV], wne[NV], wsw[NV], wse[NV];
    int o00[NV], o01[NV], o10[NV], o11[NV];  // < 0 : tap outside the image
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const float *r = rt + ((int64_t)v * p.B + b) * 12;
        float rx, ry, rz, ix, iy;
        sweep_ray(r, (float)x, (float)y, rx, ry, rz);
        sweep_coord(r, rx, ry, rz, dv, p.half_w, p.half_h, p.unn_w, p.unn_h, p.align_corners, ix,
                    iy);
        Taps t = make_taps(ix, iy, p.H, p.W);
        wnw[v] = t.nw; wne[v] = t.ne; wsw[v] = t.sw; wse[v] = t.se;
        o00[v] = (t.x0ok && t.y0ok) ? t.y0 * p.W + t.x0 : -1;
        o01[v] = (t.x1ok && t.y0ok) ? t.y0 * p.W + t.x1 : -1;
        o10[v] = (t.x0ok && t.y1ok) ? t.y1 * p.W + t.x0 : -1;
        o11[v] = (t.x1ok && t.y1ok) ? t.y1 * p.W + t.x1 : -1;
    }
    const int64_t view_stride = (int64_t)p.B * p.C * plane;
    const float *rp = ref + (int64_t)b * p.C * plane + pix;
    const float *sp = srcs + (int64_t)b * p.C * plane;
    float *op = out + (((int64_t)b * p.C) * p.D + d) * plane + pix;
    for (int c = 0; c < p.C; ++c) {
        float r = rp[(int64_t)c * plane];
        float q = r * r;
        float s = p.alias_quirk ? q : r;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const float *pl = sp + (int64_t)v * view_stride + (int64_t)c * plane;
            float v00 = ldz(pl, o00[v]);
            float v01 = ldz(pl, o01[v]);
            float v10 = ldz(pl, o10[v]);
            float v11 = ldz(pl, o11[v]);
            float w = __fmaf_rn(v11, wse[v],
                                __fmaf_rn(v10, wsw[v], __fmaf_rn(v01, wne[v], v00 * wnw[v])));
            s = s + w;
            q = q + w * w;
        }
        float sm = s / p.fV;
        op[(int64_t)c * p.D * plane] = q / p.fV - sm * sm;
    }
}

// ---------------------------------------------------------------------
// K1+K2 fused, channels-last: features [B,H,W,C] -> variance [B,D,H,W,C]
// (or the 8-channel-blocked [B,D,H,C/8,W,8] the conv0 kernel streams).
//
// A wave owns 64 consecutive voxels of one batch item's flattened (d,y,x) space.
// Phase 1: lane l evaluates the homography + bilinear tap set of voxel l for every
// source view (exact arithmetic, once per voxel) and parks it in a wave-private
// LDS record: 4 tap weights (already zero for taps outside the image, NaN if
// the coordinate is not finite -- the reference's 0*NaN) and 4 clamped byte
// offsets.  Phase 2: the wave walks its voxels 64/CQ at a time with lane =
// (voxel, channel-quad): the CQ lanes of a voxel read one whole 16*CQ-byte texel
// per tap (a full 128-B line at C=32) through an SGPR base + 32-bit offset, and
// the wave stores 1 KiB of contiguous variance per step.  No masks, no 64-bit
// integer arithmetic and no IEEE division in the inner loop.

// n / d for n < 2^32 by multiply-high (Granlund-Montgomery); host-built.
struct FastDiv {
    uint32_t m, s1, s2, d;
};
static FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    f.d = d;
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;
    f.m = (uint32_t)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
    f.s1 = l < 1 ? l : 1;
    f.s2 = l == 0 ? 0 : l - 1;
    return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv &f) {
    const uint32_t t = __umulhi(f.m, n);
    return (t + ((n - t) >> f.s1)) >> f.s2;
}

// x / V, correctly rounded, for the small integer V = number of views:
// q = RN(x*(1/V)); r = x - q*V exactly (one FMA); q' = RN(q + r*(1/V)).
// Equal to IEEE x / V for every finite x outside the subnormal-result range
// (checked exhaustively on the GPU: mvs_selftest_div_by_views_f32); tiny |x| take
// the true division so the result is the reference's in every case.
__device__ __forceinline__ float div_views_fast(float x, float fV, float rV) {
    const float q = x * rV;
    const float r = __fmaf_rn(-q, fV, x);
    return __fmaf_rn(r, rV, q);
}
__device__ __forceinline__ bool div_views_tiny(float x) { return fabsf(x) < 1e-30f; }
__device__ __forceinline__ float div_views(float x, float fV, float rV) {
    return div_views_tiny(x) ? x / fV : div_views_fast(x, fV, rV);
}

template <int CQ, int NV>
__global__ __launch_bounds__(256) void variance_fwd_cl_kernel(
    const float *__restrict__ ref, const float *__restrict__ srcs, const float *__restrict__ rt,
    const float *__restrict__ depth, SweepParams p, FastDiv fd_plane, FastDiv fd_w,
    float *__restrict__ out, int out_c8) {
    constexpr int C = CQ * 4;
    constexpr int VPS = 64 / CQ;  // voxels per step
    __shared__ float4 s_w[4][NV][64];  // tap weights nw, ne, sw, se (pre-masked)
    __shared__ uint4 s_o[4][NV][64];   // byte offset of each tap's texel (clamped)

    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const uint32_t plane = (uint32_t)(p.H * p.W);
    const uint32_t total = (uint32_t)p.D * plane;          // voxels of one batch item
    const uint32_t wave_base = (blockIdx.x * 4u + wv) * 64u;

    {   // phase 1: lane = voxel
        const uint32_t idx = min(wave_base + lane, total - 1);
        const uint32_t d = fdiv(idx, fd_plane);
        const uint32_t pix = idx - d * plane;
        const uint32_t y = fdiv(pix, fd_w);
        const uint32_t x = pix - y * (uint32_t)p.W;
        const float dv = p.depth_mode == 0 ? depth[(int64_t)b * p.D + d]
                                           : depth[(int64_t)b * total + idx];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const float *r = rt + ((int64_t)v * p.B + b) * 12;
            float rx, ry, rz, ix, iy;
            sweep_ray(r, (float)x, (float)y, rx, ry, rz);
            sweep_coord(r, rx, ry, rz, dv, p.half_w, p.half_h, p.unn_w, p.unn_h,
                        p.align_corners, ix, iy);
            Taps t = make_taps(ix, iy, p.H, p.W);
            // a tap outside the image contributes value 0 in the reference; with
            // finite features that equals weight 0.  Non-finite coordinates give
            // NaN weights there (0 * NaN) -- keep them.
            const bool fin = (fabsf(ix) <= 3.0e38f) && (fabsf(iy) <= 3.0e38f);
            const float dead = fin ? 0.0f : __int_as_float(0x7fc00000);
            const bool m00 = t.x0ok && t.y0ok, m01 = t.x1ok && t.y0ok;
            const bool m10 = t.x0ok && t.y1ok, m11 = t.x1ok && t.y1ok;
            s_w[wv][v][lane] = make_float4(m00 ? t.nw : dead, m01 ? t.ne : dead,
                                           m10 ? t.sw : dead, m11 ? t.se : dead);
            s_o[wv][v][lane] = make_uint4((uint32_t)(t.y0 * p.W + t.x0) * (C * 4u),
                                          (uint32_t)(t.y0 * p.W + t.x1) * (C * 4u),
                                          (uint32_t)(t.y1 * p.W + t.x0) * (C * 4u),
                                          (uint32_t)(t.y1 * p.W + t.x1) * (C * 4u));
        }
    }
    __syncthreads();

    const uint32_t qoff = (uint32_t)(lane % CQ) * 16u;   // byte offset of this lane's quad
    const int vsub = lane / CQ;
    const size_t fea_bytes = (size_t)plane * C * 4;       // one feature map of one batch item
    const char *ref_b = reinterpret_cast<const char *>(ref) + (size_t)b * fea_bytes;
    const float rV = 1.0f / p.fV;
#pragma unroll 1
    for (int step = 0; step < CQ; ++step) {
        const int j = step * VPS + vsub;  // voxel slot within the wave
        const uint32_t idx = wave_base + j;
        if (idx >= total) continue;
        const uint32_t d = fdiv(idx, fd_plane);
        const uint32_t pix = idx - d * plane;
        const float4 r = *reinterpret_cast<const float4 *>(ref_b + (pix * (C * 4u) + qoff));
        float4 q = make_float4(r.x * r.x, r.y * r.y, r.z * r.z, r.w * r.w);
        float4 s = p.alias_quirk ? q : r;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const float4 w = s_w[wv][v][j];
            const uint4 o = s_o[wv][v][j];
            // wave-uniform base (SGPRs) + 32-bit per-lane offset
            const char *sv = reinterpret_cast<const char *>(srcs) +
                             ((size_t)v * p.B + b) * fea_bytes;
            const float4 a = *reinterpret_cast<const float4 *>(sv + (o.x + qoff));
            const float4 bq = *reinterpret_cast<const float4 *>(sv + (o.y + qoff));
            const float4 c = *reinterpret_cast<const float4 *>(sv + (o.z + qoff));
            const float4 e = *reinterpret_cast<const float4 *>(sv + (o.w + qoff));
            const float t0 = __fmaf_rn(e.x, w.w, __fmaf_rn(c.x, w.z, __fmaf_rn(bq.x, w.y, a.x * w.x)));
            const float t1 = __fmaf_rn(e.y, w.w, __fmaf_rn(c.y, w.z, __fmaf_rn(bq.y, w.y, a.y * w.x)));
            const float t2 = __fmaf_rn(e.z, w.w, __fmaf_rn(c.z, w.z, __fmaf_rn(bq.z, w.y, a.z * w.x)));
            const float t3 = __fmaf_rn(e.w, w.w, __fmaf_rn(c.w, w.z, __fmaf_rn(bq.w, w.y, a.w * w.x)));
            s.x = s.x + t0; s.y = s.y + t1; s.z = s.z + t2; s.w = s.w + t3;
            q.x = q.x + t0 * t0; q.y = q.y + t1 * t1; q.z = q.z + t2 * t2; q.w = q.w + t3 * t3;
        }
        float4 o4;
        { const float m = div_views_fast(s.x, p.fV, rV); o4.x = div_views_fast(q.x, p.fV, rV) - m * m; }
        { const float m = div_views_fast(s.y, p.fV, rV); o4.y = div_views_fast(q.y, p.fV, rV) - m * m; }
        { const float m = div_views_fast(s.z, p.fV, rV); o4.z = div_views_fast(q.z, p.fV, rV) - m * m; }
        { const float m = div_views_fast(s.w, p.fV, rV); o4.w = div_views_fast(q.w, p.fV, rV) - m * m; }
        const bool tiny = div_views_tiny(s.x) || div_views_tiny(s.y) || div_views_tiny(s.z) ||
                          div_views_tiny(s.w) || div_views_tiny(q.x) || div_views_tiny(q.y) ||
                          div_views_tiny(q.z) || div_views_tiny(q.w);
        if (__any(tiny)) {   // wave-uniform, practically never taken: true IEEE division
            { const float m = s.x / p.fV; o4.x = q.x / p.fV - m * m; }
            { const float m = s.y / p.fV; o4.y = q.y / p.fV - m * m; }
            { const float m = s.z / p.fV; o4.z = q.z / p.fV - m * m; }
            { const float m = s.w / p.fV; o4.w = q.w / p.fV - m * m; }
        }
        float *ob = out + (size_t)b * total * C;
        if (out_c8) {   // [B,D,H,C/8,W,8]: row (d,y), 8-channel block, then x
            const uint32_t y = fdiv(pix, fd_w);
            const uint32_t x = pix - y * (uint32_t)p.W;
            const uint32_t row = d * (uint32_t)p.H + y;
            *reinterpret_cast<float4 *>(
                ob + ((size_t)(row * (C / 8) + (qoff >> 5)) * p.W + x) * 8 + ((qoff >> 2) & 7)) = o4;
        } else {
            *reinterpret_cast<float4 *>(ob + (size_t)idx * C + (qoff >> 2)) = o4;
        }
    }
}

// Exhaustive check of div_views against IEEE division: every float bit pattern.
__global__ __launch_bounds__(256) void div_selftest_kernel(float fV, unsigned long long *mismatch) {
    const float rV = 1.0f / fV;
    unsigned long long bad = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32);
         i += (uint64_t)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((uint32_t)i);
        const float a = div_views(x, fV, rV), c = x / fV;
        if (__float_as_uint(a) != __float_as_uint(c) && !(a != a && c != c)) ++bad;
    }
    if (bad) atomicAdd(mismatch, bad);
}

// ---------------------------------------------------------------------
// Backward of the fused variance (planar layouts).  var = Q/V - (S/V)^2 with
// S = ref + sum w_v, Q = ref^2 + sum w_v^2, so d var/d w_v = 2 w_v/V - 2 S/V^2
// (and the same expression with ref for the reference view, whose gradient is
// the sum over D of the broadcast at mvsnet.py:152).  Taps are recomputed.
template <int NV>
__global__ __launch_bounds__(256) void variance_bwd_planar_kernel(
    const float *__restrict__ gvar, const float *__restrict__ ref, const float *__restrict__ srcs,
    const float *__restrict__ rt, const float *__restrict__ depth, SweepParams p,
    float *__restrict__ gref, float *__restrict__ gsrcs) {
    const int64_t plane = (int64_t)p.H * p.W;
    const int64_t total = (int64_t)p.B * p.D * plane;
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    int64_t pix = idx % plane;
    int d = (int)((idx / plane) % p.D);
    int b = (int)(idx / (plane * p.D));
    int x = (int)(pix % p.W), y = (int)(pix / p.W);
    const float dv = depth_at(depth, p, b, d, pix);
    float wnw[NV], wne[NV], wsw[NV], wse[NV];
    int o00[NV], o01[NV], o10[NV], o11[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const float *r = rt + ((int64_t)v * p.B + b) * 12;
        float rx, ry, rz, ix, iy;
        sweep_ray(r, (float)x, (float)y, rx, ry, rz);
        sweep_coord(r, rx, ry, rz, dv, p.half_w, p.half_h, p.unn_w, p.unn_h, p.align_corners, ix,
                    iy);
        Taps t = make_taps(ix, iy, p.H, p.W);
        wnw[v] = t.nw; wne[v] = t.ne; wsw[v] = t.sw; wse[v] = t.se;
        o00[v] = (t.x0ok && t.y0ok) ? t.y0 * p.W + t.x0 : -1;
        o01[v] = (t.x1ok && t.y0ok) ? t.y0 * p.W + t.x1 : -1;
        o10[v] = (t.x0ok && t.y1ok) ? t.y1 * p.W + t.x0 : -1;
        o11[v] = (t.x1ok && t.y1ok) ? t.y1 * p.W + t.x1 : -1;
    }
    const int64_t view_stride = (int64_t)p.B * p.C * plane;
    const float inv_v = 1.0f / p.fV;
    for (int c = 0; c < p.C; ++c) {
        const int64_t fo = ((int64_t)b * p.C + c) * plane;
        const float g = gvar[((((int64_t)b * p.C + c) * p.D + d)) * plane + pix];
        const float r = ref[fo + pix];
        float s = r;
        float w[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const float *pl = srcs + (int64_t)v * view_stride + fo;
            float v00 = ldz(pl, o00[v]);
            float v01 = ldz(pl, o01[v]);
            float v10 = ldz(pl, o10[v]);
            float v11 = ldz(pl, o11[v]);
            w[v] = __fmaf_rn(v11, wse[v],
                             __fmaf_rn(v10, wsw[v], __fmaf_rn(v01, wne[v], v00 * wnw[v])));
            s += w[v];
        }
        const float k = 2.0f * s * inv_v * inv_v;
        atomicAdd(gref + fo + pix, g * (2.0f * r * inv_v - k));
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            float gw = g * (2.0f * w[v] * inv_v - k);
            float *pl = gsrcs + (int64_t)v * view_stride + fo;
            if (o00[v] >= 0) atomicAdd(pl + o00[v], gw * wnw[v]);
            if (o01[v] >= 0) atomicAdd(pl + o01[v], gw * wne[v]);
            if (o10[v] >= 0) atomicAdd(pl + o10[v], gw * wsw[v]);
            if (o11[v] >= 0) atomicAdd(pl + o11[v], gw * wse[v]);
        }
    }
}

// ---------------------------------------------------------------------
static SweepParams make_params(int B, int V, int C, int D, int H, int W, int depth_mode,
                               int align_corners, int alias_quirk) {
    SweepParams p;
    p.B = B; p.C = C; p.D = D; p.H = H; p.W = W; p.V = V;
    p.depth_mode = depth_mode;
    p.align_corners = align_corners;
    p.alias_quirk = alias_quirk;
    p.half_w = (float)((W - 1) / 2.0);
    p.half_h = (float)((H - 1) / 2.0);
    p.unn_w = align_corners ? (float)((W - 1) / 2.0) : (float)(W / 2.0);
    p.unn_h = align_corners ? (float)((H - 1) / 2.0) : (float)(H / 2.0);
    p.fV = (float)V;
    return p;
}

static bool grid_for(int64_t total, int per_block, unsigned &grid) {
    int64_t g = (total + per_block - 1) / per_block;
    if (g <= 0 || g > 0x7fffffffLL) return false;
    grid = (unsigned)g;
    return true;
}

template <int CQ>
static int launch_variance_cl(int NV, const float *ref, const float *srcs, const float *rt,
                              const float *depth, const SweepParams &p, float *out, int out_c8,
                              hipStream_t st) {
    const int64_t per_item = (int64_t)p.D * p.H * p.W;
    if (per_item >= (1ll << 31) || (int64_t)p.H * p.W * CQ * 16 >= (1ll << 32) || p.B > 65535) {
        set_error("mvs_costvol_variance_fwd_f32: volume too large for 32-bit indexing");
        return MVS_EINVAL;
    }
    const dim3 grid((unsigned)((per_item + 255) / 256), (unsigned)p.B);
    const FastDiv fdp = make_fastdiv((uint32_t)(p.H * p.W)), fdw = make_fastdiv((uint32_t)p.W);
#define MVS_CL_CASE(n)                                                                         \
    case n:                                                                                    \
        hipLaunchKernelGGL((variance_fwd_cl_kernel<CQ, n>), grid, dim3(256), 0, st, ref, srcs, \
                           rt, depth, p, fdp, fdw, out, out_c8);                               \
        return MVS_OK;
    switch (NV) {
        MVS_CL_CASE(1) MVS_CL_CASE(2) MVS_CL_CASE(3) MVS_CL_CASE(4) MVS_CL_CASE(5) MVS_CL_CASE(6)
        MVS_CL_CASE(7) MVS_CL_CASE(8)
    }
#undef MVS_CL_CASE
    return MVS_EUNSUPPORTED;
}

}  // namespace mvs

using namespace mvs;

extern "C" int mvs_warp_fwd_f32(const float *src_fea, const float *rot_trans,
                                const float *depth_values, int depth_mode, int B, int C, int D,
                                int H, int W, int align_corners, float *out, void *stream) {
    if (!src_fea || !rot_trans || !depth_values || !out || B <= 0 || C <= 0 || D <= 0 || H <= 1 ||
        W <= 1 || depth_mode < 0 || depth_mode > 1) {
        set_error("mvs_warp_fwd_f32: invalid argument");
        return MVS_EINVAL;
    }
    SweepParams p = make_params(B, 2, C, D, H, W, depth_mode, align_corners, 0);
    unsigned grid;
    if (!grid_for((int64_t)B * D * H * W, 256, grid)) {
        set_error("mvs_warp_fwd_f32: problem too large");
        return MVS_EINVAL;
    }
    hipLaunchKernelGGL(warp_fwd_planar_kernel, dim3(grid), dim3(256), 0, as_stream(stream),
                       src_fea, rot_trans, depth_values, p, out);
    return check_launch("mvs_warp_fwd_f32");
}

extern "C" int mvs_warp_bwd_f32(const float *grad_out, const float *rot_trans,
                                const float *depth_values, int depth_mode, int B, int C, int D,
                                int H, int W, int align_corners, float *grad_src, void *stream) {
    if (!grad_out || !rot_trans || !depth_values || !grad_src || B <= 0 || C <= 0 || D <= 0 ||
        H <= 1 || W <= 1 || depth_mode < 0 || depth_mode > 1) {
        set_error("mvs_warp_bwd_f32: invalid argument");
        return MVS_EINVAL;
    }
    SweepParams p = make_params(B, 2, C, D, H, W, depth_mode, align_corners, 0);
    unsigned grid;
    if (!grid_for((int64_t)B * D * H * W, 256, grid)) return MVS_EINVAL;
    hipStream_t st = as_stream(stream);
    if (hipMemsetAsync(grad_src, 0, sizeof(float) * (size_t)B * C * H * W, st) != hipSuccess)
        return check_launch("mvs_warp_bwd_f32 memset");
    hipLaunchKernelGGL(warp_bwd_planar_kernel, dim3(grid), dim3(256), 0, st, grad_out, rot_trans,
                       depth_values, p, grad_src);
    return check_launch("mvs_warp_bwd_f32");
}

extern "C" int mvs_costvol_variance_fwd_f32(const float *ref_fea, const float *src_feas,
                                            const float *rot_trans, const float *depth_values,
                                            int depth_mode, int B, int V, int C, int D, int H,
                                            int W, int align_corners, int alias_quirk,
                                            int fea_layout, int out_layout, float *out_var,
                                            void *stream) {
    const int NV = V - 1;
    if (!ref_fea || !src_feas || !rot_trans || !depth_values || !out_var || B <= 0 || D <= 0 ||
        H <= 1 || W <= 1 || depth_mode < 0 || depth_mode > 1) {
        set_error("mvs_costvol_variance_fwd_f32: invalid argument");
        return MVS_EINVAL;
    }
    if (NV < 1 || NV > kMaxSrcViews) {
        set_error("mvs_costvol_variance_fwd_f32: V=%d unsupported (2..%d views)", V,
                  kMaxSrcViews + 1);
        return MVS_EUNSUPPORTED;
    }
    const int out_c8 = out_layout == MVS_LAYOUT_C8;
    if (fea_layout != out_layout && !(fea_layout == MVS_LAYOUT_NHWC && out_c8)) {
        set_error("mvs_costvol_variance_fwd_f32: out_layout must equal fea_layout (or be C8 with NHWC features)");
        return MVS_EUNSUPPORTED;
    }
    SweepParams p = make_params(B, V, C, D, H, W, depth_mode, align_corners, alias_quirk);
    hipStream_t st = as_stream(stream);
    unsigned grid;
    if (!grid_for((int64_t)B * D * H * W, 256, grid)) {
        set_error("mvs_costvol_variance_fwd_f32: problem too large");
        return MVS_EINVAL;
    }
    if (fea_layout == MVS_LAYOUT_NCHW) {
        if (C <= 0) return MVS_EINVAL;
#define MVS_PL_CASE(n)                                                                            \
    case n:                                                                                       \
        hipLaunchKernelGGL((variance_fwd_planar_kernel<n>), dim3(grid), dim3(256), 0, st,         \
                           ref_fea, src_feas, rot_trans, depth_values, p, out_var);               \
        break;
        switch (NV) {
            MVS_PL_CASE(1) MVS_PL_CASE(2) MVS_PL_CASE(3) MVS_PL_CASE(4) MVS_PL_CASE(5)
            MVS_PL_CASE(6) MVS_PL_CASE(7) MVS_PL_CASE(8)
        }
#undef MVS_PL_CASE
        return check_launch("mvs_costvol_variance_fwd_f32(planar)");
    }
    if (fea_layout != MVS_LAYOUT_NHWC) return MVS_EINVAL;
    int rc;
    switch (C) {
        case 8: rc = launch_variance_cl<2>(NV, ref_fea, src_feas, rot_trans, depth_values, p, out_var, out_c8, st); break;
        case 16: rc = launch_variance_cl<4>(NV, ref_fea, src_feas, rot_trans, depth_values, p, out_var, out_c8, st); break;
        case 32: rc = launch_variance_cl<8>(NV, ref_fea, src_feas, rot_trans, depth_values, p, out_var, out_c8, st); break;
        case 64: rc = launch_variance_cl<16>(NV, ref_fea, src_feas, rot_trans, depth_values, p, out_var, out_c8, st); break;
        default:
            set_error("mvs_costvol_variance_fwd_f32: channels-last needs C in {8,16,32,64}, got %d", C);
            return MVS_EUNSUPPORTED;
    }
    if (rc != MVS_OK) return rc;
    return check_launch("mvs_costvol_variance_fwd_f32(channels-last)");
}

extern "C" int mvs_selftest_div_by_views_f32(int V, unsigned long long *mismatch_count,
                                             void *stream) {
    if (V < 1 || !mismatch_count) {
        set_error("mvs_selftest_div_by_views_f32: invalid argument");
        return MVS_EINVAL;
    }
    hipStream_t st = as_stream(stream);
    if (hipMemsetAsync(mismatch_count, 0, sizeof(unsigned long long), st) != hipSuccess)
        return check_launch("mvs_selftest_div_by_views_f32 memset");
    hipLaunchKernelGGL(div_selftest_kernel, dim3(4096), dim3(256), 0, st, (float)V, mismatch_count);
    return check_launch("mvs_selftest_div_by_views_f32");
}

extern "C" int mvs_costvol_variance_bwd_f32(const float *grad_var, const float *ref_fea,
                                            const float *src_feas, const float *rot_trans,
                                            const float *depth_values, int depth_mode, int B,
                                            int V, int C, int D, int H, int W, int align_corners,
                                            int fea_layout, int out_layout, float *grad_ref,
                                            float *grad_srcs, void *stream) {
    const int NV = V - 1;
    if (!grad_var || !ref_fea || !src_feas || !rot_trans || !depth_values || !grad_ref ||
        !grad_srcs || B <= 0 || C <= 0 || D <= 0 || H <= 1 || W <= 1) {
        set_error("mvs_costvol_variance_bwd_f32: invalid argument");
        return MVS_EINVAL;
    }
    if (NV < 1 || NV > kMaxSrcViews) return MVS_EUNSUPPORTED;
    if (fea_layout != MVS_LAYOUT_NCHW || out_layout != MVS_LAYOUT_NCHW) {
        set_error("mvs_costvol_variance_bwd_f32: only the planar layout is implemented");
        return MVS_EUNSUPPORTED;
    }
    SweepParams p = make_params(B, V, C, D, H, W, depth_mode, align_corners, 0);
    hipStream_t st = as_stream(stream);
    unsigned grid;
    if (!grid_for((int64_t)B * D * H * W, 256, grid)) return MVS_EINVAL;
    const size_t fbytes = sizeof(float) * (size_t)B * C * H * W;
    if (hipMemsetAsync(grad_ref, 0, fbytes, st) != hipSuccess ||
        hipMemsetAsync(grad_srcs, 0, fbytes * NV, st) != hipSuccess)
        return check_launch("mvs_costvol_variance_bwd_f32 memset");
#define MVS_BW_CASE(n)                                                                           \
    case n:                                                                                      \
        hipLaunchKernelGGL((variance_bwd_planar_kernel<n>), dim3(grid), dim3(256), 0, st,        \
                           grad_var, ref_fea, src_feas, rot_trans, depth_values, p, grad_ref,    \
                           grad_srcs);                                                           \
        break;
    switch (NV) {
        MVS_BW_CASE(1) MVS_BW_CASE(2) MVS_BW_CASE(3) MVS_BW_CASE(4) MVS_BW_CASE(5) MVS_BW_CASE(6)
        MVS_BW_CASE(7) MVS_BW_CASE(8)
    }
#undef MVS_BW_CASE
    return check_launch("mvs_costvol_variance_bwd_f32");
}
