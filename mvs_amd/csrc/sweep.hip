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
#ifdef MVS_TUNING
#include "../../include/mvs_hip_tuning.h"
#endif
#include "sweep_common.h"

#include <cstdlib>
#include <cstring>

namespace mvs {

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


template <int CQ, int NV>
__global__ __launch_bounds__(256) void variance_fwd_cl_kernel(
    const float *__restrict__ ref, const float *__restrict__ srcs, const float *__restrict__ rt,
    const float *__restrict__ depth, SweepParams p, FastDiv fd_plane, FastDiv fd_w,
    float *__restrict__ out, int out_c8, int ablate) {
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
            uint4 o = s_o[wv][v][j];
            if (ablate & 1) o = make_uint4(0u, 0u, 0u, 0u);   // tuning: every gather hits one line
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

// ---------------------------------------------------------------------
// K1+K2 fused with LDS-staged source tiles (the north-star form).
//
// The gather kernel above tops out on the vector-L1 path (64 B/clk/CU: 2 KiB of
// taps per voxel).  Here a block owns an 8x8-pixel tile of the reference view
// over 4 consecutive depth planes (256 voxels, lane = voxel).  For every source
// view the block computes, at run time, the bounding box of all texels its
// voxels' bilinear taps touch (an arbitrary homography: no geometry is assumed),
// copies that footprint into LDS 16 channels at a time by DMA, and samples from LDS
// (256 B/clk/CU).  A footprint that does not fit its LDS share (extreme baselines)
// falls back to global gathers for that view -- same arithmetic, slower.  Features
// come 16-channel blocked: [B,C/16,H,W,16], so a footprint row is one contiguous run
// in memory.  Arithmetic is identical to the other kernels (pre-masked weights, FMA
// order): results are bit-identical to them.
constexpr int kTileW = 8, kTileH = 8, kTileD = 4;   // square tile: compact footprints under rotation

// CORNER (shared depth planes only): the footprint box of a source view is taken from
// the projections of the block's 8 corner voxels instead of a reduction over all 256
// voxels.  Per depth plane the map pixel -> source is a homography, so (all Z > 0) the
// tile's image is the convex hull of its 4 corner images; along depth each coordinate
// is a Moebius function of d, hence monotone between the two extreme planes.  Every
// wave computes the same boxes from 8*NV lanes with shuffles only: no LDS atomics, two
// barriers fewer, and the staging loads go out before the per-voxel arithmetic.  A wave
// whose taps are not all inside the (1-texel padded) box -- never observed -- samples
// that view from global memory instead, so the result cannot depend on the argument.
//
// The footprints move HBM -> LDS by buffer-addressed DMA (buffer_load_dwordx4 ... lds):
// no staging registers (a register-staged form kept NV x 3 float4 per thread in
// flight: 2.68 vs 2.0 ms), no ds_write pass, and the copy for channel group g+1 is issued
// before group g's variances are formed and stored, so it lands under that work; the
// first group's copy lands under the per-voxel homography arithmetic.  LDS image of a view: 4 planes
// [channel quad k][texel][4 floats] -- a DMA instruction fills 64 consecutive texels of a
// plane, 16 neighbouring texels cover all 64 banks without padding, and the four
// quads of a tap sit at immediate offsets k * plane.
// texels per view: 48 KiB in total (3 workgroups per CU; the register budget allows no
// more), whole 64-lane DMA instructions where that costs little
__host__ __device__ constexpr int dma_cap(int nv, int nq = 4) {   // nq = channel quads per group
    const int raw = (48 * 1024) / (nv * 16 * nq);
    return raw >= 256 ? 256 : (raw >= 192 ? 192 : raw);
}

// taps from the planar LDS image: o?? = texel index of the tap, PL = plane stride (texels)
template <int PL, int NQ>
__device__ __forceinline__ void accumulate_taps_planar(const float *__restrict__ base, int o00,
                                                       int o01, int o10, int o11, float wnw,
                                                       float wne, float wsw, float wse,
                                                       float (&S)[4 * NQ], float (&Q)[4 * NQ]) {
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
        const float4 a = *reinterpret_cast<const float4 *>(base + (k * PL + o00) * 4);
        const float4 bq = *reinterpret_cast<const float4 *>(base + (k * PL + o01) * 4);
        const float4 c = *reinterpret_cast<const float4 *>(base + (k * PL + o10) * 4);
        const float4 e = *reinterpret_cast<const float4 *>(base + (k * PL + o11) * 4);
        const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {bq.x, bq.y, bq.z, bq.w};
        const float cv[4] = {c.x, c.y, c.z, c.w}, ev[4] = {e.x, e.y, e.z, e.w};
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            const float w = __fmaf_rn(ev[cc], wse, __fmaf_rn(cv[cc], wsw,
                                      __fmaf_rn(bv[cc], wne, av[cc] * wnw)));
            S[k * 4 + cc] = S[k * 4 + cc] + w;
            Q[k * 4 + cc] = Q[k * 4 + cc] + w * w;
        }
        if (k == 1) __builtin_amdgcn_sched_barrier(0);
    }
}

// NQ = 4: features [B,C/16,H,W,16]; NQ = 2: 8-channel maps [B,H,W,8] (the cascade's finest
// stage) -- the same layout with one group of two quads, waves 2 and 3 take every other
// DMA instruction instead of their own quad.
// LOOP: a fixed grid walks the nblk tiles (the form launched behind variance_choose_kernel: when the geometry asks for
// another kernel, 768 workgroups leave at once instead of tens of thousands being dispatched to find that out).
template <int NV, bool CORNER, int NQ = 4, bool LOOP = false>
__global__ __launch_bounds__(256, 2) void variance_fwd_dma_kernel(
    const float *__restrict__ ref16, const float *__restrict__ srcs16,
    const float *__restrict__ rt, const float *__restrict__ depth, SweepParams p,
    int tiles_x, int tiles_y, float *__restrict__ out, int out_c8, int ablate, const unsigned *__restrict__ sel, int nblk,
    unsigned *__restrict__ absmax) {
    // sel: word of the workspace header written by variance_choose_kernel (sweep_persist.hip); this kernel runs when it
    // reads 0 ("per-tile kernel"); NULL = unconditional.  absmax: NULL, or the absmax block that collects the largest |variance|
    if (sel && *sel != 0u) return;
    int vblk = blockIdx.x;
    float vmax = 0.0f;
    do {   // (one pass unless LOOP)
    constexpr int cap = dma_cap(NV, NQ);
    constexpr int NJ = (cap + 63) / 64;          // DMA instructions per plane
    constexpr int GC = 4 * NQ, NSH = 4 / NQ;     // channels per group; waves sharing a quad
    extern __shared__ __attribute__((aligned(16))) float lds[];   // NV * NQ * cap * 4 floats
    __shared__ int s_box[NV][4];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tx, ty, dc;
    {   // Block order: depth chunk fastest, and consecutive blocks on the same XCD (blockIdx
        // round-robins the 8 XCDs).  The footprints of one pixel tile move by a fraction of
        // a texel per depth plane, so the depth chunks of a tile re-read the same source lines
        // out of that XCD's L2; ordered by tile first, every depth chunk swept all source
        // maps (61 MB at config 2, > L2) again: 13x the HBM-side fetch.
        const int nwg = LOOP ? nblk : (int)gridDim.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = vblk & 7;
        int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vblk >> 3);
        const int ndc = nwg / (tiles_x * tiles_y);
        dc = bid % ndc; bid /= ndc;
        tx = bid % tiles_x;
        ty = bid / tiles_x;
    }
    const int b = blockIdx.y;
    const int px = tx * kTileW + (lane & (kTileW - 1)), py = ty * kTileH + lane / kTileW;
    const int d = dc * kTileD + wv;
    const bool live = px < p.W && py < p.H && d < p.D;
    const int cx = min(px, p.W - 1), cy = min(py, p.H - 1), cd = min(d, p.D - 1);
    const int plane = p.H * p.W;
    const int pix = cy * p.W + cx;
    const float dv = p.depth_mode == 0 ? depth[(int64_t)b * p.D + cd]
                                       : depth[((int64_t)b * p.D + cd) * plane + pix];
    const int ngroups = p.C / GC;
    const size_t grp_floats = (size_t)plane * GC;
    const unsigned lds_base = (unsigned)(uintptr_t)lds;
    const int dq = wv % NQ, dj = wv / NQ;        // this wave's quad and share of the DMA instructions

    int bx0[NV], by0[NV], bw[NV], bh[NV];
    bool staged[NV];
    // per-view source offsets (floats, inside one channel group) of this lane's texels
    int soff[NV][NJ];
    auto plan_dma = [&]() {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int n = bw[v] * bh[v];
            const unsigned inv = (65536u + bw[v] - 1) / bw[v];   // t / bw for t < 65536 / bw
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int t = min(j * 64 + lane, max(n - 1, 0));
                // 24-bit multiplies (full rate; v_mul_lo_u32 is quarter rate): t < 256, inv <= 65536,
                // rows and columns < 2^13 (H * W < 2^26 is checked at launch)
                const int ly = (int)(__umul24((unsigned)t, inv) >> 16), lx = t - __mul24(ly, bw[v]);
                soff[v][j] = (__mul24(by0[v] + ly, p.W) + (bx0[v] + lx)) * GC;
            }
        }
    };
    // wave w copies channel quad w of every view (4 waves, 4 quads)
    auto issue_dma = [&](int g) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            if (!staged[v] || (ablate & 1)) continue;
            // buffer-addressed: descriptor = this view's 16-channel group (SGPRs), per-lane
            // offsets fixed for the whole block -> no vector ALU work per copy
            const mvs_srd_t srd = make_srd(
                srcs16 + (((size_t)v * p.B + b) * ngroups + g) * grp_floats + dq * 4,
                (unsigned)(grp_floats * 4));
            const int n = bw[v] * bh[v];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (j * 64 >= n || (j % NSH) != dj) continue;    // wave-uniform
                if (j * 64 + lane < cap)                         // the last instruction may be partial
                    glds16_buf((unsigned)soff[v][j] * 4u, srd, 0u,
                               lds_base + (unsigned)(((v * NQ + dq) * cap + j * 64) * 16));
            }
        }
    };

    if constexpr (CORNER) {
        const int v0 = lane >> 3, k = lane & 7;
        const int xlo = tx * kTileW, xhi = min(tx * kTileW + kTileW - 1, p.W - 1);
        const int ylo = ty * kTileH, yhi = min(ty * kTileH + kTileH - 1, p.H - 1);
        const int dlo = dc * kTileD, dhi = min(dc * kTileD + kTileD - 1, p.D - 1);
        const float *r = rt + ((int64_t)min(v0, NV - 1) * p.B + b) * 12;
        const float cxk = (float)((k & 1) ? xhi : xlo), cyk = (float)((k & 2) ? yhi : ylo);
        const float dk = depth[(int64_t)b * p.D + ((k & 4) ? dhi : dlo)];
        float rx, ry, rz, ix, iy;
        sweep_ray(r, cxk, cyk, rx, ry, rz);
        sweep_coord(r, rx, ry, rz, dk, p.half_w, p.half_h, p.unn_w, p.unn_h, p.align_corners, ix,
                    iy);
        const bool zok = (rz * dk + r[11]) > 1e-6f && fabsf(ix) < 1.0e9f && fabsf(iy) < 1.0e9f;
        int lo_x = (int)floorf(ix) - 1, hi_x = (int)floorf(ix) + 2;
        int lo_y = (int)floorf(iy) - 1, hi_y = (int)floorf(iy) + 2;
        int bad = zok ? 0 : 1;
#pragma unroll
        for (int off = 1; off <= 4; off <<= 1) {
            lo_x = min(lo_x, __shfl_xor(lo_x, off)); hi_x = max(hi_x, __shfl_xor(hi_x, off));
            lo_y = min(lo_y, __shfl_xor(lo_y, off)); hi_y = max(hi_y, __shfl_xor(hi_y, off));
            bad |= __shfl_xor(bad, off);
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            int x0 = max(__builtin_amdgcn_readlane(lo_x, v * 8), 0);
            int x1 = min(__builtin_amdgcn_readlane(hi_x, v * 8), p.W - 1);
            int y0 = max(__builtin_amdgcn_readlane(lo_y, v * 8), 0);
            int y1 = min(__builtin_amdgcn_readlane(hi_y, v * 8), p.H - 1);
            const int vbad = __builtin_amdgcn_readlane(bad, v * 8);
            if (x1 < x0 || y1 < y0) { x0 = y0 = x1 = y1 = 0; }
            bx0[v] = x0; by0[v] = y0; bw[v] = x1 - x0 + 1; bh[v] = y1 - y0 + 1;
            staged[v] = !vbad && bw[v] * bh[v] <= cap;
        }
        plan_dma();
        issue_dma(0);      // lands under phase A
    } else {
        if (tid < NV) {
            s_box[tid][0] = 0x7fffffff; s_box[tid][1] = 0x7fffffff;
            s_box[tid][2] = -1; s_box[tid][3] = -1;
        }
        __syncthreads();
    }

    if (ablate & 64) { if (soff[0][0] == 0x7ffffff1) out[0] = 1.f; return; }   // tuning: box + plan only
    // ---- phase A: homography + tap set per source view (registers)
    float wnw[NV], wne[NV], wsw[NV], wse[NV];
    int tx0[NV], ty0[NV];
    bool wave_in[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) wave_in[v] = true;
    // (round 6) the reference's coordinates through shared / precomputed refined reciprocals -- the compiler's own division
    // sequence without its range scaling, bit for bit the same quotients (mvs_common.h: sweep_coord_shared; a wave with a Z that is
    // not a normal number takes the compiler's divisions): 23 vector instructions per view where four divisions take 44, in a
    // kernel whose tap set-up outweighs its blends at 8 or 16 channels (CasMVSNet's second and third stage)
    const float rhw = refined_rcp(p.half_w), rhh = refined_rcp(p.half_h);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const float *r = rt + ((int64_t)v * p.B + b) * 12;
        float rx, ry, rz, ix, iy;
        sweep_ray(r, (float)cx, (float)cy, rx, ry, rz);
        {
            const float X = rx * dv + r[3], Y = ry * dv + r[7], Z = rz * dv + r[11];
            if (__any(!sweep_coord_safe(Z))) sweep_coord(r, rx, ry, rz, dv, p.half_w, p.half_h, p.unn_w, p.unn_h, p.align_corners, ix, iy);
            else sweep_coord_shared(X, Y, Z, p.half_w, p.half_h, rhw, rhh, p.unn_w, p.unn_h, p.align_corners, ix, iy);
        }
        Taps t = make_taps(ix, iy, p.H, p.W);
        const bool fin = (fabsf(ix) <= 3.0e38f) && (fabsf(iy) <= 3.0e38f);
        const float dead = fin ? 0.0f : __int_as_float(0x7fc00000);
        const bool m00 = t.x0ok && t.y0ok, m01 = t.x1ok && t.y0ok;
        const bool m10 = t.x0ok && t.y1ok, m11 = t.x1ok && t.y1ok;
        wnw[v] = m00 ? t.nw : dead; wne[v] = m01 ? t.ne : dead;
        wsw[v] = m10 ? t.sw : dead; wse[v] = m11 ? t.se : dead;
        const float x0f = fminf(fmaxf(floorf(ix), -4.0f), (float)p.W + 4.0f);
        const float y0f = fminf(fmaxf(floorf(iy), -4.0f), (float)p.H + 4.0f);
        tx0[v] = fin ? (int)x0f : -4;
        ty0[v] = fin ? (int)y0f : -4;
        const bool anyx = t.x0ok || t.x1ok, anyy = t.y0ok || t.y1ok;
        int lo_x = 0x7fffffff, hi_x = -1, lo_y = 0x7fffffff, hi_y = -1;
        if (anyx && anyy) {
            lo_x = t.x0ok ? tx0[v] : tx0[v] + 1;
            hi_x = t.x1ok ? tx0[v] + 1 : tx0[v];
            lo_y = t.y0ok ? ty0[v] : ty0[v] + 1;
            hi_y = t.y1ok ? ty0[v] + 1 : ty0[v];
        }
        if constexpr (CORNER) {
            const bool inbox = !(anyx && anyy) ||
                               (lo_x >= bx0[v] && hi_x < bx0[v] + bw[v] && lo_y >= by0[v] &&
                                hi_y < by0[v] + bh[v]);
            wave_in[v] = __all(inbox);
        } else {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                lo_x = min(lo_x, __shfl_xor(lo_x, off)); hi_x = max(hi_x, __shfl_xor(hi_x, off));
                lo_y = min(lo_y, __shfl_xor(lo_y, off)); hi_y = max(hi_y, __shfl_xor(hi_y, off));
            }
            if (lane == 0) {
                atomicMin(&s_box[v][0], lo_x); atomicMin(&s_box[v][1], lo_y);
                atomicMax(&s_box[v][2], hi_x); atomicMax(&s_box[v][3], hi_y);
            }
        }
    }
    if constexpr (!CORNER) {
        __syncthreads();
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            int x0 = __builtin_amdgcn_readfirstlane(s_box[v][0]);
            int y0 = __builtin_amdgcn_readfirstlane(s_box[v][1]);
            int x1 = __builtin_amdgcn_readfirstlane(s_box[v][2]);
            int y1 = __builtin_amdgcn_readfirstlane(s_box[v][3]);
            if (x1 < x0 || y1 < y0) { x0 = y0 = x1 = y1 = 0; }
            bx0[v] = x0; by0[v] = y0; bw[v] = x1 - x0 + 1; bh[v] = y1 - y0 + 1;
            staged[v] = bw[v] * bh[v] <= cap;
        }
        plan_dma();
        issue_dma(0);
    }

    if (ablate & 128) {   // tuning: setup only
        float acc = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) acc += wnw[v] + wne[v] + wsw[v] + wse[v] + (float)(tx0[v] + ty0[v]) + (wave_in[v] ? 1.f : 0.f);
        if (acc == 1.2345e30f) out[0] = acc;
        return;
    }
    const float rV = 1.0f / p.fV;
#pragma unroll 1
    for (int g = 0; g < ngroups; ++g) {
        // make the tap positions opaque per iteration: otherwise LICM hoists every view's 4
        // LDS offsets + 4 64-bit fallback pointers out of this loop and spills
#pragma unroll
        for (int v = 0; v < NV; ++v) asm volatile("" : "+v"(tx0[v]), "+v"(ty0[v]));
        float4 ref4[NQ];
        {
            const float4 *rp = reinterpret_cast<const float4 *>(
                ref16 + ((size_t)b * ngroups + g) * grp_floats + (size_t)pix * GC);
#pragma unroll
            for (int k = 0; k < NQ; ++k) ref4[k] = rp[k];
        }
        // this group's footprints have landed for every wave
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        float S[GC], Q[GC];
#pragma unroll
        for (int k = 0; k < NQ; ++k) {
            const float4 r4 = ref4[k];
            const float rr[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                Q[k * 4 + c] = rr[c] * rr[c];
                S[k * 4 + c] = p.alias_quirk ? Q[k * 4 + c] : rr[c];
            }
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            if (ablate & 2) continue;
            if (staged[v] && wave_in[v]) {
                const int x0c = min(max(tx0[v], bx0[v]), bx0[v] + bw[v] - 1) - bx0[v];
                const int x1c = min(max(tx0[v] + 1, bx0[v]), bx0[v] + bw[v] - 1) - bx0[v];
                const int y0c = min(max(ty0[v], by0[v]), by0[v] + bh[v] - 1) - by0[v];
                const int y1c = min(max(ty0[v] + 1, by0[v]), by0[v] + bh[v] - 1) - by0[v];
                accumulate_taps_planar<cap, NQ>(lds + v * NQ * cap * 4, y0c * bw[v] + x0c,
                                            y0c * bw[v] + x1c, y1c * bw[v] + x0c,
                                            y1c * bw[v] + x1c, wnw[v], wne[v], wsw[v], wse[v], S, Q);
            } else {
                const float *base = srcs16 + (((size_t)v * p.B + b) * ngroups + g) * grp_floats;
                const int x0c = min(max(tx0[v], 0), p.W - 1), x1c = min(max(tx0[v] + 1, 0), p.W - 1);
                const int y0c = min(max(ty0[v], 0), p.H - 1), y1c = min(max(ty0[v] + 1, 0), p.H - 1);
                accumulate_taps<NQ>(base + ((size_t)y0c * p.W + x0c) * GC,
                                    base + ((size_t)y0c * p.W + x1c) * GC,
                                    base + ((size_t)y1c * p.W + x0c) * GC,
                                    base + ((size_t)y1c * p.W + x1c) * GC, wnw[v], wne[v], wsw[v],
                                    wse[v], S, Q);
            }
        }
        // every wave is done with this group's LDS image: start the next group's copy; it
        // lands while the variances below are formed and stored
        if (g + 1 < ngroups) {
            __syncthreads();
            issue_dma(g + 1);
        }
        float var[GC];
        // any |S|, |Q| outside [1e-30, 3e38] (div_views_tiny) sends the wave to the true division:
        // one min / max chain over the magnitudes instead of two compares per value (a NaN slips
        // through fminf / fmaxf, and through either division as NaN)
        float amin = 3.0e38f, amax = 0.0f;
#pragma unroll
        for (int c = 0; c < GC; ++c) {
            const float m = div_views_fast(S[c], p.fV, rV);
            var[c] = div_views_fast(Q[c], p.fV, rV) - m * m;
            amin = fminf(amin, fminf(fabsf(S[c]), fabsf(Q[c])));
            amax = fmaxf(amax, fmaxf(fabsf(S[c]), fabsf(Q[c])));
        }
        const bool tiny = !(amin >= 1e-30f && amax <= 3.0e38f);
        if (__any(tiny)) {
#pragma unroll
            for (int c = 0; c < GC; ++c) {
                const float m = S[c] / p.fV;
                var[c] = Q[c] / p.fV - m * m;
            }
        }
        if (live && !(ablate & 8)) {
            const size_t vox = ((size_t)b * p.D + d) * plane + pix;
            if (out_c8) {
                const size_t row = ((size_t)b * p.D + d) * p.H + py;
#pragma unroll
                for (int h = 0; h < NQ / 2; ++h) {
                    float *o = out + ((row * (p.C >> 3) + (g * (NQ / 2) + h)) * p.W + px) * 8;
                    reinterpret_cast<float4 *>(o)[0] = make_float4(var[h * 8 + 0], var[h * 8 + 1], var[h * 8 + 2], var[h * 8 + 3]);
                    reinterpret_cast<float4 *>(o)[1] = make_float4(var[h * 8 + 4], var[h * 8 + 5], var[h * 8 + 6], var[h * 8 + 7]);
                }
            } else {
                float4 *o = reinterpret_cast<float4 *>(out + vox * p.C + g * GC);
#pragma unroll
                for (int k = 0; k < NQ; ++k)
                    o[k] = make_float4(var[k * 4], var[k * 4 + 1], var[k * 4 + 2], var[k * 4 + 3]);
            }
#pragma unroll
            for (int c = 0; c < GC; c += 2) vmax = max_nan(max_nan(vmax, __builtin_fabsf(var[c])), __builtin_fabsf(var[c + 1]));
        }
    }
    if constexpr (LOOP) __syncthreads();   // every wave is done with this tile's LDS image before the next tile's copies
    vblk += (int)gridDim.x;
    } while (LOOP && vblk < nblk);
    publish_absmax(absmax, vmax);
}

// ---------------------------------------------------------------------
// Backward of the fused variance in the layouts of variance_fwd_dma_kernel (training, shared
// depth planes): grad_var [B,D,H,W,C] -> grad_ref, grad_srcs in the 16-channel-blocked
// feature layout.  d var / d w_v = 2 w_v / V - 2 S / V^2 (same for the reference features).
// The planar kernel sends every tap's share straight to HBM: (1 + 4 (V-1)) C atomics per
// voxel, 1.1 G at config 5, and those atomics are its whole cost.  Here a block (8x8 pixels
// x 4 depth planes) stages the source footprints as the forward kernel does, accumulates
// the gradients of the same footprints (and of its 64 reference pixels) in LDS, and flushes
// each texel ONCE: ~8x fewer global atomics.  A view whose footprint does not fit, or a wave
// with a tap outside the box, adds to HBM directly.
//
// The LDS accumulators are 64-bit fixed point, not floats: on gfx950 ds_add_f32 retires one
// lane every ~3 cycles (194 cycles per wave instruction whatever the bank pattern,
// scripts/micro/lds_atomic.hip), ds_add_u64 a whole wave in 7.  Per 16-channel group the block
// bounds its contributions, |g (2 w / V - 2 S / V^2) weight| <= 4 max|g| max|f| / V < 2^e,
// scales by 2^(52-e) (exact) and adds integers: a texel collects at most 2^8 such terms, so
// the sum stays below 2^61, every term is rounded at 2^-52 of the bound (fp32 accumulation
// rounds at 2^-24 of the running sum), and the block's sum no longer depends on the order of
// the adds.  Non-finite gradients or features poison the block's footprint with NaN.
// Channel quads per pass: a whole 16-channel group while two blocks per CU still hold a
// typical footprint (11x11 texels for a translation, more under rotation), else half a group
// -- a view whose footprint does not fit falls back to HBM atomics, ~50x slower.
__host__ __device__ constexpr int bwd_quads(int nv) { return nv <= 2 ? 4 : 2; }
__host__ __device__ constexpr int bwd_fixed_bytes(int nv) { return bwd_quads(nv) * 4 * 64 * 8 + 64; }
__host__ __device__ constexpr int bwd_cap_for(int nv, int budget) {
    const int raw = (budget - bwd_fixed_bytes(nv)) / (nv * 48 * bwd_quads(nv));   // 16 B features + 32 B sums per quad
    return raw >= 160 ? 160 : raw;
}
__host__ __device__ constexpr int bwd_blocks_per_cu(int nv) { return bwd_cap_for(nv, 80 * 1024) >= 160 ? 2 : 1; }
__host__ __device__ constexpr int bwd_cap(int nv) {   // texels per view
    return bwd_cap_for(nv, bwd_blocks_per_cu(nv) == 2 ? 80 * 1024 : 160 * 1024);
}
__host__ __device__ constexpr int bwd_lds_bytes(int nv) {
    return nv * bwd_cap(nv) * 48 * bwd_quads(nv) + bwd_fixed_bytes(nv);
}

// x * 2^k (|result| < 2^62) as a two's-complement 64-bit integer, floor rounding.
__device__ __forceinline__ unsigned long long fixed64(float y) {
    const float hi = floorf(y * 0x1p-32f);
    const float lo = __fmaf_rn(hi, -0x1p32f, y);   // exact, in [0, 2^32)
    return ((unsigned long long)(unsigned)(int)hi << 32) | (unsigned long long)(unsigned)lo;
}

__device__ __forceinline__ float wave_max(float x) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x = fmaxf(x, __shfl_xor(x, off));
    return x;
}

#ifdef MVS_TUNING
// tuning build: cycles of wave 0 per phase, summed over blocks and passes (mvs_tuning_varbwd_laps, scripts/exp_varbwd_laps.py):
// [0] set-up, [1] clear + staging, [2] bound, [3] accumulate, [4] flush, [5] passes counted
__device__ unsigned long long g_varbwd_laps[8];
#define MVS_BWD_LAP(i) do { if (tid == 0) { const long long t_ = __builtin_readcyclecounter(); atomicAdd(&g_varbwd_laps[i], (unsigned long long)(t_ - lap_t)); lap_t = t_; } } while (0)
#else
#define MVS_BWD_LAP(i) do { } while (0)
#endif

template <int NV>
__global__ __launch_bounds__(256, bwd_blocks_per_cu(NV)) void variance_bwd_dma_kernel(
    const float *__restrict__ gvar, const float *__restrict__ ref16, const float *__restrict__ srcs16,
    const float *__restrict__ rt, const float *__restrict__ depth, SweepParams p, int tiles_x,
    int tiles_y, float *__restrict__ gref16, float *__restrict__ gsrcs16) {
    constexpr int cap = bwd_cap(NV);
    constexpr int NJ = (cap + 63) / 64;
    constexpr int NQ = bwd_quads(NV), NPASS = 4 / NQ;   // quads per pass, passes per group
    typedef unsigned long long u64;
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[bwd_lds_bytes(NV)];
    float *fea = reinterpret_cast<float *>(lds_raw);                       // [NV][NQ quads][cap][4]
    u64 *grd = reinterpret_cast<u64 *>(lds_raw + NV * NQ * cap * 16);       // [NV][4 NQ channels][cap]
    u64 *rgr = grd + NV * NQ * 4 * cap;                                     // [4 NQ channels][64 pixels]
    unsigned *bound = reinterpret_cast<unsigned *>(rgr + NQ * 4 * 64);      // max|g|, max|f| (float bits)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef MVS_TUNING
    long long lap_t = __builtin_readcyclecounter();
#endif
    int tx, ty, dc;
    {
        const int nwg = gridDim.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7;
        int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + ((int)blockIdx.x >> 3);
        const int ndc = nwg / (tiles_x * tiles_y);
        dc = bid % ndc; bid /= ndc;
        tx = bid % tiles_x;
        ty = bid / tiles_x;
    }
    const int b = blockIdx.y;
    const int px = tx * kTileW + (lane & (kTileW - 1)), py = ty * kTileH + lane / kTileW;
    const int d = dc * kTileD + wv;
    const bool live = px < p.W && py < p.H && d < p.D;
    const int cx = min(px, p.W - 1), cy = min(py, p.H - 1), cd = min(d, p.D - 1);
    const int plane = p.H * p.W;
    const int pix = cy * p.W + cx;
    const float dv = depth[(int64_t)b * p.D + cd];
    const int ngroups = p.C >> 4;
    const size_t grp_floats = (size_t)plane * 16;
    const unsigned lds_base = (unsigned)(uintptr_t)lds_raw;

    // ---- footprint boxes from the 8 corner voxels (see variance_fwd_dma_kernel)
    int bx0[NV], by0[NV], bw[NV], bh[NV];
    bool staged[NV];
    {
        const int v0 = lane >> 3, k = lane & 7;
        const int xlo = tx * kTileW, xhi = min(tx * kTileW + kTileW - 1, p.W - 1);
        const int ylo = ty * kTileH, yhi = min(ty * kTileH + kTileH - 1, p.H - 1);
        const int dlo = dc * kTileD, dhi = min(dc * kTileD + kTileD - 1, p.D - 1);
        const float *r = rt + ((int64_t)min(v0, NV - 1) * p.B + b) * 12;
        const float cxk = (float)((k & 1) ? xhi : xlo), cyk = (float)((k & 2) ? yhi : ylo);
        const float dk = depth[(int64_t)b * p.D + ((k & 4) ? dhi : dlo)];
        float rx, ry, rz, ix, iy;
        sweep_ray(r, cxk, cyk, rx, ry, rz);
        sweep_coord(r, rx, ry, rz, dk, p.half_w, p.half_h, p.unn_w, p.unn_h, p.align_corners, ix, iy);
        const bool zok = (rz * dk + r[11]) > 1e-6f && fabsf(ix) < 1.0e9f && fabsf(iy) < 1.0e9f;
        int lo_x = (int)floorf(ix) - 1, hi_x = (int)floorf(ix) + 2;
        int lo_y = (int)floorf(iy) - 1, hi_y = (int)floorf(iy) + 2;
        int bad = zok ? 0 : 1;
#pragma unroll
        for (int off = 1; off <= 4; off <<= 1) {
            lo_x = min(lo_x, __shfl_xor(lo_x, off)); hi_x = max(hi_x, __shfl_xor(hi_x, off));
            lo_y = min(lo_y, __shfl_xor(lo_y, off)); hi_y = max(hi_y, __shfl_xor(hi_y, off));
            bad |= __shfl_xor(bad, off);
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            int x0 = max(__builtin_amdgcn_readlane(lo_x, v * 8), 0);
            int x1 = min(__builtin_amdgcn_readlane(hi_x, v * 8), p.W - 1);
            int y0 = max(__builtin_amdgcn_readlane(lo_y, v * 8), 0);
            int y1 = min(__builtin_amdgcn_readlane(hi_y, v * 8), p.H - 1);
            const int vbad = __builtin_amdgcn_readlane(bad, v * 8);
            if (x1 < x0 || y1 < y0) { x0 = y0 = x1 = y1 = 0; }
            bx0[v] = x0; by0[v] = y0; bw[v] = x1 - x0 + 1; bh[v] = y1 - y0 + 1;
            staged[v] = !vbad && bw[v] * bh[v] <= cap;
        }
    }
    // source offset (floats, inside a 16-channel group) of this lane's texels
    int soff[NV][NJ];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int n = bw[v] * bh[v];
        const unsigned inv = (65536u + bw[v] - 1) / bw[v];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int t = min(j * 64 + lane, max(n - 1, 0));
            const int ly = (int)(__umul24((unsigned)t, inv) >> 16), lx = t - __mul24(ly, bw[v]);
            soff[v][j] = (__mul24(by0[v] + ly, p.W) + (bx0[v] + lx)) * 16;
        }
    }

    // ---- taps of this voxel per view
    float wnw[NV], wne[NV], wsw[NV], wse[NV];
    int o00[NV], o01[NV], o10[NV], o11[NV];   // texel index inside the box (clamped), or image offset
    bool m00[NV], m01[NV], m10[NV], m11[NV];  // tap inside the image
    bool wave_in[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const float *r = rt + ((int64_t)v * p.B + b) * 12;
        float rx, ry, rz, ix, iy;
        sweep_ray(r, (float)cx, (float)cy, rx, ry, rz);
        sweep_coord(r, rx, ry, rz, dv, p.half_w, p.half_h, p.unn_w, p.unn_h, p.align_corners, ix, iy);
        Taps t = make_taps(ix, iy, p.H, p.W);
        m00[v] = t.x0ok && t.y0ok; m01[v] = t.x1ok && t.y0ok;
        m10[v] = t.x0ok && t.y1ok; m11[v] = t.x1ok && t.y1ok;
        wnw[v] = t.nw; wne[v] = t.ne; wsw[v] = t.sw; wse[v] = t.se;
        const bool any = m00[v] || m01[v] || m10[v] || m11[v];
        const int lo_x = t.x0ok ? t.x0 : t.x1, hi_x = t.x1ok ? t.x1 : t.x0;
        const int lo_y = t.y0ok ? t.y0 : t.y1, hi_y = t.y1ok ? t.y1 : t.y0;
        const bool inbox = !any || (lo_x >= bx0[v] && hi_x < bx0[v] + bw[v] && lo_y >= by0[v] &&
                                    hi_y < by0[v] + bh[v]);
        wave_in[v] = __all(inbox) && staged[v];
        const int x0c = min(max(t.x0, 0), p.W - 1), x1c = min(max(t.x1, 0), p.W - 1);
        const int y0c = min(max(t.y0, 0), p.H - 1), y1c = min(max(t.y1, 0), p.H - 1);
        if (wave_in[v]) {
            const int ax0 = min(max(x0c - bx0[v], 0), bw[v] - 1), ax1 = min(max(x1c - bx0[v], 0), bw[v] - 1);
            const int ay0 = min(max(y0c - by0[v], 0), bh[v] - 1), ay1 = min(max(y1c - by0[v], 0), bh[v] - 1);
            o00[v] = ay0 * bw[v] + ax0; o01[v] = ay0 * bw[v] + ax1;
            o10[v] = ay1 * bw[v] + ax0; o11[v] = ay1 * bw[v] + ax1;
        } else {
            o00[v] = y0c * p.W + x0c; o01[v] = y0c * p.W + x1c;
            o10[v] = y1c * p.W + x0c; o11[v] = y1c * p.W + x1c;
        }
    }

    const float inv_v = 1.0f / p.fV;
    constexpr int KU = NV <= 4 ? NQ : 1;
    constexpr int NZ = (NV * NQ * 4 * cap + NQ * 4 * 64) / 2;   // uint4 slots of the sums (+ ref sums)
    const int dq = wv % NQ, dj = wv / NQ;   // this wave's share of the staging: quad dq, every NPASS-th granule
#pragma unroll 1
    for (int pass = 0; pass < ngroups * NPASS; ++pass) {
        const int g = pass / NPASS, q0 = (pass % NPASS) * NQ;   // group, first quad of the pass
        __syncthreads();   // previous pass's flush is complete
        MVS_BWD_LAP(pass == 0 ? 0 : 4);
        for (int i = tid; i < NZ; i += 256)
            reinterpret_cast<uint4 *>(grd)[i] = make_uint4(0u, 0u, 0u, 0u);
        if (tid < 2) bound[tid] = 0u;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            if (!staged[v]) continue;
            const mvs_srd_t srd = make_srd(srcs16 + (((size_t)v * p.B + b) * ngroups + g) * grp_floats + (q0 + dq) * 4,
                                           (unsigned)(grp_floats * 4));
            const int n = bw[v] * bh[v];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (j * 64 >= n || (j % NPASS) != dj) continue;
                if (j * 64 + lane < cap)
                    glds16_buf((unsigned)soff[v][j] * 4u, srd, 0u,
                               lds_base + (unsigned)(((v * NQ + dq) * cap + j * 64) * 16));
            }
        }
        float4 ref4[NQ], gv4[NQ];
        {
            const float4 *rp = reinterpret_cast<const float4 *>(
                ref16 + ((size_t)b * ngroups + g) * grp_floats + (size_t)pix * 16 + q0 * 4);
            const float4 *gp = reinterpret_cast<const float4 *>(
                gvar + (((size_t)b * p.D + cd) * plane + pix) * p.C + g * 16 + q0 * 4);
#pragma unroll
            for (int k = 0; k < NQ; ++k) {
                ref4[k] = rp[k];
                gv4[k] = live ? gp[k] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        MVS_BWD_LAP(1);

        // the four taps of view v, channel quad k, as float4
        auto taps = [&](int v, int k, float4 &a, float4 &bq, float4 &c, float4 &e) {
            if (wave_in[v]) {
                const float *base = fea + (v * NQ + k) * cap * 4;
                a = *reinterpret_cast<const float4 *>(base + o00[v] * 4);
                bq = *reinterpret_cast<const float4 *>(base + o01[v] * 4);
                c = *reinterpret_cast<const float4 *>(base + o10[v] * 4);
                e = *reinterpret_cast<const float4 *>(base + o11[v] * 4);
            } else {
                const float *base = srcs16 + (((size_t)v * p.B + b) * ngroups + g) * grp_floats + (q0 + k) * 4;
                a = *reinterpret_cast<const float4 *>(base + (size_t)o00[v] * 16);
                bq = *reinterpret_cast<const float4 *>(base + (size_t)o01[v] * 16);
                c = *reinterpret_cast<const float4 *>(base + (size_t)o10[v] * 16);
                e = *reinterpret_cast<const float4 *>(base + (size_t)o11[v] * 16);
            }
        };
        // ---- the block's bound: max |g| and max |f| over what will meet in LDS.  fmaxf drops
        // NaNs, so the integer max of the raw magnitudes carries them (NaN patterns sort above inf).
        {
            unsigned gm = 0u, fm = 0u;
            auto mag = [](float x) { return __float_as_uint(x) & 0x7fffffffu; };
#pragma unroll
            for (int k = 0; k < NQ; ++k) {
                gm = max(max(gm, mag(gv4[k].x)), max(mag(gv4[k].y), max(mag(gv4[k].z), mag(gv4[k].w))));
                fm = max(max(fm, mag(ref4[k].x)), max(mag(ref4[k].y), max(mag(ref4[k].z), mag(ref4[k].w))));
            }
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                if (!staged[v]) continue;
                const int n = bw[v] * bh[v];
                for (int i = tid; i < NQ * cap; i += 256) {
                    const int t = i % cap;
                    if (t < n) {
                        const float4 f = *reinterpret_cast<const float4 *>(fea + (v * NQ * cap + i) * 4);
                        fm = max(max(fm, mag(f.x)), max(mag(f.y), max(mag(f.z), mag(f.w))));
                    }
                }
            }
            // Views this wave samples from HBM (footprint not staged, or a tap outside the box) enter
            // S -- hence kk = 2 S / V^2 -- just the same: their magnitudes belong in the bound too
            // (cold path; the taps are read once more here).
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                if (wave_in[v]) continue;
#pragma unroll 1
                for (int k = 0; k < NQ; ++k) {
                    float4 a, bq, c, e;
                    taps(v, k, a, bq, c, e);
                    fm = max(max(fm, max(mag(a.x), mag(a.y))), max(max(mag(a.z), mag(a.w)), max(mag(bq.x), mag(bq.y))));
                    fm = max(max(fm, max(mag(bq.z), mag(bq.w))), max(max(mag(c.x), mag(c.y)), max(mag(c.z), mag(c.w))));
                    fm = max(max(fm, max(mag(e.x), mag(e.y))), max(mag(e.z), mag(e.w)));
                }
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                gm = max(gm, (unsigned)__shfl_xor((int)gm, off));
                fm = max(fm, (unsigned)__shfl_xor((int)fm, off));
            }
            if (lane == 0) { atomicMax(bound, gm); atomicMax(bound + 1, fm); }
        }
        __syncthreads();
        MVS_BWD_LAP(2);
        float scale, inv_scale;
        bool poison;
        {
            const unsigned gm = bound[0], fm = bound[1];
            poison = gm >= 0x7f800000u || fm >= 0x7f800000u;
            const float m = 4.0f * __uint_as_float(gm) * __uint_as_float(fm) * inv_v;
            poison = poison || !(m < 3.0e38f);
            int e = (int)((__float_as_uint(m) >> 23) & 0xffu) - 126;   // m < 2^e
            e = max(e, -74);
            scale = __uint_as_float((unsigned)(127 + 52 - e) << 23);
            inv_scale = __uint_as_float((unsigned)(127 - 52 + e) << 23);
        }

        // One channel quad at a time; unrolled while the registers last (the rolled form picks
        // the quad's reference/gradient values with selects).
#pragma unroll KU
        for (int k = 0; k < NQ; ++k) {
            const float4 r4 = k == 0 ? ref4[0] : (k == 1 ? ref4[1] : (k == 2 ? ref4[NQ - 2] : ref4[NQ - 1]));
            const float4 g4 = k == 0 ? gv4[0] : (k == 1 ? gv4[1] : (k == 2 ? gv4[NQ - 2] : gv4[NQ - 1]));
            const float rr[4] = {r4.x, r4.y, r4.z, r4.w};
            const float gg[4] = {g4.x, g4.y, g4.z, g4.w};
            float S[4] = {rr[0], rr[1], rr[2], rr[3]};
            float w[NV][4];
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                float4 a, bq, c, e;
                taps(v, k, a, bq, c, e);
                const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {bq.x, bq.y, bq.z, bq.w};
                const float cv[4] = {c.x, c.y, c.z, c.w}, ev[4] = {e.x, e.y, e.z, e.w};
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    w[v][cc] = __fmaf_rn(m11[v] ? ev[cc] : 0.f, wse[v],
                               __fmaf_rn(m10[v] ? cv[cc] : 0.f, wsw[v],
                               __fmaf_rn(m01[v] ? bv[cc] : 0.f, wne[v], (m00[v] ? av[cc] : 0.f) * wnw[v])));
                    S[cc] += w[v][cc];
                }
            }
            float kk[4];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) kk[cc] = 2.0f * S[cc] * inv_v * inv_v;
            // reference features: the 4 depth planes of the block meet in LDS
            if (!poison) {
#pragma unroll
                for (int cc = 0; cc < 4; ++cc)
                    atomicAdd(rgr + (k * 4 + cc) * 64 + lane,
                              fixed64(gg[cc] * (2.0f * rr[cc] * inv_v - kk[cc]) * scale));
            }
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                float gw[4];
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) gw[cc] = gg[cc] * (2.0f * w[v][cc] * inv_v - kk[cc]);
                if (wave_in[v]) {
                    if (poison) continue;
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
                        u64 *base = grd + ((v * NQ + k) * 4 + cc) * cap;
                        const float gs = gw[cc] * scale;
                        if (m00[v]) atomicAdd(base + o00[v], fixed64(gs * wnw[v]));
                        if (m01[v]) atomicAdd(base + o01[v], fixed64(gs * wne[v]));
                        if (m10[v]) atomicAdd(base + o10[v], fixed64(gs * wsw[v]));
                        if (m11[v]) atomicAdd(base + o11[v], fixed64(gs * wse[v]));
                    }
                } else if (live) {
                    float *base = gsrcs16 + (((size_t)v * p.B + b) * ngroups + g) * grp_floats + (q0 + k) * 4;
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
                        if (m00[v]) unsafeAtomicAdd(base + (size_t)o00[v] * 16 + cc, gw[cc] * wnw[v]);
                        if (m01[v]) unsafeAtomicAdd(base + (size_t)o01[v] * 16 + cc, gw[cc] * wne[v]);
                        if (m10[v]) unsafeAtomicAdd(base + (size_t)o10[v] * 16 + cc, gw[cc] * wsw[v]);
                        if (m11[v]) unsafeAtomicAdd(base + (size_t)o11[v] * 16 + cc, gw[cc] * wse[v]);
                    }
                }
            }
        }
        __syncthreads();
        MVS_BWD_LAP(3);
        // ---- flush: every staged texel once, every reference pixel once
        const double inv_scale_d = (double)inv_scale;
        const float qnan = __uint_as_float(0x7fc00000u);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            if (!staged[v]) continue;
            const int n = bw[v] * bh[v];
            const unsigned inv = (65536u + bw[v] - 1) / bw[v];
            float *dst = gsrcs16 + (((size_t)v * p.B + b) * ngroups + g) * grp_floats + q0 * 4;
            for (int e = tid; e < n * NQ * 4; e += 256) {
                const int t = e / (NQ * 4), ch = e % (NQ * 4);
                const long long q = (long long)grd[(v * NQ * 4 + ch) * cap + t];
                if (q != 0 || poison) {
                    const float val = poison ? qnan : (float)((double)q * inv_scale_d);
                    const int ly = (int)(((unsigned)t * inv) >> 16), lx = t - ly * bw[v];
                    unsafeAtomicAdd(dst + ((size_t)(by0[v] + ly) * p.W + (bx0[v] + lx)) * 16 + ch, val);
                }
            }
        }
        for (int e = tid; e < 64 * NQ * 4; e += 256) {
            const int pl = e / (NQ * 4), ch = e % (NQ * 4);
            const int qx = tx * kTileW + (pl & (kTileW - 1)), qy = ty * kTileH + pl / kTileW;
            const long long q = (long long)rgr[ch * 64 + pl];
            if (qx < p.W && qy < p.H && (q != 0 || poison)) {
                const float val = poison ? qnan : (float)((double)q * inv_scale_d);
                unsafeAtomicAdd(gref16 + ((size_t)b * ngroups + g) * grp_floats + ((size_t)qy * p.W + qx) * 16 + q0 * 4 + ch, val);
            }
        }
#ifdef MVS_TUNING
        if (tid == 0) atomicAdd(&g_varbwd_laps[5], 1ull);
#endif
    }
    MVS_BWD_LAP(4);
}

#ifdef MVS_TUNING
extern "C" int mvs_tuning_varbwd_laps(unsigned long long *out8, int reset) {
    if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_varbwd_laps), 8 * sizeof(unsigned long long)) != hipSuccess) return MVS_ELAUNCH;
    if (reset) {
        const unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_varbwd_laps), z, sizeof(z)) != hipSuccess) return MVS_ELAUNCH;
    }
    return MVS_OK;
}
#endif

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
#ifdef MVS_TUNING
    const char *abl_env = getenv("MVS_SWEEP_ABLATE");   // tuning builds only
#else
    const char *abl_env = nullptr;
#endif
    const int ablate = abl_env ? atoi(abl_env) : 0;
#define MVS_CL_CASE(n)                                                                         \
    case n:                                                                                    \
        hipLaunchKernelGGL((variance_fwd_cl_kernel<CQ, n>), grid, dim3(256), 0, st, ref, srcs, \
                           rt, depth, p, fdp, fdw, out, out_c8, ablate);                       \
        return MVS_OK;
    switch (NV) {
        MVS_CL_CASE(1) MVS_CL_CASE(2) MVS_CL_CASE(3) MVS_CL_CASE(4) MVS_CL_CASE(5) MVS_CL_CASE(6)
        MVS_CL_CASE(7) MVS_CL_CASE(8)
    }
#undef MVS_CL_CASE
    return bare_error(MVS_EUNSUPPORTED, __func__, __LINE__);
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
    if (!grid_for((int64_t)B * D * H * W, 256, grid)) return bare_error(MVS_EINVAL, __func__, __LINE__);
    hipStream_t st = as_stream(stream);
    if (hipMemsetAsync(grad_src, 0, sizeof(float) * (size_t)B * C * H * W, st) != hipSuccess)
        return check_launch("mvs_warp_bwd_f32 memset");
    hipLaunchKernelGGL(warp_bwd_planar_kernel, dim3(grid), dim3(256), 0, st, grad_out, rot_trans,
                       depth_values, p, grad_src);
    return check_launch("mvs_warp_bwd_f32");
}


// LDS-staged per-tile kernel: features [B,C/16,H,W,16].  sel: see variance_fwd_dma_kernel.
static int launch_variance_tile_c16(const float *ref_fea, const float *src_feas, const float *rot_trans,
                                    const float *depth_values, const SweepParams &p, float *out_var, int out_c8,
                                    hipStream_t st, const unsigned *sel, unsigned *absmax, bool candidate_per_tile = false) {
    const int B = p.B, C = p.C, D = p.D, H = p.H, W = p.W, NV = p.V - 1, depth_mode = p.depth_mode;
    if (C % 16 || C > 64) {
        set_error("mvs_costvol_variance_fwd_f32: C16 features need C in {16,32,48,64}, got %d", C);
        return MVS_EUNSUPPORTED;
    }
    if ((int64_t)H * W >= (1 << 26) || H >= (1 << 23) || W >= (1 << 23) || B > 65535) return MVS_EINVAL;
    const int tiles_x = (W + kTileW - 1) / kTileW, tiles_y = (H + kTileH - 1) / kTileH;
    const int dchunks = (D + kTileD - 1) / kTileD;
    const int64_t nblk = (int64_t)tiles_x * tiles_y * dchunks;
    if (nblk > 0x7fffffffLL) return bare_error(MVS_EINVAL, __func__, __LINE__);
#ifdef MVS_TUNING
    const char *abl_env = getenv("MVS_SWEEP_ABLATE");   // tuning builds only: some values give wrong results
    const int lds_ablate = abl_env ? atoi(abl_env) : 0;
#else
    const int lds_ablate = 0;
#endif
    // Behind the chooser this kernel is a CANDIDATE that usually returns at once (the persistent kernel took the call): as one
    // block per tile its 44,000 early exits cost more than they look -- at config 2 the whole sweep takes 1.27 ms with them
    // and 1.16 ms with a fixed grid of four workgroups per CU looping over the tiles (scripts/exp_sweep_select.py, round 5),
    // which in turn is ~25 % slower where this kernel IS the one chosen (0.77 -> 0.92 ms at 48 planes x4 on the wide rig).
    // So: the looping form by default; a caller that knows the geometry picks this kernel asks for one block per tile
    // (MVS_SWEEP_TILE_CANDIDATE_PER_TILE: the Python mirror reads the chooser's verdict back asynchronously and sets it).
    // MVS_SWEEP_TILE_LOOP = <workgroups per CU> overrides (0: one block per tile always).
    static const int tile_loop = [] { const char *e = getenv("MVS_SWEEP_TILE_LOOP"); return e ? atoi(e) : 4; }();
    const bool loop = sel && tile_loop > 0 && !candidate_per_tile;
    const int loop_wgs = (tile_loop == 1 ? 3 : tile_loop) * device_cu_count();
    // (behind the chooser, sel != NULL, the chooser has cleared the word)
    if (!sel && absmax && launch_zero_words(absmax, kAbsmaxWords, st) != MVS_OK) return MVS_ELAUNCH;
    const dim3 g(loop ? (unsigned)(nblk < loop_wgs ? nblk : loop_wgs) : (unsigned)nblk, (unsigned)B);
#define MVS_LDS_CASE(n)                                                                                      \
    case n: {                                                                                                \
        const size_t shmem = (size_t)n * 4 * dma_cap(n) * 16;                                                \
        if (loop)                                                                                            \
            hipLaunchKernelGGL((variance_fwd_dma_kernel<n, true, 4, true>), g, dim3(256), shmem, st,         \
                               ref_fea, src_feas, rot_trans, depth_values, p, tiles_x,                       \
                               tiles_y, out_var, out_c8, lds_ablate, sel, (int)nblk, absmax);                        \
        else if (depth_mode == 0 && !(lds_ablate & 32))                                                      \
            hipLaunchKernelGGL((variance_fwd_dma_kernel<n, true>), g, dim3(256), shmem, st,                  \
                               ref_fea, src_feas, rot_trans, depth_values, p, tiles_x,                       \
                               tiles_y, out_var, out_c8, lds_ablate, sel, 0, absmax);                         \
        else                                                                                                 \
            hipLaunchKernelGGL((variance_fwd_dma_kernel<n, false>), g, dim3(256), shmem, st,                 \
                               ref_fea, src_feas, rot_trans, depth_values, p, tiles_x,                       \
                               tiles_y, out_var, out_c8, lds_ablate, sel, 0, absmax);                         \
        break;                                                                                               \
    }
    switch (NV) {
        MVS_LDS_CASE(1) MVS_LDS_CASE(2) MVS_LDS_CASE(3) MVS_LDS_CASE(4) MVS_LDS_CASE(5)
        MVS_LDS_CASE(6) MVS_LDS_CASE(7) MVS_LDS_CASE(8)
    }
#undef MVS_LDS_CASE
    return check_launch("mvs_costvol_variance_fwd_f32(lds)");
}

// [N, C/4, H*W, 4] -> [N, C/16, H*W, 16] (the per-tile kernel's layout), only when the chooser asked for that kernel
// (the reference view's ng_ref groups, then the source views' ng_src, into one contiguous scratch)
__global__ __launch_bounds__(256) void c4_to_c16_sel_kernel(const float4 *__restrict__ ref, const float4 *__restrict__ src,
                                                            float4 *__restrict__ out, int ng_ref, int ng_src, int plane,
                                                            const unsigned *__restrict__ sel) {
    if (sel && *sel != 0u) return;
    const int64_t total = (int64_t)(ng_ref + ng_src) * plane * 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int q = (int)(i & 3);
        const int64_t gp = i >> 2, g = gp / plane, pix = gp - g * plane;
        out[i] = g < ng_ref ? ref[(g * 4 + q) * plane + pix] : src[((g - ng_ref) * 4 + q) * plane + pix];
    }
}

extern "C" int mvs_absmax_f32(const float *x, int64_t n, void *absmax_bits, void *stream);   // conv_f16x3.hip

// absmax: NULL, or the absmax block that receives the largest |variance| (collected by the LDS-staged
// kernels as they store; one more pass over the volume behind the gather kernels)
static int variance_fwd_impl(const float *ref_fea, const float *src_feas,
                             const float *rot_trans, const float *depth_values,
                             int depth_mode, int B, int V, int C, int D, int H,
                             int W, int align_corners, int alias_quirk,
                             int fea_layout, int out_layout, float *out_var, unsigned *absmax,
                             void *stream) {
    const int NV = V - 1;
    const int64_t nvol = (int64_t)B * D * H * W * C;
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
    const bool fea_cl = fea_layout == MVS_LAYOUT_NHWC || fea_layout == MVS_LAYOUT_C16;
    if (fea_layout != out_layout && !(fea_cl && (out_c8 || out_layout == MVS_LAYOUT_NHWC))) {
        set_error("mvs_costvol_variance_fwd_f32: unsupported layout pair fea=%d out=%d", fea_layout,
                  out_layout);
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
        if (absmax) return mvs_absmax_f32(out_var, nvol, absmax, stream);
        return check_launch("mvs_costvol_variance_fwd_f32(planar)");
    }
    if (fea_layout == MVS_LAYOUT_C16)
        return launch_variance_tile_c16(ref_fea, src_feas, rot_trans, depth_values, p, out_var, out_c8, st, nullptr, absmax);
    if (fea_layout != MVS_LAYOUT_NHWC) return bare_error(MVS_EINVAL, __func__, __LINE__);
    if (C == 8 && (int64_t)H * W < (1 << 26) && H < (1 << 23) && W < (1 << 23) && B <= 65535 &&
        !getenv("MVS_SWEEP_C8_GATHER")) {
        // 8-channel maps (the cascade's finest stage): the LDS-staged kernel with one group of two quads
        const int tiles_x = (W + kTileW - 1) / kTileW, tiles_y = (H + kTileH - 1) / kTileH;
        const int dchunks = (D + kTileD - 1) / kTileD;
        const int64_t nblk = (int64_t)tiles_x * tiles_y * dchunks;
        if (nblk > 0x7fffffffLL) return bare_error(MVS_EINVAL, __func__, __LINE__);
        if (absmax && launch_zero_words(absmax, kAbsmaxWords, st) != MVS_OK) return MVS_ELAUNCH;
        const dim3 g((unsigned)nblk, (unsigned)B);
#define MVS_LDS8_CASE(n)                                                                          \
    case n: {                                                                                     \
        const size_t shmem = (size_t)n * 2 * dma_cap(n, 2) * 16;                                  \
        if (depth_mode == 0)                                                                      \
            hipLaunchKernelGGL((variance_fwd_dma_kernel<n, true, 2>), g, dim3(256), shmem, st,    \
                               ref_fea, src_feas, rot_trans, depth_values, p, tiles_x, tiles_y,   \
                               out_var, out_c8, 0, (const unsigned *)nullptr, 0, absmax);                 \
        else                                                                                      \
            hipLaunchKernelGGL((variance_fwd_dma_kernel<n, false, 2>), g, dim3(256), shmem, st,   \
                               ref_fea, src_feas, rot_trans, depth_values, p, tiles_x, tiles_y,   \
                               out_var, out_c8, 0, (const unsigned *)nullptr, 0, absmax);                 \
        break;                                                                                    \
    }
        switch (NV) {
            MVS_LDS8_CASE(1) MVS_LDS8_CASE(2) MVS_LDS8_CASE(3) MVS_LDS8_CASE(4) MVS_LDS8_CASE(5)
            MVS_LDS8_CASE(6) MVS_LDS8_CASE(7) MVS_LDS8_CASE(8)
        }
#undef MVS_LDS8_CASE
        return check_launch("mvs_costvol_variance_fwd_f32(lds, 8 channels)");
    }
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
    if (absmax) return mvs_absmax_f32(out_var, nvol, absmax, stream);
    return check_launch("mvs_costvol_variance_fwd_f32(channels-last)");
}

extern "C" int mvs_costvol_variance_fwd_f32(const float *ref_fea, const float *src_feas,
                                            const float *rot_trans, const float *depth_values,
                                            int depth_mode, int B, int V, int C, int D, int H,
                                            int W, int align_corners, int alias_quirk,
                                            int fea_layout, int out_layout, float *out_var,
                                            void *stream) {
    return variance_fwd_impl(ref_fea, src_feas, rot_trans, depth_values, depth_mode, B, V, C, D, H, W, align_corners,
                             alias_quirk, fea_layout, out_layout, out_var, nullptr, stream);
}

// Which sweep kernel: by default the DEVICE decides per call from the geometry (variance_choose_kernel,
// sweep_persist.hip): 16-plane tiles of the persistent kernel, 8-plane tiles, or the per-tile kernel above.  Round 2
// keyed this on the plane count (D >= 160 -> 16, D >= 72 -> 8), which assumed that fewer planes mean a coarser
// interval; a caller sweeping 192 planes at a x4 interval then ran 2.4x slower than the best kernel.
// MVS_SWEEP_PERSIST = 16 | 8 | 0 forces one of them (A/B runs).  Tuning builds (-DMVS_TUNING) also read
// "<waves>,<flags>,<quads>" -- the ablation flags give wrong results and are not reachable in a release build.
static int persist_forced(int *flags, int *quads) {
    *flags = 0;
    *quads = 2;
    const char *pe = getenv("MVS_SWEEP_PERSIST");
    if (!pe) return -1;
    const int nw = atoi(pe);
#ifdef MVS_TUNING
    const char *c = strchr(pe, ',');
    if (c) {
        *flags = atoi(c + 1);
        c = strchr(c + 1, ',');
        if (c) *quads = atoi(c + 1);
    }
#endif
    return (nw == 16 || nw == 8) ? nw : 0;
}

// ONE predicate for the workspace query and the launcher: the persistent kernel takes shared depth
// planes, no CVP alias quirk, C % 16 == 0, and the size limits of variance_persist_shape_ok
static bool persist_takes(const SweepParams &p, int fea_layout) {
    return (fea_layout == MVS_LAYOUT_C16 || fea_layout == MVS_LAYOUT_C4 || fea_layout == MVS_LAYOUT_NHWC) &&
           variance_persist_shape_ok(p);
}

static size_t c16_scratch_bytes(const SweepParams &p) { return (size_t)p.V * p.B * p.C * p.H * p.W * 4; }

extern "C" size_t mvs_costvol_variance_workspace_bytes2(int depth_mode, int B, int V, int C, int D, int H, int W,
                                                        int fea_layout, int alias_quirk) {
    if (B <= 0 || D <= 0 || H <= 1 || W <= 1 || V < 2) return 0;
    const SweepParams p = make_params(B, V, C, D, H, W, depth_mode, 0, alias_quirk);
    if (!persist_takes(p, fea_layout)) return 0;
    int flags, quads;
    if (persist_forced(&flags, &quads) == 0) return 0;
    const size_t q = variance_persist_workspace_bytes(p, 8) > variance_persist_workspace_bytes(p, 16)
                         ? variance_persist_workspace_bytes(p, 8) : variance_persist_workspace_bytes(p, 16);
    // C4 features: room for their 16-channel-blocked copy, should the geometry ask for the per-tile kernel
    return ((q + 255) & ~(size_t)255) + (fea_layout == MVS_LAYOUT_C4 ? c16_scratch_bytes(p) : 0);
}

extern "C" size_t mvs_costvol_variance_workspace_bytes(int depth_mode, int B, int V, int C, int D,
                                                       int H, int W, int fea_layout) {
    return mvs_costvol_variance_workspace_bytes2(depth_mode, B, V, C, D, H, W, fea_layout, 0);
}

// ho: NULL, or the hand-over of the volume as two fp16 pieces per value (sweep_common.h) -- only the device-selected route takes it
static int variance_ws_impl(const float *ref_fea, const float *src_feas, const float *rot_trans, const float *depth_values,
                            int depth_mode, int B, int V, int C, int D, int H, int W, int align_corners, int alias_quirk,
                            int fea_layout, int out_layout, int flags, float *out_var, void *workspace,
                            size_t workspace_bytes, void *absmax_bits, void *stream, const SweepHandover *ho) {
    const bool c4 = fea_layout == MVS_LAYOUT_C4;
    unsigned *const absmax = static_cast<unsigned *>(absmax_bits);
    hipStream_t st = as_stream(stream);
    if (ref_fea && src_feas && rot_trans && depth_values && out_var && B > 0 && D > 0 && H > 1 && W > 1 && V >= 2 &&
        (out_layout == MVS_LAYOUT_C8 || out_layout == MVS_LAYOUT_NHWC)) {
        const SweepParams p = make_params(B, V, C, D, H, W, depth_mode, align_corners, alias_quirk);
        int tune, quads;
        const int forced = persist_forced(&tune, &quads);
        if (ho && !(persist_takes(p, fea_layout) && forced < 0 && out_layout == MVS_LAYOUT_C8 && absmax)) {
            set_error("mvs_costvol_variance_fwd_ws3_f32: the hand-over rides on the device-selected persistent sweep (shared depth planes, "
                      "no alias quirk, mvs_costvol_variance_workspace_bytes2 > 0, no forced kernel), an MVS_LAYOUT_C8 volume and its absmax block");
            return MVS_EUNSUPPORTED;
        }
        if (persist_takes(p, fea_layout) && forced != 0) {
            const int out_c8 = out_layout == MVS_LAYOUT_C8, lay = c4 ? 1 : (fea_layout == MVS_LAYOUT_NHWC ? 2 : 0);
            if (forced > 0) {
                const int rc = launch_variance_persist(ref_fea, src_feas, rot_trans, depth_values, p, out_var, out_c8, lay,
                                                       flags & MVS_SWEEP_FAST, forced, quads, tune, workspace, workspace_bytes, st, 0, absmax);
                if (rc == MVS_OK) return check_launch("mvs_costvol_variance_fwd_ws_f32(persistent)");
                if (rc != MVS_EUNSUPPORTED) return rc;
            } else {
                const size_t need = mvs_costvol_variance_workspace_bytes2(depth_mode, B, V, C, D, H, W, fea_layout, alias_quirk);
                if (!workspace || workspace_bytes < need) {
                    set_error("mvs_costvol_variance_fwd_ws_f32: workspace of %zu bytes, need %zu", workspace_bytes, need);
                    return MVS_EWORKSPACE;
                }
                // the per-tile kernel reads 16-channel blocks: directly (C16), or from a copy made only if it is chosen (C4);
                // channels-last maps choose between the two tile depths of the persistent kernel only
                const int allow_tile = fea_layout != MVS_LAYOUT_NHWC && B <= 65535;
                int rc = launch_variance_choose(rot_trans, depth_values, p, allow_tile, absmax, workspace, st, ho);
                if (rc != MVS_OK) return rc;
#ifdef MVS_TUNING
                static const int tflags = [] { const char *e = getenv("MVS_HANDOVER_FLAGS"); return e ? atoi(e) : 0; }();
#else
                constexpr int tflags = 0;
#endif
                for (int nw = 16; nw >= 8; nw -= 8) {   // (autosel 2: no cold-path launch behind the first candidate)
                    rc = launch_variance_persist(ref_fea, src_feas, rot_trans, depth_values, p, out_var, out_c8, lay,
                                                 flags & MVS_SWEEP_FAST, nw, 2, ho ? tflags : 0, workspace, workspace_bytes, st, nw == 16 ? 2 : 1, absmax, ho);
                    if (rc != MVS_OK) return rc == MVS_EUNSUPPORTED ? bare_error(MVS_ELAUNCH, __func__, __LINE__) : rc;
                }
                if (allow_tile) {
                    const unsigned *sel = static_cast<const unsigned *>(workspace) + 1;
                    const float *r16 = ref_fea, *s16 = src_feas;
                    if (c4) {
                        float *scratch = reinterpret_cast<float *>(static_cast<char *>(workspace) + (need - c16_scratch_bytes(p)));
                        const int plane = H * W, nmaps = B * (C / 16);
                        // maps: the reference view's B maps, then the source views' (V-1) * B
                        hipLaunchKernelGGL(c4_to_c16_sel_kernel, dim3(2048), dim3(256), 0, st, reinterpret_cast<const float4 *>(ref_fea),
                                           reinterpret_cast<const float4 *>(src_feas), reinterpret_cast<float4 *>(scratch), nmaps,
                                           nmaps * (V - 1), plane, sel);
                        r16 = scratch; s16 = scratch + (size_t)B * C * plane;
                    }
                    rc = launch_variance_tile_c16(r16, s16, rot_trans, depth_values, p, out_var, out_c8, st, sel, absmax,
                                                  (flags & MVS_SWEEP_TILE_CANDIDATE_PER_TILE) != 0);
                    if (rc != MVS_OK) return rc;
                }
                if (ho) {   // the referee (sweep_common.h): writes *redo; returns at once unless the pieces do not hold
                    rc = launch_variance_redo_all(ref_fea, src_feas, rot_trans, depth_values, p, out_var, lay, flags & MVS_SWEEP_FAST,
                                                  workspace, st, *ho, absmax);
                    if (rc != MVS_OK) return rc;
                }
                return check_launch("mvs_costvol_variance_fwd_ws_f32(device-selected)");
            }
        }
    }
    if (ho) {
        set_error("mvs_costvol_variance_fwd_ws3_f32: invalid argument");
        return MVS_EINVAL;
    }
    if (c4) {
        set_error("mvs_costvol_variance_fwd_ws_f32: C4 features are the persistent kernel's layout "
                  "(shared depth planes, no alias quirk, mvs_costvol_variance_workspace_bytes2 > 0); use C16 for this shape");
        return MVS_EUNSUPPORTED;
    }
    return variance_fwd_impl(ref_fea, src_feas, rot_trans, depth_values, depth_mode, B, V, C,
                             D, H, W, align_corners, alias_quirk, fea_layout, out_layout,
                             out_var, absmax, stream);
}

extern "C" int mvs_costvol_variance_fwd_ws2_f32(const float *ref_fea, const float *src_feas,
                                                const float *rot_trans, const float *depth_values,
                                                int depth_mode, int B, int V, int C, int D, int H,
                                                int W, int align_corners, int alias_quirk,
                                                int fea_layout, int out_layout, int flags,
                                                float *out_var, void *workspace,
                                                size_t workspace_bytes, void *absmax_bits, void *stream) {
    return variance_ws_impl(ref_fea, src_feas, rot_trans, depth_values, depth_mode, B, V, C, D, H, W, align_corners, alias_quirk,
                            fea_layout, out_layout, flags, out_var, workspace, workspace_bytes, absmax_bits, stream, nullptr);
}

extern "C" size_t mvs_costvol_variance_handover_bytes(int B, int C, int D, int H, int W) {
    if (B <= 0 || C <= 0 || C % 8 || D <= 0 || H <= 0 || W <= 0) return 0;
    // 0 = no hand-over for this shape: conv0 on the pieces addresses three planes of a window with 32-bit offsets
    // (mvs_conv3d_c8p_f16x3_f32), the sweep one plane
    if (3 * pairs_geom(C, H, W, kPairsLayoutStrips).plane >= 0xffffff00LL) return 0;
    const size_t pairs = (size_t)B * D * pairs_geom(C, H, W, kPairsLayoutStrips).plane, f32 = (size_t)B * D * H * W * C * 4;
    return pairs > f32 ? pairs : f32;
}

extern "C" int mvs_costvol_variance_fwd_ws3_f32(const float *ref_fea, const float *src_feas, const float *rot_trans,
                                                const float *depth_values, int B, int V, int C, int D, int H, int W,
                                                int align_corners, int fea_layout, int flags, const void *fea_absmax,
                                                const void *reader_veto, void *out_volume, void *workspace, size_t workspace_bytes, void *var_absmax,
                                                void *hand, void *redo, void *stream) {
    if (mvs_costvol_variance_handover_bytes(B, C, D, H, W) == 0) {
        set_error("mvs_costvol_variance_fwd_ws3_f32: no hand-over for this shape (mvs_costvol_variance_handover_bytes = 0)");
        return MVS_EUNSUPPORTED;
    }
    if (!fea_absmax || !hand || !redo || !var_absmax || !out_volume) {
        set_error("mvs_costvol_variance_fwd_ws3_f32: needs the feature maps' absmax block, the volume (mvs_costvol_variance_handover_bytes), "
                  "its absmax block, the hand-over block and the redo word");
        return MVS_EINVAL;
    }
    SweepHandover ho;
    ho.fea_absmax = static_cast<const unsigned *>(fea_absmax);
    ho.hand = static_cast<unsigned *>(hand);
    ho.redo = static_cast<unsigned *>(redo);
    ho.veto = static_cast<const float *>(reader_veto);
    ho.layout = kPairsLayoutStrips;
#ifdef MVS_TUNING   // scripts/exp_handover.py: the other layouts (conv0 on them is not wired), ablation flags of the persistent kernel
    static const int lay = [] { const char *e = getenv("MVS_HANDOVER_LAYOUT"); return e ? atoi(e) : 0; }();
    if (pairs_layout_ok(lay)) ho.layout = lay;
#endif
    return variance_ws_impl(ref_fea, src_feas, rot_trans, depth_values, 0, B, V, C, D, H, W, align_corners, 0, fea_layout,
                            MVS_LAYOUT_C8, flags, static_cast<float *>(out_volume), workspace, workspace_bytes, var_absmax, stream, &ho);
}

extern "C" int mvs_costvol_variance_fwd_ws_f32(const float *ref_fea, const float *src_feas,
                                               const float *rot_trans, const float *depth_values,
                                               int depth_mode, int B, int V, int C, int D, int H,
                                               int W, int align_corners, int alias_quirk,
                                               int fea_layout, int out_layout, int flags,
                                               float *out_var, void *workspace,
                                               size_t workspace_bytes, void *stream) {
    return mvs_costvol_variance_fwd_ws2_f32(ref_fea, src_feas, rot_trans, depth_values, depth_mode, B, V, C, D, H, W,
                                            align_corners, alias_quirk, fea_layout, out_layout, flags, out_var, workspace,
                                            workspace_bytes, nullptr, stream);
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
    SweepParams p = make_params(B, V, C, D, H, W, depth_mode, align_corners, 0);
    hipStream_t st = as_stream(stream);
    if (fea_layout == MVS_LAYOUT_C16 && out_layout == MVS_LAYOUT_NHWC) {
        // channels-last training form: LDS-accumulated footprints (shared depth planes)
        if (depth_mode != 0 || C % 16 || C > 64) {
            set_error("mvs_costvol_variance_bwd_f32: the C16 form needs [B,D] depth planes and C in {16,32,48,64}");
            return MVS_EUNSUPPORTED;
        }
        const size_t fb = sizeof(float) * (size_t)B * C * H * W;
        if (hipMemsetAsync(grad_ref, 0, fb, st) != hipSuccess ||
            hipMemsetAsync(grad_srcs, 0, fb * NV, st) != hipSuccess)
            return check_launch("mvs_costvol_variance_bwd_f32 memset");
        const int tiles_x = (W + kTileW - 1) / kTileW, tiles_y = (H + kTileH - 1) / kTileH;
        const int dchunks = (D + kTileD - 1) / kTileD;
        const int64_t nblk = (int64_t)tiles_x * tiles_y * dchunks;
        if (nblk > 0x7fffffffLL || B > 65535 || (int64_t)H * W >= (1 << 26) || H >= (1 << 23) || W >= (1 << 23))
            return bare_error(MVS_EINVAL, __func__, __LINE__);
        const dim3 g((unsigned)nblk, (unsigned)B);
#define MVS_BWD_CASE(n)                                                                             \
    case n: {                                                                                       \
        hipLaunchKernelGGL((variance_bwd_dma_kernel<n>), g, dim3(256), 0, st, grad_var, ref_fea,    \
                           src_feas, rot_trans, depth_values, p, tiles_x, tiles_y, grad_ref,        \
                           grad_srcs);                                                              \
        break;                                                                                      \
    }
        switch (NV) {
            MVS_BWD_CASE(1) MVS_BWD_CASE(2) MVS_BWD_CASE(3) MVS_BWD_CASE(4) MVS_BWD_CASE(5)
            MVS_BWD_CASE(6) MVS_BWD_CASE(7) MVS_BWD_CASE(8)
        }
#undef MVS_BWD_CASE
        return check_launch("mvs_costvol_variance_bwd_f32(c16)");
    }
    if (fea_layout != MVS_LAYOUT_NCHW || out_layout != MVS_LAYOUT_NCHW) {
        set_error("mvs_costvol_variance_bwd_f32: layouts are NCHW/NCHW or C16 features with an NHWC volume");
        return MVS_EUNSUPPORTED;
    }
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
