// Pieces shared by the plane-sweep translation units (sweep.hip, sweep_persist.hip).
#pragma once
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
// tiny (subnormal-range quotient) or non-finite (inf*rV would poison the FMA)
__device__ __forceinline__ bool div_views_tiny(float x) {
    const float ax = fabsf(x);
    return !(ax >= 1e-30f && ax <= 3.0e38f);
}
__device__ __forceinline__ float div_views(float x, float fV, float rV) {
    return div_views_tiny(x) ? x / fV : div_views_fast(x, fV, rV);
}

// 16 channels of one view: bilinear blend of the four taps, then S += w, Q += w*w
// (mvsnet.py:164-165).  Four channels at a time so at most 4 float4 loads are live.
template <int NQ = 4>
__device__ __forceinline__ void accumulate_taps(const float *__restrict__ t00,
                                                const float *__restrict__ t01,
                                                const float *__restrict__ t10,
                                                const float *__restrict__ t11, float wnw, float wne,
                                                float wsw, float wse, float (&S)[4 * NQ],
                                                float (&Q)[4 * NQ]) {
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
        const float4 a = reinterpret_cast<const float4 *>(t00)[k];
        const float4 bq = reinterpret_cast<const float4 *>(t01)[k];
        const float4 c = reinterpret_cast<const float4 *>(t10)[k];
        const float4 e = reinterpret_cast<const float4 *>(t11)[k];
        const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {bq.x, bq.y, bq.z, bq.w};
        const float cv[4] = {c.x, c.y, c.z, c.w}, ev[4] = {e.x, e.y, e.z, e.w};
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            const float w = __fmaf_rn(ev[cc], wse, __fmaf_rn(cv[cc], wsw,
                                      __fmaf_rn(bv[cc], wne, av[cc] * wnw)));
            S[k * 4 + cc] = S[k * 4 + cc] + w;
            Q[k * 4 + cc] = Q[k * 4 + cc] + w * w;
        }
        if (k == 1) __builtin_amdgcn_sched_barrier(0);   // 8 tap loads in flight, not 16
    }
}

inline SweepParams make_params(int B, int V, int C, int D, int H, int W, int depth_mode,
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

inline bool grid_for(int64_t total, int per_block, unsigned &grid) {
    int64_t g = (total + per_block - 1) / per_block;
    if (g <= 0 || g > 0x7fffffffLL) return false;
    grid = (unsigned)g;
    return true;
}


// ---- hand-over of the variance volume as two fp16 pieces per value (round 6; conv_f16x3_y8p.hip has the kernel that reads it and
// the layouts): byte strides of a pairs volume
struct PairsGeom {
    int64_t plane;          // between depth planes
    int64_t chunk;          // between 8-channel chunks of a plane
    int64_t xtile;          // C8PT: between 32-voxel x tiles (0 otherwise)
    int64_t region;         // between (part, parity) blocks: H * rowpitch
    int64_t halo;           // C8PH: offset of the halo strips inside a chunk's block (behind its four regions)
    int64_t halo_part;      // C8PH: between the hi and the lo strips
    int64_t dup_lo;         // from the hi piece of a halo COPY to its lo piece (C8PT: 2 * region, C8PH: halo_part)
    int rowpitch;           // between rows of a block
    int strip;              // C8PH: bytes of one strip = H pieces
    int tiled;              // 1 = C8PT
    int strips;             // 1 = C8PH
};
// = MVS_LAYOUT_C8P / MVS_LAYOUT_C8PT / MVS_LAYOUT_C8PH (include/mvs_hip.h)
constexpr int kPairsLayoutRows = 6, kPairsLayoutTiled = 7, kPairsLayoutStrips = 8;
__host__ __device__ inline bool pairs_layout_ok(int layout) { return layout >= kPairsLayoutRows && layout <= kPairsLayoutStrips; }
__host__ __device__ inline PairsGeom pairs_geom(int C, int H, int W, int layout) {
    PairsGeom g;
    g.tiled = layout == kPairsLayoutTiled;
    g.strips = layout == kPairsLayoutStrips;
    g.rowpitch = g.tiled ? 17 * 16 : ((W + 1) / 2) * 16;
    g.region = (int64_t)H * g.rowpitch;
    g.xtile = g.tiled ? 4 * g.region : 0;
    g.strip = H * 16;
    g.halo = 4 * g.region;
    g.halo_part = g.strips ? (int64_t)((W + 31) / 32) * 2 * g.strip : 0;
    g.dup_lo = g.tiled ? 2 * g.region : g.halo_part;
    g.chunk = g.tiled ? (int64_t)((W + 31) / 32) * g.xtile : ((4 * g.region + 2 * g.halo_part + 255) & ~(int64_t)255);
    g.plane = (int64_t)(C / 8) * g.chunk;
    return g;
}
// What the sweep kernels need to hand the volume over (all device pointers; hand == NULL: the fp32 volume as ever).
//   fea_absmax  in: absmax block of ALL feature maps (reference + sources).  var = E[x^2] - E[x]^2 <= max |f|^2: the scale of
//               the pieces comes from that BOUND, known before the sweep runs (the volume's true maximum is not).
//   hand        the chooser kernel fills its 256 words with the bits of the bound when the volume is to be written as pairs
//               (a persistent candidate was chosen and the bound is finite), else with a NaN pattern = "fp32 MVS_LAYOUT_C8";
//               every later kernel decides on word 0 / the block's maximum.
//   redo        one word, written by the launch that follows the candidates (launch_variance_redo_all: the cold kernel, which
//               returns at once on a sane volume): 0 = the volume is pairs, 1 = it is fp32 MVS_LAYOUT_C8 -- because the chooser
//               took the per-tile kernel, or because the pieces do not hold (the volume's TRUE absmax block is not finite or
//               outlier-dominated, or the bound lies more than kHandoverLooseBits above it) and that launch has computed the
//               volume again in fp32.  conv0 on the pairs runs only if it is 0, the fp32 conv0 behind it only if it is 1.
struct SweepHandover {
    const unsigned *fea_absmax = nullptr;
    unsigned *hand = nullptr;
    unsigned *redo = nullptr;
    const float *veto = nullptr;   // NULL, or one float of the volume's reader: NaN there = "I cannot take pieces" (conv0's pack leaves a
                                   // NaN in its scale word when the layer's weights are not finite) -- the volume then leaves as fp32
    int layout = 0;           // kPairsLayoutRows / kPairsLayoutTiled / kPairsLayoutStrips
    int redo_all = 0;         // cold kernel only: decide *redo and, if the pieces do not hold, serve EVERY (tile, wave) in fp32
};
__device__ __forceinline__ bool hand_is_pairs(unsigned bits) { return bits < 0x7f800000u; }
// how many binary orders the a-priori bound may lie above the volume's true maximum before the pieces are declined: the pieces
// carry 2^-22 relative for values within 2^-18 of the SCALE's maximum and 2^-40 of it absolute below, so k loose bits move
// both marks by k -- at 8 a value within 2^-10 of the true maximum still has its 22 bits, and the absolute floor, 2^-32 of the
// true maximum, stays far below the 2^-22 relative error of the products that dominate a sum
constexpr int kHandoverLooseBits = 8;

// sweep_persist.hip: the persistent kernel for shared depth planes and 16-channel-blocked
// features (+ its cold-path kernel); MVS_EUNSUPPORTED (nothing launched) when the shape is not
// its own.  The workspace holds the cold path's queue.
size_t variance_persist_workspace_bytes(const SweepParams &p, int nw);
bool variance_persist_shape_ok(const SweepParams &p);     // the ONE predicate of the workspace query and the launcher
int launch_variance_persist(const float *ref16, const float *srcs16, const float *rt,
                            const float *depth, const SweepParams &p, float *out, int out_c8,
                            int fea_c4, int fast, int nw, int nq, int flags, void *workspace,
                            size_t workspace_bytes, hipStream_t st, int autosel = 0, unsigned *absmax = nullptr,
                            const SweepHandover *ho = nullptr);
// device-side choice between 16-plane tiles, 8-plane tiles and (allow_tile) the per-tile kernel of sweep.hip from the
// footprints of sample tiles; clears the workspace header (and the absmax block the candidates collect the largest
// |variance| in, if given) and writes the choice into word 1 of the header
int launch_variance_choose(const float *rt, const float *depth, const SweepParams &p, int allow_tile, unsigned *absmax,
                           void *workspace, hipStream_t st, const SweepHandover *ho = nullptr);
// the cold kernel as the hand-over's referee: writes *ho.redo; if the pieces do not hold, every (tile, wave) of the 16-plane tiling again as fp32 MVS_LAYOUT_C8 (the
// never-taken path behind a declined hand-over: global gathers, ~10x the persistent kernel's time)
int launch_variance_redo_all(const float *ref16, const float *srcs16, const float *rt, const float *depth, const SweepParams &p,
                             float *out, int fea_c4, int fast, void *workspace, hipStream_t st, const SweepHandover &ho, unsigned *absmax);

}  // namespace mvs
