// Fused plane-sweep warp + variance for shared depth planes as ONE persistent workgroup
// per CU (MVSNet/models/module.py:46-87 + mvsnet.py:152-170; the kernel north_star names).
//
// The per-tile kernel of sweep.hip (8x8 pixels x 4 planes per 256-thread block) spends two
// thirds of its life waiting: every block computes its footprint boxes, plans and issues its
// copies and then sits on their latency, and it copies 2-3 bytes of footprint per byte it
// stores.  Here a workgroup of 16 waves walks a contiguous list of tiles (16x4 pixels x 16
// depth planes, wave = plane, depth chunk fastest: a CU climbs the depth column of one pixel
// tile, its footprints slide by a fraction of a texel per step and come out of L2):
//
//   stage s = (tile, 8-channel group), LDS buffer s & 1:
//     [last stage of a tile: plan the next tile -- footprint boxes from its 8 corner voxels,
//      per-lane source offsets]
//     s_waitcnt vmcnt(<stores of the previous stage>); s_barrier
//                                  <- stage s has landed; buffer (s+1)&1 is free
//     issue the LDS-DMA of stage s+1 into buffer (s+1)&1   (lands during this stage)
//     blend the four taps of every source view out of buffer s&1, form variances, store
//
// One barrier per stage; inside the loop only the copies and the stores touch vector memory
// (the reference view's pixels ride along as one more DMA piece; camera rows and depth planes
// sit in LDS), so a counted vmcnt waits for the copies without waiting for the HBM write
// latency of the stores issued behind them.  Eight channels per stage (not 16) because a
// 16-plane tile's bounding boxes (~300 texels per view under a diagonal baseline) must fit
// twice: 2 buffers x 4 views x 512 texels x 32 B = 128 KiB.  A lane's four taps are
// ds_read_b128 at box offsets {0, 1, bw, bw + 1} texels: the box is NOT clamped to the image
// (the copy clamps the SOURCE address instead; out-of-image taps carry weight 0), and the lanes
// of one ds_read_b128 service group (MI355X: {0-3,12-15,20-27}, {4-11,16-19,28-31}, +32) own
// 16 pixels of one row, so a group's texels are consecutive and cover all 64 banks.  With
// 4-channel-blocked features (MVS_LAYOUT_C4) a footprint row of one channel quad is contiguous
// in memory and a DMA instruction touches ~19 cache lines instead of 64.
//
// EXACT = the reference's coordinate arithmetic op for op (mvs_common.h) and its
// E[x^2] - E[x]^2 with IEEE divisions: bit-identical to the other variance kernels.
// FAST  = (round 6) the SAME sampling coordinates bit for bit -- the four divisions through shared / precomputed
// refined reciprocals instead of four compiler sequences --, Q accumulated with an FMA, multiplication by 1/V
// (until round 5: one reciprocal for X/Z and Y/Z and the normalise / un-normalise pair folded into one FMA; on a
// TRAINED network those ~3e-5 texel coordinate differences moved the depth by up to 7e-3 mm against the reference,
// profiles/r06_trained_budget_switches.json).
//
// Measured at BASELINE configs[1] (22.7 M voxels x 32 channels x 4 source views): 1.46-1.56 ms
// EXACT, 1.21-1.25 ms FAST against 1.85 ms for the per-tile kernel.  SQ counters (profiles/):
// 1614 VALU instructions per wave-tile-plane FAST (2216 EXACT; per-tile kernel 2103), LDS
// array busy 46 % of the kernel, 17 % of its cycles conflicted (per-tile kernel: 54 %).
// Round 6 (reference coordinates in FAST mode, fp16 pieces + halo strips, arguments read from the kernel-argument segment): 1.42 ms,
// 1658 VALU / 677 scalar / 66 scalar-memory instructions per wave-plane (profiles/r06_sq_counters_sweep_handover.json).
#include "sweep_common.h"
#include "split2.h"
#include "conv_guard.h"

#include <cstdlib>
#include <type_traits>

namespace mvs {

constexpr int kPW = 16, kPH = 4;   // pixels of a tile; a wave owns one depth plane of it

// texels per source view and buffer: two buffers of (nv views x nq channel quads x cap x 16 B +
// nq KiB of reference pixels) inside ~140 KiB (the camera matrices and depth planes take up to
// 12 KiB more), whole 64-lane DMA instructions
__host__ __device__ constexpr int persist_cap(int nv, int nq) {
    const int raw = (140 * 1024 - 2 * nq * 1024) / (nv * nq * 32);
    const int top = 1024 / nq;
    return raw >= top ? top : (raw / 64) * 64;
}
constexpr int kPMaxCamFloats = 512, kPMaxDepthFloats = 1024;   // LDS copies of rot_trans / depth_values

struct PersistArgs {
    const float *ref16, *srcs16, *rt, *depth;
    float *out;
    unsigned *queue;        // workspace: [0] = records, [1..] = (tile << 4 | wave) for the cold kernel
    SweepParams p;
    int tiles_x, tiles_y, nchunks, total_tiles;
    int nseg, cps;          // a pixel tile's depth chunks are walked in nseg segments of cps chunks (the work units)
    int out_c8, flags;
    int fea_c4;             // features: 0 = [B,C/16,H,W,16], 1 = [B,C/4,H,W,4], 2 = [B,H,W,C]
    float sx, ox, sy, oy;   // footprint planning (chooser, DMA boxes): ix ~ (X/Z) * sx + ox
    float rhw, rhh;         // FAST: correctly rounded 1 / ((W-1)/2), 1 / ((H-1)/2) (mvs_common.h: sweep_coord_shared)
    int autosel;            // 1: run only if queue[kSelWord] names this kernel's tile depth (variance_choose_kernel)
    unsigned *absmax;       // NULL, or the absmax block (mvs_common.h) that collects the largest |variance| written
                            // (atomic max; the operand scale of mvs_conv3d_c8_f16x3_f32)
    SweepHandover ho;       // ho.hand != NULL: the volume may leave as two fp16 pieces per value (sweep_common.h)
    PairsGeom pg;           // ... in this layout; `out` then holds max(fp32, pairs) bytes
};
// Byte offset of a voxel's hi piece inside its (plane, chunk) block of a pairs volume (the lo piece: + 2 * pg.region), and of its
// second copy as a halo voxel of the neighbouring 32-voxel tile (0xffffffff: none; its lo piece: + pg.dup_lo) -- inside that tile's
// block (x-tiled layout) or in its halo strip (strips layout).
__device__ __forceinline__ void pairs_offsets(const PairsGeom &pg, int x, int y, int W, unsigned &main, unsigned &dup) {
    dup = 0xffffffffu;
    if (!pg.tiled) {
        main = (unsigned)((x & 1) * pg.region + (int64_t)y * pg.rowpitch + (x >> 1) * 16);
        if (pg.strips) {
            const int tx = x >> 5, xm = x & 31;
            if (xm == 31 && x + 1 < W) dup = (unsigned)(pg.halo + (int64_t)((tx + 1) * 2) * pg.strip + y * 16);        // left halo of tile tx + 1
            else if (xm == 0 && x > 0) dup = (unsigned)(pg.halo + (int64_t)((tx - 1) * 2 + 1) * pg.strip + y * 16);    // right halo of tile tx - 1
        }
        return;
    }
    const int tx = x >> 5, xl = (x & 31) + 1;     // local x' = 1 .. 32 of the tile's 0 .. 33
    main = (unsigned)(tx * pg.xtile + (xl & 1) * pg.region + (int64_t)y * pg.rowpitch + (xl >> 1) * 16);
    if (xl == 32 && (tx + 1) * 32 < W) dup = (unsigned)((tx + 1) * pg.xtile + (int64_t)y * pg.rowpitch);                      // x' = 0 of tile tx + 1
    else if (xl == 1 && tx > 0) dup = (unsigned)((tx - 1) * pg.xtile + pg.region + (int64_t)y * pg.rowpitch + 16 * 16);      // x' = 33 of tile tx - 1
}
// workspace header (32-bit words): [0] cold-path records, [1] the chosen tile depth (16, 8, or 0 = the per-tile
// kernel), [2..7] what the choice was made from (largest / mean footprint box of 16- and 8-plane tiles, texels; box
// samples), records from word 8
constexpr int kSelWord = 1, kQueueHdr = 8;
constexpr int kPFlagLinearLanes = 1;   // tuning: lane = (x = lane & 15, y = lane >> 4)
constexpr int kPFlagNoStore = 2;       // tuning
constexpr int kPFlagNoBlend = 4;       // tuning
constexpr int kPFlagNoDma = 8;         // tuning (results are garbage)
constexpr int kPFlagNoTaps = 32;       // tuning: constant tap set (garbage)
constexpr int kPFlagContiguous = 64;   // tuning: one contiguous, pixel-tile-major unit range per CU (round 2's first schedule)
constexpr int kPFlagNoDup = 128;       // tuning: x-tiled pairs without the halo copies (wrong results at tile borders)

typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int OFF>
__device__ __forceinline__ f32x4 lds_rd16(unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536 && OFF % 16 == 0, "ds_read_b128 offset field");
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
// wait until at most N LDS operations are outstanding (follow with an empty asm "+v" on the
// registers the landed reads wrote, so their consumers stay behind the wait)
template <int N>
__device__ __forceinline__ void lds_wait_n() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// Homography + bilinear tap set of one voxel in one source view.  EXACT (FAST = false) is the
// reference's arithmetic op for op (mvs_common.h); both kernels below call this one function, so
// a voxel gets the same taps whichever of them serves it.  Tap weights come pre-masked: 0 for a
// tap outside the image, NaN for a non-finite coordinate (the reference's 0 * NaN); tx0, ty0 =
// floor of the sampling coordinate clamped to [-4, size + 4]; has = some tap lies in the image.
// SLOW_OK = false (the persistent kernel): a FAST wave whose Z is not a normal number does NOT take the compiler's divisions
// here -- it reports `unsafe` and is handed to the cold kernel like a wave whose footprint does not fit (which calls this
// function with SLOW_OK = true): the never-taken branch with its four division sequences stays out of the hot kernel's code.
template <bool FAST, bool SLOW_OK = true>
__device__ __forceinline__ void tap_setup(const float *__restrict__ r, int cx, int cy, float dv,
                                          const SweepParams &p, float sx, float ox, float sy,
                                          float oy, float &wnw, float &wne, float &wsw, float &wse,
                                          int &tx0, int &ty0, bool &has, bool *unsafe = nullptr) {
    float ix, iy, nw, ne, sw, se;
    bool x0ok, x1ok, y0ok, y1ok, fin;
    // Branch-free on purpose (round 4): the clamp alone decides tx0 / ty0.  fmaxf drops a NaN (-> -4) and clamps +-Inf to an
    // index outside the image, so a non-finite coordinate ends up with no tap in the image and NaN weights exactly as with the
    // earlier `fin ? ... : -4` select -- whose exec-mask branch per view cost ~20 scalar instructions on every voxel.
    if constexpr (FAST) {
        // (round 6) the reference's coordinate arithmetic op for op -- so the taps ARE the reference's -- with its four divisions
        // done through shared / precomputed reciprocals (mvs_common.h: sweep_coord_shared; sx / sy here = the correctly rounded
        // reciprocals of (W-1)/2, (H-1)/2: PersistArgs.rhw / rhh); a wave with an operand outside the safe range takes the compiler's divisions
        float rx, ry, rz;
        sweep_ray(r, (float)cx, (float)cy, rx, ry, rz);
        const float X = rx * dv + r[3], Y = ry * dv + r[7], Z = rz * dv + r[11];
        if constexpr (SLOW_OK) {
            if (__any(!sweep_coord_safe(Z))) sweep_coord(r, rx, ry, rz, dv, p.half_w, p.half_h, p.unn_w, p.unn_h, p.align_corners, ix, iy);
            else sweep_coord_shared(X, Y, Z, p.half_w, p.half_h, sx, sy, p.unn_w, p.unn_h, p.align_corners, ix, iy);
        } else {
            *unsafe = *unsafe | !sweep_coord_safe(Z);
            sweep_coord_shared(X, Y, Z, p.half_w, p.half_h, sx, sy, p.unn_w, p.unn_h, p.align_corners, ix, iy);
        }
        fin = (int)(fabsf(ix) <= 3.0e38f) & (int)(fabsf(iy) <= 3.0e38f);   // (bitwise on purpose: no branch)
        const float x0f = floorf(ix), y0f = floorf(iy);
        const float wx = ix - x0f, wy = iy - y0f, ex = 1.0f - wx, ey = 1.0f - wy;
        tx0 = (int)fminf(fmaxf(x0f, -4.0f), (float)p.W + 4.0f);
        ty0 = (int)fminf(fmaxf(y0f, -4.0f), (float)p.H + 4.0f);
        x0ok = (unsigned)tx0 < (unsigned)p.W; x1ok = (unsigned)(tx0 + 1) < (unsigned)p.W;
        y0ok = (unsigned)ty0 < (unsigned)p.H; y1ok = (unsigned)(ty0 + 1) < (unsigned)p.H;
        // the four masked weights as products of masked one-dimensional factors (the factors lie in [0, 1], so a masked-out tap's
        // weight is +0 like the select's, a non-finite coordinate's NaN): four selects and four products where masking the four
        // products took four ANDs, four compares and four selects more
        const float dead = fin ? 0.0f : __int_as_float(0x7fc00000);
        const float mx0 = x0ok ? ex : dead, mx1 = x1ok ? wx : dead, my0 = y0ok ? ey : dead, my1 = y1ok ? wy : dead;
        wnw = my0 * mx0; wne = my0 * mx1; wsw = my1 * mx0; wse = my1 * mx1;
    } else {
        float rx, ry, rz;
        sweep_ray(r, (float)cx, (float)cy, rx, ry, rz);
        sweep_coord(r, rx, ry, rz, dv, p.half_w, p.half_h, p.unn_w, p.unn_h, p.align_corners, ix, iy);
        const Taps t = make_taps(ix, iy, p.H, p.W);
        fin = (int)(fabsf(ix) <= 3.0e38f) & (int)(fabsf(iy) <= 3.0e38f);   // (bitwise on purpose: no branch)
        nw = t.nw; ne = t.ne; sw = t.sw; se = t.se;
        x0ok = t.x0ok; x1ok = t.x1ok; y0ok = t.y0ok; y1ok = t.y1ok;
        tx0 = (int)fminf(fmaxf(floorf(ix), -4.0f), (float)p.W + 4.0f);
        ty0 = (int)fminf(fmaxf(floorf(iy), -4.0f), (float)p.H + 4.0f);
    }
    if constexpr (!FAST) {
        const float dead = fin ? 0.0f : __int_as_float(0x7fc00000);
        wnw = (x0ok & y0ok) ? nw : dead; wne = (x1ok & y0ok) ? ne : dead;
        wsw = (x0ok & y1ok) ? sw : dead; wse = (x1ok & y1ok) ? se : dead;
    }
    has = (x0ok | x1ok) & (y0ok | y1ok);
}

// Cold path of the persistent kernel.  A wave of it that cannot serve its voxels from LDS -- a
// footprint too large for its LDS share, a corner voxel behind the camera, a tap outside the
// box; never seen with DTU-like rigs -- appends (tile, wave) to a queue in the caller's
// workspace and moves on; this kernel, launched right behind it, drains the queue with global
// gathers (one wave per record, one channel quad at a time; same tap_setup, same blend and
// accumulation order, IEEE divisions in EXACT mode).  An empty queue costs one launch.
template <int NV, bool FAST>
__global__ __launch_bounds__(256) void variance_fwd_cold_kernel(PersistArgs a, int nw) {
    const SweepParams &p = a.p;
    int nchunks = a.nchunks;
    const unsigned hb = a.ho.hand ? a.ho.hand[0] : 0xffffffffu;
    if (a.ho.redo_all) {
        // The launch behind the candidates of a hand-over sweep: it DECIDES (every workgroup alike, from the two blocks) what the volume
        // is for its readers and tells them through *redo -- 0: two fp16 pieces per value; 1: fp32 MVS_LAYOUT_C8.
        //   the chooser picked the per-tile kernel or had no finite bound (hand = NaN pattern): the volume IS fp32 already;
        //   the pieces were written but do not hold (conv_guard.h's verdict on the volume's TRUE maximum, which the candidates
        //   collected as they stored: a NaN / Inf voxel, an outlier-dominated volume; or the bound max|f|^2 lies more than
        //   kHandoverLooseBits above that maximum): this launch computes the whole volume again, in fp32 -- every (tile, wave) of
        //   the 16-plane tiling by global gathers, ~10x the persistent kernel's time, the path of broken inputs.
        const AbsmaxVerdict tv = absmax_verdict(a.absmax);
        const bool was_pairs = hand_is_pairs(hb);
        const bool bad = was_pairs && tv.bits != 0u &&
                         (tv.code != 0 || absmax_exponent(hb) - absmax_exponent(tv.bits) > kHandoverLooseBits);
        if (blockIdx.x == 0 && threadIdx.x == 0) *a.ho.redo = (!was_pairs || bad) ? 1u : 0u;
        if (!bad) return;
    } else if (a.autosel) {   // one launch behind both candidates: the records are the chosen kernel's
        nw = (int)a.queue[kSelWord];
        if (nw == 0) return;
        nchunks = (p.D + nw - 1) / nw;
    }
    const bool pairs = !a.ho.redo_all && hand_is_pairs(hb);
    const float ps = pow2f(14 - absmax_exponent(hb));
    const unsigned count = a.ho.redo_all ? (unsigned)a.total_tiles * (unsigned)nw : a.queue[0];
    const int lane = threadIdx.x & 63;
    const int plane = p.H * p.W, ngroups = p.C >> 4;
    const size_t grp_floats = (size_t)plane * 16, map_floats = (size_t)plane * p.C;
    const unsigned tex = a.fea_c4 == 1 ? 4u : a.fea_c4 == 2 ? (unsigned)p.C : 16u;   // floats between neighbouring texels of one quad
    const float rV = 1.0f / p.fV;
    float vmax = 0.0f;
    for (unsigned rec = blockIdx.x * 4 + (threadIdx.x >> 6); rec < count; rec += gridDim.x * 4) {
        const unsigned q = a.ho.redo_all ? ((rec / (unsigned)nw) << 4 | (rec % (unsigned)nw)) : a.queue[kQueueHdr + rec];
        int t = (int)(q >> 4);
        const int wv = (int)(q & 15u);
        const int dc = t % nchunks; t /= nchunks;
        const int tx = t % a.tiles_x; t /= a.tiles_x;
        const int ty = t % a.tiles_y, b = t / a.tiles_y;
        const int px = tx * kPW + (lane & 15), py = ty * kPH + (lane >> 4), d = dc * nw + wv;
        if (d >= p.D) continue;
        const bool live = px < p.W && py < p.H;
        const int cx = min(px, p.W - 1), cy = min(py, p.H - 1);
        const int pix = cy * p.W + cx;
        const float dv = a.depth[(int64_t)b * p.D + d];
        float wnw[NV], wne[NV], wsw[NV], wse[NV];
        unsigned o00[NV], o01[NV], o10[NV], o11[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            int tx0, ty0;
            bool has;
            tap_setup<FAST>(a.rt + ((int64_t)v * p.B + b) * 12, cx, cy, dv, p, a.rhw, 0.0f, a.rhh, 0.0f,
                            wnw[v], wne[v], wsw[v], wse[v], tx0, ty0, has);
            const int x0c = min(max(tx0, 0), p.W - 1), x1c = min(max(tx0 + 1, 0), p.W - 1);
            const int y0c = min(max(ty0, 0), p.H - 1), y1c = min(max(ty0 + 1, 0), p.H - 1);
            o00[v] = (unsigned)(y0c * p.W + x0c) * tex; o01[v] = (unsigned)(y0c * p.W + x1c) * tex;
            o10[v] = (unsigned)(y1c * p.W + x0c) * tex; o11[v] = (unsigned)(y1c * p.W + x1c) * tex;
        }
        float *pl = a.out + ((size_t)b * p.D + d) * ((size_t)plane * p.C);
        unsigned char *const plp = reinterpret_cast<unsigned char *>(a.out) + ((size_t)b * p.D + d) * a.pg.plane;
        unsigned pmain = 0, pdup = 0xffffffffu;
        if (pairs) pairs_offsets(a.pg, cx, cy, p.W, pmain, pdup);
        float keep[4] = {0.f, 0.f, 0.f, 0.f};      // pairs: the even quad of an 8-channel chunk waits for the odd one
#pragma unroll 1
        for (int gk = 0; gk < ngroups * 4; ++gk) {
            const int g = gk >> 2, k = gk & 3;
            // quad gk of a feature map: C4 = plane gk of [C/4,H,W,4]; C16 = quad k of block g of [C/16,H,W,16]
            const size_t qbase = a.fea_c4 == 1 ? (size_t)gk * plane * 4 : a.fea_c4 == 2 ? (size_t)gk * 4 : (size_t)g * grp_floats + k * 4;
            const float4 r4 = *reinterpret_cast<const float4 *>(
                a.ref16 + (size_t)b * map_floats + qbase + (size_t)pix * tex);
            float S[4] = {r4.x, r4.y, r4.z, r4.w}, Q[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                Q[c] = S[c] * S[c];
                if (p.alias_quirk) S[c] = Q[c];
            }
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const float *base = a.srcs16 + ((size_t)v * p.B + b) * map_floats + qbase;
                const float4 qa = *reinterpret_cast<const float4 *>(base + o00[v]);
                const float4 qb = *reinterpret_cast<const float4 *>(base + o01[v]);
                const float4 qc = *reinterpret_cast<const float4 *>(base + o10[v]);
                const float4 qe = *reinterpret_cast<const float4 *>(base + o11[v]);
                const float av[4] = {qa.x, qa.y, qa.z, qa.w}, bv[4] = {qb.x, qb.y, qb.z, qb.w};
                const float cv[4] = {qc.x, qc.y, qc.z, qc.w}, ev[4] = {qe.x, qe.y, qe.z, qe.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float w = __fmaf_rn(ev[c], wse[v], __fmaf_rn(cv[c], wsw[v],
                                              __fmaf_rn(bv[c], wne[v], av[c] * wnw[v])));
                    S[c] = S[c] + w;
                    if constexpr (FAST) Q[c] = __fmaf_rn(w, w, Q[c]);
                    else Q[c] = Q[c] + w * w;
                }
            }
            float var[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if constexpr (FAST) {
                    const float m = S[c] * rV;
                    var[c] = __fmaf_rn(Q[c], rV, -(m * m));
                } else {
                    const float m = S[c] / p.fV;
                    var[c] = Q[c] / p.fV - m * m;
                }
            }
            if (pairs) {
                if (!(gk & 1)) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) keep[c] = var[c];
                } else if (live && !(a.flags & kPFlagNoStore)) {
                    f32x4 v0 = {keep[0], keep[1], keep[2], keep[3]}, v1 = {var[0], var[1], var[2], var[3]};
                    vmax = amax4_nan(amax4_nan(vmax, keep[0], keep[1], keep[2], keep[3]), var[0], var[1], var[2], var[3]);
                    u32x4 h, l;
                    split2_block(v0, v1, ps, h, l);
                    unsigned char *const ch = plp + (size_t)(gk >> 1) * a.pg.chunk;
                    *reinterpret_cast<u32x4 *>(ch + pmain) = h;
                    *reinterpret_cast<u32x4 *>(ch + pmain + 2 * a.pg.region) = l;
                    if (pdup != 0xffffffffu && !(a.flags & kPFlagNoDup)) {
                        *reinterpret_cast<u32x4 *>(ch + pdup) = h;
                        *reinterpret_cast<u32x4 *>(ch + pdup + a.pg.dup_lo) = l;
                    }
                }
            } else if (live && !(a.flags & kPFlagNoStore)) {
                float *o = a.out_c8
                    ? pl + ((unsigned)(py * (p.C >> 3) + (g * 2 + (k >> 1))) * (unsigned)p.W + (unsigned)px) * 8u + (k & 1) * 4
                    : pl + (unsigned)pix * (unsigned)p.C + (unsigned)(g * 16 + k * 4);
                *reinterpret_cast<float4 *>(o) = make_float4(var[0], var[1], var[2], var[3]);
                vmax = amax4_nan(vmax, var[0], var[1], var[2], var[3]);
            }
        }
    }
    publish_absmax(a.absmax, vmax);
}


// Which kernel serves this geometry?  The host cannot tell: what decides is how far a tile's footprints move
// through the source images across the tile's depth planes -- baselines against the depth range of 16 (or 8)
// consecutive planes -- and cameras and planes live on the device.  One workgroup projects the corner voxels of
// sample tiles (a 3 x 3 grid of pixel tiles x first / middle / last depth chunk) exactly as plan() does, for
// 16-plane and for 8-plane tiles, and writes the choice into the workspace header; the three candidate kernels are
// all enqueued behind it and each begins by reading that word (an early exit costs a few microseconds).
// Rule, from the times of the three kernels over interval scales x1 ... x4, 192 / 96 / 48 planes and two camera rigs
// (scripts/exp_sweep_select.py, profiles/r03_sweep_select.json): 16-plane tiles while the sampled boxes average at most
// 0.31 of a view's LDS share (160 of 512 texels; beyond that the copies grow faster than the blends they feed and the
// near chunks start to go cold), else 8-plane tiles up to 0.36 (184 texels), else the per-tile kernel of sweep.hip.  The
// MEAN decides: the largest box sits at the nearest planes of the image corners and says little about the volume.
// (Round 6: 0.52 -> 0.36.  With the reference's coordinate arithmetic in FAST mode and the cold kernel's share of an 8-plane
// sweep the per-tile kernel now wins wherever the 8-plane boxes average more than ~0.37 of the share -- all four such cases
// of the sweep above, by 5-17 %, CasMVSNet's first stage among them: profiles/r06_sweep_select.json.)
__global__ __launch_bounds__(1024) void variance_choose_kernel(PersistArgs a, int NV, int cap, int allow_tile, unsigned *hdr) {
    __shared__ int s_max[2], s_sum[2], s_cnt[2];
    const SweepParams &p = a.p;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    __shared__ unsigned s_fmax, s_choice;
    if (tid < 2) { s_max[tid] = 0; s_sum[tid] = 0; s_cnt[tid] = 0; }
    if (tid == 0) { s_fmax = 0u; if (a.ho.redo) *a.ho.redo = 0u; }
    if (tid < kAbsmaxWords && a.absmax) a.absmax[tid] = 0u;
    if (tid < kQueueHdr) hdr[tid] = 0u;      // the candidates' queue header (was a memset node in front of this kernel)
    __syncthreads();
    const int tiles_x = (p.W + kPW - 1) / kPW, tiles_y = (p.H + kPH - 1) / kPH;
    for (int t = wv; t < 54 * p.B; t += 16) {
        const int b = t / 54, tt = t % 54, which = tt / 27, s = tt % 27;
        const int nw = which == 0 ? 16 : 8, nchunks = (p.D + nw - 1) / nw;
        const int gx = s % 3, gy = (s / 3) % 3, gc = s / 9;
        const int ptx = gx == 0 ? 0 : (gx == 1 ? tiles_x / 2 : tiles_x - 1), pty = gy == 0 ? 0 : (gy == 1 ? tiles_y / 2 : tiles_y - 1);
        const int pdc = gc == 0 ? 0 : (gc == 1 ? nchunks / 2 : nchunks - 1);
        const int v0 = min(lane >> 3, NV - 1), k = lane & 7;
        const int xlo = ptx * kPW, xhi = min(xlo + kPW - 1, p.W - 1);
        const int ylo = pty * kPH, yhi = min(ylo + kPH - 1, p.H - 1);
        const int dlo = pdc * nw, dhi = min(dlo + nw - 1, p.D - 1);
        const float *r = a.rt + ((int64_t)v0 * p.B + b) * 12;
        const float cxk = (float)((k & 1) ? xhi : xlo), cyk = (float)((k & 2) ? yhi : ylo);
        const float dk = a.depth[(int64_t)b * p.D + ((k & 4) ? dhi : dlo)];
        const float rx = __fmaf_rn(r[0], cxk, __fmaf_rn(r[1], cyk, r[2]));
        const float ry = __fmaf_rn(r[4], cxk, __fmaf_rn(r[5], cyk, r[6]));
        const float rz = __fmaf_rn(r[8], cxk, __fmaf_rn(r[9], cyk, r[10]));
        const float X = __fmaf_rn(rx, dk, r[3]), Y = __fmaf_rn(ry, dk, r[7]), Z = __fmaf_rn(rz, dk, r[11]);
        const float inv = __builtin_amdgcn_rcpf(Z);
        const float ix = __fmaf_rn(X * inv, a.sx, a.ox), iy = __fmaf_rn(Y * inv, a.sy, a.oy);
        const bool zok = Z > 1e-6f && fabsf(ix) < 1.0e9f && fabsf(iy) < 1.0e9f;
        const int fxi = zok ? (int)floorf(ix) : 0, fyi = zok ? (int)floorf(iy) : 0;
        int lo_x = fxi - 1, hi_x = fxi + 2, lo_y = fyi - 1, hi_y = fyi + 2;
        int bad = zok ? 0 : 1;
#pragma unroll
        for (int off = 1; off <= 4; off <<= 1) {
            lo_x = min(lo_x, __shfl_xor(lo_x, off)); hi_x = max(hi_x, __shfl_xor(hi_x, off));
            lo_y = min(lo_y, __shfl_xor(lo_y, off)); hi_y = max(hi_y, __shfl_xor(hi_y, off));
            bad |= __shfl_xor(bad, off);
        }
        int x0 = max(lo_x, -1), x1 = min(hi_x, p.W), y0 = max(lo_y, -1), y1 = min(hi_y, p.H);
        if (x1 <= x0 || y1 <= y0) { x0 = 0; y0 = 0; x1 = 1; y1 = 1; }
        const int area = bad ? 4 * cap : min((x1 - x0 + 1) * (y1 - y0 + 1), 4 * cap);
        if (k == 0 && (lane >> 3) < NV) {
            atomicMax(&s_max[which], area);
            atomicAdd(&s_sum[which], area);
            atomicAdd(&s_cnt[which], 1);
        }
    }
    __syncthreads();
    if (tid == 0) {
        const float m16 = (float)s_sum[0] / (float)max(s_cnt[0], 1), m8 = (float)s_sum[1] / (float)max(s_cnt[1], 1);
        unsigned choice;
        if (m16 <= 0.31f * (float)cap) choice = 16;
        else if (m8 <= 0.36f * (float)cap || !allow_tile) choice = 8;
        else choice = 0;
        hdr[kSelWord] = choice;
        s_choice = choice;
        hdr[2] = (unsigned)s_max[0]; hdr[3] = (unsigned)(m16 + 0.5f); hdr[4] = (unsigned)s_max[1]; hdr[5] = (unsigned)(m8 + 0.5f);
        hdr[6] = (unsigned)s_cnt[0]; hdr[7] = (unsigned)cap;
    }
    // the hand-over block (sweep_common.h): the bound max |f|^2 of the variance when a persistent candidate runs and the bound is
    // finite, else a NaN pattern = "the volume leaves as fp32"
    if (a.ho.hand) {
        if (tid < kAbsmaxWords) atomicMax(&s_fmax, a.ho.fea_absmax[tid]);
        __syncthreads();
        const float f = __uint_as_float(s_fmax), bound = f * f;
        const unsigned bb = __float_as_uint(bound);
        const float vt = a.ho.veto ? *a.ho.veto : 0.0f;
        const bool ok = s_choice != 0u && s_fmax < 0x7f800000u && bb < 0x7f800000u && vt == vt;
        if (tid < kAbsmaxWords) a.ho.hand[tid] = ok ? bb : 0x7fc00000u;
    }
}

template <int NV, int NW, int NQ, bool FAST>
__global__ __launch_bounds__(NW * 64) void variance_fwd_persist_kernel(PersistArgs a) {
    constexpr int cap = persist_cap(NV, NQ);
    constexpr int NJ = cap / 64;                 // DMA instructions per (view, quad) plane
    constexpr int WQ = NW / NQ;                  // waves sharing a channel quad
    constexpr int NP = (NJ + WQ - 1) / WQ;       // pieces of a plane one wave may copy
    constexpr int GC = 4 * NQ;                   // channels per stage
    constexpr int NST = NQ;                      // 16-byte stores per lane and stage
    constexpr unsigned kViewBytes = (unsigned)NQ * cap * 16u;
    constexpr unsigned kRefOff = NV * kViewBytes;
    constexpr unsigned kBufBytes = kRefOff + NQ * 1024u;
    static_assert(NW % NQ == 0 && NV <= 8 && (NQ == 2 || NQ == 4), "8 corner lanes per view");
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[2 * kBufBytes];
    __shared__ float s_cam[kPMaxCamFloats];      // rot_trans [NV][B][12]
    __shared__ float s_depth[kPMaxDepthFloats];  // depth_values [B][D]

    const SweepParams &p = a.p;
    // The arguments are ~70 words; held in scalar registers for the whole kernel they push the allocator 140 registers over its
    // budget, and every spilled scalar comes back as a v_readlane -- a VECTOR instruction, in a kernel bound by vector issue
    // (round 6: ~350 of a wave-plane's ~1980 vector instructions were such reloads).  What the planning, scheduling and per-tile code
    // needs is therefore read from the kernel-argument segment WHERE IT IS USED (`ka->`: volatile scalar loads, scalar-memory pipe).
    typedef const volatile PersistArgs __attribute__((address_space(4))) *KernArgPtr;
    const KernArgPtr ka = (KernArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
    if (a.autosel && a.queue[kSelWord] != (unsigned)NW) return;   // the geometry asked for another kernel
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kq = wv % NQ, mj = wv / NQ;
    const unsigned lds_base = (unsigned)(uintptr_t)lds_raw;
    const int nstage = p.C / GC;                 // stages per tile
    const unsigned tstride = a.fea_c4 == 1 ? 16u : a.fea_c4 == 2 ? (unsigned)p.C * 4u : 64u;   // bytes between neighbouring texels of one quad

    int lx, ly;   // this lane's pixel inside the 16x4 tile
    if (a.flags & kPFlagLinearLanes) {
        lx = lane & 15; ly = lane >> 4;
    } else {
        const unsigned l5 = lane & 31, g0 = 0x0FF0F00Fu, below = (1u << l5) - 1u;
        const bool in0 = (g0 >> l5) & 1u;
        lx = __popc((in0 ? g0 : ~g0) & below);
        ly = (lane >> 5) * 2 + (in0 ? 0 : 1);
    }

    // Work units = (depth segment, pixel tile), segment-major.  XCD x (= blockIdx & 7) owns a contiguous
    // range of the unit list and its CUs take that range round-robin: the ~32 units in flight on an XCD
    // are neighbouring pixel tiles at the same depths, their footprints overlap and stay within its 4 MiB
    // L2 (with one contiguous range per CU the CUs of an XCD were 7 pixel tiles apart and every footprint
    // came from the Infinity Cache: 2.0 GB of fabric reads per launch).
    const int ptiles = a.tiles_x * a.tiles_y * p.B, nunits = ptiles * a.nseg;
    int u, u_end, u_step;
    const bool contiguous = a.flags & kPFlagContiguous;
    {
        const int nb = gridDim.x;
        if (contiguous) {
            const int q = nb >> 3, r = nb & 7, xcd = blockIdx.x & 7;
            const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + ((int)blockIdx.x >> 3);
            u = (int)((int64_t)nunits * L / nb); u_end = (int)((int64_t)nunits * (L + 1) / nb); u_step = 1;
        } else if ((nb & 7) == 0) {
            const int xcd = blockIdx.x & 7;
            u = (int)((int64_t)nunits * xcd / 8) + ((int)blockIdx.x >> 3);
            u_end = (int)((int64_t)nunits * (xcd + 1) / 8);
            u_step = nb >> 3;
        } else {
            u = blockIdx.x; u_end = nunits; u_step = nb;
        }
    }
    if (u >= u_end) return;
    // hand-over (sweep_common.h): the chooser has left the bound in a.ho.hand, or a NaN pattern = store fp32 as ever
    const unsigned hb = a.ho.hand ? (unsigned)__builtin_amdgcn_readfirstlane((int)a.ho.hand[0]) : 0xffffffffu;
    const bool pairs = hand_is_pairs(hb);
    const float ps = pow2f(14 - absmax_exponent(hb));

    // camera rows and depth planes into LDS: inside the loop nothing but the copies and the
    // stores touches vector memory, so a counted vmcnt can tell them apart
    for (int i = tid; i < NV * p.B * 12; i += NW * 64) s_cam[i] = a.rt[i];
    for (int i = tid; i < p.B * p.D; i += NW * 64) s_depth[i] = a.depth[i];
    __syncthreads();

    // ---- the planned tile: the one whose copies are issued next
    int pdc, ptx, pty, pb, seg_end;
    struct TileIt { int u, dc, seg_end, tx, ty, b; };
    auto it_open = [&](TileIt &t) {   // unit t.u -> its pixel tile and first chunk
        const int k_tx = ka->tiles_x, k_ty = ka->tiles_y, k_nseg = ka->nseg, k_cps = ka->cps, k_nch = ka->nchunks;
        const int k_ptiles = k_tx * k_ty * ka->p.B;
        const int sg = contiguous ? t.u % k_nseg : t.u / k_ptiles;
        int qq = contiguous ? t.u / k_nseg : t.u - sg * k_ptiles;
        t.tx = qq % k_tx; qq /= k_tx;
        t.ty = qq % k_ty; t.b = qq / k_ty;
        t.dc = sg * k_cps;
        t.seg_end = min(t.dc + k_cps, k_nch);
    };
    auto it_advance = [&](TileIt &t) {   // -> false behind this workgroup's last tile
        if (++t.dc < t.seg_end) return true;
        t.u += u_step;
        if (t.u >= u_end) return false;
        it_open(t);
        return true;
    };
    auto open_unit = [&]() {
        TileIt t; t.u = u; it_open(t);
        pdc = t.dc; ptx = t.tx; pty = t.ty; pb = t.b; seg_end = t.seg_end;
    };
    open_unit();
    int pbx0[NV], pby0[NV], pbw[NV], pbh[NV];
    unsigned pstaged = 0;
    unsigned soff[NV][NP];
    unsigned roff = 0;
    int tcount = 0;     // index of the planned tile in this workgroup's sequence

    // The footprint boxes of a tile are the same for all its waves, so they are PLANNED ONCE PER WORKGROUP (round 4; every wave
    // used to re-derive them for every tile: ~250 of a wave's ~2960 instructions per tile): in a round, wave w plans the w-th
    // tile from `first` on and leaves (x0, y0, bw, bh | staged << 30) per view in slot (first_index + w) mod 2 NW of s_plan;
    // load_plan() below picks a tile's entry up.  A round is written while the previous round's last tile is still being
    // loaded, hence two rounds of slots; a barrier always lies between a slot's write and its first read.
    __shared__ __attribute__((aligned(16))) int s_plan[2 * NW][NV][4];
    auto plan_round = [&](TileIt t, int first_index) {
        bool ok = true;
        for (int i = 0; i < wv && ok; ++i) ok = it_advance(t);
        if (!ok) return;
        const int v0 = min(lane >> 3, NV - 1), k = lane & 7;
        const int kW = ka->p.W, kH = ka->p.H, kD = ka->p.D, kB = ka->p.B;
        const float k_sx = ka->sx, k_ox = ka->ox, k_sy = ka->sy, k_oy = ka->oy;
        const int xlo = t.tx * kPW, xhi = min(xlo + kPW - 1, kW - 1);
        const int ylo = t.ty * kPH, yhi = min(ylo + kPH - 1, kH - 1);
        const int dlo = t.dc * NW, dhi = min(dlo + NW - 1, kD - 1);
        const float *r = s_cam + (v0 * kB + t.b) * 12;
        const float cxk = (float)((k & 1) ? xhi : xlo), cyk = (float)((k & 2) ? yhi : ylo);
        const float dk = s_depth[t.b * kD + ((k & 4) ? dhi : dlo)];
        // Per depth plane pixel -> source is a homography, so (all Z > 0) the tile's image is
        // the convex hull of its corner images; along depth each coordinate is a Moebius
        // function of d, monotone between the extreme planes.  Approximate arithmetic is
        // enough: the box is padded by a texel and every wave checks its own taps against it.
        const float rx = __fmaf_rn(r[0], cxk, __fmaf_rn(r[1], cyk, r[2]));
        const float ry = __fmaf_rn(r[4], cxk, __fmaf_rn(r[5], cyk, r[6]));
        const float rz = __fmaf_rn(r[8], cxk, __fmaf_rn(r[9], cyk, r[10]));
        const float X = __fmaf_rn(rx, dk, r[3]), Y = __fmaf_rn(ry, dk, r[7]), Z = __fmaf_rn(rz, dk, r[11]);
        const float inv = __builtin_amdgcn_rcpf(Z);
        const float ix = __fmaf_rn(X * inv, k_sx, k_ox), iy = __fmaf_rn(Y * inv, k_sy, k_oy);
        const bool zok = Z > 1e-6f && fabsf(ix) < 1.0e9f && fabsf(iy) < 1.0e9f;
        const int fxi = zok ? (int)floorf(ix) : 0, fyi = zok ? (int)floorf(iy) : 0;
        int lo_x = fxi - 1, hi_x = fxi + 2, lo_y = fyi - 1, hi_y = fyi + 2;
        int bad = zok ? 0 : 1;
#pragma unroll
        for (int off = 1; off <= 4; off <<= 1) {
            lo_x = min(lo_x, __shfl_xor(lo_x, off)); hi_x = max(hi_x, __shfl_xor(hi_x, off));
            lo_y = min(lo_y, __shfl_xor(lo_y, off)); hi_y = max(hi_y, __shfl_xor(hi_y, off));
            bad |= __shfl_xor(bad, off);
        }
        // one texel beyond the image on every side stays in the box, so a tap pair that
        // straddles the border is addressed like any other
        int x0 = max(lo_x, -1), x1 = min(hi_x, kW), y0 = max(lo_y, -1), y1 = min(hi_y, kH);
        if (x1 <= x0 || y1 <= y0) { x0 = 0; y0 = 0; x1 = 1; y1 = 1; }   // nothing of this view in sight
        const int bw = x1 - x0 + 1, bh = y1 - y0 + 1;
        const int staged = (!bad && bw * bh <= cap) ? 1 : 0;
        if (k == 0 && (lane >> 3) < NV)
            *reinterpret_cast<int4 *>(&s_plan[(first_index + wv) & (2 * NW - 1)][lane >> 3][0]) = make_int4(x0, y0, bw, bh | (staged << 30));
    };
    // the planned tile's entry -> the box scalars, the staged-view mask and this wave's per-lane source offsets
    auto load_plan = [&]() {
        pstaged = 0;
        const int kW = ka->p.W, kH = ka->p.H;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int4 e = *reinterpret_cast<const int4 *>(&s_plan[tcount & (2 * NW - 1)][v][0]);
            const int x0 = __builtin_amdgcn_readfirstlane(e.x), y0 = __builtin_amdgcn_readfirstlane(e.y);
            const int bw = __builtin_amdgcn_readfirstlane(e.z), bhs = __builtin_amdgcn_readfirstlane(e.w);
            pbx0[v] = x0; pby0[v] = y0; pbw[v] = bw; pbh[v] = bhs & 0xffff;
            pstaged |= (unsigned)(bhs >> 30) << v;
            const float rb = __builtin_amdgcn_rcpf((float)bw);
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int t = (mj + i * WQ) * 64 + lane;
                const int tyy = (int)(((float)t + 0.5f) * rb);      // t / bw (t < 2^10: exact)
                const int txx = t - __mul24(tyy, bw);
                const int sxx = min(max(x0 + txx, 0), kW - 1), syy = min(max(y0 + tyy, 0), kH - 1);
                soff[v][i] = (unsigned)(__mul24(syy, kW) + sxx) * tstride;
            }
        }
        roff = (unsigned)(__mul24(min(pty * kPH + ly, kH - 1), kW) + min(ptx * kPW + lx, kW - 1)) * tstride;
    };

    // Stage st of the planned tile = channel quads [st * NQ, st * NQ + NQ); this wave copies quad
    // st * NQ + kq.  ONE buffer descriptor per tensor for the whole kernel; view, batch item and
    // quad travel in the 32-bit scalar offset of the instruction, so a copy costs a handful of
    // scalar instructions (the CU has one scalar unit for its 16 waves).
    auto issue_dma = [&](int st, unsigned buf_off) {
        if (a.flags & kPFlagNoDma) return;
        const int kB = ka->p.B, k_c4 = ka->fea_c4;
        const unsigned k_plane = (unsigned)(ka->p.H * ka->p.W);
        const unsigned map_bytes = k_plane * (unsigned)ka->p.C * 4u;
        const mvs_srd_t srd_src = make_srd(ka->srcs16, map_bytes * (unsigned)(NV * kB));
        const mvs_srd_t srd_ref = make_srd(ka->ref16, map_bytes * (unsigned)kB);
        const int q = st * NQ + kq;
        const unsigned qoff = k_c4 == 1 ? (unsigned)q * k_plane * 16u
                            : k_c4 == 2 ? (unsigned)q * 16u
                                        : (unsigned)(q >> 2) * k_plane * 64u + (unsigned)(q & 3) * 16u;
        const unsigned boff = (unsigned)pb * map_bytes + qoff;
        const unsigned ldst = lds_base + buf_off + (unsigned)kq * cap * 16u;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            if (!((pstaged >> v) & 1u)) continue;
            const unsigned so = (unsigned)(v * kB) * map_bytes + boff;
            const int n = pbw[v] * pbh[v];
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int j = mj + i * WQ;
                if (j >= NJ || j * 64 >= n) continue;    // wave-uniform
                glds16_buf(soff[v][i], srd_src, so, ldst + (unsigned)((v * NQ * cap + j * 64) * 16));
            }
        }
        if (mj == 0)   // the reference view's 64 pixels, slot = consumer lane
            glds16_buf(roff, srd_ref, boff, lds_base + buf_off + kRefOff + (unsigned)kq * 1024u);
    };

    {
        TileIt t0; t0.u = u; it_open(t0);
        plan_round(t0, 0);
    }
    __syncthreads();
    load_plan();
    issue_dma(0, 0u);
    unsigned buf_off = 0;
    const float rV = 1.0f / p.fV;
    int stored = 0;        // store instructions this wave has issued after the last copy it issued (0: wait for everything)
    float vmax = 0.0f;     // largest |variance| this lane has stored

#pragma unroll 1
    for (;;) {
        const int T = ((pb * ka->tiles_y + pty) * ka->tiles_x + ptx) * ka->nchunks + pdc;   // tile id (cold-path records)
        // ---- adopt the planned tile; per-voxel homography + tap set of every source view
        SweepParams pl;     // the parameters, read here for this tile's set-up (see `ka` above)
        pl.B = ka->p.B; pl.C = ka->p.C; pl.D = ka->p.D; pl.H = ka->p.H; pl.W = ka->p.W; pl.V = NV + 1;
        pl.depth_mode = 0; pl.align_corners = ka->p.align_corners; pl.alias_quirk = 0;
        pl.half_w = ka->p.half_w; pl.half_h = ka->p.half_h; pl.unn_w = ka->p.unn_w; pl.unn_h = ka->p.unn_h; pl.fV = (float)(NV + 1);
        const float k_rhw = ka->rhw, k_rhh = ka->rhh;
        const int cb = pb;
        const int px = ptx * kPW + lx, py = pty * kPH + ly, d = pdc * NW + wv;
        const bool live = px < pl.W && py < pl.H && d < pl.D;
        const bool wave_live = d < pl.D;
        const int cx = min(px, pl.W - 1), cy = min(py, pl.H - 1), cd = min(d, pl.D - 1);
        const int pix = cy * pl.W + cx;
        const unsigned cstaged = pstaged;
        int cbw16[NV];
        float wnw[NV], wne[NV], wsw[NV], wse[NV];
        unsigned aoff[NV];
        unsigned win = 0;
        if (a.flags & kPFlagNoTaps) {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                wnw[v] = wne[v] = wsw[v] = wse[v] = 0.25f;
                aoff[v] = (unsigned)v * kViewBytes + (unsigned)lane * 16u;
                cbw16[v] = pbw[v] * 16;
            }
            win = (1u << NV) - 1u;
        } else {
            const float dv = s_depth[cb * pl.D + cd];
            bool unsafe = false;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                int tx0, ty0;
                bool has;
                tap_setup<FAST, false>(s_cam + (v * pl.B + cb) * 12, cx, cy, dv, pl, k_rhw, 0.0f, k_rhh, 0.0f,
                                       wnw[v], wne[v], wsw[v], wse[v], tx0, ty0, has, &unsafe);
                const int bx0 = pbx0[v], by0 = pby0[v], bw = pbw[v], bh = pbh[v];
                const bool inbox = !has | ((tx0 >= bx0) & (tx0 + 1 < bx0 + bw) & (ty0 >= by0) & (ty0 + 1 < by0 + bh));
                if (__all(inbox)) win |= 1u << v;
                const int ccx = min(max(tx0, bx0), bx0 + bw - 2) - bx0;
                const int ccy = min(max(ty0, by0), by0 + bh - 2) - by0;
                aoff[v] = (unsigned)v * kViewBytes + (unsigned)(__mul24(ccy, bw) + ccx) * 16u;
                cbw16[v] = bw * 16;
                __builtin_amdgcn_sched_barrier(0);   // one view at a time
            }
            if (__any(unsafe)) win = 0u;             // (FAST: a Z outside the shared-reciprocal division's range -> the cold kernel)
        }
        // a wave that cannot serve this tile from LDS hands its plane to the cold kernel
        const bool hot = (cstaged & win) == (1u << NV) - 1u;
        if (!hot && wave_live && lane == 0) {
            const unsigned slot = atomicAdd(a.queue, 1u);
            a.queue[kQueueHdr + slot] = ((unsigned)T << 4) | (unsigned)wv;
            stored = 0;   // more vector-memory traffic behind the last copy: wait for all of it
        }
        const bool has_next = pdc + 1 < seg_end || u + u_step < u_end;
        const bool any_live = __ballot(live) != 0ull;
        float *const k_out = ka->out;
        float *const plf = k_out + ((size_t)cb * pl.D + cd) * ((size_t)(pl.H * pl.W) * pl.C);   // wave-uniform plane base
        unsigned char *const plp = reinterpret_cast<unsigned char *>(k_out) + ((size_t)cb * pl.D + cd) * ka->pg.plane;   // ... of a pairs volume
        unsigned pmain = 0, pdup = 0xffffffffu;
        if (pairs) {
            PairsGeom pgl;
            pgl.plane = 0; pgl.chunk = 0; pgl.xtile = ka->pg.xtile; pgl.region = ka->pg.region; pgl.halo = ka->pg.halo; pgl.halo_part = 0;
            pgl.dup_lo = 0; pgl.rowpitch = ka->pg.rowpitch; pgl.strip = ka->pg.strip; pgl.tiled = ka->pg.tiled; pgl.strips = ka->pg.strips;
            pairs_offsets(pgl, cx, cy, pl.W, pmain, pdup);
        }
        if (!live || (a.flags & kPFlagNoDup)) pdup = 0xffffffffu;
        // Both pieces of a border column's halo copy leave in ONE store instruction: the border lane stores its hi piece, the lane
        // beside it in the quad (x ^ 1: never a border lane itself) its lo piece, handed over by DPP.  (Two instructions of four
        // lanes each cost the sweep 0.05-0.08 ms at configs[1]: profiles/r06_handover_sweep.json.)
        const unsigned pdup_nb = (unsigned)__builtin_amdgcn_update_dpp(-1, (int)pdup, 0xB1, 0xf, 0xf, false);      // quad_perm [1, 0, 3, 2]
        const bool carries = pdup == 0xffffffffu && pdup_nb != 0xffffffffu;
        if (carries) pdup = pdup_nb + (unsigned)ka->pg.dup_lo;
        const bool any_dup = pairs && __ballot(pdup != 0xffffffffu) != 0ull;

#pragma unroll 1
        for (int st = 0; st < nstage; ++st) {
            const bool last = st + 1 == nstage;
            // opaque per iteration: otherwise every tap address of every view is hoisted out of
            // this loop and held in registers
#pragma unroll
            for (int v = 0; v < NV; ++v) asm volatile("" : "+v"(aoff[v]));
            if (last && has_next) {   // the next tile's plan before the barrier: off the critical path
                if (++pdc == seg_end) {
                    u += u_step;
                    open_unit();
                }
                ++tcount;
                load_plan();
                if ((tcount & (NW - 1)) == NW - 1) {   // the last tile of a round is planned: now the next round's (a barrier before their first use)
                    TileIt t; t.u = u; t.dc = pdc; t.seg_end = seg_end; t.tx = ptx; t.ty = pty; t.b = pb;
                    if (it_advance(t)) plan_round(t, tcount + 1);
                }
            }
            // This stage's copies have landed (vector memory retires in order: at most the NST
            // stores issued behind them are still in flight -- waiting for those too would put a
            // full HBM write latency into every stage), for every wave, and nobody still reads
            // the other buffer.
            if (stored == NST) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
            else if (stored == NST + 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST + 1) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (!last) issue_dma(st + 1, buf_off ^ kBufBytes);
            else if (has_next) issue_dma(0, buf_off ^ kBufBytes);
            stored = 0;

            if (wave_live && hot && !(a.flags & kPFlagNoBlend)) {
                // Taps by hand-issued ds_read_b128, one (view, channel quad) batch of four ahead of
                // the batch being blended: left to the compiler, a branch-free body hoists every
                // read of a stage above the arithmetic.  Nothing else that counts in lgkmcnt may
                // sit inside this section.
                float S[GC], Q[GC];
                const unsigned la = lds_base + buf_off;
                constexpr int PD = 3;                     // batches in flight (1 being blended)
                constexpr int NB = NV * NQ;
                f32x4 rr[NQ], tb[PD][4];
                static_for<0, NQ>([&](auto K) {
                    constexpr int k = K;
                    rr[k] = lds_rd16<k * 1024>(la + kRefOff + (unsigned)lane * 16u);
                });
                auto issue_batch = [&](auto I) {
                    constexpr int i = I;
                    constexpr int v = i / NQ, k = i % NQ, sl = i % PD;
                    const unsigned a0 = la + aoff[v], a1 = a0 + (unsigned)cbw16[v];
                    tb[sl][0] = lds_rd16<k * cap * 16>(a0);
                    tb[sl][1] = lds_rd16<k * cap * 16 + 16>(a0);
                    tb[sl][2] = lds_rd16<k * cap * 16>(a1);
                    tb[sl][3] = lds_rd16<k * cap * 16 + 16>(a1);
                };
                static_for<0, (PD - 1 < NB ? PD - 1 : NB)>(issue_batch);   // prologue: PD-1 batches out
                {   // the reference pixels are the oldest reads
                    constexpr int ahead = (PD - 1 < NB ? PD - 1 : NB);
                    lds_wait_n<4 * ahead>();
                    static_for<0, NQ>([&](auto K) {
                        constexpr int k = K;
                        asm volatile("" : "+v"(rr[k]));
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            Q[k * 4 + c] = rr[k][c] * rr[k][c];
                            S[k * 4 + c] = rr[k][c];
                        }
                    });
                }
                static_for<0, NB>([&](auto J) {
                    constexpr int j = J, v = j / NQ, k = j % NQ, sl = j % PD;
                    if constexpr (j + PD - 1 < NB) issue_batch(std::integral_constant<int, j + PD - 1>{});
                    // batches issued after batch j and still allowed in flight
                    constexpr int newer = (j + PD - 1 < NB ? PD - 1 : NB - 1 - j);
                    lds_wait_n<4 * newer>();
                    asm volatile("" : "+v"(tb[sl][0]), "+v"(tb[sl][1]), "+v"(tb[sl][2]), "+v"(tb[sl][3]));
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float w = __fmaf_rn(tb[sl][3][c], wse[v], __fmaf_rn(tb[sl][2][c], wsw[v],
                                                  __fmaf_rn(tb[sl][1][c], wne[v], tb[sl][0][c] * wnw[v])));
                        S[k * 4 + c] = S[k * 4 + c] + w;
                        if constexpr (FAST) Q[k * 4 + c] = __fmaf_rn(w, w, Q[k * 4 + c]);
                        else Q[k * 4 + c] = Q[k * 4 + c] + w * w;
                    }
                    // this batch is blended before the next reads go out (and the arithmetic is
                    // not sunk into the store's `if`, leaving every tap register live)
#pragma unroll
                    for (int c = 0; c < 4; ++c) asm volatile("" : "+v"(S[k * 4 + c]), "+v"(Q[k * 4 + c]));
                });
                float var[GC];
                if constexpr (FAST) {
#pragma unroll
                    for (int c = 0; c < GC; ++c) {
                        const float m = S[c] * rV;
                        var[c] = __fmaf_rn(Q[c], rV, -(m * m));
                    }
                } else {
                    // any |S|, |Q| outside [1e-30, 3e38] sends the wave to the true division
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
                }
#pragma unroll
                for (int c = 0; c < GC; ++c) asm volatile("" : "+v"(var[c]));   // formed here, not inside the `if`
                if (any_live && !(a.flags & kPFlagNoStore) && pairs) {
                    // hand-over: scale by the power of two of the bound, split into two fp16 pieces (split2.h), one 16-byte store per
                    // piece -- the same NST = 2 store instructions as the fp32 form; a tile-border column once more into the neighbouring
                    // tile's halo strip / block (one more instruction, eight lanes)
                    static_assert(GC == 8, "a stage = one 8-channel chunk");
                    f32x4 v0 = {var[0], var[1], var[2], var[3]}, v1 = {var[4], var[5], var[6], var[7]};
                    if (live) {
#pragma unroll
                        for (int c = 0; c < GC; c += 2) vmax = max_nan(max_nan(vmax, __builtin_fabsf(var[c])), __builtin_fabsf(var[c + 1]));
                    }
                    u32x4 hp, lp;
                    split2_block(v0, v1, ps, hp, lp);
                    unsigned char *const ch = plp + (size_t)st * ka->pg.chunk;
                    if (live) {
                        *reinterpret_cast<u32x4 *>(ch + pmain) = hp;
                        *reinterpret_cast<u32x4 *>(ch + pmain + 2 * ka->pg.region) = lp;
                    }
                    stored = NST;
                    if (any_dup) {
                        u32x4 dd;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const unsigned nb = (unsigned)__builtin_amdgcn_update_dpp(0, (int)lp[i], 0xB1, 0xf, 0xf, false);
                            dd[i] = carries ? nb : hp[i];
                        }
                        if (pdup != 0xffffffffu) *reinterpret_cast<u32x4 *>(ch + pdup) = dd;
                        stored = NST + 1;
                    }
                } else if (any_live && !(a.flags & kPFlagNoStore)) {
                    // exactly NST store instructions per wave (lanes outside the volume masked off)
                    const int k_c8 = ka->out_c8, kC = ka->p.C, kW = ka->p.W;
                    float *o = k_c8
                        ? plf + ((unsigned)(cy * (kC >> 3) + st * (NQ / 2)) * (unsigned)kW + (unsigned)cx) * 8u   // [B,D,H,C/8,W,8]
                        : plf + (unsigned)pix * (unsigned)kC + (unsigned)(st * GC);
                    const unsigned step = k_c8 ? (unsigned)kW * 8u : 8u;   // floats between 8-channel blocks
                    if (live) {
#pragma unroll
                        for (int h = 0; h < NQ / 2; ++h) {
                            reinterpret_cast<float4 *>(o + h * step)[0] = make_float4(var[h * 8 + 0], var[h * 8 + 1], var[h * 8 + 2], var[h * 8 + 3]);
                            reinterpret_cast<float4 *>(o + h * step)[1] = make_float4(var[h * 8 + 4], var[h * 8 + 5], var[h * 8 + 6], var[h * 8 + 7]);
                        }
#pragma unroll
                        for (int c = 0; c < GC; c += 2) vmax = max_nan(max_nan(vmax, __builtin_fabsf(var[c])), __builtin_fabsf(var[c + 1]));
                    }
                    stored = NST;
                }
            }
            buf_off ^= kBufBytes;
        }
        if (!has_next) break;
    }
    publish_absmax(a.absmax, vmax);
}

// Launch table.  MVS_EUNSUPPORTED (nothing launched) when the shape is not this kernel's.
template <int NW, int NQ, bool FAST>
static int launch_persist_nv(int NV, const PersistArgs &a, int grid, hipStream_t st) {
#define MVS_PERSIST_CASE(n)                                                                           \
    case n:                                                                                           \
        hipLaunchKernelGGL((variance_fwd_persist_kernel<n, NW, NQ, FAST>), dim3(grid), dim3(NW * 64), \
                           0, st, a);                                                                 \
        if (a.autosel != 2)   /* 2: another candidate follows, the cold kernel is launched behind it */ \
            hipLaunchKernelGGL((variance_fwd_cold_kernel<n, FAST>), dim3(grid), dim3(256), 0, st, a,  \
                               NW);                                                                   \
        return MVS_OK;
    switch (NV) {
        MVS_PERSIST_CASE(1) MVS_PERSIST_CASE(2) MVS_PERSIST_CASE(3) MVS_PERSIST_CASE(4)
        MVS_PERSIST_CASE(5) MVS_PERSIST_CASE(6) MVS_PERSIST_CASE(7) MVS_PERSIST_CASE(8)
    }
#undef MVS_PERSIST_CASE
    return MVS_EUNSUPPORTED;
}

static bool persist_shape_ok(const SweepParams &p) {
    const int NV = p.V - 1;
    return p.depth_mode == 0 && !p.alias_quirk && p.C % 16 == 0 && p.C <= 64 && NV >= 1 && NV <= kMaxSrcViews &&
           (int64_t)p.H * p.W < (1 << 26) && p.H < 32760 && p.W < 32760 &&
           (int64_t)p.H * p.W * p.C * 4 * NV * p.B < (1ll << 32) && (int64_t)NV * p.B * 12 <= kPMaxCamFloats &&
           (int64_t)p.B * p.D <= kPMaxDepthFloats;
}

static int64_t persist_tiles(const SweepParams &p, int nw) {
    return (int64_t)((p.W + kPW - 1) / kPW) * ((p.H + kPH - 1) / kPH) * ((p.D + nw - 1) / nw) * p.B;
}

size_t variance_persist_workspace_bytes(const SweepParams &p, int nw) {
    if (!persist_shape_ok(p)) return 0;
    const int64_t total = persist_tiles(p, nw);
    if (total >= (1 << 27)) return 0;
    return 4 * (size_t)kQueueHdr + 4 * (size_t)total * nw;   // header + one record per (tile, wave)
}

bool variance_persist_shape_ok(const SweepParams &p) { return persist_shape_ok(p) && persist_tiles(p, 8) < (1 << 27); }

// the chooser in front of the candidate kernels (all of them launched with autosel)
int launch_variance_choose(const float *rt, const float *depth, const SweepParams &p, int allow_tile, unsigned *absmax, void *workspace,
                           hipStream_t st, const SweepHandover *ho) {
    PersistArgs a{};
    a.rt = rt; a.depth = depth; a.p = p; a.absmax = absmax;
    if (ho) a.ho = *ho;
    a.rhw = (float)(1.0 / (double)p.half_w); a.rhh = (float)(1.0 / (double)p.half_h);
    if (p.align_corners) { a.sx = 1.0f; a.ox = 0.0f; a.sy = 1.0f; a.oy = 0.0f; }
    else {
        a.sx = (float)((double)p.W / (double)(p.W - 1)); a.ox = -0.5f;
        a.sy = (float)((double)p.H / (double)(p.H - 1)); a.oy = -0.5f;
    }
    const int NV = p.V - 1;
    hipLaunchKernelGGL(variance_choose_kernel, dim3(1), dim3(1024), 0, st, a, NV, persist_cap(NV, 2), allow_tile,
                       static_cast<unsigned *>(workspace));
    return check_launch("variance_choose_kernel");
}

int launch_variance_persist(const float *ref16, const float *srcs16, const float *rt,
                            const float *depth, const SweepParams &p, float *out, int out_c8,
                            int fea_c4, int fast, int nw, int nq, int flags, void *workspace,
                            size_t workspace_bytes, hipStream_t st, int autosel, unsigned *absmax, const SweepHandover *ho) {
    if ((nw != 8 && nw != 16) || nq != 2) return MVS_EUNSUPPORTED;
    const size_t need = variance_persist_workspace_bytes(p, nw);
    if (need == 0) return MVS_EUNSUPPORTED;
    if (!workspace || workspace_bytes < need) {
        set_error("mvs_costvol_variance_fwd_ws_f32: workspace of %zu bytes, need %zu", workspace_bytes, need);
        return MVS_EWORKSPACE;
    }
    PersistArgs a{};
    a.ref16 = ref16; a.srcs16 = srcs16; a.rt = rt; a.depth = depth; a.out = out; a.p = p;
    if (ho) {
        if (!autosel || !out_c8 || p.C % 8) return bare_error(MVS_EINVAL, __func__, __LINE__);   // the chooser writes ho->hand
        a.ho = *ho;
        a.pg = pairs_geom(p.C, p.H, p.W, ho->layout);
        if (a.pg.plane >= 0xffffffffLL) return bare_error(MVS_EINVAL, __func__, __LINE__);       // 32-bit per-lane offsets
    }
    a.queue = static_cast<unsigned *>(workspace);
    a.tiles_x = (p.W + kPW - 1) / kPW;
    a.tiles_y = (p.H + kPH - 1) / kPH;
    a.nchunks = (p.D + nw - 1) / nw;
    a.total_tiles = (int)persist_tiles(p, nw);
    {   // segments: enough work units for ~32 per CU (balance), each as long as that allows (a CU that stays
        // on a pixel tile re-reads its sliding footprints out of L2).  (Round 6: 16 -> 32 per CU.  At configs[1] that is 7400 units
        // of three depth chunks instead of 5550 of four: a CU's share is 28.9 +- 0.1 units instead of 21.7 +- 0.3 and the last
        // round of units is shorter -- the op's median 1.50-1.53 -> 1.42-1.44 ms on one box, alternating; 48 and 128: no better.)
        const int64_t ptiles = (int64_t)a.tiles_x * a.tiles_y * p.B;
        const int ncu = device_cu_count();
        int nseg = (int)((32ll * ncu + ptiles - 1) / ptiles);
        nseg = nseg < 1 ? 1 : (nseg > a.nchunks ? a.nchunks : nseg);
        a.cps = (a.nchunks + nseg - 1) / nseg;
        a.nseg = (a.nchunks + a.cps - 1) / a.cps;
    }
    a.out_c8 = out_c8;
    a.flags = flags;
    a.fea_c4 = fea_c4;
    a.autosel = autosel;
    a.absmax = absmax;
    a.rhw = (float)(1.0 / (double)p.half_w); a.rhh = (float)(1.0 / (double)p.half_h);
    if (p.align_corners) { a.sx = 1.0f; a.ox = 0.0f; a.sy = 1.0f; a.oy = 0.0f; }
    else {
        a.sx = (float)((double)p.W / (double)(p.W - 1)); a.ox = -0.5f;
        a.sy = (float)((double)p.H / (double)(p.H - 1)); a.oy = -0.5f;
    }
    // (with autosel the chooser has cleared the header)
    if (!autosel && launch_zero_words(workspace, kQueueHdr, st) != MVS_OK) return MVS_ELAUNCH;
    if (!autosel && absmax && launch_zero_words(absmax, kAbsmaxWords, st) != MVS_OK) return MVS_ELAUNCH;
    const int grid = device_cu_count();
    const int NV = p.V - 1;
#define MVS_PERSIST_PICK(W_, Q_)                                                             \
    if (nw == W_ && nq == Q_)                                                                \
        return fast ? launch_persist_nv<W_, Q_, true>(NV, a, grid, st)                       \
                    : launch_persist_nv<W_, Q_, false>(NV, a, grid, st);
    MVS_PERSIST_PICK(16, 2) MVS_PERSIST_PICK(8, 2)
#undef MVS_PERSIST_PICK
    return MVS_EUNSUPPORTED;
}

int launch_variance_redo_all(const float *ref16, const float *srcs16, const float *rt, const float *depth, const SweepParams &p,
                             float *out, int fea_c4, int fast, void *workspace, hipStream_t st, const SweepHandover &ho, unsigned *absmax) {
    if (!persist_shape_ok(p) || !ho.redo || !ho.hand || !absmax) return bare_error(MVS_EINVAL, __func__, __LINE__);
    const int nw = 16, NV = p.V - 1;
    PersistArgs a{};
    a.ref16 = ref16; a.srcs16 = srcs16; a.rt = rt; a.depth = depth; a.out = out; a.p = p;
    a.queue = static_cast<unsigned *>(workspace);
    a.tiles_x = (p.W + kPW - 1) / kPW;
    a.tiles_y = (p.H + kPH - 1) / kPH;
    a.nchunks = (p.D + nw - 1) / nw;
    a.total_tiles = (int)persist_tiles(p, nw);
    a.out_c8 = 1; a.fea_c4 = fea_c4; a.absmax = absmax;
    a.ho = ho; a.ho.redo_all = 1;
    a.rhw = (float)(1.0 / (double)p.half_w); a.rhh = (float)(1.0 / (double)p.half_h);
    if (p.align_corners) { a.sx = 1.0f; a.ox = 0.0f; a.sy = 1.0f; a.oy = 0.0f; }
    else {
        a.sx = (float)((double)p.W / (double)(p.W - 1)); a.ox = -0.5f;
        a.sy = (float)((double)p.H / (double)(p.H - 1)); a.oy = -0.5f;
    }
    // one workgroup per CU: the launch is in every forward and normally only DECIDES (every workgroup reads the two blocks and
    // returns: 2048 of them took ~19 us, 256 take a third of that); when it does have to recompute the volume its loops stride the grid
    const int grid = device_cu_count();
#define MVS_REDO_CASE(n)                                                                                                      \
    case n:                                                                                                                   \
        if (fast) hipLaunchKernelGGL((variance_fwd_cold_kernel<n, true>), dim3(grid), dim3(256), 0, st, a, nw);               \
        else hipLaunchKernelGGL((variance_fwd_cold_kernel<n, false>), dim3(grid), dim3(256), 0, st, a, nw);                   \
        break;
    switch (NV) {
        MVS_REDO_CASE(1) MVS_REDO_CASE(2) MVS_REDO_CASE(3) MVS_REDO_CASE(4)
        MVS_REDO_CASE(5) MVS_REDO_CASE(6) MVS_REDO_CASE(7) MVS_REDO_CASE(8)
        default: return MVS_EUNSUPPORTED;
    }
#undef MVS_REDO_CASE
    return check_launch("variance_fwd_cold_kernel (redo)");
}

}  // namespace mvs
