// Shared by conv3d_mfma.hip and conv2d_mfma.hip: launch arguments and tile order of the
// convolution kernels, and the PERSISTENT DMA-fed MFMA convolution kernel (3D 3x3x3 layers, and
// 2D k x k layers as volumes whose planes are the images and whose kernel is one plane deep).
#ifndef MVS_CONV_PERSISTENT_H
#define MVS_CONV_PERSISTENT_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "mvs_common.h"

namespace mvs {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int round_up_c(int v, int m) { return (v + m - 1) / m * m; }

struct ConvArgs {
    const float *in, *wpk, *scale, *shift, *residual;
    float *out;
    int B, D, H, W;        // input dims
    int Do, Ho, Wo;        // output dims
    int tiles_x, tiles_y, tiles_z;
    int relu;
    int in_c8;   // input is [B,D,H,C/8,W,8] (8-channel blocked) instead of [B,D,H,W,C]
    int ystrip;  // tile order: 0 = x, y, z; n > 0 = y within strips of n tile rows, then z, then x
    int res_up2; // residual is [B,Do,Ho/2,Wo/2,C]: added through a nearest x2 upsample in y and x (FPN top-down path)
    int out_c4 = 0;   // 2D layers: write [image,C/4,Ho,Wo,4] (4-channel blocked, the sweep kernel's fastest input) instead of [image,Ho,Wo,C]
    unsigned *out_absmax = nullptr;   // the absmax block (mvs_common.h) the largest magnitude stored is max-ed into, or NULL
    const unsigned *run_flag = nullptr;   // Cout = 1 kernels: a device word; the launch returns at once when it is 0 (mvs_common.h: conv_run_flag)
};

// XCD-aware bijective remap: consecutive tiles land on the same XCD (same L2)
// so halo re-reads of neighbouring tiles hit that L2 (guide T1).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// Tile owned by a workgroup.  Tiles that share halo planes should run close together in
// time on the same XCD so the re-read hits that XCD's 4-MiB L2: with `ystrip` the order is
// y inside a strip of tile rows (fastest), then z, then x -- the y and z neighbours of a
// tile are then at most one strip column (a few MiB of input) away instead of a whole
// z-slab (tens of MiB).
struct TileIdx { int tx, ty, tz, b; };
__device__ __forceinline__ TileIdx decode_ordered_tile(const ConvArgs &a, int bid) {
    TileIdx t;
    if (a.ystrip <= 0) {
        t.tx = bid % a.tiles_x; bid /= a.tiles_x;
        t.ty = bid % a.tiles_y; bid /= a.tiles_y;
        t.tz = bid % a.tiles_z;
        t.b = bid / a.tiles_z;
    } else {
        const int per_b = a.tiles_x * a.tiles_y * a.tiles_z;
        t.b = bid / per_b; bid -= t.b * per_b;
        const int full = a.ystrip * a.tiles_z * a.tiles_x;
        const int s = bid / full; bid -= s * full;
        const int y0 = s * a.ystrip;
        const int hs = min(a.ystrip, a.tiles_y - y0);   // the last strip may be short
        t.ty = y0 + bid % hs; bid /= hs;
        t.tz = bid % a.tiles_z;
        t.tx = bid / a.tiles_z;
    }
    return t;
}
__device__ __forceinline__ TileIdx decode_tile(const ConvArgs &a, int blk, int nblk) {
    return decode_ordered_tile(a, xcd_remap(blk, nblk));
}

// ---------------------------------------------------------------------
// conv0-class layers (Cout = 8, stride 1, 8-channel-blocked input) as a PERSISTENT,
// DMA-fed kernel: one 512-thread workgroup per CU walks a list of (4,4,32)-voxel output
// tiles.  Phase timestamps of the kernel above showed a third of every workgroup's life
// in its VALU-heavy prologue / staging / epilogue, crawling beside the partner
// workgroup's MFMA stream (VALU issue on a SIMD is arbitrated by age), so the matrix
// pipe idled ~30 %.  Here
//   * the whole layer's A fragments (Cin/8 x 18 KiB) live in LDS for the kernel's
//     lifetime: no global load is ever waited on inside the MFMA loop;
//   * the input halo of an 8-channel chunk goes HBM -> LDS by buffer-addressed DMA
//     (buffer_load_dwordx4 ... lds: no staging VGPRs, no ds_write pass, no vector ALU work per
//     copy, 5 DMA instructions per thread per chunk, zero fill by the range check) into
//     the buffer the MFMAs are NOT reading; chunk k+1 (of this tile or the next) is in
//     flight during the MFMAs of chunk k, and there is one barrier per chunk;
//   * all 8 waves do both jobs, so the two waves of a SIMD are always in the same phase.
// LDS image of a chunk: [h = channel half][voxel][4 channels], voxel order (z,y,x') with
// x de-interleaved as in MODE 2; lane (n,kq) reads channels 2kq, 2kq+1 of voxel n with
// one ds_read_b64 -- the same channel <-> (k-step, kq) assignment as the kernel above,
// so the packed weights are shared.

typedef float f32x2 __attribute__((ext_vector_type(2)));

// One ds_read_b64 at a compile-time offset from a per-lane LDS byte address.  Written
// as asm so that it STAYS a ds_read_b64 (64-bank rules, two 32-lane groups: conflict-free
// for the 16-byte voxel stride used below): hipcc merges neighbouring b64 loads into
// ds_read2_b64, which is serviced under 32-bank rules in 16-lane groups -- a 2-way
// conflict on this layout that makes the reads, not the MFMAs, set the pace.  The
// compiler does not see these loads complete: lds_wait_n<N>() is the matching s_waitcnt.
template <int OFF>
__device__ __forceinline__ f32x2 lds_read_b64(unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536 && OFF % 8 == 0, "ds_read_b64 offset field");
    f32x2 v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
// wait until at most N LDS operations are outstanding (pair with an empty asm "+v" on the
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

// KD x KH x KH kernel: 3x3x3 for the CostRegNet layers; KD = 1 for FeatureNet's 2D layers
// (3x3, or 5x5 stride 2), whose images are the planes of the "volume" (never strided).
template <int CIN_, int COUT_ = 8, int MODE_ = 2, int TZ_ = 4, int TY_ = 4, int KD_ = 3, int KH_ = 3>
struct PersistCfg {
    static constexpr int CIN = CIN_, COUT = COUT_, MODE = MODE_, TZ = TZ_, TY = TY_, KD = KD_, KH = KH_;
    static constexpr int NCHUNK = CIN / 8;
    static constexpr int NKX = KH + (MODE == 2 ? 1 : 0), NTAPS = KD * KH * NKX;
    static constexpr int MT = (MODE == 2) ? 1 : COUT / 16;
    static constexpr int SZY = (MODE == 1) ? 2 : 1;          // conv stride in y
    static constexpr int SZ = (MODE == 1 && KD == 3) ? 2 : 1; // ... and in z
    static constexpr int XOUT = (MODE == 2) ? 32 : 16;
    static constexpr int SX = (MODE == 0) ? 1 : 2;           // x step of a wave's B reads
    static constexpr int XT = 15 * SX + NKX, XH = (XT + 1) / 2, XTP = (SX == 2) ? 2 * XH : XT;
    static constexpr int YT = (TY - 1) * SZY + KH, ZT = (TZ - 1) * SZ + KD;
    static constexpr int NVOX = ZT * YT * XTP;
    static constexpr int PLANE = round_up_c(NVOX, 64);        // voxels per channel half
    static constexpr int NDMA = 2 * PLANE / 64;               // wave-instructions per chunk
    static constexpr int IPW = (NDMA + 7) / 8;                // per wave
    static constexpr int ROWS = TZ * TY, RPW = ROWS / 8;      // (z,y) output rows per wave
    static constexpr int W_FLOATS = NCHUNK * NTAPS * MT * 64 * 2;
    static constexpr int BUF_FLOATS = 2 * PLANE * 4;
    static constexpr int LDS_FLOATS = W_FLOATS + 2 * BUF_FLOATS;
    // (MODE 0, stride 1 with Cout >= 16, is expressible but not instantiated: conv2 gained 9 %
    // and its weights would need the 8-channel-chunk packing)
    static_assert(MODE >= 0 && MODE <= 2, "MODE 0: stride 1; MODE 1: stride 2; MODE 2: Cout = 8 shifted form");
    static_assert(MODE != 2 || COUT == 8, "MODE 2 is the Cout = 8 form");
    static_assert(MODE == 2 || COUT % 16 == 0, "MODE 0/1 need whole 16-channel M tiles");
    static_assert(ROWS % 8 == 0, "rows split over 8 waves");
    static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS budget");
};

template <class P, int ABL = 0>
__global__ __launch_bounds__(512) void conv3d_c8_persistent_kernel(ConvArgs a, int ntiles) {
    constexpr int CIN = P::CIN, COUT = P::COUT, MODE = P::MODE, MT = P::MT, RPW = P::RPW;
    constexpr int NTAPS = P::NTAPS, NKX = P::NKX, XT = P::XT, XH = P::XH, XTP = P::XTP, YT = P::YT;
    constexpr int PLANE = P::PLANE, IPW = P::IPW, SZY = P::SZY;
    __shared__ __attribute__((aligned(16))) float lds[P::LDS_FLOATS];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4;
    const unsigned lds_base = (unsigned)(uintptr_t)lds;

    // ---- tiles of this workgroup: XCD x (= blockIdx & 7) owns a contiguous range of the
    // ordered tile list, its workgroups take that range round-robin, so the tiles in
    // flight on one XCD are neighbours and share halo lines in that XCD's L2
    int t_cur, t_end, t_step;
    float vmax = 0.0f;      // largest magnitude this lane has stored (-> a.out_absmax)
    {
        const int nb = gridDim.x;
        if ((nb & 7) == 0) {
            const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, per = nb >> 3;
            const int lo = (int)((int64_t)ntiles * xcd / 8), hi = (int)((int64_t)ntiles * (xcd + 1) / 8);
            t_cur = lo + j; t_end = hi; t_step = per;
        } else {
            t_cur = blockIdx.x; t_end = ntiles; t_step = nb;
        }
    }

    // ---- once: all A fragments -> LDS; tile-independent halo coordinates of this
    // thread's DMA items (item i of wave w is DMA instruction g = i*8 + w of the chunk:
    // channel half h = g / (PLANE/64), voxels (g % (PLANE/64))*64 + lane)
    {
        constexpr int NGR = P::W_FLOATS / 4;        // 16-byte granules
        constexpr int NWI = (NGR + 63) / 64;        // wave-instructions (the last may be partial)
        for (int i = wv; i < NWI; i += 8)
            if (i * 64 + lane < NGR)
                glds16(a.wpk + ((size_t)i * 64 + lane) * 4, lds_base + (unsigned)i * 1024u);
    }
    int loc[IPW];        // lx | ly << 8 | lz << 16 | h << 24 | invalid << 31 (as sign)
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
        const int g = min(i * 8 + wv, P::NDMA - 1);
        const int h = g / (PLANE / 64), vb = g % (PLANE / 64);
        const int v = vb * 64 + lane;
        const int vc = min(v, P::NVOX - 1);
        const int lxp = vc % XTP, t2 = vc / XTP;
        const int ly = t2 % YT, lz = t2 / YT;
        // x de-interleaved (evens, then odds) where a wave's B reads step by 2
        const int lx = (P::SX == 1) ? lxp : (lxp < XH ? 2 * lxp : 2 * (lxp - XH) + 1);
        loc[i] = lx | (ly << 8) | (lz << 16) | (h << 24) | ((v < P::NVOX && lx < XT) ? 0 : (int)0x80000000);
    }

    // input addressing: 8-channel-blocked [D,H,C/8,W,8] or channels-last [D,H,W,C]
    const int64_t plane_in = (int64_t)a.H * a.W * CIN;   // floats per z-plane
    const int row_in = a.W * CIN;                        // floats per (z,y) row
    const int vox_in = a.in_c8 ? 8 : CIN;                // floats between x neighbours
    const int ch_step = a.in_c8 ? a.W * 8 : 8;           // floats between 8-channel chunks

    // geometry of one tile.  The copies are buffer-addressed: a resource descriptor whose base
    // is the first halo plane of the tile (wave-uniform, SGPRs), a per-lane byte offset that
    // is the same for every channel chunk, and the chunk offset in an SGPR -- re-issuing the
    // copy for the next chunk costs no vector ALU work (a wave's VALU issue crawls beside its
    // partner's MFMA stream).  Halo voxels outside the volume get an offset past
    // num_records: the hardware returns zeros for them.
    unsigned voff[IPW];
    mvs_srd_t srd;
    TileIdx tile;
    const unsigned window_bytes = (unsigned)min((int64_t)P::ZT * plane_in * 4, (int64_t)0xffffff00u);
    auto geometry = [&](int t) {
        tile = decode_ordered_tile(a, t);
        const int ix0 = tile.tx * P::XOUT * (MODE == 1 ? 2 : 1) - P::KH / 2;
        const int iy0 = tile.ty * P::TY * SZY - P::KH / 2, iz0 = tile.tz * P::TZ * P::SZ - P::KD / 2;
        srd = make_srd(a.in + ((int64_t)tile.b * a.D + iz0) * plane_in, window_bytes);
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            const int gx = ix0 + (loc[i] & 255), gy = iy0 + ((loc[i] >> 8) & 255);
            const int lz = (loc[i] >> 16) & 255, h = (loc[i] >> 24) & 1;
            const bool ok = loc[i] >= 0 && (unsigned)gx < (unsigned)a.W && (unsigned)gy < (unsigned)a.H &&
                            (unsigned)(iz0 + lz) < (unsigned)a.D;
            voff[i] = ok ? (unsigned)(((int64_t)lz * plane_in + (int64_t)gy * row_in + gx * vox_in + h * 4) * 4)
                         : 0xffffff00u;
        }
    };
    auto issue = [&](int ch, int parity) {
        const unsigned base = lds_base + (unsigned)(P::W_FLOATS + parity * P::BUF_FLOATS) * 4u;
        const unsigned soff = (unsigned)(ch * ch_step * 4);
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            if (P::NDMA % 8 != 0 && i * 8 + wv >= P::NDMA) continue;   // wave-uniform
            glds16_buf(voff[i], srd, soff, base + (unsigned)(i * 8 + wv) * 1024u);
        }
    };

    // BatchNorm(eval) affine of this lane's output channels (4 per M tile)
    float4 sc[MT], sh[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int c0 = (MODE == 2) ? (kq & 1) * 4 : m * 16 + kq * 4;
        sc[m] = a.scale ? *reinterpret_cast<const float4 *>(a.scale + c0) : make_float4(1.f, 1.f, 1.f, 1.f);
        sh[m] = a.shift ? *reinterpret_cast<const float4 *>(a.shift + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
    }

    // per-lane LDS read bases (floats): B = plane kq>>1, voxel n, channel pair kq&1
    const int rdB = ((kq >> 1) * PLANE + n) * 4 + (kq & 1) * 2;
    int rowoff[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int row = wv * RPW + r;
        rowoff[r] = (((row / P::TY) * P::SZ) * YT + (row % P::TY) * SZY) * XTP * 4;
    }
    const int rdA = lane * 2;

    f32x4 acc[RPW][MT];
#pragma unroll
    for (int r = 0; r < RPW; ++r)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[r][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ABL & 16 (tuning): cycles of wave 0 per phase, summed over the tiles of this
    // workgroup, into the buffer passed as `residual`: [wait, barrier, issue, mfma, stores, drain]
    long long tsum[6] = {0, 0, 0, 0, 0, 0};
    long long tprev = 0;
    if constexpr (ABL & 16) tprev = clock64();
#define MVS_LAP(k) do { if constexpr (ABL & 16) { const long long tn = clock64(); tsum[k] += tn - tprev; tprev = tn; } } while (0)

    int parity = 0;
    if (t_cur < t_end) {
        geometry(t_cur);
        issue(0, 0);
    }
    while (t_cur < t_end) {
        const TileIdx cur = tile;
        const int t_next = t_cur + t_step;
#pragma unroll 1
        for (int ch = 0; ch < P::NCHUNK; ++ch) {
            // this chunk's DMA (issued one phase ago) has landed for every wave, and
            // every wave is done reading the other buffer
            MVS_LAP(4);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            MVS_LAP(0);
            __syncthreads();
            MVS_LAP(1);
            // next chunk (of this tile, or the first of the next one) -> the other buffer.
            // (Issuing it a few taps into the MFMA stream, staggering it between the two
            // waves of a SIMD, deferring a tile's stores into the next tile's stream, or
            // hoisting the geometry were each measured: no gain.)
            if (ch + 1 < P::NCHUNK) {
                issue(ch + 1, parity ^ 1);
            } else if (t_next < t_end) {
                geometry(t_next);
                issue(0, parity ^ 1);
            }
            MVS_LAP(2);
            // ---- MFMA stream of the chunk: per tap MT A reads and one B read per row,
            // software-pipelined PD taps ahead through PD+1 register slots; LDS returns in
            // order, so "tap t has landed" is lgkmcnt <= (MT+RPW) x (taps issued after it)
            const unsigned aA = lds_base + (unsigned)(ch * (NTAPS * MT * 64 * 2) + rdA) * 4u;
            unsigned aB[RPW];
#pragma unroll
            for (int r = 0; r < RPW; ++r)
                aB[r] = lds_base + (unsigned)(P::W_FLOATS + parity * P::BUF_FLOATS + rdB + rowoff[r]) * 4u;
            constexpr int NRD = MT + RPW;
            constexpr int PD = (15 / NRD >= 3) ? 3 : (15 / NRD >= 1 ? 15 / NRD : 1);   // lgkmcnt counts to 15
            static_assert(NRD * PD <= 15, "outstanding LDS reads exceed the lgkmcnt field");
            f32x2 fa[PD + 1][MT], fb[PD + 1][RPW];
            auto fetch = [&](auto tc) {
                constexpr int t = decltype(tc)::value;
                constexpr int kz = t / (P::KH * NKX), ky = (t / NKX) % P::KH, kx = t % NKX;
                constexpr int xoff = (P::SX == 1) ? kx : (kx & 1) * XH + (kx >> 1);
                constexpr int boff = ((kz * YT + ky) * XTP + xoff) * 16;
                static_for<0, MT>([&](auto mc) {
                    constexpr int m = decltype(mc)::value;
                    fa[t % (PD + 1)][m] = lds_read_b64<(t * MT + m) * 512>(aA);
                });
#pragma unroll
                for (int r = 0; r < RPW; ++r) fb[t % (PD + 1)][r] = lds_read_b64<boff>(aB[r]);
            };
            static_for<0, PD>(fetch);
            static_for<0, NTAPS>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                constexpr int sl = t % (PD + 1);
                if constexpr (t + PD < NTAPS) fetch(std::integral_constant<int, t + PD>{});
                constexpr int newer = (t + PD < NTAPS ? PD : NTAPS - 1 - t);
                lds_wait_n<NRD * newer>();
#pragma unroll
                for (int m = 0; m < MT; ++m) asm volatile("" : "+v"(fa[sl][m]));
#pragma unroll
                for (int r = 0; r < RPW; ++r) asm volatile("" : "+v"(fb[sl][r]));
                if constexpr (ABL & 8) {
#pragma unroll
                    for (int r = 0; r < RPW; ++r) asm volatile("" ::"v"(fa[sl][0]), "v"(fb[sl][r]));
                } else {
                    // same accumulation order as conv3d_mfma_kernel: per (tap, row, M tile) k-step 0, 1
#pragma unroll
                    for (int r = 0; r < RPW; ++r)
#pragma unroll
                        for (int m = 0; m < MT; ++m) {
                            acc[r][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[sl][m].x, fb[sl][r].x, acc[r][m], 0, 0, 0);
                            acc[r][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[sl][m].y, fb[sl][r].y, acc[r][m], 0, 0, 0);
                        }
                }
            });
            parity ^= 1;
            MVS_LAP(3);
        }
        // ---- epilogue of `cur`: BN affine, ReLU, skip add, one 16-byte store per lane, row, M tile
        if constexpr (ABL & 16) {
            asm volatile("" : "+v"(acc[0][0]));
            asm volatile("s_nop 0" ::: "memory");
            MVS_LAP(5);
        }
        {
            const int ox = (MODE == 2) ? cur.tx * P::XOUT + 2 * n + (kq >> 1) : cur.tx * P::XOUT + n;
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int row = wv * RPW + r;
                const int oz = cur.tz * P::TZ + row / P::TY, oy = cur.ty * P::TY + row % P::TY;
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    f32x4 v = acc[r][m];
                    acc[r][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (oz >= a.Do || oy >= a.Ho || ox >= a.Wo) continue;
                    const int c0 = (MODE == 2) ? (kq & 1) * 4 : m * 16 + kq * 4;
                    v[0] = v[0] * sc[m].x + sh[m].x; v[1] = v[1] * sc[m].y + sh[m].y;
                    v[2] = v[2] * sc[m].z + sh[m].z; v[3] = v[3] * sc[m].w + sh[m].w;
                    if (a.relu == 1) {
                        v[0] = relu_nan(v[0]); v[1] = relu_nan(v[1]);
                        v[2] = relu_nan(v[2]); v[3] = relu_nan(v[3]);
                    } else if (a.relu == 2) {   // LeakyReLU(0.1): CVP-MVSNet's feature pyramid
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : v[j] * 0.1f;
                    }
                    // out_c4: a lane's four channels are one block of [image,C/4,Ho,Wo,4]; the 16 lanes of an
                    // MFMA column group then write 256 contiguous bytes
                    const int64_t o = a.out_c4
                        ? (((((int64_t)cur.b * a.Do + oz) * (COUT / 4) + (c0 >> 2)) * a.Ho + oy) * a.Wo + ox) * 4
                        : ((((int64_t)cur.b * a.Do + oz) * a.Ho + oy) * a.Wo + ox) * COUT + c0;
                    if (a.residual && !(ABL & 16)) {
                        const int64_t ro = a.res_up2 ? ((((int64_t)cur.b * a.Do + oz) * (a.Ho >> 1) + (oy >> 1)) * (a.Wo >> 1) +
                                                        (ox >> 1)) * COUT + c0 : o;
                        const float4 rs = *reinterpret_cast<const float4 *>(a.residual + ro);
                        v[0] += rs.x; v[1] += rs.y; v[2] += rs.z; v[3] += rs.w;
                    }
                    *reinterpret_cast<float4 *>(a.out + o) = make_float4(v[0], v[1], v[2], v[3]);
                    vmax = amax4_nan(vmax, v[0], v[1], v[2], v[3]);
                }
            }
        }
        t_cur = t_next;
    }
    publish_absmax(a.out_absmax, vmax);
    if constexpr (ABL & 16) {
        MVS_LAP(4);
        if (tid == 0) {
            long long *dbg = reinterpret_cast<long long *>(const_cast<float *>(a.residual)) + (int64_t)blockIdx.x * 8;
            for (int k = 0; k < 6; ++k) dbg[k] = tsum[k];
        }
    }
#undef MVS_LAP
}

}  // namespace mvs
#endif  // MVS_CONV_PERSISTENT_H
