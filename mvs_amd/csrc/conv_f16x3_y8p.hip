// conv0 (3x3x3, Cout = 8, stride 1; mvsnet.py:66, module.py:26-33) on a variance volume that ARRIVES as two fp16 pieces per
// value, with eight-row tiles (round 6; VERDICT r05 item 1: the hand-over of a pre-split volume AND the larger tile).
//
// conv3d_c8_f16x3_zs_kernel spends a third of every step between two barriers turning its fp32 staging buffer into fp16 part
// planes (split pass 615-740 cycles + the second barrier, with the matrix pipe idle: scripts/exp_conv0_y8.py laps), and vector
// work cannot be hidden beside an MFMA stream on this hardware (HISTORY.md section B: four overlap designs that lost).  The
// producer -- the plane-sweep kernel, which holds every variance in a register anyway, or mvs_c8_to_c8p_f32 -- writes
// x 2^(14 - exponent(bound)) as hi = fp16(.) and lo = fp16(. - hi) (conv_f16x3.hip has the arithmetic and the error bound);
// this kernel's copy waves bring the pieces straight into the ring of plane slots the MFMA phase reads: no staging buffer, no
// split pass, ONE barrier per step.  Tiles are (2, 8, 32) voxels marched two planes at a time as in conv_f16x3_y8.hip (same MFMA
// phase, same packed weights, same results bit for bit as mvs_conv3d_c8_f16x3_f32 on the fp32 volume with the same block).
//
// Memory layouts of the pairs (16-byte piece = 8 fp16 channels of one voxel and part; parity = of the voxel's x):
//   MVS_LAYOUT_C8P  (6): [B, D, C/8, part (hi, lo), parity (x & 1), H, ceil(W/2)] pieces -- 4 bytes per element like fp32.  The 17
//                        pieces of a (row, part, parity) a tile needs are one 272-byte run.
//   MVS_LAYOUT_C8PT (7): [B, D, C/8, ceil(W/32), part, local parity, H, 17] pieces -- x-tiled with the two halo voxels of a
//                        32-voxel tile stored in ITS block too (local x' = x - 32 tx + 1 in 0..33: parity x' & 1, piece x' >> 1;
//                        +6 % bytes): the ten rows of a (plane, part, parity) a tile needs are ONE run of 2720 bytes, which is
//                        what an LDS-DMA copy wants (its rate is set by the cache lines an instruction touches).
// LDS: 2 x 18 KiB weights + 2 parts x 2 parities x 2 NPAIR plane slots x 2720 B: NPAIR = 4 (copies two steps ahead) 124 KiB,
// NPAIR = 5 (three steps ahead) 146 KiB.
#include "conv_split_common.h"
#include "conv_guard.h"
#include "sweep_common.h"

#include <cstdlib>

namespace mvs {

constexpr int kYPChunkBytes = 9 * 2 * 1024;   // A fragments of one 8-channel chunk (the pack of conv_f16x3.hip)
constexpr int kYPRowVox = 17, kYPRows = 10, kYPT = 8, kYPCopyWaves = 4, kYPThreads = 512 + 64 * kYPCopyWaves;

__device__ __forceinline__ void wait_vmcnt_dyn(int n) {   // wave-uniform n: at most n vector-memory operations outstanding
    switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
    case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
    case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
    case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    case 17: asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); break;
    case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
    case 19: asm volatile("s_waitcnt vmcnt(19)" ::: "memory"); break;
    case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
    case 21: asm volatile("s_waitcnt vmcnt(21)" ::: "memory"); break;
    case 22: asm volatile("s_waitcnt vmcnt(22)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(23)" ::: "memory"); break;
    }
}

template <int CIN, int NPAIR, int ABL = 0>
__global__ __launch_bounds__(kYPThreads) void conv3d_c8p_f16x3_kernel(ConvArgs a, PairsGeom pg, int ngroups,
                                                                      const unsigned *__restrict__ in_absmax,
                                                                      const unsigned *__restrict__ redo,
                                                                      unsigned *__restrict__ out_absmax) {
    // hand-over protocol (sweep_common.h): *redo != 0 = the volume is fp32 after all; the fp32 kernel enqueued behind this launch
    // (mvs_conv3d_c8_f16x3_f32 under that word as its run-only-if flag) serves it
    if (redo && *redo != 0u) return;
    constexpr int NCHUNK = CIN / 8, YT = kYPRows, T = kYPT, AHEAD = NPAIR - 2;
    constexpr int NSLOT = 2 * NPAIR;
    constexpr int WBYTES = kYPChunkBytes, WCOPIES = WBYTES / 1024;
    constexpr int SLOT = YT * kYPRowVox * 16;                       // 2720: one plane of one (part, parity)
    constexpr int RSTR = (NSLOT * SLOT + 255) / 256 * 256;          // (part, parity) regions a multiple of 256 bytes apart
    constexpr int SPART = 2 * RSTR, SBYTES = 2 * SPART;
    constexpr int S_OFF = 2 * WBYTES;
    constexpr int PPS = 2 * YT * kYPRowVox;                         // 340 pieces of a region per step (two planes)
    constexpr int IPR = (PPS + 63) / 64;                            // 6 copy instructions per region and step
    static_assert(S_OFF + SBYTES <= 160 * 1024 && SPART + 3 * kYPRowVox * 16 < 65536, "LDS budget / ds offset field");
    static_assert(kYPCopyWaves == 4, "one copy wave per (part, parity) region");
    __shared__ __attribute__((aligned(16))) unsigned char lds[S_OFF + SBYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4;
    const unsigned lds_base = (unsigned)(uintptr_t)lds;
    const bool copier = wv >= 8;
    const int cw = wv - 8;       // copy wave = region: part = cw >> 1, local parity = cw & 1

    // the producer scaled by 2^(14 - exponent(block)); the weights' scale is the trailer of their pack
    const int xe = absmax_exponent(load_absmax(in_absmax));
    const float isx = pow2f(xe - 14);
    // (non-finite weights leave a NaN here: every output becomes NaN.  A hand-over sweep that was given this word as its veto,
    // mvs_costvol_variance_fwd_ws3_f32, never lets it come to that: the volume is then fp32 and conv_f16x3.hip's guard serves it)
    const float isw = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(
        __builtin_bit_cast(int, a.wpk[(size_t)NCHUNK * (WBYTES / 4)])));

    int g0, g_step, ngw;
    {
        const int nb = gridDim.x;
        if ((nb & 7) == 0) {
            const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3, per = nb >> 3;
            const int lo = (int)((int64_t)ngroups * xcd / 8), hi = (int)((int64_t)ngroups * (xcd + 1) / 8);
            g0 = lo + jb; g_step = per; ngw = (hi - g0 + per - 1) / per;
        } else {
            g0 = blockIdx.x; g_step = nb; ngw = (ngroups - g0 + nb - 1) / nb;
        }
        if (ngw < 0) ngw = 0;
    }
    if (ngw == 0) return;
    const int ngz = (a.tiles_z + T - 1) / T;
    struct Grp { int tx, ty, zg, b; };
    auto decode = [&](int g) {
        Grp r;
        r.ty = g % a.tiles_y; g /= a.tiles_y;
        r.tx = g % a.tiles_x; g /= a.tiles_x;
        r.zg = g % ngz; r.b = g / ngz;
        return r;
    };

    if (copier) {
        // ================================================================ copy waves
        // Wave cw copies region (part, local parity) = (cw >> 1, cw & 1): per step the 340 pieces of two planes -- piece q = plane
        // q / 170, row (q % 170) / 17, index i = (q % 170) % 17 -- to the step's pair of slots, six instructions.
        const int part = cw >> 1, lpar = cw & 1;
        int qrow[IPR], qidx[IPR], qpl[IPR];
#pragma unroll
        for (int i = 0; i < IPR; ++i) {
            const int q = i * 64 + lane, qc = q < PPS ? q : 0, r = qc % (YT * kYPRowVox);
            qpl[i] = q < PPS ? qc / (YT * kYPRowVox) : 99;          // 99: no such piece
            qrow[i] = r / kYPRowVox; qidx[i] = r % kYPRowVox;
        }
        const unsigned window_bytes = (unsigned)min((int64_t)2 * pg.plane, (int64_t)0xffffff00u);
        unsigned voff[IPR];       // byte offset from (plane zs, chunk 0, this tile / region); 0xffffff00 = outside the volume in x or y
        Grp cg{0, 0, 0, 0};
        unsigned tile_off = 0;    // C8PT: the x tile's block; both: the region's block
        auto geometry = [&](int g) {
            cg = decode(g);
            const int iy0 = cg.ty * 8 - 1;
            // local x' = 2 i + lpar counts from the tile's halo origin 32 tx - 1
            tile_off = (unsigned)(pg.tiled ? cg.tx * pg.xtile + (part * 2 + lpar) * pg.region
                                           : (part * 2 + (1 - lpar)) * pg.region);          // C8P / C8PH: global parity = 1 - local parity
#pragma unroll
            for (int i = 0; i < IPR; ++i) {
                const int gy = iy0 + qrow[i], gx = cg.tx * 32 - 1 + 2 * qidx[i] + lpar;
                const bool ok = qpl[i] < 2 && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
                const int col = pg.tiled ? qidx[i] : (gx >> 1);
                unsigned o = (unsigned)(qpl[i] * pg.plane + (int64_t)gy * pg.rowpitch + col * 16);
                // C8PH: the tile's two halo columns (x' = 0: first piece of local parity 0; x' = 33: last piece of local parity 1) come
                // from its halo strips -- H pieces in a row, so the ten rows of a plane share two cache lines instead of taking one each
                const bool halo = pg.strips && (lpar == 0 ? qidx[i] == 0 : qidx[i] == kYPRowVox - 1);
                if (halo) o = (unsigned)(qpl[i] * pg.plane + pg.halo + part * pg.halo_part + (int64_t)(cg.tx * 2 + lpar) * pg.strip + gy * 16) - tile_off;
                voff[i] = ok ? o : 0xffffff00u;
            }
        };
        auto issue_halo = [&](int p, int ch, int pair) {              // planes 2p, 2p + 1 of the column (global z = 16 zg - 1 + ...)
            const int zs = cg.zg * 16 - 1 + 2 * p;
            const mvs_srd_t srd = make_srd(reinterpret_cast<const unsigned char *>(a.in) + ((int64_t)cg.b * a.D + zs) * pg.plane, window_bytes);
            const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane((int)(ch * pg.chunk) + (int)tile_off);
            const unsigned dst = lds_base + (unsigned)(S_OFF + part * SPART + lpar * RSTR + pair * 2 * SLOT);
#pragma unroll
            for (int i = 0; i < IPR; ++i) {
                const bool ok = (unsigned)(zs + qpl[i]) < (unsigned)a.D;
                // (the last instruction of a region: 20 of its 64 lanes -- EXEC keeps the others from zero-filling the next slots)
                if (qpl[i] < 2) glds16_buf(ok ? voff[i] : 0xffffff00u, srd, soff, dst + i * 1024);
            }
            return IPR;
        };
        const unsigned char *wsrc = reinterpret_cast<const unsigned char *>(a.wpk);
        constexpr int WHALF = (WCOPIES + 1) / 2, NC = kYPCopyWaves;
        auto issue_weights = [&](int ch, int sel, int lo, int hi) {   // wave-copies [lo, hi) of a chunk's A fragments
            int cnt = 0;
#pragma unroll
            for (int i = 0; i < (WCOPIES + NC - 1) / NC; ++i) {
                const int g = i * NC + cw;
                if (g >= lo && g < hi) {
                    glds16(wsrc + (size_t)ch * WBYTES + (size_t)g * 1024 + lane * 16, lds_base + (unsigned)(sel * WBYTES + g * 1024));
                    ++cnt;
                }
            }
            return cnt;
        };
        int it_k = 0, it_ch = 0, it_p = 0, it_nvalid = 0, it_pair = 0;
        bool it_done = false;
        auto it_open = [&]() { geometry(g0 + it_k * g_step); it_nvalid = min(T, a.tiles_z - cg.zg * T); };
        auto it_issue = [&]() {
            if (it_done) return 0;
            const int cnt = issue_halo(it_p, it_ch, it_pair);
            it_pair = it_pair == NPAIR - 1 ? 0 : it_pair + 1;
            if (++it_p > it_nvalid) {
                it_p = 0;
                if (++it_ch >= NCHUNK) {
                    it_ch = 0;
                    if (++it_k >= ngw) it_done = true; else it_open();
                }
            }
            return cnt;
        };
        int wsel = 0;
        // While step s is multiplied (pairs s - 1 and s of the ring) the planes of steps s + 1 .. s + AHEAD are in flight or landed in
        // the other AHEAD pairs; behind the barrier of step s the batch of step s + AHEAD goes into pair s - 2.
        int fifo[AHEAD - 1];      // operations issued in the batches behind the step about to be multiplied, oldest first
        it_open();
        issue_weights(0, 0, 0, WCOPIES);
        it_issue();
#pragma unroll
        for (int i = 0; i < AHEAD - 1; ++i) fifo[i] = it_issue();
        for (int k = 0; k < ngw; ++k) {
            const int nvalid = min(T, a.tiles_z - decode(g0 + k * g_step).zg * T);
#pragma unroll 1
            for (int ch = 0; ch < NCHUNK; ++ch) {
                const bool more = ch + 1 < NCHUNK || k + 1 < ngw;
                const int nch = ch + 1 < NCHUNK ? ch + 1 : 0;
#pragma unroll 1
                for (int p = 0; p <= nvalid; ++p) {
                    int newer = 0;
#pragma unroll
                    for (int i = 0; i < AHEAD - 1; ++i) newer += fifo[i];
                    wait_vmcnt_dyn(newer);         // this step's planes have landed; those of the next AHEAD - 1 steps may be in flight
                    __syncthreads();               // ... for every copy wave; and step s - 1 is multiplied: its older pair of slots is free
                    int cnt = 0;                   // the next chunk's weights: in halves behind the first two steps (a one-tile column: at once)
                    if (more && p == 0) cnt += issue_weights(nch, wsel ^ 1, 0, nvalid == 1 ? WCOPIES : WHALF);
                    if (more && p == 1 && nvalid > 1) cnt += issue_weights(nch, wsel ^ 1, WHALF, WCOPIES);
                    cnt += it_issue();
#pragma unroll
                    for (int i = 0; i + 2 < AHEAD; ++i) fifo[i] = fifo[i + 1];
                    fifo[AHEAD - 2] = cnt;
                }
                wsel ^= 1;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }

    // ================================================================ multiplying waves
    float4 sc, sh;
    {
        const int c0 = (kq & 1) * 4;
        sc = a.scale ? *reinterpret_cast<const float4 *>(a.scale + c0) : make_float4(1.f, 1.f, 1.f, 1.f);
        sh = a.shift ? *reinterpret_cast<const float4 *>(a.shift + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int z0 = wv >> 2, y0 = (wv & 3) * 2;          // the wave's output rows: (z0, y0) and (z0, y0 + 1)
    const int ex = 2 * n + (kq >> 1);
    const int eoff = ((z0 * a.Ho + y0) * a.Wo + ex) * 8 + (kq & 1) * 4;
    // this lane's B voxel of row y0 inside a plane slot: local x' = 2n + kq -> region kq & 1, piece n + (kq >> 1)
    const unsigned aB = lds_base + (unsigned)(S_OFF + (kq & 1) * RSTR + (y0 * kYPRowVox + n + (kq >> 1)) * 16);

    f32x4 acc[T][2];
#pragma unroll
    for (int j = 0; j < T; ++j) acc[j][0] = acc[j][1] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int wsel = 0, pair = 0;
    float vmax = 0.0f;
    long long tsum[3] = {0, 0, 0};   // tuning build, ABL & 128: cycles in the barrier / MFMA phase / the rest
    long long tprev = 0;
    if constexpr (ABL & 128) tprev = clock64();
#define MVS_LAP(k) do { if constexpr (ABL & 128) { const long long tn = clock64(); tsum[k] += tn - tprev; tprev = tn; } } while (0)
    for (int k = 0; k < ngw; ++k) {
        const Grp cur = decode(__builtin_amdgcn_readfirstlane(g0 + k * g_step));
        const int nvalid = min(T, a.tiles_z - cur.zg * T);
#pragma unroll 1
        for (int ch = 0; ch < NCHUNK; ++ch) {
            static_for<0, T + 1>([&](auto pc) {
                constexpr int p = decltype(pc)::value;
                if (p > nvalid) return;   // wave-uniform
                const int prev = pair == 0 ? NPAIR - 1 : pair - 1;
                MVS_LAP(2);
                __syncthreads();
                MVS_LAP(0);
                if constexpr (p >= 1 && !(ABL & 2)) {
                    constexpr int j = p - 1;
                    // ---- MFMA phase (conv_f16x3.hip / conv_f16x3_y8.hip): nine blocks (kz, ky) of six MFMAs alternating between the
                    // two accumulators; plane q = z0 + kz of the tile: q < 2 in the previous step's pair of slots, else in this step's
                    unsigned aBz[3];
#pragma unroll
                    for (int kz = 0; kz < 3; ++kz) {
                        const int q = z0 + kz;
                        aBz[kz] = aB + (unsigned)(((q >> 1 ? pair : prev) * 2 + (q & 1)) * SLOT);
                    }
                    const unsigned aA = lds_base + (unsigned)(wsel * WBYTES + lane * 16);
                    f16x8 bsr[4][2], Aw[2][2];
                    auto rd = [&](auto ic, auto cc) {
                        constexpr int i = decltype(ic)::value, c = decltype(cc)::value, kz = c / 3, ky = c % 3;
                        constexpr bool two = ky == 0;
                        constexpr int nin = two ? 6 : 4;
                        if constexpr (i == 0 || i == nin - 1) {
                            constexpr int sp = i == 0 ? 0 : 1;
                            Aw[c & 1][sp] = __builtin_bit_cast(f16x8, lds_read_b128<(c * 2 + sp) * 1024>(aA));
                        } else {
                            constexpr int q = i - 1;
                            constexpr int sp = two ? q / 2 : q, iy = two ? ky + (q & 1) : ky + 1, g = kz * 4 + iy;
                            bsr[g & 3][sp] = __builtin_bit_cast(f16x8, lds_read_b128<iy * kYPRowVox * 16 + sp * SPART>(aBz[kz]));
                        }
                    };
                    static_for<0, 6>([&](auto ic) { rd(ic, std::integral_constant<int, 0>{}); });
                    __builtin_amdgcn_sched_barrier(0);
                    static_for<0, 9>([&](auto cc) {
                        constexpr int c = decltype(cc)::value, kz = c / 3, ky = c % 3, g0r = kz * 4 + ky, g1r = g0r + 1;
                        constexpr int nin = ky == 0 ? 6 : 4;
                        constexpr int nout = c == 8 ? 0 : ((c + 1) % 3 == 0 ? 6 : 4);
                        static_for<0, 6>([&](auto mc) {
                            constexpr int m = decltype(mc)::value, t = m / 2, r = m % 2;
                            constexpr int as = t == 2 ? 1 : 0, bp = t == 1 ? 1 : 0;         // ah bh, ah bl, al bh
                            if constexpr (r == 0) {
                                constexpr int need = t == 0 ? (nin == 6 ? 3 : 2) : (t == 1 ? nin - 1 : nin);
                                constexpr int issued = m < nout ? m : nout;
                                lds_wait_n<nin - need + issued>();
                                asm volatile("" : "+v"(Aw[c & 1][as]), "+v"(bsr[g0r & 3][bp]), "+v"(bsr[g1r & 3][bp]));
                            }
                            const f16x8 &bb = bsr[(r == 0 ? g0r : g1r) & 3][bp];
                            f32x4 &cc2 = acc[j][r];
                            cc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Aw[c & 1][as], bb, cc2, 0, 0, 0);
                            __builtin_amdgcn_sched_barrier(0);
                            if constexpr (m < nout) {
                                rd(mc, std::integral_constant<int, (c + 1) % 9>{});
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        });
                    });
                    if constexpr (ABL & 128) {
                        f32x4 &c0 = acc[j][0], &c1 = acc[j][1];
                        asm volatile("" : "+v"(c0), "+v"(c1));
                        asm volatile("s_nop 0" ::: "memory");
                    }
                    MVS_LAP(1);
                }
                pair = pair == NPAIR - 1 ? 0 : pair + 1;
            });
            wsel ^= 1;
        }
        // ---- epilogue of the column: undo the operand scales, BN affine, ReLU, one 16-byte store per lane and row
        const int tb = __builtin_amdgcn_readfirstlane(cur.b), oy0 = __builtin_amdgcn_readfirstlane(cur.ty) * 8;
        const int ox0 = __builtin_amdgcn_readfirstlane(cur.tx) * 32, ozg = __builtin_amdgcn_readfirstlane(cur.zg) * 16;
        static_for<0, T>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if (j >= nvalid) return;
            const int oz0 = ozg + 2 * j;
            const int64_t base = ((((int64_t)tb * a.Do + oz0) * a.Ho + oy0) * a.Wo + ox0) * 8;
            float *const ob = a.out + base;
            const float *const rp = (a.residual && !(ABL & 128)) ? a.residual + base : nullptr;
            const bool xz_in = oz0 + z0 < a.Do && ox0 + ex < a.Wo;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                f32x4 v = acc[j][r];
                acc[j][r] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (!xz_in || oy0 + y0 + r >= a.Ho) continue;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = (v[i] * isx) * isw;
                v[0] = v[0] * sc.x + sh.x; v[1] = v[1] * sc.y + sh.y;
                v[2] = v[2] * sc.z + sh.z; v[3] = v[3] * sc.w + sh.w;
                if (a.relu == 1) {
                    v[0] = relu_nan(v[0]); v[1] = relu_nan(v[1]);
                    v[2] = relu_nan(v[2]); v[3] = relu_nan(v[3]);
                }
                const int o = eoff + r * a.Wo * 8;
                if (rp) {
                    const float4 rs = *reinterpret_cast<const float4 *>(rp + o);
                    v[0] += rs.x; v[1] += rs.y; v[2] += rs.z; v[3] += rs.w;
                }
                *reinterpret_cast<float4 *>(ob + o) = make_float4(v[0], v[1], v[2], v[3]);
                vmax = amax4_nan(vmax, v[0], v[1], v[2], v[3]);
            }
        });
    }
    publish_absmax(out_absmax, vmax);
    if constexpr (ABL & 128) {
        MVS_LAP(2);
        if (lane == 0) {
            long long *dbg = reinterpret_cast<long long *>(const_cast<float *>(a.residual)) + ((int64_t)blockIdx.x * 8 + wv) * 8;
            for (int k = 0; k < 3; ++k) dbg[k] = tsum[k];
        }
    }
#undef MVS_LAP
}

// fp32 volume [B,D,H,C/8,W,8] (MVS_LAYOUT_C8) -> fp16 pairs of x * 2^(14 - exponent(block)) in MVS_LAYOUT_C8P / C8PT.
// One thread per voxel-chunk (8 values).  The block must bound |x| (values beyond it overflow fp16).
__global__ __launch_bounds__(256) void c8_to_c8p_kernel(const float *__restrict__ in, const unsigned *__restrict__ absmax,
                                                        int D, int H, int W, int G, PairsGeom pg, int64_t total,
                                                        unsigned char *__restrict__ out) {
    const float s = pow2f(14 - absmax_exponent(load_absmax(absmax)));
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int64_t r = i;
        const int x = (int)(r % W); r /= W;
        const int g = (int)(r % G); r /= G;
        const int y = (int)(r % H); r /= H;            // r = b * D + d
        f32x4 v0 = *reinterpret_cast<const f32x4 *>(in + i * 8), v1 = *reinterpret_cast<const f32x4 *>(in + i * 8 + 4);
        u32x4 h, l;
        split2_block(v0, v1, s, h, l);
        unsigned char *const pl = out + r * pg.plane + g * pg.chunk;
        if (!pg.tiled) {
            unsigned char *o = pl + (x & 1) * pg.region + (int64_t)y * pg.rowpitch + (x >> 1) * 16;
            *reinterpret_cast<u32x4 *>(o) = h;
            *reinterpret_cast<u32x4 *>(o + 2 * pg.region) = l;
            if (pg.strips) {      // a tile-border column once more, into the neighbouring tile's halo strip
                const int tx = x >> 5, xm = x & 31;
                int64_t so = -1;
                if (xm == 31 && x + 1 < W) so = (int64_t)((tx + 1) * 2) * pg.strip;
                else if (xm == 0 && x > 0) so = (int64_t)((tx - 1) * 2 + 1) * pg.strip;
                if (so >= 0) {
                    unsigned char *d = pl + pg.halo + so + y * 16;
                    *reinterpret_cast<u32x4 *>(d) = h;
                    *reinterpret_cast<u32x4 *>(d + pg.halo_part) = l;
                }
            }
        } else {
            // own tile, and as a halo voxel of the neighbour: x' = 32 of tile tx is x' = 0 of tile tx + 1; x' = 1 is x' = 33 of tx - 1
            const int tx = x >> 5, xl = (x & 31) + 1;
#pragma unroll
            for (int dup = 0; dup < 2; ++dup) {
                int t = tx, xx = xl;
                if (dup) {
                    if (xl == 32) { t = tx + 1; xx = 0; }
                    else if (xl == 1) { t = tx - 1; xx = 33; }
                    else break;
                    if (t < 0 || t * 32 >= W) break;
                }
                unsigned char *o = pl + t * pg.xtile + (xx & 1) * pg.region + (int64_t)y * pg.rowpitch + (xx >> 1) * 16;
                *reinterpret_cast<u32x4 *>(o) = h;
                *reinterpret_cast<u32x4 *>(o + 2 * pg.region) = l;
            }
        }
    }
}

}  // namespace mvs

using namespace mvs;

extern "C" size_t mvs_conv3d_f16x3_packed_bytes(int Cin);

// the word of a conv0 pack that is NaN when the layer's weights are not finite -- what a hand-over sweep takes as its veto
extern "C" const void *mvs_conv3d_f16x3_pack_veto_word(const void *packed, int Cin) {
    if (!packed || mvs_conv3d_f16x3_packed_bytes(Cin) == 0) return nullptr;
    return static_cast<const unsigned char *>(packed) + (size_t)(Cin / 8) * kYPChunkBytes;
}

extern "C" size_t mvs_c8p_bytes(int B, int C, int D, int H, int W, int layout) {
    if (B <= 0 || C <= 0 || C % 8 || D <= 0 || H <= 0 || W <= 0 || !pairs_layout_ok(layout)) return 0;
    return (size_t)B * D * pairs_geom(C, H, W, layout).plane;
}

extern "C" int mvs_c8_to_c8p_f32(const float *in, const void *absmax, int B, int C, int D, int H, int W, int layout, void *out, void *stream) {
    if (!in || !absmax || !out || mvs_c8p_bytes(B, C, D, H, W, layout) == 0) {
        set_error("mvs_c8_to_c8p_f32: invalid argument");
        return MVS_EINVAL;
    }
    const int64_t total = (int64_t)B * D * H * (C / 8) * W;
    const int64_t blocks = (total + 255) / 256;
    hipLaunchKernelGGL(c8_to_c8p_kernel, dim3((unsigned)(blocks > 16384 ? 16384 : blocks)), dim3(256), 0, as_stream(stream), in,
                       static_cast<const unsigned *>(absmax), D, H, W, C / 8, pairs_geom(C, H, W, layout), total, static_cast<unsigned char *>(out));
    return check_launch("mvs_c8_to_c8p_f32");
}

extern "C" int mvs_conv3d_c8p_f16x3_f32(const void *in_pairs, const void *in_absmax, const void *redo, const void *packed, const float *scale,
                                        const float *shift, const float *residual, int relu, int B, int Cin,
                                        int D, int H, int W, int layout, int npair, float *out, void *out_absmax, void *stream) {
#ifdef MVS_TUNING
    const bool npair_ok = npair == 4 || npair == 5;
#else
    const bool npair_ok = npair == 4;        // (copies three steps ahead, npair 5, measured no faster: tuning build only)
#endif
    if (!in_pairs || !in_absmax || !packed || !out || mvs_c8p_bytes(B, Cin, D, H, W, layout) == 0 || mvs_conv3d_f16x3_packed_bytes(Cin) == 0 ||
        !npair_ok) {
        set_error("mvs_conv3d_c8p_f16x3_f32: invalid argument (Cin in {8, 16, 32}, Cout = 8, stride 1; in_pairs = MVS_LAYOUT_C8P / C8PT volume, "
                  "in_absmax = the block it was scaled by, packed = mvs_conv3d_pack_weights_f16x3_f32, npair 4)");
        return MVS_EINVAL;
    }
    const PairsGeom pg = pairs_geom(Cin, H, W, layout);
    if (3 * pg.plane >= 0xffffff00LL) return bare_error(MVS_EINVAL, __func__, __LINE__);
    ConvArgs a;
    a.in = static_cast<const float *>(in_pairs); a.wpk = static_cast<const float *>(packed); a.scale = scale; a.shift = shift;
    a.residual = residual; a.out = out;
    a.B = B; a.D = D; a.H = H; a.W = W;
    a.Do = D; a.Ho = H; a.Wo = W;
    a.tiles_x = (W + 31) / 32; a.tiles_y = (H + 7) / 8; a.tiles_z = (D + 1) / 2;
    a.relu = relu; a.in_c8 = 1; a.ystrip = 8; a.res_up2 = 0;
    const int64_t ng = (int64_t)B * a.tiles_x * a.tiles_y * ((a.tiles_z + kYPT - 1) / kYPT);
    if (ng <= 0 || ng > 0x7fffffffLL) return bare_error(MVS_EINVAL, __func__, __LINE__);
    const int n_cu = device_cu_count();
    hipStream_t st = as_stream(stream);
    const dim3 grid((unsigned)(ng < n_cu ? ng : n_cu)), blk(kYPThreads);
    const unsigned *mx = static_cast<const unsigned *>(in_absmax);
    unsigned *omx = static_cast<unsigned *>(out_absmax);
    const unsigned *rd = static_cast<const unsigned *>(redo);
#define MVS_YP_LAUNCH(C, NP, AB) hipLaunchKernelGGL((conv3d_c8p_f16x3_kernel<C, NP, AB>), grid, blk, 0, st, a, pg, (int)ng, mx, rd, omx)
#ifdef MVS_TUNING
    static const int abl = [] { const char *e = getenv("MVS_CONV_SPLIT_ABL"); return e ? atoi(e) : 0; }();
    if ((abl & 128) && Cin == 32) {
        if (!residual) return bare_error(MVS_EINVAL, __func__, __LINE__);
        if (npair == 4) MVS_YP_LAUNCH(32, 4, 128); else MVS_YP_LAUNCH(32, 5, 128);
        return check_launch("mvs_conv3d_c8p_f16x3_f32");
    }
    if ((abl & 2) && Cin == 32) {   // no MFMA phase: copies and barriers only (wrong results)
        if (npair == 4) MVS_YP_LAUNCH(32, 4, 2); else MVS_YP_LAUNCH(32, 5, 2);
        return check_launch("mvs_conv3d_c8p_f16x3_f32");
    }
#endif
    if (npair == 4) {
        if (Cin == 32) MVS_YP_LAUNCH(32, 4, 0); else if (Cin == 16) MVS_YP_LAUNCH(16, 4, 0); else MVS_YP_LAUNCH(8, 4, 0);
    }
#ifdef MVS_TUNING
    else {
        if (Cin == 32) MVS_YP_LAUNCH(32, 5, 0); else if (Cin == 16) MVS_YP_LAUNCH(16, 5, 0); else MVS_YP_LAUNCH(8, 5, 0);
    }
#endif
#undef MVS_YP_LAUNCH
    return check_launch("mvs_conv3d_c8p_f16x3_f32");
}

// conv0 on the volume of a hand-over sweep (include/mvs_hip.h): the kernel above on the pieces (returns at once if *redo != 0), then
// mvs_conv3d_c8_f16x3_f32 on the fp32 volume under *redo as its run-only-if word (returns at once if *redo == 0) -- exactly one of
// the two writes `out` and maxes into out_absmax.  No host synchronisation.
extern "C" int mvs_conv3d_c8_handed_f16x3_f32(const void *volume, const void *hand, const void *redo, const void *var_absmax,
                                              const void *packed, const float *scale, const float *shift, const float *residual, int relu,
                                              int B, int Cin, int D, int H, int W, float *out, void *out_absmax, void *stream) {
    if (!hand || !redo || !var_absmax) {
        set_error("mvs_conv3d_c8_handed_f16x3_f32: needs the hand-over block, the redo word and the volume's absmax block of "
                  "mvs_costvol_variance_fwd_ws3_f32");
        return MVS_EINVAL;
    }
    int rc = mvs_conv3d_c8p_f16x3_f32(volume, hand, redo, packed, scale, shift, residual, relu, B, Cin, D, H, W, kPairsLayoutStrips, 4, out,
                                      out_absmax, stream);
    if (rc != MVS_OK) return rc;
    struct FlagScope {
        explicit FlagScope(const void *f) { conv_run_flag() = static_cast<const unsigned *>(f); }
        ~FlagScope() { conv_run_flag() = nullptr; }
    } scope(redo);
    return mvs_conv3d_c8_f16x3_f32(static_cast<const float *>(volume), var_absmax, packed, scale, shift, residual, relu, B, Cin, D, H, W, out,
                                   out_absmax, stream);
}
