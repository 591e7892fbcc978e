// conv0-class layers (3x3x3, Cout = 8, stride 1, 8-channel-blocked input; mvsnet.py:66, module.py:26-33) on two-piece fp16
// operands -- the arithmetic, the operand scale, the packed weights and the range guard of conv_f16x3.hip, bit for bit --
// with EIGHT-row tiles marched two planes at a time (round 6).
//
// conv3d_c8_f16x3_zs_kernel is bound by what a CU can copy: its (4, 4, 32)-voxel tiles need (6, 6, 34)-voxel halos, so
// every input voxel is requested 18/16 x 6/4 x 34/32 = 1.79 times (5.7 GB per launch at config 2, 1.0 ms with everything
// but the copies removed), and the tile could not grow in y because the fp32 staging of four planes x six rows, twice,
// took 80 of the 156 KiB.  Here a tile is (2, 8, 32) voxels: wave w of the eight multiplying waves still owns two
// output rows that share their weight fragments -- (z0 = w >> 2, y0 = 2 (w & 3)) and (z0, y0 + 1) -- and the MFMA
// phase is the same 54 products per wave and step, but a step now brings TWO new planes of ten rows: 18/16 x 10/8 x
// 34/32 = 1.49 requests per voxel (-17 % copied and split bytes), 21.3 KiB per staging buffer instead of 40.
// A work unit is a column of T = 8 z-neighbouring tiles (16 output planes, 18 input planes); per 8-channel chunk the
// column is marched in nvalid + 1 steps of two planes each (step 0 only fills the ring), the accumulators of all eight
// tiles stay in registers across the chunk loop (64 registers) and a chunk's weights are fetched once per column.
// Steps form one stream s = 0, 1, ... across chunks and columns: staging buffer = s & 1, ring pair = s mod NPAIR.
//
//   NB = 2 (4 ring slots): barrier, split pass (staging -> two fp16 part planes), barrier, MFMA phase -- the schedule
//          of conv_f16x3.hip; the copy waves wait for a step's rows just before its first barrier, with the next
//          step's copies still in flight (counted vmcnt), so a step's copies have ~1.5 steps to land.
//   NB = 1 (6 ring slots): ONE barrier per step: behind it every wave splits the rows of step s + 1 into the pair of
//          slots nobody reads, then multiplies step s.  The second barrier and its ~470 cycles of waiting for the
//          slowest wave are gone; the copies of step s + 2 have one step to land.
// LDS: 2 x 18 KiB weights + 2 x 22 KiB staging + 2 parts x (4 | 6) slots x (10 rows x 17 voxels x 2 parities x 16 B)
//      = 123 / 145 KiB.
#include "conv_split_common.h"
#include "conv_guard.h"

#include <cstdlib>

namespace mvs {

constexpr int kY8ChunkBytes = 9 * 2 * 1024;   // A fragments of one 8-channel chunk (the pack of conv_f16x3.hip)
constexpr int kY8RowVox = 17, kY8Rows = 10, kY8T = 8, kY8CopyWaves = 4, kY8Threads = 512 + 64 * kY8CopyWaves;

__device__ __forceinline__ void wait_vmcnt_upto(int n) {   // wave-uniform n: at most n vector-memory operations outstanding
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
    default: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    }
}

template <int CIN, int NB, int ABL = 0>
__global__ __launch_bounds__(kY8Threads) void conv3d_c8_f16x3_y8_kernel(ConvArgs a, int ngroups,
                                                                        const unsigned *__restrict__ in_absmax,
                                                                        unsigned *__restrict__ out_absmax,
                                                                        unsigned long long *guard_cnt) {
    constexpr int NCHUNK = CIN / 8, YT = kY8Rows, T = kY8T, NC = kY8CopyWaves, NT = kY8Threads;
    constexpr int NPAIR = NB == 2 ? 2 : 3, NSLOT = 2 * NPAIR;
    constexpr int ROWP = 68;                                        // 16-byte pieces per (z, y) row: 34 voxels x 2 halves
    constexpr int NPIECE = 2 * YT * ROWP;                           // a step: two planes
    constexpr int NCOPY = (NPIECE + 63) / 64, IPW = (NCOPY + NC - 1) / NC;
    constexpr int WBYTES = kY8ChunkBytes, WCOPIES = WBYTES / 1024;
    constexpr int FBYTES = NCOPY * 1024;
    constexpr int SLOT = YT * kY8RowVox * 16;                       // bytes of one plane slot of one parity inside a part
    constexpr int ODDB = (NSLOT * SLOT + 255) / 256 * 256;          // odd-x half of a part: a multiple of 256 bytes behind the even one
    constexpr int SPART = 2 * ODDB, SBYTES = 2 * SPART;
    constexpr int F_OFF = 2 * WBYTES, S_OFF = F_OFF + 2 * FBYTES;
    static_assert(S_OFF + SBYTES <= 160 * 1024, "LDS budget");
    static_assert(SPART + 3 * kY8RowVox * 16 + SLOT * NSLOT < 65536 + SLOT * NSLOT, "ds offset field");
    __shared__ __attribute__((aligned(16))) unsigned char lds[S_OFF + SBYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4;
    const unsigned lds_base = (unsigned)(uintptr_t)lds;
    const bool copier = wv >= 8;
    const int cw = wv - 8;

    // operand scale of the input and what undoes it and the weights' scale; the range guard (conv_f16x3.hip, conv_guard.h)
    const AbsmaxVerdict verdict = absmax_verdict(in_absmax);
    const int xe = absmax_exponent(verdict.bits);
    const float sx = pow2f(14 - xe), isx = pow2f(xe - 14);
    const float isw = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(
        __builtin_bit_cast(int, a.wpk[(size_t)NCHUNK * (WBYTES / 4)])));
    if (verdict.code != 0 || isw != isw) {
        GuardConv g;
        g.in = a.in; g.w = a.wpk + (size_t)NCHUNK * (WBYTES / 4) + 4; g.scale = a.scale; g.shift = a.shift; g.residual = a.residual;
        g.out = a.out; g.out_absmax = out_absmax; g.counter = guard_cnt;
        g.B = a.B; g.D = a.D; g.H = a.H; g.W = a.W; g.Cin = CIN; g.Do = a.Do; g.Ho = a.Ho; g.Wo = a.Wo;
        g.ldc = 8; g.co0 = 0; g.nco = 8; g.kd = 3; g.kh = 3; g.stride = 1; g.transposed = 0; g.relu = a.relu; g.in_c8 = 1; g.out_c4 = 0;
        guard_direct_conv(g);
        return;
    }

    // this workgroup's columns: g0 + k * g_step, k < ngw (XCD x owns a contiguous slice of the ty-fastest column list)
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

    // ---- the split pass: staging piece P (row = P / 68 = zl * 10 + y, q = P % 68 -> voxel x = q / 2, channel half q & 1)
    // -> 8 bytes of each part at slot(2 pair + zl) + (y * 17 + x / 2) * 16 (+ ODDB for odd x) + half * 8
    constexpr int NPS = (NPIECE + NT - 1) / NT;                     // 2
    static_assert(NPS == 2, "one split block per thread");
    unsigned spos[NPS];
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps) {
        const int P = ps * NT + tid;
        const int Pc = P < NPIECE ? P : 0;
        const int row = Pc / ROWP, q = Pc % ROWP, x = q >> 1;
        spos[ps] = (unsigned)((row / YT) * SLOT + ((row % YT) * kY8RowVox + (x >> 1)) * 16 + (x & 1) * ODDB + (q & 1) * 8);
    }
    const bool p1_live = NT + tid < NPIECE;
    auto split_pass = [&](int buf, int pair) {                        // staging buffer `buf` -> ring pair `pair`
        f32x4 x0, x1;
        const unsigned fp = lds_base + (unsigned)(F_OFF + buf * FBYTES + tid * 16);
        x0 = lds_read_b128<0>(fp);
        x1 = lds_read_b128<0>(fp + (p1_live ? NT * 16 : 0));
        lds_wait_n<0>();
        if constexpr (ABL & 1) return;                               // tuning: barriers and copies only
        asm volatile("" : "+v"(x0), "+v"(x1));
        u32x4 hu, lu;
        split2_block(x0, x1, sx, hu, lu);
        const unsigned sbase = lds_base + (unsigned)(S_OFF + pair * 2 * SLOT);
        lds_write_b64<0>(sbase + spos[0], hu[0], hu[1]);
        lds_write_b64<SPART>(sbase + spos[0], lu[0], lu[1]);
        if (p1_live) {
            lds_write_b64<0>(sbase + spos[1], hu[2], hu[3]);
            lds_write_b64<SPART>(sbase + spos[1], lu[2], lu[3]);
        }
        lds_wait_n<0>();
    };

    if (copier) {
        // ================================================================ copy waves
        int loc[IPW];
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            const int P = (i * NC + cw) * 64 + lane;
            const int Pc = P < NPIECE ? P : 0;
            const int row = Pc / ROWP, q = Pc % ROWP;
            loc[i] = (q >> 1) | ((row % YT) << 8) | ((row / YT) << 16) | ((q & 1) << 24);
        }
        const int64_t plane_in = (int64_t)a.H * a.W * CIN;
        const int row_in = a.W * CIN;
        const unsigned window_bytes = (unsigned)min((int64_t)2 * plane_in * 4, (int64_t)0xffffff00u);
        unsigned voff[IPW];       // byte offset from the first plane of a step; 0xffffff00 = outside the image in x or y
        Grp cg{0, 0, 0, 0};
        auto geometry = [&](int g) {
            cg = decode(g);
            const int ix0 = cg.tx * 32 - 1, iy0 = cg.ty * 8 - 1;
#pragma unroll
            for (int i = 0; i < IPW; ++i) {
                const int gx = ix0 + (loc[i] & 255), gy = iy0 + ((loc[i] >> 8) & 255);
                const int lz = (loc[i] >> 16) & 255, h = (loc[i] >> 24) & 1;
                const bool ok = (unsigned)gx < (unsigned)a.W && (unsigned)gy < (unsigned)a.H;
                voff[i] = ok ? (unsigned)(((int64_t)lz * plane_in + (int64_t)gy * row_in + gx * 8 + h * 4) * 4) : 0xffffff00u;
            }
        };
        auto issue_halo = [&](int p, int ch, int buf) {               // planes 2p, 2p + 1 of the column (global z = 16 zg - 1 + ...)
            const int zs = cg.zg * 16 - 1 + 2 * p;
            const mvs_srd_t srd = make_srd(a.in + ((int64_t)cg.b * a.D + zs) * plane_in, window_bytes);
            const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(ch * a.W * 32);   // (the one-barrier form keeps the iterator in a VGPR)
            int cnt = 0;
#pragma unroll
            for (int i = 0; i < IPW; ++i) {
                if (i * NC + cw >= NCOPY) continue;   // wave-uniform
                const int P = (i * NC + cw) * 64 + lane, lz = (loc[i] >> 16) & 255;
                const bool ok = P < NPIECE && (unsigned)(zs + lz) < (unsigned)a.D;
                glds16_buf(ok ? voff[i] : 0xffffff00u, srd, soff, lds_base + (unsigned)(F_OFF + buf * FBYTES + (i * NC + cw) * 1024));
                ++cnt;
            }
            return cnt;
        };
        const unsigned char *wsrc = reinterpret_cast<const unsigned char *>(a.wpk);
        constexpr int WHALF = (WCOPIES + 1) / 2;
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
        // issue iterator: runs two steps ahead of the step being split
        int it_k = 0, it_ch = 0, it_p = 0, it_nvalid = 0, it_buf = 0;
        bool it_done = false;
        auto it_open = [&]() { geometry(g0 + it_k * g_step); it_nvalid = min(T, a.tiles_z - cg.zg * T); };
        auto it_issue = [&]() {
            if (it_done) return 0;
            const int cnt = issue_halo(it_p, it_ch, it_buf);
            it_buf ^= 1;
            if (++it_p > it_nvalid) {
                it_p = 0;
                if (++it_ch >= NCHUNK) {
                    it_ch = 0;
                    if (++it_k >= ngw) it_done = true; else it_open();
                }
            }
            return cnt;
        };
        int wsel = 0, buf = 0, pair = 0;
        it_open();
        issue_weights(0, 0, 0, WCOPIES);
        it_issue();
        int newer = it_issue();        // operations issued behind the rows of the step about to be split
        if constexpr (NB == 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            split_pass(0, 0);
        }
        for (int k = 0; k < ngw; ++k) {
            const int nvalid = min(T, a.tiles_z - decode(g0 + k * g_step).zg * T);
#pragma unroll 1
            for (int ch = 0; ch < NCHUNK; ++ch) {
                const bool more = ch + 1 < NCHUNK || k + 1 < ngw;
                const int nch = ch + 1 < NCHUNK ? ch + 1 : 0;
#pragma unroll 1
                for (int p = 0; p <= nvalid; ++p) {
                    if constexpr (NB == 2) {
                        wait_vmcnt_upto(newer);    // this step's rows have landed; the next step's may still be in flight
                        __syncthreads();
                        split_pass(buf, pair);
                        __syncthreads();           // ... and have been split: that staging buffer is free
                        int cnt = 0;
                        if (more && p == 0) cnt += issue_weights(nch, wsel ^ 1, 0, WHALF);
                        if (more && p == 1) cnt += issue_weights(nch, wsel ^ 1, WHALF, WCOPIES);
                        cnt += it_issue();
                        newer = cnt;
                    } else {
                        __syncthreads();           // step s split, step s - 1 multiplied: staging buffer s & 1 and ring pair (s + 1) % 3 are free
                        if (more && p == 0) issue_weights(nch, wsel ^ 1, 0, WHALF);
                        if (more && p == 1) issue_weights(nch, wsel ^ 1, WHALF, WCOPIES);
                        it_issue();                // rows of step s + 2 -> staging buffer s & 1
                        const bool last = !more && p == nvalid;
                        if (!last) split_pass(buf ^ 1, pair == NPAIR - 1 ? 0 : pair + 1);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                    buf ^= 1;
                    pair = pair == NPAIR - 1 ? 0 : pair + 1;
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
    // this lane's B voxel of row y0 inside a plane slot: x = 2n + kq
    const unsigned aB = lds_base + (unsigned)(S_OFF + (y0 * kY8RowVox + n + (kq >> 1)) * 16 + (kq & 1) * ODDB);

    f32x4 acc[T][2];
#pragma unroll
    for (int j = 0; j < T; ++j) acc[j][0] = acc[j][1] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int wsel = 0, buf = 0, pair = 0;
    float vmax = 0.0f;       // largest magnitude this lane has stored (-> out_absmax, the next layer's operand scale)
    long long tsum[5] = {0, 0, 0, 0, 0};   // tuning build, ABL & 128: cycles in barrier 1 / split pass / barrier 2 / MFMA phase / the rest
    long long tprev = 0;
#define MVS_LAP(k) do { if constexpr (ABL & 128) { const long long tn = clock64(); tsum[k] += tn - tprev; tprev = tn; } } while (0)
    if constexpr (NB == 1) {
        __syncthreads();
        split_pass(0, 0);
    }
    if constexpr (ABL & 128) tprev = clock64();
    for (int k = 0; k < ngw; ++k) {
        const Grp cur = decode(__builtin_amdgcn_readfirstlane(g0 + k * g_step));
        const int nvalid = min(T, a.tiles_z - cur.zg * T);
#pragma unroll 1
        for (int ch = 0; ch < NCHUNK; ++ch) {
            const bool more = ch + 1 < NCHUNK || k + 1 < ngw;
            static_for<0, T + 1>([&](auto pc) {
                constexpr int p = decltype(pc)::value;
                if (p > nvalid) return;   // wave-uniform
                const int prev = pair == 0 ? NPAIR - 1 : pair - 1;
                MVS_LAP(4);
                if constexpr (NB == 2) {
                    __syncthreads();
                    MVS_LAP(0);
                    split_pass(buf, pair);
                    MVS_LAP(1);
                    __syncthreads();
                    MVS_LAP(2);
                } else {
                    __syncthreads();
                    MVS_LAP(0);
                    const bool last = !more && p == nvalid;
                    if (!last) split_pass(buf ^ 1, pair == NPAIR - 1 ? 0 : pair + 1);
                    MVS_LAP(1);
                }
                if constexpr (p >= 1 && !(ABL & 2)) {
                    constexpr int j = p - 1;
                    // ---- MFMA phase (conv_f16x3.hip): nine blocks c = (kz, ky) of six MFMAs -- the weight pair A(kz, ky) against
                    // input row (kz, ky) into output row 0 and against input row (kz, ky + 1) into output row 1, alternating
                    // between the two accumulators: ah bh, ah bl, al bh.  Plane q = z0 + kz of the tile: q < 2 in the previous
                    // step's pair of slots, else in this step's.
                    unsigned aBz[3];
#pragma unroll
                    for (int kz = 0; kz < 3; ++kz) {
                        const int q = z0 + kz;
                        aBz[kz] = aB + (unsigned)(((q >> 1 ? pair : prev) * 2 + (q & 1)) * SLOT);
                    }
                    const unsigned aA = lds_base + (unsigned)(wsel * WBYTES + lane * 16);
                    f16x8 bsr[4][2], Aw[2][2];        // input rows g = kz * 4 + iy in slot g % 4; weight pairs c in slot c % 2
                    auto rd = [&](auto ic, auto cc) {   // i-th read of the set that block c needs
                        constexpr int i = decltype(ic)::value, c = decltype(cc)::value, kz = c / 3, ky = c % 3;
                        constexpr bool two = ky == 0;                     // both rows are new at a kz change
                        constexpr int nin = two ? 6 : 4;
                        if constexpr (i == 0 || i == nin - 1) {
                            constexpr int sp = i == 0 ? 0 : 1;
                            Aw[c & 1][sp] = __builtin_bit_cast(f16x8, lds_read_b128<(c * 2 + sp) * 1024>(aA));
                        } else {
                            constexpr int q = i - 1;                       // two: h(g0) h(g1) l(g0) l(g1); else h(g1) l(g1)
                            constexpr int sp = two ? q / 2 : q, iy = two ? ky + (q & 1) : ky + 1, g = kz * 4 + iy;
                            bsr[g & 3][sp] = __builtin_bit_cast(f16x8, lds_read_b128<iy * kY8RowVox * 16 + sp * SPART>(aBz[kz]));
                        }
                    };
                    static_for<0, 6>([&](auto ic) { rd(ic, std::integral_constant<int, 0>{}); });
                    __builtin_amdgcn_sched_barrier(0);
                    static_for<0, 9>([&](auto cc) {
                        constexpr int c = decltype(cc)::value, kz = c / 3, ky = c % 3, g0r = kz * 4 + ky, g1r = g0r + 1;
                        constexpr int nin = ky == 0 ? 6 : 4;                               // reads this block waits for
                        constexpr int nout = c == 8 ? 0 : ((c + 1) % 3 == 0 ? 6 : 4);      // reads it issues for block c + 1
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
                    MVS_LAP(3);
                }
                buf ^= 1;
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
        MVS_LAP(4);
        if (lane == 0) {
            long long *dbg = reinterpret_cast<long long *>(const_cast<float *>(a.residual)) + ((int64_t)blockIdx.x * 8 + wv) * 8;
            for (int k = 0; k < 5; ++k) dbg[k] = tsum[k];
        }
    }
#undef MVS_LAP
}

// launcher, called by mvs_conv3d_c8_f16x3_f32 (conv_f16x3.hip): variant 2 = two barriers per step, 1 = one
int launch_conv3d_c8_f16x3_y8(ConvArgs a, int Cin, int variant, const unsigned *in_absmax, unsigned *out_absmax,
                              unsigned long long *guard_cnt, hipStream_t st) {
    a.tiles_x = (a.W + 31) / 32; a.tiles_y = (a.H + 7) / 8; a.tiles_z = (a.D + 1) / 2;
    const int64_t ng = (int64_t)a.B * a.tiles_x * a.tiles_y * ((a.tiles_z + kY8T - 1) / kY8T);
    if (ng <= 0 || ng > 0x7fffffffLL) return bare_error(MVS_EINVAL, __func__, __LINE__);
    const int n_cu = device_cu_count();
    const dim3 grid((unsigned)(ng < n_cu ? ng : n_cu)), blk(kY8Threads);
#define MVS_Y8_LAUNCH(C, NBV, AB) hipLaunchKernelGGL((conv3d_c8_f16x3_y8_kernel<C, NBV, AB>), grid, blk, 0, st, a, (int)ng, in_absmax, out_absmax, guard_cnt)
#ifdef MVS_TUNING   // wrong results by design: 1 = no split work, 2 = no MFMA phase, 3 = copies and barriers only
    static const int abl = [] { const char *e = getenv("MVS_CONV_SPLIT_ABL"); return e ? atoi(e) : 0; }();
    if ((abl & 128) && Cin == 32) {   // phase stamps: cycle counters written through `residual` (scripts/exp_conv0_y8.py laps)
        if (!a.residual) return bare_error(MVS_EINVAL, __func__, __LINE__);
        if (variant == 2) MVS_Y8_LAUNCH(32, 2, 128); else MVS_Y8_LAUNCH(32, 1, 128);
        return check_launch("mvs_conv3d_c8_f16x3_f32 (y8)");
    }
    if ((abl & 3) && Cin == 32) {
        const int m = abl & 3;
        if (variant == 2) { if (m == 1) MVS_Y8_LAUNCH(32, 2, 1); else if (m == 2) MVS_Y8_LAUNCH(32, 2, 2); else MVS_Y8_LAUNCH(32, 2, 3); }
        else { if (m == 1) MVS_Y8_LAUNCH(32, 1, 1); else if (m == 2) MVS_Y8_LAUNCH(32, 1, 2); else MVS_Y8_LAUNCH(32, 1, 3); }
        return check_launch("mvs_conv3d_c8_f16x3_f32 (y8)");
    }
#endif
    if (variant == 2) {
        if (Cin == 32) MVS_Y8_LAUNCH(32, 2, 0); else if (Cin == 16) MVS_Y8_LAUNCH(16, 2, 0); else MVS_Y8_LAUNCH(8, 2, 0);
    } else {
        if (Cin == 32) MVS_Y8_LAUNCH(32, 1, 0); else if (Cin == 16) MVS_Y8_LAUNCH(16, 1, 0); else MVS_Y8_LAUNCH(8, 1, 0);
    }
#undef MVS_Y8_LAUNCH
    return check_launch("mvs_conv3d_c8_f16x3_f32 (y8)");
}

}  // namespace mvs
