// conv0-class layers (3x3x3, Cout = 8, stride 1, 8-channel-blocked input: MVSNet's conv0 32 -> 8 and the cascade's
// first layers; mvsnet.py:66, module.py:26-33) on the FP16 matrix pipe with TWO-piece operands: three products per
// fp32 product instead of the six of conv_bf16x6.hip.
//
// An fp32 number scaled by a power of two into fp16's range is hi + lo + e with hi = fp16(x s) (11 significand bits),
// lo = fp16(x s - hi) (the next 11; the subtraction is exact) and |e| <= 2^-23 |x s| -- one unit in fp32's own last
// place, rms about a third of that.  a b = ah bh + ah bl + al bh to 2^-22 relative worst case (al bl dropped), every
// product exact in the matrix pipe's fp32 accumulator.  Measured against a float64 convolution on variance-like data
// the operand error (rms 2.9e-8 at |y| <= 1.8) is a quarter of what ATen's fp32 convolution of the reference loses in
// its own accumulation (1.26e-7) -- the fp32 accumulation of the matrix pipe, the same in both split forms, is what
// sets this kernel's distance from the float64 answer, not the operands (tests: test_conv3d_f16x3_*; full-size error
// budget in DESIGN section 2).  What the two-piece form needs and the three-piece bf16 form does not is RANGE: fp16
// has 5 exponent bits.  The caller passes the largest magnitude of the input (an absmax block, mvs_common.h, filled by the kernel
// that produced the volume: mvs_costvol_variance_fwd_ws_f32, or by mvs_absmax_f32) and the kernel scales by
// s = 2^(14 - exponent(max)); the weights are scaled the same way when they are packed.  An element below 2^-18 of
// the maximum has its lo piece in fp16's subnormals: its absolute error stays below 2^-40 of the maximum.
//
// Structure = conv3d_c8_bf16x6_zs_kernel (groups of four z-neighbouring (4,4,32)-voxel tiles, ring of six plane
// slots, four copy waves, split pass between two barriers, alternating-accumulator MFMA phase), with
//   * the split as v_fma_mixlo/mixhi_f16 (scale and round in one instruction), v_fma_mix_f32 (x s - hi straight from
//     the packed pair) and v_cvt_pk_f16_f32: 20 vector instructions per 8 values instead of 28 + hazard nops;
//   * 54 MFMAs (v_mfma_f32_16x16x32_f16) and 42 ds_read_b128 per wave and step instead of 108 and 63;
//   * LDS 2 x 18 KiB weights + 2 x 40 KiB fp32 staging (copies run two steps ahead) + 40 KiB fp16 parts = 156 KiB.
#include "conv_split_common.h"
#include "conv_guard.h"

#include <cstdlib>

namespace mvs {

constexpr int kF16ChunkBytes = 9 * 2 * 1024;   // A fragments of one 8-channel chunk: (kz,ky) x (hi,lo) x 1 KiB
constexpr int kFRowVox = 17, kFOddBase = 624, kFHaloPlane = 1280;
constexpr int kFGroup = 4, kFCopyWaves = 4, kFThreads = 512 + 64 * kFCopyWaves;

template <int CIN, int ABL = 0>
__global__ __launch_bounds__(kFThreads) void conv3d_c8_f16x3_zs_kernel(ConvArgs a, int ngroups,
                                                                       const unsigned *__restrict__ in_absmax,
                                                                       unsigned *__restrict__ out_absmax,
                                                                       unsigned long long *guard_cnt) {
    constexpr int NCHUNK = CIN / 8, YT = 6, PLANE = kFHaloPlane, T = kFGroup;
    constexpr int NC = kFCopyWaves, NT = kFThreads;
    constexpr int ROWP = 68;                                        // 16-byte pieces per (z, y) row: 34 voxels x 2 halves
    constexpr int NPIECE0 = 36 * ROWP, NPIECE1 = 24 * ROWP;         // step 0: six planes; later steps: four
    constexpr int NCOPY0 = (NPIECE0 + 63) / 64, NCOPY1 = (NPIECE1 + 63) / 64, IPW = (NCOPY0 + NC - 1) / NC;
    constexpr int WBYTES = kF16ChunkBytes, WCOPIES = WBYTES / 1024;
    constexpr int FBYTES = 2 * PLANE * 16, SPART = PLANE * 16, SBYTES = 2 * SPART;
    constexpr int F_OFF = 2 * WBYTES, S_OFF = F_OFF + 2 * FBYTES;   // TWO staging buffers: a step's copies get a whole step to land
    constexpr int SLOT = YT * kFRowVox * 16;                        // bytes of one plane slot inside a part
    static_assert(NPIECE0 * 16 <= FBYTES && S_OFF + SBYTES <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(16))) unsigned char lds[S_OFF + SBYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4;
    const unsigned lds_base = (unsigned)(uintptr_t)lds;
    const bool copier = wv >= 8;
    const int cw = wv - 8;       // copy wave index

    if (a.run_flag && *a.run_flag == 0u) return;      // (uniform: a launch in front of this one did the work -- conv_f16x3_y8p.hip on a handed-over volume)
    // operand scale of the input and what undoes it and the weights' scale (the trailer of the packed weights)
    const AbsmaxVerdict verdict = absmax_verdict(in_absmax);
    const int xe = absmax_exponent(verdict.bits);
    const float sx = pow2f(14 - xe), isx = pow2f(xe - 14);
    const float isw = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(
        __builtin_bit_cast(int, a.wpk[(size_t)NCHUNK * (WBYTES / 4)])));
    // the range guard (conv_guard.h): a non-finite or outlier-dominated input, or non-finite weights (the pack kernel then
    // left a NaN where the scale goes) -- the layer in plain fp32 on the original weights, which lie behind the trailer
    if (verdict.code != 0 || isw != isw) {
        GuardConv g;
        g.in = a.in; g.w = a.wpk + (size_t)NCHUNK * (WBYTES / 4) + 4; g.scale = a.scale; g.shift = a.shift; g.residual = a.residual;
        g.out = a.out; g.out_absmax = out_absmax; g.counter = guard_cnt;
        g.B = a.B; g.D = a.D; g.H = a.H; g.W = a.W; g.Cin = CIN; g.Do = a.Do; g.Ho = a.Ho; g.Wo = a.Wo;
        g.ldc = 8; g.co0 = 0; g.nco = 8; g.kd = 3; g.kh = 3; g.stride = 1; g.transposed = 0; g.relu = a.relu; g.in_c8 = 1; g.out_c4 = 0;
        guard_direct_conv(g);
        return;
    }

    // this workgroup's groups: g0 + k * g_step, k < ngw
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
    const int ngz = (a.tiles_z + T - 1) / T;
    struct Grp { int tx, ty, zg, b; };
    auto decode = [&](int g) {
        Grp r;
        r.ty = g % a.tiles_y; g /= a.tiles_y;
        r.tx = g % a.tiles_x; g /= a.tiles_x;
        r.zg = g % ngz; r.b = g / ngz;
        return r;
    };

    // the split pass: staging piece P (row = P / 68 = zl * 6 + y, q = P % 68 -> voxel x = q / 2, channel half q & 1)
    // -> 8 bytes of each part at slot((zl + first slot of the step) mod 6) + (y * 17 + x / 2 (+ kFOddBase for odd x)) * 16
    // + half * 8.  The ragged tail (96 of 1632 pieces: a third round for two waves) goes to the two OLDEST waves: they win the issue
    // arbitration during the pass, so the round costs the workgroup least there (given to the copy waves -- the youngest -- as in
    // the bf16 kernel, every other wave waited ~400 cycles for them at the second barrier: 3493 -> 3391 cycles per step).
    const int tidr = tid;
    constexpr int NPS = (NPIECE0 + NT - 1) / NT;
    unsigned spos[NPS];          // in-slot byte position | zl << 16
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps) {
        const int P = ps * NT + tidr;
        const int Pc = P < NPIECE0 ? P : 0;
        const int row = Pc / ROWP, q = Pc % ROWP, x = q >> 1;
        spos[ps] = (unsigned)(((row % YT) * kFRowVox + (x >> 1) + (x & 1) * kFOddBase) * 16 + (q & 1) * 8) | ((unsigned)(row / YT) << 16);
    }
    auto split_pass = [&](auto jc, int par) {                         // par: which staging buffer holds this step's rows
        constexpr int j = decltype(jc)::value;
        constexpr int NP = j == 0 ? NPIECE0 : NPIECE1, NPSJ = (NP + NT - 1) / NT;
        constexpr int ZF = j == 0 ? 0 : (4 * j + 2) % 6;             // ring slot of the first staged plane
        f32x4 x[NPSJ];
        const unsigned fp = lds_base + (unsigned)(F_OFF + par * FBYTES + tidr * 16);
        const int wbase = tidr & ~63;                                 // wave-uniform
        static_for<0, NPSJ>([&](auto pc) {
            constexpr int ps = decltype(pc)::value;
            if (ps * NT + wbase < NP) x[ps] = lds_read_b128<ps * NT * 16>(fp);
        });
        lds_wait_n<0>();
        if constexpr (ABL & 1) { lds_wait_n<0>(); return; }          // tuning: barriers and copies only
        static_for<0, (NPSJ + 1) / 2>([&](auto pc) {
            constexpr int p0 = 2 * decltype(pc)::value, p1 = (p0 + 1 < NPSJ) ? p0 + 1 : p0;
            if (p0 * NT + wbase >= NP) return;                        // the whole wave has nothing here
            asm volatile("" : "+v"(x[p0]), "+v"(x[p1]));
            u32x4 hu, lu;
            split2_block(x[p0], x[p1], sx, hu, lu);
            auto dest = [&](unsigned sp) {
                unsigned s = (sp >> 16) + ZF;
                s = min(s, s - 6u);                                   // mod 6 (s < 12)
                return lds_base + (unsigned)S_OFF + (sp & 0xffffu) + s * (unsigned)SLOT;
            };
            if (p0 * NT + tidr < NP) {
                const unsigned sp = dest(spos[p0]);
                lds_write_b64<0>(sp, hu[0], hu[1]);
                lds_write_b64<SPART>(sp, lu[0], lu[1]);
            }
            if (p1 != p0 && p1 * NT + tidr < NP) {
                const unsigned sp = dest(spos[p1]);
                lds_write_b64<0>(sp, hu[2], hu[3]);
                lds_write_b64<SPART>(sp, lu[2], lu[3]);
            }
        });
        lds_wait_n<0>();
    };

    if (copier) {
        // ================================================================ copy waves
        int loc[IPW];
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            const int P = (i * NC + cw) * 64 + lane;
            const int Pc = P < NPIECE0 ? P : 0;
            const int row = Pc / ROWP, q = Pc % ROWP;
            loc[i] = (q >> 1) | ((row % YT) << 8) | ((row / YT) << 16) | ((q & 1) << 24);
        }
        const int64_t plane_in = (int64_t)a.H * a.W * CIN;
        const int row_in = a.W * CIN;
        const unsigned window_bytes = (unsigned)min((int64_t)6 * plane_in * 4, (int64_t)0xffffff00u);
        unsigned voff[IPW];       // byte offset from the first staged plane; 0xffffff00 = outside the image in x or y
        Grp cg{0, 0, 0, 0};
        auto geometry = [&](int g) {
            cg = decode(g);
            const int ix0 = cg.tx * 32 - 1, iy0 = cg.ty * 4 - 1;
#pragma unroll
            for (int i = 0; i < IPW; ++i) {
                const int gx = ix0 + (loc[i] & 255), gy = iy0 + ((loc[i] >> 8) & 255);
                const int lz = (loc[i] >> 16) & 255, h = (loc[i] >> 24) & 1;
                const bool ok = (unsigned)gx < (unsigned)a.W && (unsigned)gy < (unsigned)a.H;
                voff[i] = ok ? (unsigned)(((int64_t)lz * plane_in + (int64_t)gy * row_in + gx * 8 + h * 4) * 4) : 0xffffff00u;
            }
        };
        auto issue_halo = [&](int j, int ch, int par) {
            const bool first = j == 0;
            const int NP = first ? NPIECE0 : NPIECE1, NCP = first ? NCOPY0 : NCOPY1;
            const int zs = cg.zg * 16 + (first ? -1 : 4 * j + 1);    // first staged plane
            const mvs_srd_t srd = make_srd(a.in + ((int64_t)cg.b * a.D + zs) * plane_in, window_bytes);
            const unsigned soff = (unsigned)(ch * a.W * 32);
#pragma unroll
            for (int i = 0; i < IPW; ++i) {
                if (i * NC + cw >= NCP) continue;   // wave-uniform
                const int P = (i * NC + cw) * 64 + lane, lz = (loc[i] >> 16) & 255;
                const bool ok = P < NP && (unsigned)(zs + lz) < (unsigned)a.D;
                glds16_buf(ok ? voff[i] : 0xffffff00u, srd, soff,
                           lds_base + (unsigned)(F_OFF + par * FBYTES + (i * NC + cw) * 1024));
            }
        };
        const unsigned char *wsrc = reinterpret_cast<const unsigned char *>(a.wpk);
        constexpr int WHALF = (WCOPIES + 1) / 2;
        auto issue_weights = [&](int ch, int sel, int lo, int hi) {   // wave-copies [lo, hi) of a chunk's A fragments
#pragma unroll
            for (int i = 0; i < (WCOPIES + NC - 1) / NC; ++i) {
                const int g = i * NC + cw;
                if (g >= lo && g < hi) glds16(wsrc + (size_t)ch * WBYTES + (size_t)g * 1024 + lane * 16,
                                              lds_base + (unsigned)(sel * WBYTES + g * 1024));
            }
        };
        // The copies run TWO steps ahead of the step being split and multiplied: behind the second barrier of step s
        // the staging buffer s & 1 is free, and the rows of step s + 2 go there; the rows of step s + 1, issued a step
        // earlier, are waited for before the first barrier of step s + 1.  (One staging buffer, copies issued behind the
        // second barrier of step s for step s + 1: with the MFMA phase halved they no longer landed inside it -- every
        // wave waited ~700 cycles per step at the first barrier.)  The weights of the next chunk follow the first step(s)
        // of a chunk, when the buffer of the previous chunk is free.
        int it_k = 0, it_ch = 0, it_j = 0, it_nvalid = 0, it_par = 0;
        bool it_done = ngw <= 0;
        auto it_open = [&]() { geometry(g0 + it_k * g_step); it_nvalid = min(T, a.tiles_z - cg.zg * T); };
        auto it_issue = [&]() {
            if (it_done) return;
            issue_halo(it_j, it_ch, it_par);
            it_par ^= 1;
            if (++it_j >= it_nvalid) {
                it_j = 0;
                if (++it_ch >= NCHUNK) {
                    it_ch = 0;
                    if (++it_k >= ngw) it_done = true; else it_open();
                }
            }
        };
        int wsel = 0, par = 0;
        if (ngw > 0) {
            it_open();
            issue_weights(0, 0, 0, WCOPIES);
            it_issue();
            it_issue();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        for (int k = 0; k < ngw; ++k) {
            const int nvalid = min(T, a.tiles_z - decode(g0 + k * g_step).zg * T);
#pragma unroll 1
            for (int ch = 0; ch < NCHUNK; ++ch) {
                const bool more = ch + 1 < NCHUNK || k + 1 < ngw;
                const int nch = ch + 1 < NCHUNK ? ch + 1 : 0;
                static_for<0, T>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    if (j >= nvalid) return;   // wave-uniform
                    __syncthreads();           // the rows of this step are in its staging buffer (this wave has waited for them)
                    split_pass(jc, par);
                    __syncthreads();           // ... and have been split: that staging buffer is free
                    par ^= 1;
                    if (nvalid == 1) {         // the next step is already the next chunk: its weights must land with the rows
                        if (more) issue_weights(nch, wsel ^ 1, 0, WCOPIES);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    } else {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        if (more && j == 0) issue_weights(nch, wsel ^ 1, 0, WHALF);
                        if (more && j == 1) issue_weights(nch, wsel ^ 1, WHALF, WCOPIES);
                    }
                    it_issue();
                });
                wsel ^= 1;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }

    float4 sc, sh;
    {
        const int c0 = (kq & 1) * 4;
        sc = a.scale ? *reinterpret_cast<const float4 *>(a.scale + c0) : make_float4(1.f, 1.f, 1.f, 1.f);
        sh = a.shift ? *reinterpret_cast<const float4 *>(a.shift + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
    }

    const int z0 = wv >> 1, y0 = (wv & 1) * 2;          // the wave's output rows: (z0, y0) and (z0, y0 + 1)
    const int ex = 2 * n + (kq >> 1);
    const int eoff = ((z0 * a.Ho + y0) * a.Wo + ex) * 8 + (kq & 1) * 4;
    // this lane's B voxel of row y0 inside a plane slot: x = 2n + kq
    const unsigned aB = lds_base + (unsigned)(S_OFF + (y0 * kFRowVox + n + (kq >> 1) + (kq & 1) * kFOddBase) * 16);

    f32x4 acc[T][2];
#pragma unroll
    for (int j = 0; j < T; ++j) acc[j][0] = acc[j][1] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int wsel = 0, par = 0;
    float vmax = 0.0f;       // largest magnitude this lane has stored (-> out_absmax, the next layer's operand scale)
    long long tsum[7] = {0, 0, 0, 0, 0, 0, 0};
    long long tprev = 0;
    if constexpr (ABL & 128) tprev = clock64();
#define MVS_LAP(k) do { if constexpr (ABL & 128) { const long long tn = clock64(); tsum[k] += tn - tprev; tprev = tn; } } while (0)
    for (int k = 0; k < ngw; ++k) {
        const Grp cur = decode(__builtin_amdgcn_readfirstlane(g0 + k * g_step));
        const int nvalid = min(T, a.tiles_z - cur.zg * T);
#pragma unroll 1
        for (int ch = 0; ch < NCHUNK; ++ch) {
            static_for<0, T>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                if (j >= nvalid) return;   // wave-uniform
                MVS_LAP(6);
                __syncthreads();
                MVS_LAP(1);
                split_pass(jc, par);
                par ^= 1;
                MVS_LAP(2);
                __syncthreads();
                MVS_LAP(3);
                // ---- MFMA phase: nine blocks c = (kz, ky) of six MFMAs -- the weight pair A(kz, ky) against input row
                // (kz, ky) into output row 0 and against input row (kz, ky + 1) into output row 1, alternating between the
                // two accumulators: ah bh, ah bl, al bh.  Plane z0 + kz of the tile sits in ring slot (4j + z0 + kz) mod 6.
                // The reads of block c + 1 (A hi, the new row(s) hi, lo, A lo: 4, or 6 at a kz change) ride one per MFMA
                // behind block c's MFMAs, with counted waits in the order block c + 1 consumes them.
                unsigned aBz[3];
#pragma unroll
                for (int kz = 0; kz < 3; ++kz) {
                    int s = (4 * j) % 6 + z0 + kz;
                    s = s >= 6 ? s - 6 : s;
                    aBz[kz] = aB + (unsigned)(s * SLOT);
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
                        bsr[g & 3][sp] = __builtin_bit_cast(f16x8, lds_read_b128<iy * kFRowVox * 16 + sp * SPART>(aBz[kz]));
                    }
                };
                if constexpr (!(ABL & 2)) {
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
                }
                if constexpr (ABL & 128) {
                    f32x4 &c0 = acc[j][0], &c1 = acc[j][1];
                    asm volatile("" : "+v"(c0), "+v"(c1));
                    asm volatile("s_nop 0" ::: "memory");
                }
                MVS_LAP(5);
            });
            wsel ^= 1;
        }
        // ---- epilogue of the group: undo the operand scales, BN affine, ReLU, one 16-byte store per lane and row
        const int tb = __builtin_amdgcn_readfirstlane(cur.b), oy0 = __builtin_amdgcn_readfirstlane(cur.ty) * 4;
        const int ox0 = __builtin_amdgcn_readfirstlane(cur.tx) * 32, ozg = __builtin_amdgcn_readfirstlane(cur.zg) * 16;
        static_for<0, T>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if (j >= nvalid) return;
            const int oz0 = ozg + 4 * j;
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
        MVS_LAP(6);
        if (lane == 0) {
            long long *dbg = reinterpret_cast<long long *>(const_cast<float *>(a.residual)) + ((int64_t)blockIdx.x * 8 + wv) * 8;
            for (int k = 0; k < 7; ++k) dbg[k] = tsum[k];
        }
    }
#undef MVS_LAP
}

// largest magnitude of n floats into an absmax block (a NaN wins: its bit pattern is above every number's)
__global__ __launch_bounds__(256) void absmax_kernel(const float *__restrict__ x, int64_t n, unsigned *__restrict__ out) {
    unsigned m = 0;
    const int64_t n4 = n >> 2, stride = (int64_t)gridDim.x * 256;
    const uint4 *x4 = reinterpret_cast<const uint4 *>(x);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const uint4 v = x4[i];
        m = max(max(m, v.x & 0x7fffffffu), max(v.y & 0x7fffffffu, max(v.z & 0x7fffffffu, v.w & 0x7fffffffu)));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = max(m, __float_as_uint(x[n4 * 4 + threadIdx.x]) & 0x7fffffffu);
#pragma unroll
    for (int o = 32; o; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out + (blockIdx.x & (kAbsmaxWords - 1)), m);
}

// ... of a small array (a layer's weights) into ONE word, by one workgroup
__global__ __launch_bounds__(1024) void absmax_word_kernel(const float *__restrict__ x, int n, unsigned *__restrict__ out) {
    __shared__ unsigned s_m[16];
    unsigned m = 0;
    for (int i = threadIdx.x; i < n; i += 1024) m = max(m, __float_as_uint(x[i]) & 0x7fffffffu);
#pragma unroll
    for (int o = 32; o; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 16; ++i) m = max(m, s_m[i]);
        *out = m;
    }
}

// PyTorch-layout weight (8, Cin, 3, 3, 3) -> [chunk][kz*3+ky][hi,lo][lane][8 fp16] of w * 2^(14 - exponent(max |w|)),
// then one float: what undoes that scale.  Lane (m, kq): row m = (cout = m & 7, x-shift = m >> 3), k = kq * 8 + c:
// x-tap kx' = kq of the 4-tap window, channel c of the chunk; zero outside the 3 real taps.
__global__ __launch_bounds__(256) void pack_f16x3_kernel(const float *__restrict__ w, int Cin, unsigned short *__restrict__ out,
                                                         int total, const unsigned *__restrict__ wmax) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int e = absmax_exponent(*wmax);
    // what undoes the scale; NaN = "the weights are not finite", which sends every launch to the guard's fp32 path
    if (i == 0) *reinterpret_cast<float *>(out + (size_t)total * 2) = *wmax >= 0x7f800000u ? __builtin_nanf("") : pow2f(e - 14);
    // (the fp32 weights behind the 16-byte trailer, conv_guard.h: launch_guard_weights from the host side of this pack)
    if (i >= total) return;
    const int j = i & 7, lane = (i >> 3) & 63, t = (i >> 9) % 9, ch = i / (9 * 512);
    const int m = lane & 15, kq = lane >> 4, co = m & 7, sft = m >> 3, kx = kq - sft;
    const int kz = t / 3, ky = t % 3, cin = ch * 8 + j;
    float x = 0.0f;
    if (kx >= 0 && kx <= 2) x = w[((int64_t)co * Cin + cin) * 27 + kz * 9 + ky * 3 + kx] * pow2f(14 - e);
    const _Float16 h = (_Float16)x;
    const _Float16 l = (_Float16)(x - (float)h);
    unsigned short *o = out + ((size_t)(ch * 9 + t) * 2) * 512 + lane * 8 + j;
    o[0] = __builtin_bit_cast(unsigned short, h);
    o[512] = __builtin_bit_cast(unsigned short, l);
}

}  // namespace mvs

namespace mvs {
int launch_conv3d_c8_f16x3_y8(ConvArgs a, int Cin, int variant, const unsigned *in_absmax, unsigned *out_absmax,
                              unsigned long long *guard_cnt, hipStream_t st);   // conv_f16x3_y8.hip
}

using namespace mvs;

static bool f16x3_shape_ok(int Cin) { return Cin == 8 || Cin == 16 || Cin == 32; }

extern "C" size_t mvs_conv3d_f16x3_packed_bytes(int Cin) {
    // fragments + 16-byte trailer (scale, -, -, largest weight) + the original (8, Cin, 27) fp32 weights for the range guard
    return f16x3_shape_ok(Cin) ? (size_t)(Cin / 8) * kF16ChunkBytes + 16 + (size_t)8 * Cin * 27 * 4 : 0;
}

namespace mvs {
__global__ __launch_bounds__(256) void guard_weights_kernel(const float *__restrict__ w, int transposed, int Cin, int Cout, int ntap,
                                                            float *__restrict__ dst) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Cin * Cout * ntap) return;
    const int co = i % Cout, ci = (i / Cout) % Cin, t = i / (Cout * Cin);
    dst[i] = transposed ? w[((int64_t)ci * Cout + co) * ntap + t] : w[((int64_t)co * Cin + ci) * ntap + t];
}
int launch_guard_weights(const float *w, int transposed, int Cin, int Cout, int ntap, float *dst, hipStream_t st) {
    const int total = Cin * Cout * ntap;
    hipLaunchKernelGGL(guard_weights_kernel, dim3((total + 255) / 256), dim3(256), 0, st, w, transposed, Cin, Cout, ntap, dst);
    return check_launch("guard_weights_kernel");
}
// (weights) one word; n below 2^31
int launch_absmax_word(const float *x, int64_t n, unsigned *word, hipStream_t st) {
    hipLaunchKernelGGL(absmax_word_kernel, dim3(1), dim3(1024), 0, st, x, (int)n, word);
    return check_launch("absmax_word_kernel");
}
}  // namespace mvs

extern "C" int mvs_absmax_f32(const float *x, int64_t n, void *absmax, void *stream) {
    void *const absmax_bits = absmax;
    if (!x || !absmax_bits || n <= 0 || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(absmax) & 15)) {
        set_error("mvs_absmax_f32: needs 16-byte aligned device arrays: the data and the MVS_ABSMAX_WORDS-word block of the result");
        return MVS_EINVAL;
    }
    hipStream_t st = as_stream(stream);
    if (launch_zero_words(absmax_bits, kAbsmaxWords, st) != MVS_OK) return MVS_ELAUNCH;
    const int64_t blocks = (n / 4 + 255) / 256;
    const int nb = (int)(blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks));
    hipLaunchKernelGGL(absmax_kernel, dim3(nb), dim3(256), 0, st, x, n, static_cast<unsigned *>(absmax_bits));
    return check_launch("mvs_absmax_f32");
}

extern "C" int mvs_conv3d_pack_weights_f16x3_f32(const float *weight, int Cin, void *packed, void *stream) {
    if (!weight || !packed || !f16x3_shape_ok(Cin)) {
        set_error("mvs_conv3d_pack_weights_f16x3_f32: needs a (8, Cin, 3, 3, 3) weight with Cin in {8, 16, 32}");
        return MVS_EINVAL;
    }
    // the weights' largest magnitude goes through the last word of the trailer, which the pack kernel does not write
    unsigned *wmax = reinterpret_cast<unsigned *>(static_cast<unsigned char *>(packed) + (size_t)(Cin / 8) * kF16ChunkBytes + 12);
    const int rc = launch_absmax_word(weight, (int64_t)8 * Cin * 27, wmax, as_stream(stream));
    if (rc != MVS_OK) return rc;
    const int total = (Cin / 8) * 9 * 512;
    hipLaunchKernelGGL(pack_f16x3_kernel, dim3((total + 255) / 256), dim3(256), 0, as_stream(stream), weight, Cin,
                       static_cast<unsigned short *>(packed), total, wmax);
    const int rc2 = check_launch("mvs_conv3d_pack_weights_f16x3_f32");
    if (rc2 != MVS_OK) return rc2;
    return launch_guard_weights(weight, 0, Cin, 8, 27,
                                reinterpret_cast<float *>(static_cast<unsigned char *>(packed) + (size_t)(Cin / 8) * kF16ChunkBytes + 16), as_stream(stream));
}

extern "C" int mvs_conv3d_c8_f16x3_f32(const float *in, const void *in_absmax, const void *packed, const float *scale,
                                       const float *shift, const float *residual, int relu, int B, int Cin,
                                       int D, int H, int W, float *out, void *out_absmax, void *stream) {
    if (!in || !in_absmax || !packed || !out || B <= 0 || D <= 0 || H <= 0 || W <= 0 || !f16x3_shape_ok(Cin)) {
        set_error("mvs_conv3d_c8_f16x3_f32: invalid argument (Cin in {8, 16, 32}, Cout = 8, stride 1, 8-channel-blocked input, "
                  "in_absmax = the absmax block of the input)");
        return MVS_EINVAL;
    }
    if ((int64_t)9 * H * W * Cin * 4 >= 0xffffff00LL) return bare_error(MVS_EINVAL, __func__, __LINE__);
    ConvArgs a;
    a.in = in; a.wpk = static_cast<const float *>(packed); a.scale = scale; a.shift = shift; a.residual = residual;
    a.out = out;
    a.B = B; a.D = D; a.H = H; a.W = W;
    a.Do = D; a.Ho = H; a.Wo = W;
    a.tiles_x = (W + 31) / 32; a.tiles_y = (H + 3) / 4; a.tiles_z = (D + 3) / 4;
    a.relu = relu; a.in_c8 = 1; a.ystrip = 8; a.res_up2 = 0;
    a.run_flag = conv_run_flag();
    const int64_t ng = (int64_t)B * a.tiles_x * a.tiles_y * ((a.tiles_z + kFGroup - 1) / kFGroup);
    if (ng <= 0 || ng > 0x7fffffffLL) return bare_error(MVS_EINVAL, __func__, __LINE__);
    const int n_cu = device_cu_count();
    hipStream_t st = as_stream(stream);
    const dim3 grid((unsigned)(ng < n_cu ? ng : n_cu)), blk(kFThreads);
    const unsigned *mx = static_cast<const unsigned *>(in_absmax);
    unsigned *omx = static_cast<unsigned *>(out_absmax);
    unsigned long long *const gc = guard_counter();
#ifdef MVS_TUNING
    // eight-row tiles with the staging buffer kept (conv_f16x3_y8.hip; bit-identical results, no faster: profiles/r06_conv0_y8.json):
    // MVS_CONV0_Y8 = 2 (two barriers per step), 1 (one), 0 = this file's kernel
    static const int y8 = [] { const char *e = getenv("MVS_CONV0_Y8"); return e ? atoi(e) : 0; }();
    if (y8 == 1 || y8 == 2) return launch_conv3d_c8_f16x3_y8(a, Cin, y8, mx, omx, gc, st);
    // phase-stamp build: cycle counters written through `residual` (scripts/exp_conv0_f16.py)
    static const int abl = [] { const char *e = getenv("MVS_CONV_SPLIT_ABL"); return e ? atoi(e) : 0; }();
    if ((abl & 128) && Cin == 32) {
        if (!residual) return bare_error(MVS_EINVAL, __func__, __LINE__);
        hipLaunchKernelGGL((conv3d_c8_f16x3_zs_kernel<32, 128>), grid, blk, 0, st, a, (int)ng, mx, omx, gc);
        return check_launch("mvs_conv3d_c8_f16x3_f32");
    }
    if ((abl & 3) && Cin == 32) {   // wrong results by design: 1 = no split work, 2 = no MFMA phase, 3 = copies and barriers only
        if ((abl & 3) == 1) hipLaunchKernelGGL((conv3d_c8_f16x3_zs_kernel<32, 1>), grid, blk, 0, st, a, (int)ng, mx, omx, gc);
        else if ((abl & 3) == 2) hipLaunchKernelGGL((conv3d_c8_f16x3_zs_kernel<32, 2>), grid, blk, 0, st, a, (int)ng, mx, omx, gc);
        else hipLaunchKernelGGL((conv3d_c8_f16x3_zs_kernel<32, 3>), grid, blk, 0, st, a, (int)ng, mx, omx, gc);
        return check_launch("mvs_conv3d_c8_f16x3_f32");
    }
#endif
    if (Cin == 32) hipLaunchKernelGGL((conv3d_c8_f16x3_zs_kernel<32>), grid, blk, 0, st, a, (int)ng, mx, omx, gc);
    else if (Cin == 16) hipLaunchKernelGGL((conv3d_c8_f16x3_zs_kernel<16>), grid, blk, 0, st, a, (int)ng, mx, omx, gc);
    else hipLaunchKernelGGL((conv3d_c8_f16x3_zs_kernel<8>), grid, blk, 0, st, a, (int)ng, mx, omx, gc);
    return check_launch("mvs_conv3d_c8_f16x3_f32");
}
