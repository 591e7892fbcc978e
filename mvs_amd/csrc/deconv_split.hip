// Transposed 3x3x3 convolutions, stride 2, pad 1, output_padding 1 (CostRegNet conv7 / conv9 / conv11,
// mvsnet.py:76-81,89-91; the cascade's and CVP-MVSNet's up-path) on the BF16 matrix pipe at FP32 accuracy, as a
// persistent copy-wave kernel.  Arithmetic and error bound: conv_split_common.h / conv_bf16x6.hip; structure of a
// step (copy waves, split pass between two barriers, MFMA items): conv_split.hip.
//
// out[o] += in[i] w[k] with o = 2i - 1 + k: per dimension an even output o = 2j sees one tap (k = 1, i = j), an odd
// output o = 2j + 1 two (k = 2, i = j) and (k = 0, i = j + 1).  The 8 output parity classes are 8 small convolutions
// on the INPUT grid (27 class-taps in all, no structural zeros); a tile is (TZ, TY, 16) input voxels plus one halo
// voxel on the high side of each axis, a step holds 16 input channels (two 8-channel chunk planes), a K = 32 MFMA
// step = four (tap, chunk) slots of one class.
//   Cout = 8 (conv11): the two x parities share the MFMA -- rows = (px, 8 channels), slots = (tz, ty, dx, chunk);
//     4 classes (pz, py) of 1, 2, 2, 4 K-steps; a lane stores 16 bytes of voxel 2x + px.
//   Cout = 16 per launch (conv9; conv7 as two launches): 8 classes of 1, 1, 1, 2, 1, 2, 2, 4 K-steps.
// BatchNorm affine, ReLU and the skip add (AFTER the ReLU, mvsnet.py:89-91) in the epilogue.
#include "conv_split_common.h"
#include "conv_guard.h"

#include <cstdlib>

namespace mvs {

struct DeconvArgs {
    const float *in;          // [B, D, H, W, Cin]
    const unsigned char *wpk; // [step][K-step][part][lane][8 bf16]
    const float *scale, *shift, *residual;
    float *out;               // [B, 2D, 2H, 2W, ldc]; this launch writes channels [0, 8 or 16)
    int B, D, H, W, ldc;
    int tiles_x, tiles_y, tiles_z, ystrip;
    int relu;
    // two-piece fp16 form (NP = 2): the input's absmax block (mvs_common.h), what undoes the weights' scale (device float)
    const unsigned *in_absmax;
    const float *w_iscale;
    unsigned *out_absmax;     // NULL, or the absmax block that collects the largest magnitude this launch stores (any form)
    // NP = 2, the range guard (conv_guard.h): the original fp32 weights (behind the trailer of the packed weights), the first
    // output channel of this launch, the device counter of launches that fell back
    const float *w_f32;
    int co0;
    unsigned long long *guard_cnt;
    const unsigned *run_flag;   // NULL, or a device word: the launch returns at once when it is 0 (mvs_common.h: conv_run_flag)
};

// per-dimension tap t of parity p -> (kernel index, input offset)
__host__ __device__ constexpr int dsp_k(int p, int t) { return p == 0 ? 1 : (t == 0 ? 2 : 0); }
__host__ __device__ constexpr int dsp_d(int p, int t) { return (p == 1 && t == 1) ? 1 : 0; }

// NP: operand pieces -- 3 = bf16 hi/mid/lo, six products; 2 = scaled fp16 hi/lo, three products (conv_f16x3.hip)
template <int CIN_, bool PXM_, int NP_ = 3>
struct DeconvSplitCfg {
    static constexpr int CIN = CIN_, NP = NP_;
    static constexpr bool PXM = PXM_;
    static constexpr int COUT = PXM ? 8 : 16, NCLS = PXM ? 4 : 8, CPS = 2, NSTEP = CIN / 16;
    static constexpr int TZ = PXM ? 4 : 2, TY = 4, ZT = TZ + 1, YT = TY + 1, XP = 17, NVOX = ZT * YT * XP;
    static constexpr int NVP = (NVOX + 15) / 16 * 16;
    static constexpr int RB = TZ * TY, RPW = RB / 8;
    static constexpr int NPIECE = 2 * NVP * CPS, NCOPY = (NPIECE + 63) / 64;
    static constexpr int FBYTES = NCOPY * 1024, SPART = NVP * 16 * CPS, SBYTES = NP * SPART;
    // class c = (pz, py [, px]); taps (tz, ty, tx) with tx fastest; slots = taps x chunks, 4 slots per K-step
    static constexpr int pz(int c) { return PXM ? c >> 1 : c >> 2; }
    static constexpr int py(int c) { return PXM ? c & 1 : (c >> 1) & 1; }
    static constexpr int px(int c) { return PXM ? 1 : c & 1; }          // (PXM: both dx taps are always there)
    static constexpr int nx(int c) { return PXM ? 2 : 1 + px(c); }
    static constexpr int ntap(int c) { return (1 + pz(c)) * (1 + py(c)) * nx(c); }
    static constexpr int nk(int c) { return (ntap(c) * CPS + 3) / 4; }
    static constexpr int kbase(int c) { int s = 0; for (int i = 0; i < c; ++i) s += nk(i); return s; }
    static constexpr int NK = kbase(NCLS);
    static constexpr int cls_of(int g) { int c = 0; while (g >= kbase(c + 1)) ++c; return c; }
    static constexpr int WBYTES = NK * NP * 1024;
    static constexpr int F_OFF = 2 * WBYTES, S_OFF = F_OFF + FBYTES, AFF_OFF = S_OFF + SBYTES, LDS_BYTES = AFF_OFF + 2 * COUT * 4;
    static_assert(RB % 8 == 0 && LDS_BYTES <= 160 * 1024 && SPART * 2 + 16 * 1024 < 65536, "tile / LDS budget");
};

constexpr int kDeconvCopyWaves = 4, kDeconvThreads = 512 + 64 * kDeconvCopyWaves;

template <class C>
__global__ __launch_bounds__(kDeconvThreads) void deconv_split_kernel(DeconvArgs a, int ntiles) {
    constexpr int CIN = C::CIN, NSTEP = C::NSTEP, NK = C::NK, NCLS = C::NCLS, RPW = C::RPW;
    constexpr int YT = C::YT, XP = C::XP, NVOX = C::NVOX, NPIECE = C::NPIECE, NCOPY = C::NCOPY;
    constexpr int SPART = C::SPART, WBYTES = C::WBYTES, F_OFF = C::F_OFF, S_OFF = C::S_OFF;
    constexpr bool PXM = C::PXM;
    constexpr int NC = kDeconvCopyWaves, IPW = (NCOPY + NC - 1) / NC, NT = kDeconvThreads;
    constexpr int NWC = (WBYTES / 1024 + NC - 1) / NC;
    __shared__ __attribute__((aligned(16))) unsigned char lds[C::LDS_BYTES];

    if (a.run_flag && *a.run_flag == 0u) return;      // (uniform: the fused tail kernel in front of this launch did the work)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4;
    const unsigned lds_base = (unsigned)(uintptr_t)lds;
    const bool copier = wv >= 8;
    const int cw = wv - 8;
    constexpr int NP = C::NP;
    float sx = 1.0f, unscale = 1.0f, unscale2 = 1.0f;     // NP = 2: operand scale of the input; what undoes it and the weights' scale
    if constexpr (NP == 2) {
        const AbsmaxVerdict verdict = absmax_verdict(a.in_absmax);
        const int xe = absmax_exponent(verdict.bits);
        const float isw = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, *a.w_iscale)));
        sx = pow2f(14 - xe);
        // what undoes both operand scales is 2^E: the part within +-60 rides in the per-channel scale of the epilogue, the rest
        // (inputs or weights beyond 1e+-18: folded into one float it would underflow) is a second multiply there
        const int E = xe - 14 + (int)((__builtin_bit_cast(unsigned, isw) >> 23) & 255u) - 127;
        const int e1 = E < -60 ? -60 : (E > 60 ? 60 : E), e2 = E - e1 < -126 ? -126 : (E - e1 > 127 ? 127 : E - e1);
        unscale = pow2f(e1);
        unscale2 = pow2f(e2);
        // the range guard (conv_guard.h): a non-finite or outlier-dominated input, or non-finite weights -> plain fp32
        if (verdict.code != 0 || isw != isw) {
            GuardConv g;
            g.in = a.in; g.w = a.w_f32; g.scale = a.scale; g.shift = a.shift; g.residual = a.residual; g.out = a.out;
            g.out_absmax = a.out_absmax; g.counter = a.guard_cnt;
            g.B = a.B; g.D = a.D; g.H = a.H; g.W = a.W; g.Cin = CIN; g.Do = 2 * a.D; g.Ho = 2 * a.H; g.Wo = 2 * a.W;
            g.ldc = a.ldc; g.co0 = a.co0; g.nco = C::COUT; g.kd = 3; g.kh = 3; g.stride = 2; g.transposed = 1;
            g.relu = a.relu ? 1 : 0; g.in_c8 = 0; g.out_c4 = 0;
            guard_direct_conv(g);
            return;
        }
    }
    if (tid < 2 * C::COUT) {
        const int c = tid % C::COUT;
        const float v = tid < C::COUT ? (a.scale ? a.scale[c] : 1.0f) * unscale : (a.shift ? a.shift[c] : 0.0f);
        *reinterpret_cast<float *>(lds + C::AFF_OFF + tid * 4) = v;
    }

    int t0, t_step, ntw;
    {
        const int nb = gridDim.x;
        if ((nb & 7) == 0) {
            const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, per = nb >> 3;
            const int lo = (int)((int64_t)ntiles * xcd / 8), hi = (int)((int64_t)ntiles * (xcd + 1) / 8);
            t0 = lo + j; t_step = per; ntw = (hi - t0 + per - 1) / per;
        } else {
            t0 = blockIdx.x; t_step = nb; ntw = (ntiles - t0 + nb - 1) / nb;
        }
        if (ntw < 0) ntw = 0;
    }
    // ordered-tile decode (y inside strips of tile rows, then z, then x) through float reciprocals + one correction
    // step: exact for tile indices below 2^23 (the launcher guarantees it)
    const int per_b = a.tiles_x * a.tiles_y * a.tiles_z, full = a.ystrip * a.tiles_z * a.tiles_x;
    const float r_per_b = 1.0f / (float)per_b, r_full = 1.0f / (float)full, r_tz = 1.0f / (float)a.tiles_z;
    const int hs_last = a.tiles_y % a.ystrip ? a.tiles_y % a.ystrip : a.ystrip;
    const float r_hs = 1.0f / (float)a.ystrip, r_hs_last = 1.0f / (float)hs_last;
    auto divmod = [](int nn, int d, float rd, int &q, int &r) {
        q = (int)((float)nn * rd);
        r = nn - q * d;
        if (r < 0) { q -= 1; r += d; } else if (r >= d) { q += 1; r -= d; }
    };
    auto decode = [&](int t) {
        TileIdx ti;
        int rem, strip, rem2, tyl, rem3;
        divmod(t, per_b, r_per_b, ti.b, rem);
        divmod(rem, full, r_full, strip, rem2);
        const int ys = strip * a.ystrip;
        const bool last = ys + a.ystrip > a.tiles_y;
        divmod(rem2, last ? hs_last : a.ystrip, last ? r_hs_last : r_hs, rem3, tyl);
        divmod(rem3, a.tiles_z, r_tz, ti.tx, ti.tz);
        ti.ty = ys + tyl;
        return ti;
    };

    // split pass (all waves): piece P = ps*NT + tid -> 16 bytes of the fp32 buffer -> 8 bytes of each bf16 part
    constexpr int NPS = (NPIECE + NT - 1) / NT;
    auto split_pass = [&]() {
        f32x4 x[NPS];
        const unsigned fp = lds_base + (unsigned)(F_OFF + tid * 16), sp = lds_base + (unsigned)(S_OFF + tid * 8);
        static_for<0, NPS>([&](auto pc) {
            constexpr int ps = decltype(pc)::value;
            x[ps] = lds_read_b128<ps * NT * 16>(fp);
        });
        lds_wait_n<0>();
        static_for<0, (NPS + 1) / 2>([&](auto pc) {
            constexpr int p0 = 2 * decltype(pc)::value, p1 = (p0 + 1 < NPS) ? p0 + 1 : p0;
            f32x4 &x0 = x[p0], &x1 = x[p1];
            asm volatile("" : "+v"(x0), "+v"(x1));
            u32x4 hu, mu, lu;
            if constexpr (NP == 2) {
                split2_block(x0, x1, sx, hu, mu);
            } else {
                bf16x8 h, m, l;
                split3_block(x0, x1, h, m, l);
                hu = __builtin_bit_cast(u32x4, h); mu = __builtin_bit_cast(u32x4, m); lu = __builtin_bit_cast(u32x4, l);
            }
            if (p0 * NT + tid < NPIECE) {
                lds_write_b64<p0 * NT * 8>(sp, hu[0], hu[1]);
                lds_write_b64<p0 * NT * 8 + SPART>(sp, mu[0], mu[1]);
                if constexpr (NP == 3) lds_write_b64<p0 * NT * 8 + 2 * SPART>(sp, lu[0], lu[1]);
            }
            if (p1 != p0 && p1 * NT + tid < NPIECE) {
                lds_write_b64<p1 * NT * 8>(sp, hu[2], hu[3]);
                lds_write_b64<p1 * NT * 8 + SPART>(sp, mu[2], mu[3]);
                if constexpr (NP == 3) lds_write_b64<p1 * NT * 8 + 2 * SPART>(sp, lu[2], lu[3]);
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
            const bool ok = P < NPIECE;
            const int vraw = (P % (2 * C::NVP)) >> 1, c = ok ? P / (2 * C::NVP) : 0;
            const bool okv = ok && vraw < NVOX;
            const int v = okv ? vraw : 0;
            loc[i] = (v % XP) | (((v / XP) % YT) << 8) | ((v / (XP * YT)) << 16) | ((P & 1) << 24) | (c << 25) |
                     (okv ? 0 : (int)0x80000000);
        }
        const int64_t plane_in = (int64_t)a.H * a.W * CIN;
        const int row_in = a.W * CIN;
        const unsigned window_bytes = (unsigned)min((int64_t)C::ZT * plane_in * 4, (int64_t)0xffffff00u);
        unsigned voff[IPW];
        mvs_srd_t srd = make_srd(a.in, 0);
        auto geometry = [&](int t) {
            const TileIdx tile = decode(t);
            const int ix0 = tile.tx * 16, iy0 = tile.ty * C::TY, iz0 = tile.tz * C::TZ;
            srd = make_srd(a.in + ((int64_t)tile.b * a.D + iz0) * plane_in, window_bytes);
#pragma unroll
            for (int i = 0; i < IPW; ++i) {
                const int gx = ix0 + (loc[i] & 255), gy = iy0 + ((loc[i] >> 8) & 255);
                const int lz = (loc[i] >> 16) & 255, h = (loc[i] >> 24) & 1, c = (loc[i] >> 25) & 3;
                const bool ok = loc[i] >= 0 && gx < a.W && gy < a.H && iz0 + lz < a.D;
                voff[i] = ok ? (unsigned)(((int64_t)lz * plane_in + (int64_t)gy * row_in + gx * CIN + c * 8 + h * 4) * 4) : 0xffffff00u;
            }
        };
        auto issue_halo = [&](int st) {
            const unsigned soff = (unsigned)(st * 64);
#pragma unroll
            for (int i = 0; i < IPW; ++i) {
                if (i * NC + cw >= NCOPY) continue;   // wave-uniform
                glds16_buf(voff[i], srd, soff, lds_base + (unsigned)(F_OFF + (i * NC + cw) * 1024));
            }
        };
        auto issue_weights = [&](int st, int sel) {
#pragma unroll
            for (int i = 0; i < NWC; ++i) {
                const int g = i * NC + cw;
                if (g < WBYTES / 1024) glds16(a.wpk + (size_t)st * WBYTES + (size_t)g * 1024 + lane * 16,
                                              lds_base + (unsigned)(sel * WBYTES + g * 1024));
            }
        };
        int wsel = 0;
        if (ntw > 0) {
            geometry(t0);
            issue_halo(0);
            issue_weights(0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        for (int k = 0; k < ntw; ++k) {
#pragma unroll 1
            for (int st = 0; st < NSTEP; ++st) {
                __syncthreads();           // the halo of this step is in the fp32 buffer (this wave has waited for it)
                split_pass();
                __syncthreads();           // ... and has been split: the fp32 buffer is free
                if (st + 1 < NSTEP) {
                    issue_halo(st + 1);
                    issue_weights(st + 1, wsel ^ 1);
                } else if (k + 1 < ntw) {
                    geometry(t0 + (k + 1) * t_step);
                    issue_halo(0);
                    if (NSTEP > 1) issue_weights(0, wsel ^ 1);      // (one step per tile: the weights never change)
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (NSTEP > 1) wsel ^= 1;
            }
        }
        return;
    }

    // ================================================================ multiplying waves
    // row block rb = wv*RPW + r -> input row (z, y); this lane's B voxel of tap (dz, dy, dx): (z + dz, y + dy, n + dx)
    unsigned rbo[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int rb = wv * RPW + r;
        rbo[r] = (unsigned)((((rb / C::TY) * YT + rb % C::TY) * XP + n) * 16);
    }
    // slot 4 ks + kq of K-step g (of class c): (tap, chunk); slots past the class's taps read voxel 0 against zero weights
    unsigned tapo[NK];
    static_for<0, NK>([&](auto gc) {
        constexpr int g = decltype(gc)::value, c = C::cls_of(g), ks = g - C::kbase(c);
        const int sl = 4 * ks + kq, ti = sl / C::CPS, cc = sl % C::CPS;
        const int tx = ti % C::nx(c), ty = (ti / C::nx(c)) % (1 + C::py(c)), tz = ti / (C::nx(c) * (1 + C::py(c)));
        const int dz = C::pz(c) == 1 && tz == 1 ? 1 : 0, dy = C::py(c) == 1 && ty == 1 ? 1 : 0;
        const int dx = PXM ? tx : (C::px(c) == 1 && tx == 1 ? 1 : 0);
        tapo[g] = sl < C::ntap(c) * C::CPS ? (unsigned)((((dz * YT + dy) * XP + dx) + cc * C::NVP) * 16) : 0u;
    });

    f32x4 acc[NCLS][RPW];
#pragma unroll
    for (int c = 0; c < NCLS; ++c)
#pragma unroll
        for (int r = 0; r < RPW; ++r) acc[c][r] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // skip-connection values of every output this lane will write: requested at the top of a tile's last step, consumed
    // in the epilogue -- their HBM latency (the 727 MB of c0 stream from memory) hides under the split pass and the MFMAs
    // instead of standing between the last MFMA and the first store
    float4 res[NCLS][RPW];
    const int c0 = PXM ? (kq & 1) * 4 : kq * 4;
    const int Do = 2 * a.D, Ho = 2 * a.H, Wo = 2 * a.W;
    auto out_offset = [&](const TileIdx &cur, int c, int r, bool &inside) {
        const int rb = wv * RPW + r;
        const int jz = cur.tz * C::TZ + rb / C::TY, jy = cur.ty * C::TY + rb % C::TY, jx = cur.tx * 16 + n;
        inside = jz < a.D && jy < a.H && jx < a.W;
        const int oz = 2 * jz + C::pz(c), oy = 2 * jy + C::py(c), ox = 2 * jx + (PXM ? (kq >> 1) : C::px(c));
        return ((((int64_t)cur.b * Do + oz) * Ho + oy) * Wo + ox) * a.ldc + c0;
    };
    int wsel = 0;
    float vmax = 0.0f;       // largest magnitude this lane has stored (-> a.out_absmax)
    for (int k = 0; k < ntw; ++k) {
        const TileIdx cur = decode(t0 + k * t_step);
#pragma unroll 1
        for (int st = 0; st < NSTEP; ++st) {
            if (st == NSTEP - 1 && a.residual) {
#pragma unroll
                for (int c = 0; c < NCLS; ++c)
#pragma unroll
                    for (int r = 0; r < RPW; ++r) {
                        bool inside;
                        const int64_t o = out_offset(cur, c, r, inside);
                        res[c][r] = inside ? *reinterpret_cast<const float4 *>(a.residual + o) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
            }
            __syncthreads();
            split_pass();
            __syncthreads();
            // ---- MFMA phase: items (K-step g, row block r); the reads of the next item go out before the MFMAs of this one
            const unsigned aA = lds_base + (unsigned)(wsel * WBYTES + lane * 16);
            const unsigned aS = lds_base + (unsigned)S_OFF;
            bf16x8 A[2][NP], Bf[2][NP];      // (NP = 2: the same registers hold fp16 pairs)
            auto read_a = [&](auto gc) {
                constexpr int g = decltype(gc)::value;
                static_for<0, NP>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    A[g & 1][i] = __builtin_bit_cast(bf16x8, lds_read_b128<(g * NP + i) * 1024>(aA));
                });
            };
            auto read_b = [&](auto ic) {
                constexpr int it = decltype(ic)::value, g = it / RPW, r = it % RPW;
                const unsigned ad = aS + rbo[r] + tapo[g];
                static_for<0, NP>([&](auto pc) {
                    constexpr int sp = decltype(pc)::value;
                    Bf[it & 1][sp] = __builtin_bit_cast(bf16x8, lds_read_b128<sp * SPART>(ad));
                });
            };
            read_a(std::integral_constant<int, 0>{});
            read_b(std::integral_constant<int, 0>{});
            static_for<0, NK * RPW>([&](auto ic) {
                constexpr int it = decltype(ic)::value, g = it / RPW, r = it % RPW, c = C::cls_of(g);
                lds_wait_n<0>();
                static_for<0, NP>([&](auto pc) {
                    bf16x8 &b0 = Bf[it & 1][decltype(pc)::value];
                    asm volatile("" : "+v"(b0));
                });
                if constexpr (r == 0) {
                    static_for<0, NP>([&](auto pc) {
                        bf16x8 &a0 = A[g & 1][decltype(pc)::value];
                        asm volatile("" : "+v"(a0));
                    });
                }
                if constexpr (it + 1 < NK * RPW) {
                    if constexpr (r == RPW - 1) read_a(std::integral_constant<int, g + 1>{});
                    read_b(std::integral_constant<int, it + 1>{});
                }
                __builtin_amdgcn_sched_barrier(0);   // the reads go out BEFORE this item's MFMAs
                f32x4 &cc = acc[c][r];
                if constexpr (NP == 2) {
                    // three partial products, small terms first: al bh, ah bl, ah bh
                    const f16x8 ah = __builtin_bit_cast(f16x8, A[g & 1][0]), al = __builtin_bit_cast(f16x8, A[g & 1][1]);
                    const f16x8 fh = __builtin_bit_cast(f16x8, Bf[it & 1][0]), fl = __builtin_bit_cast(f16x8, Bf[it & 1][1]);
                    cc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, fh, cc, 0, 0, 0);
                    cc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, fl, cc, 0, 0, 0);
                    cc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, fh, cc, 0, 0, 0);
                } else {
                const bf16x8 bh = Bf[it & 1][0], bm = Bf[it & 1][1], bl = Bf[it & 1][NP - 1];
                // six partial products, small terms first
                cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[g & 1][1], bm, cc, 0, 0, 0);
                cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[g & 1][NP - 1], bh, cc, 0, 0, 0);
                cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[g & 1][0], bl, cc, 0, 0, 0);
                cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[g & 1][1], bh, cc, 0, 0, 0);
                cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[g & 1][0], bm, cc, 0, 0, 0);
                cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[g & 1][0], bh, cc, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            if (NSTEP > 1) wsel ^= 1;
        }
        // ---- epilogue of the tile: affine, ReLU, skip add, one 16-byte store per lane, class and row block
        {
            const float4 sc = *reinterpret_cast<const float4 *>(lds + C::AFF_OFF + c0 * 4);
            const float4 sh = *reinterpret_cast<const float4 *>(lds + C::AFF_OFF + (C::COUT + c0) * 4);
#pragma unroll
            for (int c = 0; c < NCLS; ++c) {
#pragma unroll
                for (int r = 0; r < RPW; ++r) {
                    f32x4 v = acc[c][r];
                    acc[c][r] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    bool inside;
                    const int64_t o = out_offset(cur, c, r, inside);
                    if (!inside) continue;
                    if (unscale2 != 1.0f) { v[0] *= unscale2; v[1] *= unscale2; v[2] *= unscale2; v[3] *= unscale2; }   // (wave-uniform; extreme magnitudes only)
                    v[0] = v[0] * sc.x + sh.x; v[1] = v[1] * sc.y + sh.y;
                    v[2] = v[2] * sc.z + sh.z; v[3] = v[3] * sc.w + sh.w;
                    if (a.relu) {
                        v[0] = relu_nan(v[0]); v[1] = relu_nan(v[1]);
                        v[2] = relu_nan(v[2]); v[3] = relu_nan(v[3]);
                    }
                    if (a.residual) {
                        v[0] += res[c][r].x; v[1] += res[c][r].y; v[2] += res[c][r].z; v[3] += res[c][r].w;
                    }
                    *reinterpret_cast<float4 *>(a.out + o) = make_float4(v[0], v[1], v[2], v[3]);
                    vmax = amax4_nan(vmax, v[0], v[1], v[2], v[3]);
                }
            }
        }
    }
    publish_absmax(a.out_absmax, vmax);
}

// PyTorch ConvTranspose3d weight (Cin, Cout_total, 3, 3, 3), output channels [co0, co0 + 8 or 16) ->
// [step][K-step g][part][lane][8 bf16]; lane (mrow, kq): slot 4 ks + kq of g's class = (tap (tz, ty, tx), chunk cc),
// input channel (step*2 + cc)*8 + j; row mrow = output channel (pxm: (px', channel))
// F16: [..][hi,lo][lane][8 fp16] of w * 2^(14 - exponent(*wmax)); thread 0 writes what undoes the scale into *iscale
template <bool PXM, bool F16 = false>
__global__ __launch_bounds__(256) void pack_deconv_split_kernel(const float *__restrict__ w, int Cin, int Cout, int co0,
                                                                unsigned short *__restrict__ out, int total,
                                                                const unsigned *__restrict__ wmax = nullptr, float *__restrict__ iscale = nullptr) {
    using C = DeconvSplitCfg<16, PXM>;      // (the class tables do not depend on Cin)
    const int i = blockIdx.x * 256 + threadIdx.x;
    int we = 0;
    if constexpr (F16) {
        we = absmax_exponent(*wmax);
        // what undoes the scale; NaN = "the weights are not finite", which sends every launch to the guard's fp32 path
        if (i == 0) *iscale = *wmax >= 0x7f800000u ? __builtin_nanf("") : pow2f(we - 14);
    }
    if (i >= total) return;
    const int j = i & 7, lane = (i >> 3) & 63;
    const int g = (i >> 9) % C::NK, st = (i >> 9) / C::NK;
    int c = 0;
    while (g >= C::kbase(c + 1)) ++c;
    const int ks = g - C::kbase(c);
    const int mrow = lane & 15, kq = lane >> 4, sl = 4 * ks + kq, ti = sl / 2, cc = sl % 2;
    float x = 0.0f;
    if (sl < C::ntap(c) * 2) {
        const int tx = ti % C::nx(c), ty = (ti / C::nx(c)) % (1 + C::py(c)), tz = ti / (C::nx(c) * (1 + C::py(c)));
        const int kz = dsp_k(C::pz(c), tz), ky = dsp_k(C::py(c), ty);
        int kx, co;
        bool live = true;
        if (PXM) {
            const int pxr = mrow >> 3;                  // this row's x parity
            co = co0 + (mrow & 7);
            if (pxr == 0) { live = tx == 0; kx = 1; } else { kx = tx == 0 ? 2 : 0; }
        } else {
            co = co0 + mrow;
            kx = dsp_k(C::px(c), tx);
        }
        const int ci = (st * 2 + cc) * 8 + j;
        if (live) x = w[(((int64_t)ci * Cout + co) * 3 + kz) * 9 + ky * 3 + kx];
    }
    if constexpr (F16) {
        x *= pow2f(14 - we);
        const _Float16 h = (_Float16)x;
        const _Float16 l = (_Float16)(x - (float)h);
        unsigned short *o = out + ((size_t)(st * C::NK + g) * 2) * 512 + lane * 8 + j;
        o[0] = __builtin_bit_cast(unsigned short, h);
        o[512] = __builtin_bit_cast(unsigned short, l);
        return;
    }
    const __bf16 h = (__bf16)x;
    const float r1 = x - (float)h;
    const __bf16 mm = (__bf16)r1;
    const float r2 = r1 - (float)mm;
    const __bf16 l = (__bf16)r2;
    unsigned short *o = out + ((size_t)(st * C::NK + g) * 3) * 512 + lane * 8 + j;
    o[0] = __builtin_bit_cast(unsigned short, h);
    o[512] = __builtin_bit_cast(unsigned short, mm);
    o[1024] = __builtin_bit_cast(unsigned short, l);
}

template <class C>
static int launch_deconv_split(const DeconvArgs &a0, hipStream_t st) {
    DeconvArgs a = a0;
    a.tiles_x = (a.W + 15) / 16; a.tiles_y = (a.H + C::TY - 1) / C::TY; a.tiles_z = (a.D + C::TZ - 1) / C::TZ;
    a.ystrip = 4;
    const int64_t nt = (int64_t)a.B * a.tiles_x * a.tiles_y * a.tiles_z;
    if (nt <= 0 || nt >= (1 << 23)) return bare_error(MVS_EUNSUPPORTED, __func__, __LINE__);   // the kernel's float tile decode: callers fall back to mvs_conv3d_f32 / mvs_conv2d_f32
    const int n_cu = device_cu_count();
    hipLaunchKernelGGL((deconv_split_kernel<C>), dim3((unsigned)(nt < n_cu ? nt : n_cu)), dim3(kDeconvThreads), 0, st, a, (int)nt);
    return check_launch("mvs_deconv_split_f32");
}

}  // namespace mvs

using namespace mvs;

extern "C" int mvs_deconv_split_supported(int Cin, int Cout) {
    return (Cin == 16 || Cin == 32 || Cin == 64) && (Cout == 8 || Cout == 16 || Cout == 32);
}

static int deconv_nk(int Cout) { return Cout == 8 ? DeconvSplitCfg<16, true>::NK : DeconvSplitCfg<16, false>::NK; }

static size_t deconv_packed_bytes(int Cin, int Cout, int np) {
    if (!mvs_deconv_split_supported(Cin, Cout)) return 0;
    const int launches = Cout == 8 ? 1 : Cout / 16;
    return (size_t)launches * (Cin / 16) * deconv_nk(Cout) * np * 1024;
}

extern "C" size_t mvs_deconv_split_packed_bytes(int Cin, int Cout) { return deconv_packed_bytes(Cin, Cout, 3); }

// two-piece fp16 form: the fragments + a 16-byte trailer (what undoes the weights' scale; the weights' largest magnitude)
// + the original fp32 weights (the range guard's, conv_guard.h)
extern "C" size_t mvs_deconv_split_f16_packed_bytes(int Cin, int Cout) {
    const size_t n = deconv_packed_bytes(Cin, Cout, 2);
    return n ? n + 16 + (size_t)Cin * Cout * 27 * 4 : 0;
}

namespace mvs { int launch_absmax_word(const float *x, int64_t n, unsigned *word, hipStream_t st); }   // conv_f16x3.hip

extern "C" int mvs_deconv_split_pack_weights_f16_f32(const float *weight, int Cin, int Cout, void *packed, void *stream) {
    if (!weight || !packed || !mvs_deconv_split_supported(Cin, Cout)) {
        set_error("mvs_deconv_split_pack_weights_f16_f32: needs a (Cin, Cout, 3, 3, 3) weight with Cin in {16, 32, 64}, Cout in {8, 16, 32}");
        return MVS_EINVAL;
    }
    const int nk = deconv_nk(Cout), step = Cout == 8 ? 8 : 16;
    const size_t per_launch = (size_t)(Cin / 16) * nk * 2 * 1024, body = deconv_packed_bytes(Cin, Cout, 2);
    unsigned char *pk = static_cast<unsigned char *>(packed);
    unsigned *wmax = reinterpret_cast<unsigned *>(pk + body + 4);
    float *iscale = reinterpret_cast<float *>(pk + body);
    const int rc = launch_absmax_word(weight, (int64_t)Cin * Cout * 27, wmax, as_stream(stream));
    if (rc != MVS_OK) return rc;
    const int total = (Cin / 16) * nk * 512;
    for (int co0 = 0; co0 < Cout; co0 += step) {
        unsigned short *dst = reinterpret_cast<unsigned short *>(pk + (co0 / step) * per_launch);
        if (Cout == 8)
            hipLaunchKernelGGL((pack_deconv_split_kernel<true, true>), dim3((total + 255) / 256), dim3(256), 0, as_stream(stream), weight, Cin, Cout, co0, dst, total, wmax, iscale);
        else
            hipLaunchKernelGGL((pack_deconv_split_kernel<false, true>), dim3((total + 255) / 256), dim3(256), 0, as_stream(stream), weight, Cin, Cout, co0, dst, total, wmax, iscale);
    }
    if (launch_guard_weights(weight, 1, Cin, Cout, 27, reinterpret_cast<float *>(pk + body + 16), as_stream(stream)) != MVS_OK)
        return bare_error(MVS_ELAUNCH, __func__, __LINE__);
    return check_launch("mvs_deconv_split_pack_weights_f16_f32");
}

extern "C" int mvs_deconv_split_pack_weights_f32(const float *weight, int Cin, int Cout, void *packed, void *stream) {
    if (!weight || !packed || !mvs_deconv_split_supported(Cin, Cout)) {
        set_error("mvs_deconv_split_pack_weights_f32: needs a (Cin, Cout, 3, 3, 3) weight with Cin in {16, 32, 64}, Cout in {8, 16, 32}");
        return MVS_EINVAL;
    }
    const int nk = deconv_nk(Cout), step = Cout == 8 ? 8 : 16;
    const size_t per_launch = (size_t)(Cin / 16) * nk * 3 * 1024;
    const int total = (Cin / 16) * nk * 512;
    for (int co0 = 0; co0 < Cout; co0 += step) {
        unsigned short *dst = reinterpret_cast<unsigned short *>(static_cast<unsigned char *>(packed) + (co0 / step) * per_launch);
        if (Cout == 8)
            hipLaunchKernelGGL((pack_deconv_split_kernel<true, false>), dim3((total + 255) / 256), dim3(256), 0, as_stream(stream), weight, Cin, Cout, co0, dst, total, nullptr, nullptr);
        else
            hipLaunchKernelGGL((pack_deconv_split_kernel<false, false>), dim3((total + 255) / 256), dim3(256), 0, as_stream(stream), weight, Cin, Cout, co0, dst, total, nullptr, nullptr);
    }
    return check_launch("mvs_deconv_split_pack_weights_f32");
}

// np = 3: the bf16 form; np = 2: the fp16 form (in_absmax required); out_absmax: NULL, or the absmax block the largest magnitude
// of `out` is max-ed INTO (the caller clears it)
static int deconv_split_impl(const float *in, const void *in_absmax, const void *packed, const float *scale, const float *shift,
                             const float *residual, int relu, int B, int Cin, int Cout, int D, int H, int W,
                             float *out, void *out_absmax, int np, void *stream) {
    if (!in || (np == 2 && !in_absmax) || !packed || !out || B <= 0 || D <= 0 || H <= 0 || W <= 0 || !mvs_deconv_split_supported(Cin, Cout)) {
        set_error("mvs_deconv_split_f32: invalid argument (Cin in {16, 32, 64}; Cout in {8, 16, 32}; stride 2; channels-last)");
        return MVS_EINVAL;
    }
    if ((int64_t)5 * H * W * Cin * 4 >= 0xffffff00LL) return bare_error(MVS_EUNSUPPORTED, __func__, __LINE__);   // 32-bit halo offsets: callers fall back to the fp32 kernels
    const int nk = deconv_nk(Cout), step = Cout == 8 ? 8 : 16;
    const size_t per_launch = (size_t)(Cin / 16) * nk * np * 1024;
    hipStream_t st = as_stream(stream);
    for (int co0 = 0; co0 < Cout; co0 += step) {
        DeconvArgs a;
        a.in_absmax = static_cast<const unsigned *>(in_absmax);
        a.w_iscale = reinterpret_cast<const float *>(static_cast<const unsigned char *>(packed) + deconv_packed_bytes(Cin, Cout, 2));
        a.out_absmax = static_cast<unsigned *>(out_absmax);
        a.w_f32 = a.w_iscale + 4; a.co0 = co0;
        a.guard_cnt = np == 2 ? guard_counter() : nullptr;
        a.run_flag = conv_run_flag();
        a.in = in; a.wpk = static_cast<const unsigned char *>(packed) + (co0 / step) * per_launch;
        a.scale = scale ? scale + co0 : nullptr; a.shift = shift ? shift + co0 : nullptr;
        a.residual = residual ? residual + co0 : nullptr; a.out = out + co0;
        a.B = B; a.D = D; a.H = H; a.W = W; a.ldc = Cout; a.relu = relu;
        int rc = MVS_EUNSUPPORTED;
#define MVS_DECONV_CASE(CI, PX) if (Cin == CI && (Cout == 8) == PX) \
        rc = np == 2 ? launch_deconv_split<DeconvSplitCfg<CI, PX, 2>>(a, st) : launch_deconv_split<DeconvSplitCfg<CI, PX>>(a, st);
        MVS_DECONV_CASE(16, true) MVS_DECONV_CASE(32, true) MVS_DECONV_CASE(64, true)
        MVS_DECONV_CASE(16, false) MVS_DECONV_CASE(32, false) MVS_DECONV_CASE(64, false)
#undef MVS_DECONV_CASE
        if (rc != MVS_OK) return rc;
    }
    return MVS_OK;
}

extern "C" int mvs_deconv_split_f32(const float *in, const void *packed, const float *scale, const float *shift,
                                    const float *residual, int relu, int B, int Cin, int Cout, int D, int H, int W,
                                    float *out, void *stream) {
    return deconv_split_impl(in, nullptr, packed, scale, shift, residual, relu, B, Cin, Cout, D, H, W, out, nullptr, 3, stream);
}

extern "C" int mvs_deconv_split_f16_f32(const float *in, const void *in_absmax, const void *packed, const float *scale,
                                        const float *shift, const float *residual, int relu, int B, int Cin, int Cout, int D,
                                        int H, int W, float *out, void *out_absmax, void *stream) {
    return deconv_split_impl(in, in_absmax, packed, scale, shift, residual, relu, B, Cin, Cout, D, H, W, out, out_absmax, 2, stream);
}
