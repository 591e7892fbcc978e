// conv0 (3x3x3, Cout = 8, stride 1; mvsnet.py:66, module.py:26-33) on a volume that ARRIVES as fp16 pairs: the producer (the
// plane-sweep kernel, or mvs_c8_to_c8h_f32) has already scaled every fp32 value by the power of two of an absmax block and
// split it into hi + lo (conv_f16x3.hip has the arithmetic and its error bound).  MVS_LAYOUT_C8H:
//     [B, D, H, C/8, part (hi, lo), parity (x & 1), ceil(W/2), 8 fp16]
// -- the same 4 bytes per element as the fp32 volume, laid out so that the four runs of a halo row a tile needs (hi / lo x
// even / odd x, 17 voxels each) are contiguous in memory AND in LDS: the copy waves bring them straight into the ring of
// plane slots the MFMA phase reads.  What conv3d_c8_f16x3_zs_kernel spends between its two barriers per step -- reading the
// fp32 staging buffer, splitting, writing the parts: ~1600 of 3500 cycles with the matrix pipe idle -- is gone: no staging
// buffer, no split pass, ONE barrier per step.
//
// A step = one 8-channel chunk of one (4,4,32)-voxel tile of a group of four z-neighbouring tiles, as there.  The planes a
// workgroup consumes form one stream (6 for a (group, chunk)'s first tile, 4 for each further one); stream position t lives in
// ring slot t mod 18.  The copies of step s + 2 are issued behind the barrier of step s: the 18 slots hold the six planes step s
// reads, what step s + 1 adds and what step s + 2 adds (6 + 6 + 6 in the worst case, single-tile groups).
// LDS: 2 x 18 KiB weights + 2 parts x 2 parities x 18 slots x (6 rows x 17 voxels x 16 B) = 151 KiB.
#include "conv_split_common.h"
#include "../../include/mvs_hip_tuning.h"

#include <cstdlib>

namespace mvs {

constexpr int kPChunkBytes = 9 * 2 * 1024;   // A fragments of one 8-channel chunk: (kz,ky) x (hi,lo) x 1 KiB (the pack of conv_f16x3.hip)
constexpr int kPGroup = 4, kPCopyWaves = 4, kPThreads = 512 + 64 * kPCopyWaves;
constexpr int kPSlots = 18, kPRowVox = 17, kPRows = 6, kPSlotBytes = kPRows * kPRowVox * 16;   // 1632
// parity planes a multiple of 256 bytes apart: the two 8-lane halves of a ds_read_b128 service group (x even / x odd) then
// read complementary 16-byte slots
constexpr int kPParityBytes = (kPSlots * kPSlotBytes + 255) / 256 * 256, kPPartBytes = 2 * kPParityBytes;

template <int CIN, int ABL = 0>
__global__ __launch_bounds__(kPThreads) void conv3d_c8h_f16x3_kernel(ConvArgs a, int ngroups,
                                                                     const unsigned *__restrict__ in_absmax,
                                                                     unsigned *__restrict__ out_absmax) {
    constexpr int NCHUNK = CIN / 8, T = kPGroup, NC = kPCopyWaves;
    constexpr int WBYTES = kPChunkBytes, WCOPIES = WBYTES / 1024;
    constexpr int S_OFF = 2 * WBYTES, SLOT = kPSlotBytes, QS = kPParityBytes, PS = kPPartBytes;
    constexpr int PIECES = kPRows * kPRowVox;        // 102 16-byte pieces of one (plane, part, parity): 64 + 38
    static_assert(S_OFF + 2 * PS <= 160 * 1024 && PS + 6 * 272 < 65536, "LDS budget / offset field");
    __shared__ __attribute__((aligned(16))) unsigned char lds[S_OFF + 2 * PS];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4;
    const unsigned lds_base = (unsigned)(uintptr_t)lds;
    const bool copier = wv >= 8;
    const int cw = wv - 8;

    // the producer scaled by 2^(14 - exponent(block)); the weights' scale is the trailer of their pack
    const int xe = absmax_exponent(load_absmax(in_absmax));
    const float isx = pow2f(xe - 14);
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
    const int ngz = (a.tiles_z + T - 1) / T;
    struct Grp { int tx, ty, zg, b; };
    auto decode = [&](int g) {
        Grp r;
        r.ty = g % a.tiles_y; g /= a.tiles_y;
        r.tx = g % a.tiles_x; g /= a.tiles_x;
        r.zg = g % ngz; r.b = g / ngz;
        return r;
    };
    auto mod18 = [](int t) { return t % kPSlots; };

    if (copier) {
        // ================================================================ copy waves
        // One instruction = (plane, part, local parity, half): pieces q = half * 64 + lane < 102 of that run -- row y = q / 17,
        // voxel i = q % 17 -- to slot(plane) of the (part, parity) plane.  Local x' = 2 i + parity counts from the tile's halo
        // origin 32 tx - 1: local even = global odd.  Instruction m of a step: plane m / 8, kind m % 8; wave cw takes m = cw, cw + 4, ...
        const int Wh = (a.W + 1) >> 1;
        const int64_t sub_bytes = (int64_t)Wh * 16, crow_bytes = 4 * sub_bytes;          // one (part, parity) run / one chunk of a row
        const int64_t row_bytes = crow_bytes * NCHUNK, plane_bytes = row_bytes * a.H;
        const unsigned window_bytes = (unsigned)min((int64_t)6 * plane_bytes, (int64_t)0xffffff00u);
        // this wave's kinds: cw, cw + NC, ... (8 / NC of them): part = kind >> 2, parity = (kind >> 1) & 1, half = kind & 1
        constexpr int NU = 8 / NC;
        unsigned voff[NU];
        bool live[NU];
        Grp cg{0, 0, 0, 0};
        auto geometry = [&](int g) {
            cg = decode(g);
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int kind = cw + NC * u, part = kind >> 2, par = (kind >> 1) & 1, half = kind & 1;
                const int q = half * 64 + lane;
                live[u] = q < PIECES;
                const int qc = live[u] ? q : 0, y = qc / kPRowVox, i = qc % kPRowVox;
                const int gpar = 1 - par;                                      // global parity of local parity
                const int xh = cg.tx * 16 - 1 + i + par;                       // local even: global odd from 32 tx - 1; local odd: even from 32 tx
                const int gx = 2 * xh + gpar, gy = cg.ty * 4 - 1 + y;
                const bool ok = xh >= 0 && gx < a.W && (unsigned)gy < (unsigned)a.H;
                voff[u] = ok ? (unsigned)((int64_t)gy * row_bytes + ((int64_t)(part * 2 + gpar) * Wh + xh) * 16) : 0xffffff00u;
            }
        };
        // stream position of the issue iterator's (group, chunk) and the iterator itself
        int it_k = 0, it_ch = 0, it_j = 0, it_nvalid = 0, it_t0 = 0;
        bool it_done = ngw <= 0;
        auto it_open = [&]() { geometry(g0 + it_k * g_step); it_nvalid = min(T, a.tiles_z - cg.zg * T); };
        auto it_issue = [&]() {
            if (it_done) return;
            const bool first = it_j == 0;
            const int np = first ? 6 : 4, zl0 = first ? 0 : 4 * it_j + 2;             // planes of this step (local index: z = 16 zg - 1 + zl)
            const int zs = cg.zg * 16 - 1 + zl0;
            const mvs_srd_t srd = make_srd(reinterpret_cast<const unsigned char *>(a.in) + ((int64_t)cg.b * a.D + zs) * plane_bytes, window_bytes);
            const unsigned choff = (unsigned)(it_ch * crow_bytes);
#pragma unroll
            for (int p = 0; p < 6; ++p) {
                if (p >= np) break;                                                 // wave-uniform
                const bool zin = (unsigned)(zs + p) < (unsigned)a.D;
                const unsigned slot_off = (unsigned)(mod18(it_t0 + zl0 + p) * SLOT);
                const unsigned soff = choff + (unsigned)(p * plane_bytes);
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const int kind = cw + NC * u, part = kind >> 2, par = (kind >> 1) & 1, half = kind & 1;
                    if (live[u])
                        glds16_buf(zin ? voff[u] : 0xffffff00u, srd, soff,
                                   lds_base + (unsigned)(S_OFF + part * PS + par * QS + half * 1024) + slot_off);
                }
            }
            if (++it_j >= it_nvalid) {
                it_t0 = mod18(it_t0 + 4 * it_nvalid + 2);
                it_j = 0;
                if (++it_ch >= NCHUNK) {
                    it_ch = 0;
                    if (++it_k >= ngw) it_done = true; else it_open();
                }
            }
        };
        const unsigned char *wsrc = reinterpret_cast<const unsigned char *>(a.wpk);
        constexpr int WHALF = (WCOPIES + 1) / 2;
        auto issue_weights = [&](int ch, int sel, int lo, int hi) {
#pragma unroll
            for (int i = 0; i < (WCOPIES + NC - 1) / NC; ++i) {
                const int g = i * NC + cw;
                if (g >= lo && g < hi) glds16(wsrc + (size_t)ch * WBYTES + (size_t)g * 1024 + lane * 16,
                                              lds_base + (unsigned)(sel * WBYTES + g * 1024));
            }
        };
        int wsel = 0;
        if (ngw > 0) {
            it_open();
            issue_weights(0, 0, 0, WCOPIES);
            it_issue();
            it_issue();
        }
        for (int k = 0; k < ngw; ++k) {
            const int nvalid = min(T, a.tiles_z - decode(g0 + k * g_step).zg * T);
#pragma unroll 1
            for (int ch = 0; ch < NCHUNK; ++ch) {
                const bool more = ch + 1 < NCHUNK || k + 1 < ngw;
                const int nch = ch + 1 < NCHUNK ? ch + 1 : 0;
#pragma unroll 1
                for (int j = 0; j < nvalid; ++j) {
                    // everything this wave has issued for step (k, ch, j) -- and, a step ahead, for the next one -- has landed
                    // once the older of the two batches is waited for; the simple form: all of it
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();           // step (k, ch, j): its planes are in the ring, the previous step's reads are done
                    if (nvalid == 1) {         // the next step is already the next chunk: its weights go out first
                        if (more) issue_weights(nch, wsel ^ 1, 0, WCOPIES);
                    } else {
                        if (more && j == 0) issue_weights(nch, wsel ^ 1, 0, WHALF);
                        if (more && j == 1) issue_weights(nch, wsel ^ 1, WHALF, WCOPIES);
                    }
                    it_issue();
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
    const int z0 = wv >> 1, y0 = (wv & 1) * 2;          // the wave's output rows: (z0, y0) and (z0, y0 + 1)
    const int ex = 2 * n + (kq >> 1);
    const int eoff = ((z0 * a.Ho + y0) * a.Wo + ex) * 8 + (kq & 1) * 4;
    // this lane's B voxel of row y0 inside a plane slot: local x' = 2n + kq -> parity kq & 1, voxel n + (kq >> 1)
    const unsigned aB = lds_base + (unsigned)(S_OFF + (kq & 1) * QS + (y0 * kPRowVox + n + (kq >> 1)) * 16);

    f32x4 acc[T][2];
#pragma unroll
    for (int j = 0; j < T; ++j) acc[j][0] = acc[j][1] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int wsel = 0, t0 = 0;
    float vmax = 0.0f;
    long long tsum[4] = {0, 0, 0, 0};
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
                MVS_LAP(2);
                __syncthreads();
                MVS_LAP(0);
                // ---- MFMA phase (as conv3d_c8_f16x3_zs_kernel): nine blocks (kz, ky) of six MFMAs alternating between the two
                // accumulators; plane z0 + kz of tile j = stream position t0 + 4j + z0 + kz
                unsigned aBz[3];
#pragma unroll
                for (int kz = 0; kz < 3; ++kz) aBz[kz] = aB + (unsigned)(mod18(t0 + 4 * j + z0 + kz) * SLOT);
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
                        bsr[g & 3][sp] = __builtin_bit_cast(f16x8, lds_read_b128<iy * kPRowVox * 16 + sp * PS>(aBz[kz]));
                    }
                };
                if constexpr (!(ABL & 2)) {
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
                }
                if constexpr (ABL & 128) {
                    f32x4 &c0 = acc[j][0], &c1 = acc[j][1];
                    asm volatile("" : "+v"(c0), "+v"(c1));
                    asm volatile("s_nop 0" ::: "memory");
                }
                MVS_LAP(1);
            });
            wsel ^= 1;
            t0 = mod18(t0 + 4 * nvalid + 2);
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
                    v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f);
                    v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
                }
                const int o = eoff + r * a.Wo * 8;
                if (rp) {
                    const float4 rs = *reinterpret_cast<const float4 *>(rp + o);
                    v[0] += rs.x; v[1] += rs.y; v[2] += rs.z; v[3] += rs.w;
                }
                *reinterpret_cast<float4 *>(ob + o) = make_float4(v[0], v[1], v[2], v[3]);
                vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
            }
        });
        MVS_LAP(3);
    }
    publish_absmax(out_absmax, vmax);
    if constexpr (ABL & 128) {
        if (lane == 0) {
            long long *dbg = reinterpret_cast<long long *>(const_cast<float *>(a.residual)) + ((int64_t)blockIdx.x * 8 + wv) * 8;
            for (int k = 0; k < 4; ++k) dbg[k] = tsum[k];
        }
    }
#undef MVS_LAP
}

// fp32 volume [B,D,H,C/8,W,8] (MVS_LAYOUT_C8) -> fp16 pairs [B,D,H,C/8,2,2,ceil(W/2),8] (MVS_LAYOUT_C8H) of x * 2^(14 - exponent(block)).
// One thread per voxel-chunk (8 values).  The block must bound |x| (values beyond it overflow fp16).
__global__ __launch_bounds__(256) void c8_to_c8h_kernel(const float *__restrict__ in, const unsigned *__restrict__ absmax,
                                                        int W, int64_t rows, unsigned short *__restrict__ out) {
    const float s = pow2f(14 - absmax_exponent(load_absmax(absmax)));
    const int Wh = (W + 1) >> 1;
    const int64_t total = rows * W;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / W;
        const int x = (int)(i - row * W);
        f32x4 v0 = *reinterpret_cast<const f32x4 *>(in + i * 8), v1 = *reinterpret_cast<const f32x4 *>(in + i * 8 + 4);
        u32x4 h, l;
        split2_block(v0, v1, s, h, l);
        unsigned short *o = out + ((row * 4 + (x & 1)) * Wh + (x >> 1)) * 8;        // part 0
        *reinterpret_cast<u32x4 *>(o) = h;
        *reinterpret_cast<u32x4 *>(o + (int64_t)2 * Wh * 8) = l;                    // part 1
    }
}

}  // namespace mvs

using namespace mvs;

extern "C" size_t mvs_conv3d_f16x3_packed_bytes(int Cin);

extern "C" size_t mvs_c8h_bytes(int B, int C, int D, int H, int W) {
    if (B <= 0 || C <= 0 || C % 8 || D <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)B * D * H * (C / 8) * 4 * ((W + 1) / 2) * 16;
}

extern "C" int mvs_c8_to_c8h_f32(const float *in, const void *absmax, int B, int C, int D, int H, int W, void *out, void *stream) {
    if (!in || !absmax || !out || mvs_c8h_bytes(B, C, D, H, W) == 0) {
        set_error("mvs_c8_to_c8h_f32: invalid argument");
        return MVS_EINVAL;
    }
    const int64_t rows = (int64_t)B * D * H * (C / 8);
    const int64_t blocks = (rows * W + 255) / 256;
    hipLaunchKernelGGL(c8_to_c8h_kernel, dim3((unsigned)(blocks > 16384 ? 16384 : blocks)), dim3(256), 0, as_stream(stream), in,
                       static_cast<const unsigned *>(absmax), W, rows, static_cast<unsigned short *>(out));
    return check_launch("mvs_c8_to_c8h_f32");
}

extern "C" int mvs_conv3d_c8h_f16x3_f32(const void *in_pairs, const void *in_absmax, const void *packed, const float *scale,
                                        const float *shift, const float *residual, int relu, int B, int Cin,
                                        int D, int H, int W, float *out, void *out_absmax, void *stream) {
    if (!in_pairs || !in_absmax || !packed || !out || B <= 0 || D <= 0 || H <= 0 || W <= 0 || mvs_conv3d_f16x3_packed_bytes(Cin) == 0) {
        set_error("mvs_conv3d_c8h_f16x3_f32: invalid argument (Cin in {8, 16, 32}, Cout = 8, stride 1; in_pairs = MVS_LAYOUT_C8H volume, "
                  "in_absmax = the absmax block it was scaled by, packed = mvs_conv3d_pack_weights_f16x3_f32)");
        return MVS_EINVAL;
    }
    if ((int64_t)9 * H * W * Cin * 4 >= 0xffffff00LL) return bare_error(MVS_EINVAL, __func__, __LINE__);
    ConvArgs a;
    a.in = static_cast<const float *>(in_pairs); a.wpk = static_cast<const float *>(packed); a.scale = scale; a.shift = shift;
    a.residual = residual; a.out = out;
    a.B = B; a.D = D; a.H = H; a.W = W;
    a.Do = D; a.Ho = H; a.Wo = W;
    a.tiles_x = (W + 31) / 32; a.tiles_y = (H + 3) / 4; a.tiles_z = (D + 3) / 4;
    a.relu = relu; a.in_c8 = 1; a.ystrip = 8; a.res_up2 = 0;
    const int64_t ng = (int64_t)B * a.tiles_x * a.tiles_y * ((a.tiles_z + kPGroup - 1) / kPGroup);
    if (ng <= 0 || ng > 0x7fffffffLL) return bare_error(MVS_EINVAL, __func__, __LINE__);
    const int n_cu = device_cu_count();
    hipStream_t st = as_stream(stream);
    const dim3 grid((unsigned)(ng < n_cu ? ng : n_cu)), blk(kPThreads);
    const unsigned *mx = static_cast<const unsigned *>(in_absmax);
    unsigned *omx = static_cast<unsigned *>(out_absmax);
#ifdef MVS_TUNING   // phase-stamp / ablation builds (scripts/exp_conv0_pairs.py)
    static const int abl = [] { const char *e = getenv("MVS_CONV_SPLIT_ABL"); return e ? atoi(e) : 0; }();
    if ((abl & 128) && Cin == 32) {
        if (!residual) return bare_error(MVS_EINVAL, __func__, __LINE__);
        hipLaunchKernelGGL((conv3d_c8h_f16x3_kernel<32, 128>), grid, blk, 0, st, a, (int)ng, mx, omx);
        return check_launch("mvs_conv3d_c8h_f16x3_f32");
    }
    if ((abl & 2) && Cin == 32) {   // no MFMA phase: copies and barriers only (wrong results)
        hipLaunchKernelGGL((conv3d_c8h_f16x3_kernel<32, 2>), grid, blk, 0, st, a, (int)ng, mx, omx);
        return check_launch("mvs_conv3d_c8h_f16x3_f32");
    }
#endif
    if (Cin == 32) hipLaunchKernelGGL((conv3d_c8h_f16x3_kernel<32>), grid, blk, 0, st, a, (int)ng, mx, omx);
    else if (Cin == 16) hipLaunchKernelGGL((conv3d_c8h_f16x3_kernel<16>), grid, blk, 0, st, a, (int)ng, mx, omx);
    else hipLaunchKernelGGL((conv3d_c8h_f16x3_kernel<8>), grid, blk, 0, st, a, (int)ng, mx, omx);
    return check_launch("mvs_conv3d_c8h_f16x3_f32");
}
