// conv0-class layers (3x3x3, Cout = 8, stride 1, 8-channel-blocked input: MVSNet's conv0 32 -> 8 and
// the cascade's first layers; mvsnet.py:66, module.py:26-33) on the BF16 matrix pipe at FP32 accuracy.
//
// gfx950 has no reduced-precision fast path for fp32 operands (no xf32), and its fp32 MFMA runs at the
// vector rate: conv0 is 314 GFLOP at 157 TFLOP/s = 2.0 ms before the 25 % padding of its Cout = 8 shifted
// form.  The bf16 MFMA is 16x faster.  Every fp32 number is EXACTLY the sum of three bf16 numbers
// (8 + 8 + 8 significand bits: hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid), each
// subtraction exact), so a product a b = (ah + am + al)(bh + bm + bl); the six terms ah bh, ah bm, am bh,
// ah bl, al bh, am bm carry it to 2^-25 relative (the three dropped terms are below fp32's own product
// rounding), each term is exact in the fp32 accumulator's multiplier, and the matrix pipe sums in fp32.
// Measured on the golden conv0 case against fp64: max error 2.4e-10 for the six-term form with exact
// accumulation vs 2.1e-9 for ATen's fp32 convolution -- the split form is not the less accurate one.
// Six bf16 MFMAs of K = 32 replace eight fp32 MFMAs of K = 4: 2.7x less matrix time.
//
// Structure = the persistent DMA-fed kernel of conv_persistent.h ((4,4,32)-voxel tiles, strip order per XCD, same
// epilogue), a step being one 8-channel chunk of one tile:
//   * four COPY WAVES beside the eight multiplying ones issue the LDS-DMA copies of the next step's fp32 halo (39 KiB,
//     the 36 (z, y) rows as they lie in memory: neighbouring lanes = neighbouring 16-byte pieces) and, at a chunk
//     change, of the next weight chunk -- a copy blocks the wave that issues it for ~250 cycles;
//   * a split pass between two barriers: every halo voxel once, fp32 -> hi, mid, lo -> three bf16 planes in LDS laid
//     out [row][even x ... | odd x ...], so that a B fragment (lane (n, kq) = voxel x = 2n + kq, 8 channels) is one
//     conflict-free ds_read_b128 per part (splitting in registers at fragment-read time was 5x redundant);
//   * one v_mfma_f32_16x16x32_bf16 covers the FOUR x-taps of the shifted Cout = 8 form x 8 channels (K = 32); a
//     wave's two output rows are y-neighbours, so the 12 distinct (z, y) input rows of a chunk are read once and
//     used by both rows: 108 MFMAs and 63 ds_read_b128 per wave and step;
//   * the weights pre-split on the host side of the call (mvs_conv3d_pack_weights_bf16x6_f32) into A fragments
//     [chunk][kz,ky][hi,mid,lo][lane][8 bf16], 27 KiB per chunk, double-buffered; groups of four tiles share a
//     chunk's weights and keep their accumulators in registers across the chunk loop.
// Measurements, per-phase cycle counts and the variants that did not beat this form: DESIGN.md sections 4 and 6.
#include "conv_split_common.h"

#include <cstdlib>

namespace mvs {

constexpr int kSplitChunkBytes = 9 * 3 * 1024;   // A fragments of one 8-channel chunk: (kz,ky) x (hi,mid,lo) x 1 KiB

// Halo of one 8-channel chunk in LDS: [channel half][voxel][4 floats]; voxels = the 36 (z, y) rows' 17 even-x
// voxels, then -- from voxel kOddBase, a multiple of 16 -- the rows' 17 odd-x voxels.  A wave's B read (lane
// (n, kq) -> x = 2n + kq of one row) then touches, in each 16-lane service group of ds_read_b128, sixteen
// different 16-byte slots modulo 256 bytes: conflict-free (an odd half starting 17 voxels after its even
// half, as in the fp32 kernel's b64 layout, collides in one slot per group).
constexpr int kRowVox = 17, kRows = 36, kOddBase = 624, kHaloVox = kOddBase + kRows * kRowVox, kHaloPlane = 1280;
static_assert(kOddBase % 16 == 0 && kOddBase >= kRows * kRowVox && kHaloVox <= kHaloPlane, "halo layout");


// tiles of one group: they share each weight chunk in LDS (one 27-KiB copy per chunk and GROUP instead of per
// chunk and tile) and keep their accumulators in registers across the chunk loop
constexpr int kGroup = 4;

// Besides the eight waves that split and multiply, the workgroup has kCopyWaves waves that do nothing but issue the
// LDS-DMA copies of the next step (and wait for them).  Issued by the eight working waves themselves -- five 1-KiB
// pieces each, in one burst behind the barrier -- the copies cost every wave ~1300 cycles of a 7700-cycle step: a
// vector-memory instruction blocks its wave at issue while the address unit is busy, wherever in the stream it sits.
constexpr int kCopyWaves = 4, kSplitKernelThreads = 512 + 64 * kCopyWaves;

template <int CIN, bool DOT2, int ABL = 0>
__global__ __launch_bounds__(kSplitKernelThreads) void conv3d_c8_bf16x6_kernel(ConvArgs a, int ntiles) {
    constexpr int NCHUNK = CIN / 8, YT = 6, PLANE = kHaloPlane, T = kGroup;
    constexpr int NC = kCopyWaves;
    constexpr int NCOPY = (kRows * 68 + 63) / 64, IPW = (NCOPY + NC - 1) / NC;     // 39 wave-copies per chunk
    // LDS: two weight chunks (double-buffered), the fp32 halo as the copy engine delivers it, its three bf16 parts
    constexpr int WBYTES = kSplitChunkBytes, FBYTES = 2 * PLANE * 16, SPART = PLANE * 16, SBYTES = 3 * SPART;
    constexpr int F_OFF = 2 * WBYTES, S_OFF = F_OFF + FBYTES;
    static_assert(S_OFF + SBYTES <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(16))) unsigned char lds[S_OFF + SBYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4;
    const unsigned lds_base = (unsigned)(uintptr_t)lds;
    const int abl = a.res_up2;   // tuning: 1 = no chunk compute, 2 = no halo copies, 4 = no weight copies
    const bool copier = wv >= 8;
    const int cw = wv - 8;       // copy wave index

    // this workgroup's tiles: t0 + k * t_step, k < ntw (XCD x owns a contiguous range of the ordered tile list)
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

    // fp32 halo buffer = the 36 (z, y) rows as they lie in memory: 34 voxels x 32 bytes = 68 16-byte pieces per
    // row, piece P = row * 68 + q at byte 16 P.  Neighbouring lanes copy neighbouring pieces, so a 64-byte line
    // is requested once (a de-interleaved destination -- even-x voxels first, channel halves apart -- made every
    // lane its own 64-byte request for 16 useful bytes: 4x the requests, and the copy rate is set by requests).
    // A copy wave's items: copy g = i*NC + cw brings pieces g*64 + lane.
    constexpr int ROWP = 68, NPIECE = kRows * ROWP;               // 2448 pieces = 38.25 wave-copies
    int loc[IPW];
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
        const int P = ((i * NC + cw) & 63) * 64 + lane;
        const bool ok = P < NPIECE;
        const int Pc = ok ? P : 0;
        const int row = Pc / ROWP, q = Pc % ROWP;
        loc[i] = (q >> 1) | ((row % YT) << 8) | ((row / YT) << 16) | ((q & 1) << 24) | (ok ? 0 : (int)0x80000000);
    }
    // the bf16 parts: position p of voxel (row, x) = row*17 + x/2 (+ kOddBase for odd x).  This thread's items of
    // the split pass (every wave of the workgroup takes part, the copy waves too -- they are idle between the two
    // barriers of a step): piece P = ps*NT + tid -> 8 bytes at p*16 + half*8 of each part
    constexpr int NT = kSplitKernelThreads, NPS = (NPIECE + NT - 1) / NT;
    unsigned spos[NPS];
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps) {
        const int P = ps * NT + tid;
        const int Pc = P < NPIECE ? P : 0;
        const int row = Pc / ROWP, q = Pc % ROWP, x = q >> 1;
        spos[ps] = P < NPIECE ? (unsigned)((row * kRowVox + (x >> 1) + (x & 1) * kOddBase) * 16 + (q & 1) * 8) : 0xffffffffu;
    }
    const int64_t plane_in = (int64_t)a.H * a.W * CIN;
    const int row_in = a.W * CIN;
    const unsigned window_bytes = (unsigned)min((int64_t)6 * plane_in * 4, (int64_t)0xffffff00u);
    unsigned voff[T][IPW];
    mvs_srd_t srd[T];
    auto geometry = [&](auto jc, int t) {
        constexpr int j = decltype(jc)::value;
        const TileIdx tile = decode_ordered_tile(a, t);
        const int ix0 = tile.tx * 32 - 1, iy0 = tile.ty * 4 - 1, iz0 = tile.tz * 4 - 1;
        srd[j] = make_srd(a.in + ((int64_t)tile.b * a.D + iz0) * plane_in, window_bytes);
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            const int gx = ix0 + (loc[i] & 255), gy = iy0 + ((loc[i] >> 8) & 255);
            const int lz = (loc[i] >> 16) & 255, h = (loc[i] >> 24) & 1;
            const bool ok = loc[i] >= 0 && (unsigned)gx < (unsigned)a.W && (unsigned)gy < (unsigned)a.H &&
                            (unsigned)(iz0 + lz) < (unsigned)a.D;
            voff[j][i] = ok ? (unsigned)(((int64_t)lz * plane_in + (int64_t)gy * row_in + gx * 8 + h * 4) * 4) : 0xffffff00u;
        }
    };
    auto issue_halo = [&](auto jc, int ch) {
        constexpr int j = decltype(jc)::value;
        if (abl & 2) return;
        const unsigned soff = (unsigned)(ch * a.W * 32);
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            if (i * NC + cw >= NCOPY) continue;   // wave-uniform
            glds16_buf(voff[j][i], srd[j], soff, lds_base + (unsigned)(F_OFF + (i * NC + cw) * 1024));
        }
    };
    const unsigned char *wsrc = reinterpret_cast<const unsigned char *>(a.wpk);
    auto issue_weights = [&](int ch, int sel) {   // 27 KiB = 27 wave-copies over the copy waves
        if (abl & 4) return;
#pragma unroll
        for (int i = 0; i < (27 + NC - 1) / NC; ++i) {
            const int g = i * NC + cw;
            if (g < 27) glds16(wsrc + (size_t)ch * WBYTES + (size_t)g * 1024 + lane * 16,
                               lds_base + (unsigned)(sel * WBYTES + g * 1024));
        }
    };

    // split pass: every halo voxel once (the MFMA phase reads each ~5 times), fp32 -> hi, mid, lo
    auto split_pass = [&]() {
        f32x4 x[NPS];
        const unsigned fp = lds_base + (unsigned)(F_OFF + tid * 16);
        static_for<0, NPS>([&](auto pc) {
            constexpr int ps = decltype(pc)::value;
            x[ps] = lds_read_b128<ps * NT * 16>(fp);
        });
        lds_wait_n<0>();
        static_for<0, (NPS + 1) / 2>([&](auto pc) {
            constexpr int p0 = 2 * decltype(pc)::value, p1 = (p0 + 1 < NPS) ? p0 + 1 : p0;
            asm volatile("" : "+v"(x[p0]), "+v"(x[p1]));
            bf16x8 h, m, l;
            if constexpr (ABL & 8) {
                h = __builtin_bit_cast(bf16x8, x[p0]); m = __builtin_bit_cast(bf16x8, x[p1]); l = h;
            } else if constexpr (DOT2) split3_block(x[p0], x[p1], h, m, l);
            else split3<false>(x[p0], x[p1], h, m, l);
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 hu = __builtin_bit_cast(u32x4, h), mu = __builtin_bit_cast(u32x4, m), lu = __builtin_bit_cast(u32x4, l);
            if (spos[p0] != 0xffffffffu) {
                const unsigned sp = lds_base + (unsigned)S_OFF + spos[p0];
                lds_write_b64<0>(sp, hu[0], hu[1]);
                lds_write_b64<SPART>(sp, mu[0], mu[1]);
                lds_write_b64<2 * SPART>(sp, lu[0], lu[1]);
            }
            if (p1 != p0 && spos[p1] != 0xffffffffu) {
                const unsigned sp = lds_base + (unsigned)S_OFF + spos[p1];
                lds_write_b64<0>(sp, hu[2], hu[3]);
                lds_write_b64<SPART>(sp, mu[2], mu[3]);
                lds_write_b64<2 * SPART>(sp, lu[2], lu[3]);
            }
        });
        lds_wait_n<0>();
    };

    if (copier) {
        // ================================================================ copy waves: the walk of steps, two barriers each
        int wsel = 0;
        if (ntw > 0) {
            geometry(std::integral_constant<int, 0>{}, t0);
            issue_halo(std::integral_constant<int, 0>{}, 0);
            issue_weights(0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        for (int k0 = 0; k0 < ntw; k0 += T) {
            const int nvalid = min(T, ntw - k0);
#pragma unroll 1
            for (int ch = 0; ch < NCHUNK; ++ch) {
                static_for<0, T>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    if (j >= nvalid) return;   // wave-uniform
                    __syncthreads();           // the halo of this step is in the fp32 buffer (this wave has waited for it)
                    if (!(abl & 1)) split_pass();
                    __syncthreads();           // ... and has been split: the fp32 buffer is free
                    if (j + 1 < nvalid) {
                        if (ch == 0) geometry(std::integral_constant<int, (j + 1) % T>{}, t0 + (k0 + j + 1) * t_step);
                        issue_halo(std::integral_constant<int, (j + 1) % T>{}, ch);
                    } else if (ch + 1 < NCHUNK) {
                        issue_halo(std::integral_constant<int, 0>{}, ch + 1);
                        issue_weights(ch + 1, wsel ^ 1);
                    } else if (k0 + T < ntw) {
                        geometry(std::integral_constant<int, 0>{}, t0 + (k0 + T) * t_step);
                        issue_halo(std::integral_constant<int, 0>{}, 0);
                        issue_weights(0, wsel ^ 1);
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                });
                wsel ^= 1;
            }
        }
        return;
    }

    float4 sc, sh;
    {
        const int c0 = (kq & 1) * 4;
        sc = a.scale ? *reinterpret_cast<const float4 *>(a.scale + c0) : make_float4(1.f, 1.f, 1.f, 1.f);
        sh = a.shift ? *reinterpret_cast<const float4 *>(a.shift + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
    }

    const int z0 = wv >> 1, y0 = (wv & 1) * 2;          // the wave's output rows: (z0, y0) and (z0, y0 + 1)
    // epilogue: this lane's x inside a tile and its element offset from the tile's first output (row r adds Wo * 8)
    const int ex = 2 * n + (kq >> 1);
    const int eoff = ((z0 * a.Ho + y0) * a.Wo + ex) * 8 + (kq & 1) * 4;
    // this lane's B voxel of halo row (z0, y0): x = 2n + kq
    const unsigned aB = lds_base + (unsigned)(S_OFF + ((z0 * YT + y0) * kRowVox + n + (kq >> 1) + (kq & 1) * kOddBase) * 16);

    f32x4 acc[T][2];
#pragma unroll
    for (int j = 0; j < T; ++j) acc[j][0] = acc[j][1] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int wsel = 0;
    // ABL & 128 (tuning): cycles of each wave per phase, summed over the workgroup's steps, into the buffer passed
    // as `residual` (int64 [workgroup][wave][8]): copy wait, barrier, split pass, barrier, copy issue, MFMA phase, epilogue
    long long tsum[7] = {0, 0, 0, 0, 0, 0, 0};
    long long tprev = 0;
    if constexpr (ABL & 128) tprev = clock64();
#define MVS_LAP(k) do { if constexpr (ABL & 128) { const long long tn = clock64(); tsum[k] += tn - tprev; tprev = tn; } } while (0)
    for (int k0 = 0; k0 < ntw; k0 += T) {
        const int nvalid = min(T, ntw - k0);
#pragma unroll 1
        for (int ch = 0; ch < NCHUNK; ++ch) {
            static_for<0, T>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                if (j >= nvalid) return;   // wave-uniform
                // ---- the copies of this step have landed; every wave is done with the previous step's bf16 parts
                MVS_LAP(6);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                MVS_LAP(0);
                __syncthreads();
                MVS_LAP(1);
                if (!(abl & 1)) split_pass();
                MVS_LAP(2);
                __syncthreads();
                MVS_LAP(3);
                // (the copy waves now request the next step's halo and, at a chunk change, the next weight chunk)
                MVS_LAP(4);
                if (abl & 1) return;
                // ---- MFMA phase: 12 input-row fragments f = (kz, iy) into row 0 (ky = iy) and row 1 (ky = iy - 1);
                // the reads of fragment f+1 (and of the weight triple it brings in) go out before the MFMAs of f
                const unsigned aA = lds_base + (unsigned)(wsel * WBYTES + lane * 16);
                bf16x8 bs[2][3], A[3][3];
                auto read_b = [&](auto fc) {
                    constexpr int f = decltype(fc)::value, off = ((f / 4) * YT + f % 4) * kRowVox * 16;
                    static_for<0, 3>([&](auto pc) {
                        constexpr int sp = decltype(pc)::value;
                        bs[f & 1][sp] = __builtin_bit_cast(bf16x8, lds_read_b128<off + sp * SPART>(aB));
                    });
                };
                auto read_a = [&](auto kzc, auto kyc) {
                    constexpr int kz = decltype(kzc)::value, ky = decltype(kyc)::value;
                    static_for<0, 3>([&](auto pc) {
                        constexpr int sp = decltype(pc)::value;
                        A[ky][sp] = __builtin_bit_cast(bf16x8, lds_read_b128<((kz * 3 + ky) * 3 + sp) * 1024>(aA));
                    });
                };
                read_a(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
                read_b(std::integral_constant<int, 0>{});
                static_for<0, 12>([&](auto fc) {
                    constexpr int f = decltype(fc)::value, iy = f % 4;
                    constexpr int f1 = f + 1, kz1 = f1 / 4, iy1 = f1 % 4;
                    lds_wait_n<0>();
                    asm volatile("" : "+v"(bs[f & 1][0]), "+v"(bs[f & 1][1]), "+v"(bs[f & 1][2]));
                    if constexpr (iy <= 2) asm volatile("" : "+v"(A[iy][0]), "+v"(A[iy][1]), "+v"(A[iy][2]));
                    if constexpr (f1 < 12) {
                        if constexpr (iy1 <= 2) read_a(std::integral_constant<int, kz1>{}, std::integral_constant<int, iy1>{});
                        read_b(std::integral_constant<int, f1>{});
                    }
                    __builtin_amdgcn_sched_barrier(0);   // the reads go out BEFORE this fragment's MFMAs
                    const bf16x8 bh = bs[f & 1][0], bm = bs[f & 1][1], bl = bs[f & 1][2];
                    // six partial products per output row, small terms first; rows interleaved so that
                    // neighbouring MFMAs do not wait on each other's accumulator
                    static_for<0, 6>([&](auto tc) {
                        constexpr int t = decltype(tc)::value;
                        constexpr int as = (t == 0 || t == 3) ? 1 : (t == 1 ? 2 : 0);       // am al ah am ah ah
                        static_for<0, 2>([&](auto rc) {
                            constexpr int r = decltype(rc)::value, ky = iy - r;
                            if constexpr (ky >= 0 && ky <= 2) {
                                const bf16x8 &bb = (t == 0 || t == 4) ? bm : (t == 2 ? bl : bh);   // bm bh bl bh bm bh
                                const bf16x8 &aa = A[ky][as];
                                f32x4 &cc = acc[j][r];
                                if constexpr (ABL & 16) asm volatile("" ::"v"(aa), "v"(bb));
                                else cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aa, bb, cc, 0, 0, 0);
                            }
                        });
                    });
                    __builtin_amdgcn_sched_barrier(0);
                });
                if constexpr (ABL & 128) {
                    f32x4 &c0 = acc[j][0], &c1 = acc[j][1];
                    asm volatile("" : "+v"(c0), "+v"(c1));
                    asm volatile("s_nop 0" ::: "memory");
                }
                MVS_LAP(5);
            });
            wsel ^= 1;
        }
        // ---- epilogue of the group: BN affine, ReLU, one 16-byte store per lane and row (as the fp32 kernel's MODE 2).
        // The matrix pipe idles here and a wave issues an instruction every ~7 cycles: the tile's base address is scalar
        // arithmetic, a lane adds a precomputed in-tile offset.
        static_for<0, T>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if (j >= nvalid) return;
            const TileIdx cur = decode_ordered_tile(a, __builtin_amdgcn_readfirstlane(t0 + (k0 + j) * t_step));
            const int tb = __builtin_amdgcn_readfirstlane(cur.b), oz0 = __builtin_amdgcn_readfirstlane(cur.tz) * 4;
            const int oy0 = __builtin_amdgcn_readfirstlane(cur.ty) * 4, ox0 = __builtin_amdgcn_readfirstlane(cur.tx) * 32;
            const int64_t base = ((((int64_t)tb * a.Do + oz0) * a.Ho + oy0) * a.Wo + ox0) * 8;
            float *const ob = a.out + base;
            const float *const rp = (a.residual && !(ABL & 128)) ? a.residual + base : nullptr;
            const bool xz_in = oz0 + z0 < a.Do && ox0 + ex < a.Wo;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                f32x4 v = acc[j][r];
                acc[j][r] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (!xz_in || oy0 + y0 + r >= a.Ho) continue;
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
            }
        });
    }
    if constexpr (ABL & 128) {
        MVS_LAP(6);
        if (lane == 0) {
            long long *dbg = reinterpret_cast<long long *>(const_cast<float *>(a.residual)) + ((int64_t)blockIdx.x * 8 + wv) * 8;
            for (int k = 0; k < 7; ++k) dbg[k] = tsum[k];
        }
    }
#undef MVS_LAP
}

// ---------------------------------------------------------------------------------------------------------------
// The same layer with a Z-SLIDING WINDOW (round 3).  A work unit is a GROUP of four z-neighbouring (4,4,32)-voxel tiles:
// 16 output planes of one (y, x) tile.  Per 8-channel chunk the group's 18 halo planes enter LDS ONCE: step (j, chunk)
// brings planes 16 zg + 4j + 1 ... + 4 (four planes = 24 rows; step 0 the six planes -1 ... 4) instead of a tile's six,
// and the bf16 parts live in a ring of six plane slots, slot(z) = (z + 1 - 16 zg) mod 6 -- a step overwrites the four
// slots its predecessor no longer needs.  25 % fewer copied and split bytes per output (27 rows per step on average
// instead of 36): in the per-tile form the copies of a step (39 KiB at the ~11 bytes/cycle a CU sustains) did not
// land inside the 3240-cycle MFMA phase they were meant to hide under -- the barrier in front of the split pass
// waited for them -- and the split pass is bound by its LDS writes.  Results are bit-identical to the per-tile
// kernel: same fragments, same MFMA order per output.
// Group order: ty fastest, then tx, then z group, then batch; XCD x owns a contiguous range of the list and its
// 32 workgroups walk 32 consecutive groups at a time -- y-neighbours, which share two of six halo rows in the XCD's L2.
template <int CIN, int ABL = 0>
__global__ __launch_bounds__(kSplitKernelThreads) void conv3d_c8_bf16x6_zs_kernel(ConvArgs a, int ngroups) {
    constexpr int NCHUNK = CIN / 8, YT = 6, PLANE = kHaloPlane, T = kGroup;
    constexpr int NC = kCopyWaves, NT = kSplitKernelThreads;
    constexpr int ROWP = 68;                                        // 16-byte pieces per (z, y) row: 34 voxels x 2 halves
    constexpr int NPIECE0 = 36 * ROWP, NPIECE1 = 24 * ROWP;         // step 0: six planes; later steps: four
    constexpr int NCOPY0 = (NPIECE0 + 63) / 64, NCOPY1 = (NPIECE1 + 63) / 64, IPW = (NCOPY0 + NC - 1) / NC;
    constexpr int WBYTES = kSplitChunkBytes, FBYTES = 2 * PLANE * 16, SPART = PLANE * 16, SBYTES = 3 * SPART;
    constexpr int F_OFF = 2 * WBYTES, S_OFF = F_OFF + FBYTES;
    constexpr int SLOT = YT * kRowVox * 16;                         // bytes of one plane slot inside a part
    static_assert(NPIECE0 * 16 <= FBYTES && S_OFF + SBYTES <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(16))) unsigned char lds[S_OFF + SBYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4;
    const unsigned lds_base = (unsigned)(uintptr_t)lds;
    const bool copier = wv >= 8;
    const int cw = wv - 8;       // copy wave index

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
    // -> 8 bytes of each part at slot((zl + first slot of the step) mod 6) + ((y * 17 + x / 2 (+ kOddBase for odd x)) * 16
    // + half * 8.  Thread order rotated so that the copy waves -- idle between the two barriers -- own the ragged tail.
    const int tidr = tid < 512 ? tid + 256 : tid - 512;
    constexpr int NPS = (NPIECE0 + NT - 1) / NT;
    unsigned spos[NPS];          // in-slot byte position | zl << 16
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps) {
        const int P = ps * NT + tidr;
        const int Pc = P < NPIECE0 ? P : 0;
        const int row = Pc / ROWP, q = Pc % ROWP, x = q >> 1;
        spos[ps] = (unsigned)(((row % YT) * kRowVox + (x >> 1) + (x & 1) * kOddBase) * 16 + (q & 1) * 8) | ((unsigned)(row / YT) << 16);
    }
    auto split_pass = [&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int NP = j == 0 ? NPIECE0 : NPIECE1, NPSJ = (NP + NT - 1) / NT;
        constexpr int ZF = j == 0 ? 0 : (4 * j + 2) % 6;             // ring slot of the first staged plane
        f32x4 x[NPSJ];
        const unsigned fp = lds_base + (unsigned)(F_OFF + tidr * 16);
        const int wbase = tidr & ~63;                                 // wave-uniform
        static_for<0, NPSJ>([&](auto pc) {
            constexpr int ps = decltype(pc)::value;
            if (ps * NT + wbase < NP) x[ps] = lds_read_b128<ps * NT * 16>(fp);
        });
        lds_wait_n<0>();
        static_for<0, (NPSJ + 1) / 2>([&](auto pc) {
            constexpr int p0 = 2 * decltype(pc)::value, p1 = (p0 + 1 < NPSJ) ? p0 + 1 : p0;
            if (p0 * NT + wbase >= NP) return;                        // the whole wave has nothing here
            asm volatile("" : "+v"(x[p0]), "+v"(x[p1]));
            bf16x8 h, m, l;
            split3_block(x[p0], x[p1], h, m, l);
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 hu = __builtin_bit_cast(u32x4, h), mu = __builtin_bit_cast(u32x4, m), lu = __builtin_bit_cast(u32x4, l);
            auto dest = [&](unsigned sp) {
                unsigned s = (sp >> 16) + ZF;
                s = min(s, s - 6u);                                   // mod 6 (s < 12)
                return lds_base + (unsigned)S_OFF + (sp & 0xffffu) + s * (unsigned)SLOT;
            };
            if (p0 * NT + tidr < NP) {
                const unsigned sp = dest(spos[p0]);
                lds_write_b64<0>(sp, hu[0], hu[1]);
                lds_write_b64<SPART>(sp, mu[0], mu[1]);
                lds_write_b64<2 * SPART>(sp, lu[0], lu[1]);
            }
            if (p1 != p0 && p1 * NT + tidr < NP) {
                const unsigned sp = dest(spos[p1]);
                lds_write_b64<0>(sp, hu[2], hu[3]);
                lds_write_b64<SPART>(sp, mu[2], mu[3]);
                lds_write_b64<2 * SPART>(sp, lu[2], lu[3]);
            }
        });
        lds_wait_n<0>();
    };

    if (copier) {
        // ================================================================ copy waves
        // staging piece P = (i * NC + cw) * 64 + lane: the rows as they lie in memory, neighbouring lanes =
        // neighbouring pieces (a 64-byte line is requested once)
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
        auto issue_halo = [&](auto jc, int ch) {
            constexpr int j = decltype(jc)::value;
            constexpr int NP = j == 0 ? NPIECE0 : NPIECE1, NCP = j == 0 ? NCOPY0 : NCOPY1;
            const int zs = cg.zg * 16 + (j == 0 ? -1 : 4 * j + 1);   // first staged plane
            const mvs_srd_t srd = make_srd(a.in + ((int64_t)cg.b * a.D + zs) * plane_in, window_bytes);
            const unsigned soff = (unsigned)(ch * a.W * 32);
#pragma unroll
            for (int i = 0; i < IPW; ++i) {
                if (i * NC + cw >= NCP) continue;   // wave-uniform
                const int P = (i * NC + cw) * 64 + lane, lz = (loc[i] >> 16) & 255;
                const bool ok = P < NP && (unsigned)(zs + lz) < (unsigned)a.D;
                glds16_buf(ok ? voff[i] : 0xffffff00u, srd, soff, lds_base + (unsigned)(F_OFF + (i * NC + cw) * 1024));
            }
        };
        const unsigned char *wsrc = reinterpret_cast<const unsigned char *>(a.wpk);
        auto issue_weights = [&](int ch, int sel) {   // 27 KiB = 27 wave-copies over the copy waves
#pragma unroll
            for (int i = 0; i < (27 + NC - 1) / NC; ++i) {
                const int g = i * NC + cw;
                if (g < 27) glds16(wsrc + (size_t)ch * WBYTES + (size_t)g * 1024 + lane * 16,
                                   lds_base + (unsigned)(sel * WBYTES + g * 1024));
            }
        };
        int wsel = 0;
        if (ngw > 0) {
            geometry(g0);
            issue_halo(std::integral_constant<int, 0>{}, 0);
            issue_weights(0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        for (int k = 0; k < ngw; ++k) {
            const int nvalid = min(T, a.tiles_z - cg.zg * T);
#pragma unroll 1
            for (int ch = 0; ch < NCHUNK; ++ch) {
                static_for<0, T>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    if (j >= nvalid) return;   // wave-uniform
                    __syncthreads();           // the rows of this step are in the staging buffer (this wave has waited for them)
                    split_pass(jc);
                    __syncthreads();           // ... and have been split: the staging buffer is free
                    if (j + 1 < nvalid) {
                        issue_halo(std::integral_constant<int, (j + 1) % T>{}, ch);
                    } else if (ch + 1 < NCHUNK) {
                        issue_halo(std::integral_constant<int, 0>{}, ch + 1);
                        issue_weights(ch + 1, wsel ^ 1);
                    } else if (k + 1 < ngw) {
                        geometry(g0 + (k + 1) * g_step);
                        issue_halo(std::integral_constant<int, 0>{}, 0);
                        issue_weights(0, wsel ^ 1);
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                });
                wsel ^= 1;
            }
        }
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
    const unsigned aB = lds_base + (unsigned)(S_OFF + (y0 * kRowVox + n + (kq >> 1) + (kq & 1) * kOddBase) * 16);

    f32x4 acc[T][2];
#pragma unroll
    for (int j = 0; j < T; ++j) acc[j][0] = acc[j][1] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int wsel = 0;
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
                split_pass(jc);
                MVS_LAP(2);
                __syncthreads();
                MVS_LAP(3);
                // ---- MFMA phase: 12 input-row fragments f = (kz, iy) into row 0 (ky = iy) and row 1 (ky = iy - 1);
                // plane z0 + kz of the tile sits in ring slot (4j + z0 + kz) mod 6
                unsigned aBz[3];
#pragma unroll
                for (int kz = 0; kz < 3; ++kz) {
                    int s = (4 * j) % 6 + z0 + kz;
                    s = s >= 6 ? s - 6 : s;
                    aBz[kz] = aB + (unsigned)(s * SLOT);
                }
                const unsigned aA = lds_base + (unsigned)(wsel * WBYTES + lane * 16);
                // Nine blocks c = (kz, ky) of twelve MFMAs: the weight triple A(kz, ky) against input row (kz, ky) into
                // output row 0 and against input row (kz, ky + 1) into output row 1, ALTERNATING between the two
                // accumulators (each accumulator sees its products in the same order as before: bit-identical).  Back-to-
                // back MFMAs on ONE accumulator wait for each other; the fragment-by-fragment form had two such runs of
                // six per kz and issued its six reads as a block in front of each fragment -- fine while the partner wave
                // of the SIMD fills the gaps, but the older wave wins the arbitration, finishes its 108 MFMAs in ~2400
                // cycles and leaves the younger one alone at 62 % of the pipe's rate (phase laps).  Here
                // the reads of block c + 1 ride in the issue shadow of block c's MFMAs 4 ... 11, ordered as block c + 1
                // consumes them (mid, mid | lo, hi | hi, lo) with counted waits.
                bf16x8 bsr[4][3], Aw[2][3];        // input rows g = kz * 4 + iy in slot g % 4; weight triples c in slot c % 2
                auto rd = [&](auto ic, auto cc) {   // i-th read of the set that block c needs
                    constexpr int i = decltype(ic)::value, c = decltype(cc)::value, kz = c / 3, ky = c % 3;
                    constexpr bool two = ky == 0;                     // both rows are new at a kz change
                    constexpr int per = two ? 3 : 2, what = i % per, lvl = i / per;   // what: 0 = A, 1.. = B rows
                    constexpr int spA = lvl == 0 ? 1 : (lvl == 1 ? 2 : 0), spB = lvl == 0 ? 1 : (lvl == 1 ? 0 : 2);
                    if constexpr (what == 0) {
                        Aw[c & 1][spA] = __builtin_bit_cast(bf16x8, lds_read_b128<(c * 3 + spA) * 1024>(aA));
                    } else {
                        constexpr int iy = two ? ky + what - 1 : ky + 1, g = kz * 4 + iy;
                        bsr[g & 3][spB] = __builtin_bit_cast(bf16x8, lds_read_b128<iy * kRowVox * 16 + spB * SPART>(aBz[kz]));
                    }
                };
                static_for<0, 9>([&](auto ic) { rd(ic, std::integral_constant<int, 0>{}); });
                __builtin_amdgcn_sched_barrier(0);
                static_for<0, 9>([&](auto cc) {
                    constexpr int c = decltype(cc)::value, kz = c / 3, ky = c % 3, g0 = kz * 4 + ky, g1 = g0 + 1;
                    constexpr int nin = ky == 0 ? 9 : 6;                               // reads this block waits for
                    constexpr int nout = c == 8 ? 0 : ((c + 1) % 3 == 0 ? 9 : 6);      // reads it issues for block c + 1
                    static_for<0, 12>([&](auto mc) {
                        constexpr int m = decltype(mc)::value, t = m / 2, r = m % 2;
                        constexpr int as = (t == 0 || t == 3) ? 1 : (t == 1 ? 2 : 0);   // am al ah am ah ah
                        constexpr int bp = (t == 0 || t == 4) ? 1 : (t == 2 ? 2 : 0);   // bm bh bl bh bm bh
                        if constexpr (m == 0 || m == 2 || m == 4) {
                            lds_wait_n<(m == 0 ? nin - nin / 3 : (m == 2 ? nin / 3 : 0))>();
                            asm volatile("" : "+v"(Aw[c & 1][as]), "+v"(bsr[g0 & 3][bp]), "+v"(bsr[g1 & 3][bp]));
                        }
                        const bf16x8 &bb = bsr[(r == 0 ? g0 : g1) & 3][bp];
                        f32x4 &cc2 = acc[j][r];
                        cc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Aw[c & 1][as], bb, cc2, 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (m >= 4 && m - 4 < nout) {
                            rd(std::integral_constant<int, m - 4>{}, std::integral_constant<int, (c + 1) % 9>{});
                            if constexpr (m == 11 && nout == 9) rd(std::integral_constant<int, 8>{}, std::integral_constant<int, (c + 1) % 9>{});
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    });
                });
                if constexpr (ABL & 128) {
                    f32x4 &c0 = acc[j][0], &c1 = acc[j][1];
                    asm volatile("" : "+v"(c0), "+v"(c1));
                    asm volatile("s_nop 0" ::: "memory");
                }
                MVS_LAP(5);
            });
            wsel ^= 1;
        }
        // ---- epilogue of the group: BN affine, ReLU, one 16-byte store per lane and row
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
            }
        });
    }
    if constexpr (ABL & 128) {
        MVS_LAP(6);
        if (lane == 0) {
            long long *dbg = reinterpret_cast<long long *>(const_cast<float *>(a.residual)) + ((int64_t)blockIdx.x * 8 + wv) * 8;
            for (int k = 0; k < 7; ++k) dbg[k] = tsum[k];
        }
    }
#undef MVS_LAP
}

// PyTorch-layout weight (8, Cin, 3, 3, 3) -> [chunk][kz*3+ky][split][lane][8 bf16]; lane (m, kq): row m =
// (cout = m & 7, x-shift = m >> 3), k = kq * 8 + c: x-tap kx' = kq of the 4-tap window, channel c of the
// chunk; the weight is w[cout][chunk*8 + c][kz][ky][kx' - shift] (zero outside the 3 real taps).
__global__ __launch_bounds__(256) void pack_bf16x6_kernel(const float *__restrict__ w, int Cin,
                                                          unsigned short *__restrict__ out, int total) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int j = i & 7, lane = (i >> 3) & 63, t = (i >> 9) % 9, ch = i / (9 * 512);
    const int m = lane & 15, kq = lane >> 4, co = m & 7, sft = m >> 3, kx = kq - sft;
    const int kz = t / 3, ky = t % 3, cin = ch * 8 + j;
    float x = 0.0f;
    if (kx >= 0 && kx <= 2) x = w[((int64_t)co * Cin + cin) * 27 + kz * 9 + ky * 3 + kx];
    const __bf16 h = (__bf16)x;
    const float r1 = x - (float)h;
    const __bf16 mm = (__bf16)r1;
    const float r2 = r1 - (float)mm;
    const __bf16 l = (__bf16)r2;
    unsigned short *o = out + ((size_t)(ch * 9 + t) * 3) * 512 + lane * 8 + j;
    o[0] = __builtin_bit_cast(unsigned short, h);
    o[512] = __builtin_bit_cast(unsigned short, mm);
    o[1024] = __builtin_bit_cast(unsigned short, l);
}

}  // namespace mvs

using namespace mvs;

static bool split_shape_ok(int Cin) { return Cin == 8 || Cin == 16 || Cin == 32; }

extern "C" size_t mvs_conv3d_bf16x6_packed_bytes(int Cin) {
    return split_shape_ok(Cin) ? (size_t)(Cin / 8) * kSplitChunkBytes : 0;
}

extern "C" int mvs_conv3d_pack_weights_bf16x6_f32(const float *weight, int Cin, void *packed, void *stream) {
    if (!weight || !packed || !split_shape_ok(Cin)) {
        set_error("mvs_conv3d_pack_weights_bf16x6_f32: needs a (8, Cin, 3, 3, 3) weight with Cin in {8, 16, 32}");
        return MVS_EINVAL;
    }
    const int total = (Cin / 8) * 9 * 512;
    hipLaunchKernelGGL(pack_bf16x6_kernel, dim3((total + 255) / 256), dim3(256), 0, as_stream(stream), weight, Cin,
                       static_cast<unsigned short *>(packed), total);
    return check_launch("mvs_conv3d_pack_weights_bf16x6_f32");
}

extern "C" int mvs_conv3d_c8_bf16x6_f32(const float *in, const void *packed, const float *scale,
                                        const float *shift, const float *residual, int relu, int B, int Cin,
                                        int D, int H, int W, float *out, void *stream) {
    if (!in || !packed || !out || B <= 0 || D <= 0 || H <= 0 || W <= 0 || !split_shape_ok(Cin)) {
        set_error("mvs_conv3d_c8_bf16x6_f32: invalid argument (Cin in {8, 16, 32}, Cout = 8, stride 1, 8-channel-blocked input)");
        return MVS_EINVAL;
    }
    if ((int64_t)9 * H * W * Cin * 4 >= 0xffffff00LL) return bare_error(MVS_EINVAL, __func__, __LINE__);
    ConvArgs a;
    a.in = in; a.wpk = static_cast<const float *>(packed); a.scale = scale; a.shift = shift; a.residual = residual;
    a.out = out;
    a.B = B; a.D = D; a.H = H; a.W = W;
    a.Do = D; a.Ho = H; a.Wo = W;
    a.tiles_x = (W + 31) / 32; a.tiles_y = (H + 3) / 4; a.tiles_z = (D + 3) / 4;
    a.relu = relu; a.in_c8 = 1; a.ystrip = 8;
#ifdef MVS_TUNING   // ablation / phase-stamp builds: wrong results by design, cycle counters written through `residual`
    static const int abl = [] { const char *e = getenv("MVS_CONV_SPLIT_ABL"); return e ? atoi(e) : 0; }();
#else
    constexpr int abl = 0;
#endif
    a.res_up2 = abl & 7;
    const int64_t nt = (int64_t)B * a.tiles_x * a.tiles_y * a.tiles_z;
    if (nt <= 0 || nt > 0x7fffffffLL) return bare_error(MVS_EINVAL, __func__, __LINE__);
    const int n_cu = device_cu_count();
    hipStream_t st = as_stream(stream);
    // MVS_CONV0_ZSLIDE=0: the per-tile kernel of round 2 (A/B; bit-identical results)
    static const bool zslide = [] { const char *e = getenv("MVS_CONV0_ZSLIDE"); return !(e && e[0] == '0'); }();
    if (zslide && !(abl & (128 | 24 | 7))) {
        const int64_t ng = (int64_t)B * a.tiles_x * a.tiles_y * ((a.tiles_z + kGroup - 1) / kGroup);
        const dim3 gridz((unsigned)(ng < n_cu ? ng : n_cu)), blkz(kSplitKernelThreads);
        if (Cin == 32) hipLaunchKernelGGL((conv3d_c8_bf16x6_zs_kernel<32>), gridz, blkz, 0, st, a, (int)ng);
        else if (Cin == 16) hipLaunchKernelGGL((conv3d_c8_bf16x6_zs_kernel<16>), gridz, blkz, 0, st, a, (int)ng);
        else hipLaunchKernelGGL((conv3d_c8_bf16x6_zs_kernel<8>), gridz, blkz, 0, st, a, (int)ng);
        return check_launch("mvs_conv3d_c8_bf16x6_f32");
    }
    if (zslide && (abl & 128) && Cin == 32 && (abl & 64)) {   // tuning build of the z-sliding kernel: phase cycles into `residual`
        if (!residual) return bare_error(MVS_EINVAL, __func__, __LINE__);
        const int64_t ng = (int64_t)B * a.tiles_x * a.tiles_y * ((a.tiles_z + kGroup - 1) / kGroup);
        const dim3 gridz((unsigned)(ng < n_cu ? ng : n_cu)), blkz(kSplitKernelThreads);
        hipLaunchKernelGGL((conv3d_c8_bf16x6_zs_kernel<32, 128>), gridz, blkz, 0, st, a, (int)ng);
        return check_launch("mvs_conv3d_c8_bf16x6_f32");
    }
    const dim3 grid((unsigned)(nt < n_cu ? nt : n_cu)), blk(kSplitKernelThreads);
    // MVS_CONV_SPLIT_DOT2=0: the split written with plain conversions and subtractions (same operands bit for
    // bit; kept to cross-check the v_dot2c form)
    static const bool dot2 = [] { const char *e = getenv("MVS_CONV_SPLIT_DOT2"); return !(e && e[0] == '0'); }();
#define MVS_LAUNCH_SPLIT(C)                                                                                    \
    do {                                                                                                       \
        if (dot2) hipLaunchKernelGGL((conv3d_c8_bf16x6_kernel<C, true>), grid, blk, 0, st, a, (int)nt);        \
        else hipLaunchKernelGGL((conv3d_c8_bf16x6_kernel<C, false>), grid, blk, 0, st, a, (int)nt);           \
    } while (0)
    if (Cin == 32 && (abl & 128)) {   // tuning build: per-phase cycle counts into `residual` (scripts/exp_conv_split_time.py)
        if (!residual) return bare_error(MVS_EINVAL, __func__, __LINE__);
        hipLaunchKernelGGL((conv3d_c8_bf16x6_kernel<32, true, 128>), grid, blk, 0, st, a, (int)nt);
    } else if (Cin == 32 && (abl & 24)) {   // tuning builds: 8 = no split, 16 = no MFMAs (wrong results)
        if ((abl & 24) == 8) hipLaunchKernelGGL((conv3d_c8_bf16x6_kernel<32, true, 8>), grid, blk, 0, st, a, (int)nt);
        else if ((abl & 24) == 16) hipLaunchKernelGGL((conv3d_c8_bf16x6_kernel<32, true, 16>), grid, blk, 0, st, a, (int)nt);
        else hipLaunchKernelGGL((conv3d_c8_bf16x6_kernel<32, true, 24>), grid, blk, 0, st, a, (int)nt);
    } else if (Cin == 32) MVS_LAUNCH_SPLIT(32);
    else if (Cin == 16) MVS_LAUNCH_SPLIT(16);
    else MVS_LAUNCH_SPLIT(8);
#undef MVS_LAUNCH_SPLIT
    return check_launch("mvs_conv3d_c8_bf16x6_f32");
}
