// FeatureNet's first two layers in one kernel: conv0 (3 -> 8, 3x3) + BN + ReLU and conv1 (8 -> 8, 3x3) + BN + ReLU
// (MVSNet/models/mvsnet.py:11-12,33-34; the same head opens CasMVSNet's FeatureNet, CasMVSNet/models/module.py:318-321).
//
// As two launches the pair moves 0.30 GB of conv0 output out to HBM and back in (1600 x 1184, 5 views) for 0.11 GB of
// image in and 0.30 GB out.  Here a persistent workgroup (4 waves, two workgroups per CU) walks 32 x 16-pixel tiles:
//   copy      the image halo of the NEXT tile, 3 planes x 20 rows x 40 columns, by LDS-DMA (double-buffered; zero
//             fill outside the image by the buffer range check = conv0's zero padding)
//   conv0     (round 5) on the BF16 matrix pipe too, exact three-piece operands: the image tile is split once into pixel-major
//             planes [row][column][3 channels + a zero] of bf16 pieces, so that 8 consecutive K slots = two neighbouring
//             columns x 4 channels are ONE ds_read_b128; MFMA rows = (x-shift, 8 output channels) against a 4-column window
//             (the Cout = 8 shifted form), K = 3 kernel rows x 4 window columns x 4 channels = 48 of 64 slots in two K-steps,
//             N = 16 column pairs of one row; 18 rows + 2 items for the 17th column pair (N = rows there) = 20 items of 12
//             MFMAs per tile.  BN + ReLU, every result split exactly into three bf16 numbers (conv_split_common.h) and
//             written -- zeros outside the image: conv1 pads conv0's OUTPUT -- as the operand planes of conv1's matrix
//             stream.  (Rounds 2-4 ran conv0 on the vector ALU: 216 FMAs per pixel, ~1190 instructions per wave and tile,
//             0.09-0.11 ms of the kernel's 0.21; now 0.06 of 0.185: items in pairs -- four accumulators in
//             flight, one 8-value split for both -- with the next pair's MFMAs issued before a pair's epilogue.)
//   conv1     on the BF16 matrix pipe at fp32 accuracy (six partial products per fp32 product): the Cout = 8 "shifted"
//             form -- MFMA rows = channel x x-shift, 16 even-x pixels per column group -- with K = 4 x-taps x 8
//             channels, i.e. one v_mfma_f32_16x16x32_bf16 per kernel row and product
// The first version multiplied conv1 in fp32 (v_mfma_f32_16x16x4_f32): its phases ADDED UP (copies + stores 0.10, conv0
// 0.06-0.09, conv1 0.09 ms of 0.22-0.25), because beside an fp32 MFMA stream the other wave of a SIMD issues no vector
// instruction at all (scripts/micro/coissue.hip); beside bf16 MFMAs it keeps two thirds of its rate, and the bf16 stream
// is 2.7x shorter.  The two launches this replaces take 0.42 ms.  MVS_HEAD_ABL (1 no conv0, 2 no conv1, 4 no stores,
// 8 no copies) is the tuning switch.
#include "conv_split_common.h"

#include <cstdlib>

namespace mvs {

namespace {
constexpr int kTH = 16;                                  // tile rows (x: 32)
constexpr int kHR = kTH + 2, kHT = 34, kHXH = 17;        // conv1's halo of a tile; half of its width (x de-interleaved)
constexpr int kBPartBytes = kHR * kHT * 16;              // one bf16 part of the halo: 8 channels = 16 bytes per pixel
constexpr int kIR = kTH + 4, kIP = 40;                   // image tile: rows y0-2 .. y0+17, columns x0-4 .. x0+35
constexpr int kNPos = kIR * (kIP / 4);                   // (row, 4-column group) positions of the image tile: 200
constexpr int kHeadWaves = 4, kHeadThreads = 256, kRPW = kTH / kHeadWaves;
// a wave copies -- and later splits -- its OWN 64 positions, all three channels: [wave][channel][lane][4 floats]
constexpr int kImgFloats = kHeadWaves * 3 * 256;
static_assert(kHeadWaves * 64 >= kNPos, "positions per wave");
constexpr int kPRowBytes = kIP * 8;                      // image pieces: [row][column][4 bf16] -- one row
constexpr int kPPartBytes = (kIR + 1) * kPRowBytes;      // ... one piece plane (+ a zero row: kernel row 3 of the second K-step)
constexpr int kW1Bytes = 3 * 3 * 1024;                   // conv1 A fragments: [kernel row][part][lane][8 bf16]
constexpr int kW0Floats = 16;                            // conv0's scale, shift
constexpr int kItems = kHR + 2, kIPW0 = kItems / kHeadWaves;   // conv0: 18 row items + 2 column items, five per wave
static_assert(kItems % kHeadWaves == 0, "items per wave");
constexpr int kW1Off = 0, kBOff = kW1Off + kW1Bytes / 4, kImgOff = kBOff + 3 * kBPartBytes / 4;
constexpr int kPOff = kImgOff + kImgFloats, kW0Off = kPOff + 3 * kPPartBytes / 4, kHeadLdsFloats = kW0Off + kW0Floats;
static_assert(2 * (kHeadLdsFloats * 4 + 512) <= 160 * 1024, "two workgroups per CU");
}  // namespace

struct HeadArgs {
    const float *img;      // [N,3,H,W]
    const float *w0, *scale0, *shift0;   // conv0: PyTorch layout (8,3,3,3); BN affine
    const unsigned char *wpk1;           // conv1: mvs_feature_head_pack_weights_f32
    const float *scale1, *shift1;
    float *out;            // [N,H,W,8]
    int N, H, W, tiles_x, tiles_y, ystrip;
    int abl;               // tuning: 1 no conv0, 2 no conv1, 4 no stores, 8 no copies (results are garbage)
    unsigned *out_absmax;  // NULL, or the absmax block (mvs_common.h) the largest magnitude stored is max-ed into
};

__global__ __launch_bounds__(kHeadThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void feature_head_kernel(HeadArgs a, int ntiles) {
    __shared__ __attribute__((aligned(16))) float lds[kHeadLdsFloats];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4;
    const unsigned lds_base = (unsigned)(uintptr_t)lds;

    int t_cur, t_end, t_step;
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
    // once: conv1's A fragments, conv0's affine, zeros in the image pieces (their pad row stays zero)
    for (int i = tid; i < kW1Bytes / 16; i += kHeadThreads)
        reinterpret_cast<float4 *>(lds + kW1Off)[i] = reinterpret_cast<const float4 *>(a.wpk1)[i];
    if (tid < kW0Floats) lds[kW0Off + tid] = tid < 8 ? (a.scale0 ? a.scale0[tid] : 1.0f) : (a.shift0 ? a.shift0[tid - 8] : 0.0f);
    for (int i = tid; i < 3 * kPPartBytes / 16; i += kHeadThreads) reinterpret_cast<float4 *>(lds + kPOff)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    // conv0's A fragments in registers: lane (m, kq), K-step s, element i: K slot 8 s + 2 kq + (i >> 2) = (kernel row ky = slot >> 2,
    // window column kxw = slot & 3), channel i & 3; MFMA row m = (x-shift m >> 3, output channel m & 7): w0[co][c][ky][kxw - shift]
    bf16x8 A0[2][3];
    {
        const int m = lane & 15, co = m & 7, sft = m >> 3;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            f32x4 x0, x1;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int slot = 8 * st + 2 * kq + (i >> 2), ky = slot >> 2, kx = (slot & 3) - sft, c = i & 3;
                const float v = (c < 3 && ky < 3 && kx >= 0 && kx < 3) ? a.w0[((co * 3 + c) * 3 + ky) * 3 + kx] : 0.0f;
                if (i < 4) x0[i] = v; else x1[i - 4] = v;
            }
            split3_block(x0, x1, A0[st][0], A0[st][1], A0[st][2]);
        }
    }

    // tile-independent coordinates of this wave's copies: position p = 64 wv + lane = (row, 4-column group), channels 0..2
    const int pos = wv * 64 + lane;
    const bool pos_ok = pos < kNPos;
    const int prow = pos_ok ? pos / (kIP / 4) : 0, pgrp = pos_ok ? pos % (kIP / 4) : 0;
    const int plane = a.H * a.W;
    struct Tile { int tx, ty, b; };
    auto decode = [&](int t) {
        Tile r;
        const int per_b = a.tiles_x * a.tiles_y;
        r.b = t / per_b; t -= r.b * per_b;
        const int full = a.ystrip * a.tiles_x;
        const int s = t / full; t -= s * full;
        const int y0 = s * a.ystrip, hs = min(a.ystrip, a.tiles_y - y0);
        r.ty = y0 + t % hs;
        r.tx = t / hs;
        return r;
    };
    Tile nxt = {0, 0, 0};
    auto issue = [&](int t) {
        nxt = decode(t);
        if (a.abl & 8) return;
        const mvs_srd_t srd = make_srd(a.img + (int64_t)nxt.b * 3 * plane, (unsigned)(3 * plane) * 4u);
        const int gx = nxt.tx * 32 - 4 + pgrp * 4, gy = nxt.ty * kTH - 2 + prow;
        const bool ok = pos_ok && (unsigned)gx < (unsigned)a.W && (unsigned)gy < (unsigned)a.H;
        const unsigned base = lds_base + (unsigned)(kImgOff * 4 + wv * 3 * 1024);
#pragma unroll
        for (int c = 0; c < 3; ++c)
            glds16_buf(ok ? (unsigned)((c * plane + gy * a.W + gx) * 4) : 0xffffff00u, srd, 0u, base + (unsigned)c * 1024u);
    };
    // this wave's positions of the staged image -> the three piece planes: 4 pixels x (3 channels + 0) per lane
    auto split_own = [&]() {
        if (!pos_ok) return;
        const float *im = lds + kImgOff + wv * 3 * 256 + lane * 4;
        const float4 c0v = *reinterpret_cast<const float4 *>(im), c1v = *reinterpret_cast<const float4 *>(im + 256),
                     c2v = *reinterpret_cast<const float4 *>(im + 512);
        const unsigned dst = lds_base + (unsigned)(kPOff * 4 + prow * kPRowBytes + pgrp * 32);
        f32x4 p0 = {c0v.x, c1v.x, c2v.x, 0.f}, p1 = {c0v.y, c1v.y, c2v.y, 0.f};
        f32x4 p2 = {c0v.z, c1v.z, c2v.z, 0.f}, p3 = {c0v.w, c1v.w, c2v.w, 0.f};
        bf16x8 h, m, l;
        split3_block(p0, p1, h, m, l);
        lds_write_b128<0>(dst, h); lds_write_b128<kPPartBytes>(dst, m); lds_write_b128<2 * kPPartBytes>(dst, l);
        split3_block(p2, p3, h, m, l);
        lds_write_b128<16>(dst, h); lds_write_b128<16 + kPPartBytes>(dst, m); lds_write_b128<16 + 2 * kPPartBytes>(dst, l);
        lds_wait_n<0>();
    };

    // conv1's BN affine of this lane's four channels
    const int c0 = (kq & 1) * 4;
    const float4 sc1 = a.scale1 ? *reinterpret_cast<const float4 *>(a.scale1 + c0) : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 sh1 = a.shift1 ? *reinterpret_cast<const float4 *>(a.shift1 + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
    // conv1 operand addresses: A = [kernel row][part][lane]; B = this lane's pixel x = 2 n + kq (x de-interleaved) of the
    // wave's first row, every part; rows, kernel rows and parts are compile-time offsets
    const unsigned aA = lds_base + (unsigned)(kW1Off * 4 + lane * 16);
    const unsigned aB = lds_base + (unsigned)(kBOff * 4 + ((wv * kRPW) * kHT + (kq & 1) * kHXH + (kq >> 1) + n) * 16);

    if (t_cur < t_end) {
        issue(t_cur);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();         // the zeroed piece planes, conv1's fragments, the affine
    if (t_cur < t_end) split_own();
    float vmax = 0.0f;
    while (t_cur < t_end) {
        const Tile cur = nxt;
        const int t_next = t_cur + t_step;
        __syncthreads();     // A: the image pieces of this tile are complete; every wave is done with conv1 of the previous tile
        if (t_next < t_end) issue(t_next);       // (each wave's staging slots are its own: free since its split)

        // ---- conv0 + BN + ReLU -> conv1's operand planes
        if (!(a.abl & 1)) {
            const float4 sc0 = *reinterpret_cast<const float4 *>(lds + kW0Off + c0), sh0 = *reinterpret_cast<const float4 *>(lds + kW0Off + 8 + c0);
            const int sft = kq >> 1;
            // items in pairs (four independent accumulators in flight; one 8-value split serves both); the reads of the next
            // pair go out before the MFMAs of this one
            int rowk[kIPW0 + 1], pairk[kIPW0 + 1];
            unsigned adk[kIPW0 + 1];
#pragma unroll
            for (int k = 0; k < kIPW0 + 1; ++k) {
                const int it = wv + kHeadWaves * (k < kIPW0 ? k : kIPW0 - 1);   // wave-uniform (the odd item out is paired with itself)
                // row items: row = it, column pair = n; column items (the 17th pair): rows 0..15 / 2..17, pair 16
                rowk[k] = it < kHR ? it : (it - kHR) * 2 + n;
                pairk[k] = it < kHR ? n : 16;
                adk[k] = lds_base + (unsigned)(kPOff * 4 + (rowk[k] + (kq >> 1)) * kPRowBytes + (2 * pairk[k] + 2 + 2 * (kq & 1)) * 8);
            }
            bf16x8 B0[2][2][2][3];      // [pair parity][item of the pair][K-step][part]
            auto read_pair = [&](auto qc) {
                constexpr int q = decltype(qc)::value;
                static_for<0, 2>([&](auto ic) {
                    constexpr int i = decltype(ic)::value, k = 2 * q + i;
                    if constexpr (k < kIPW0) {
                        static_for<0, 2>([&](auto sc_) {
                            constexpr int st = decltype(sc_)::value;
                            static_for<0, 3>([&](auto pc) {
                                constexpr int p = decltype(pc)::value;
                                B0[q & 1][i][st][p] = __builtin_bit_cast(bf16x8, lds_read_b128<st * 2 * kPRowBytes + p * kPPartBytes>(adk[k]));
                            });
                        });
                    }
                });
            };
            constexpr int NPAIR = (kIPW0 + 1) / 2;
            f32x4 acc[2][2][2];       // [pair parity][item][K-step]: per item two accumulators, six partial products each
            auto mfma_pair = [&](auto qc) {
                constexpr int q = decltype(qc)::value;
                constexpr bool two = 2 * q + 1 < kIPW0;
                lds_wait_n<0>();
                static_for<0, 2>([&](auto sc_) {
                    constexpr int st = decltype(sc_)::value;
                    asm volatile("" : "+v"(B0[q & 1][0][st][0]), "+v"(B0[q & 1][0][st][1]), "+v"(B0[q & 1][0][st][2]));
                    if constexpr (two) asm volatile("" : "+v"(B0[q & 1][1][st][0]), "+v"(B0[q & 1][1][st][1]), "+v"(B0[q & 1][1][st][2]));
                });
                if constexpr (q + 1 < NPAIR) read_pair(std::integral_constant<int, q + 1>{});
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int st = 0; st < 2; ++st) acc[q & 1][i][st] = (f32x4){0.f, 0.f, 0.f, 0.f};
                constexpr int PA[6] = {1, 2, 0, 1, 0, 0}, PB[6] = {1, 0, 2, 0, 1, 0};      // small terms first
                static_for<0, 6>([&](auto pc) {
                    constexpr int pr = decltype(pc)::value;
                    static_for<0, two ? 2 : 1>([&](auto ic) {
                        constexpr int i = decltype(ic)::value;
                        acc[q & 1][i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A0[0][PA[pr]], B0[q & 1][i][0][PB[pr]], acc[q & 1][i][0], 0, 0, 0);
                        acc[q & 1][i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A0[1][PA[pr]], B0[q & 1][i][1][PB[pr]], acc[q & 1][i][1], 0, 0, 0);
                    });
                });
                __builtin_amdgcn_sched_barrier(0);
            };
            // this lane: 4 output channels of pixel (row, column 2 pair + shift) of the 18 x 34 region, per item
            auto epilogue_pair = [&](auto qc) {
                constexpr int q = decltype(qc)::value, k0 = 2 * q;
                constexpr bool two = 2 * q + 1 < kIPW0;
                f32x4 v[2];
                unsigned bw[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int k = (i == 1 && !two) ? k0 : k0 + i;
                    const int row = rowk[k], hcol = 2 * pairk[k] + sft;
                    const int gx = cur.tx * 32 - 1 + hcol, gy = cur.ty * kTH - 1 + row;
                    const bool in_img = (unsigned)gx < (unsigned)a.W && (unsigned)gy < (unsigned)a.H;
                    const f32x4 e = acc[q & 1][i][0], o = acc[q & 1][i][1];
                    v[i][0] = relu_nan(fmaf(e[0] + o[0], sc0.x, sh0.x)); v[i][1] = relu_nan(fmaf(e[1] + o[1], sc0.y, sh0.y));
                    v[i][2] = relu_nan(fmaf(e[2] + o[2], sc0.z, sh0.z)); v[i][3] = relu_nan(fmaf(e[3] + o[3], sc0.w, sh0.w));
                    if (!in_img || (i == 1 && !two)) v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    const int hxd = (hcol & 1) ? kHXH + (hcol >> 1) : (hcol >> 1);
                    bw[i] = lds_base + (unsigned)(kBOff * 4 + (row * kHT + hxd) * 16 + (kq & 1) * 8);
                }
                bf16x8 ph, pm, pl;
                split3_block(v[0], v[1], ph, pm, pl);
                const u32x4 hu = __builtin_bit_cast(u32x4, ph), mu = __builtin_bit_cast(u32x4, pm), lu = __builtin_bit_cast(u32x4, pl);
                lds_write_b64<0>(bw[0], hu[0], hu[1]);
                lds_write_b64<kBPartBytes>(bw[0], mu[0], mu[1]);
                lds_write_b64<2 * kBPartBytes>(bw[0], lu[0], lu[1]);
                if constexpr (two) {
                    lds_write_b64<0>(bw[1], hu[2], hu[3]);
                    lds_write_b64<kBPartBytes>(bw[1], mu[2], mu[3]);
                    lds_write_b64<2 * kBPartBytes>(bw[1], lu[2], lu[3]);
                }
            };
            // the MFMAs of pair q + 1 are issued before the vector work of pair q's epilogue: it runs in their shadow
            read_pair(std::integral_constant<int, 0>{});
            mfma_pair(std::integral_constant<int, 0>{});
            static_for<1, NPAIR>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                mfma_pair(std::integral_constant<int, q>{});
                epilogue_pair(std::integral_constant<int, q - 1>{});
            });
            epilogue_pair(std::integral_constant<int, NPAIR - 1>{});
            lds_wait_n<0>();
        }
        __syncthreads();     // B: conv1's operand planes are complete; the image pieces are free

        // ---- conv1: items (kernel row ky, row r): three B reads (one per part) one item ahead of its six MFMAs; the A
        // fragments of a kernel row (three reads) go out with the last item of the row before
        f32x4 acc1[kRPW];
#pragma unroll
        for (int r = 0; r < kRPW; ++r) acc1[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (!(a.abl & 2)) {
            bf16x8 A[2][3], Bf[2][3];
            auto read_a = [&](auto kc) {
                constexpr int ky = decltype(kc)::value;
                static_for<0, 3>([&](auto pc) {
                    constexpr int p = decltype(pc)::value;
                    A[ky & 1][p] = __builtin_bit_cast(bf16x8, lds_read_b128<(ky * 3 + p) * 1024>(aA));
                });
            };
            auto read_b = [&](auto ic) {
                constexpr int it = decltype(ic)::value, ky = it / kRPW, r = it % kRPW;
                static_for<0, 3>([&](auto pc) {
                    constexpr int p = decltype(pc)::value;
                    Bf[it & 1][p] = __builtin_bit_cast(bf16x8, lds_read_b128<(r + ky) * kHT * 16 + p * kBPartBytes>(aB));
                });
            };
            read_a(std::integral_constant<int, 0>{});
            read_b(std::integral_constant<int, 0>{});
            static_for<0, 3 * kRPW>([&](auto ic) {
                constexpr int it = decltype(ic)::value, ky = it / kRPW, r = it % kRPW;
                lds_wait_n<0>();
                {
                    bf16x8 &b0 = Bf[it & 1][0], &b1 = Bf[it & 1][1], &b2 = Bf[it & 1][2];
                    asm volatile("" : "+v"(b0), "+v"(b1), "+v"(b2));
                }
                if constexpr (r == 0) {
                    bf16x8 &a0 = A[ky & 1][0], &a1 = A[ky & 1][1], &a2 = A[ky & 1][2];
                    asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2));
                }
                if constexpr (it + 1 < 3 * kRPW) {
                    if constexpr (r == kRPW - 1) read_a(std::integral_constant<int, ky + 1>{});
                    read_b(std::integral_constant<int, it + 1>{});
                }
                __builtin_amdgcn_sched_barrier(0);   // the reads go out BEFORE this item's MFMAs
                const bf16x8 ah = A[ky & 1][0], am = A[ky & 1][1], al = A[ky & 1][2];
                const bf16x8 bh = Bf[it & 1][0], bm = Bf[it & 1][1], bl = Bf[it & 1][2];
                f32x4 &cc = acc1[r];
                // six partial products, small terms first (as conv_split.hip)
                cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, cc, 0, 0, 0);
                cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, cc, 0, 0, 0);
                cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, cc, 0, 0, 0);
                cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, cc, 0, 0, 0);
                cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, cc, 0, 0, 0);
                cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, cc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        // ---- the next tile's image (this wave's own copies) has had conv0 and conv1 to land: split it into the piece planes
        if (t_next < t_end) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (!(a.abl & 8)) split_own();
        }
        // ---- epilogue: BN affine, ReLU, one 16-byte store per lane and row
        {
            const int ox = cur.tx * 32 + 2 * n + (kq >> 1);
#pragma unroll
            for (int r = 0; r < kRPW; ++r) {
                const int oy = cur.ty * kTH + wv * kRPW + r;
                if (oy >= a.H || ox >= a.W || (a.abl & 4)) continue;
                f32x4 v = acc1[r];
                v[0] = relu_nan(v[0] * sc1.x + sh1.x); v[1] = relu_nan(v[1] * sc1.y + sh1.y);
                v[2] = relu_nan(v[2] * sc1.z + sh1.z); v[3] = relu_nan(v[3] * sc1.w + sh1.w);
                const int64_t o = (((int64_t)cur.b * a.H + oy) * a.W + ox) * 8 + c0;
                *reinterpret_cast<float4 *>(a.out + o) = make_float4(v[0], v[1], v[2], v[3]);
                vmax = max_nan(max_nan(max_nan(vmax, v[0]), v[1]), max_nan(v[2], v[3]));       // (after the ReLU: non-negative)
            }
        }
        t_cur = t_next;
    }
    publish_absmax(a.out_absmax, vmax);
}

// conv1's weight (8, 8, 3, 3) -> [kernel row ky][part][lane][8 bf16]: lane (m, kq) = MFMA row m = (x-shift m >> 3, channel
// m & 7), K slots kq * 8 + j = (x-tap kq, input channel j): w[m & 7][j][ky][kq - shift] (zero outside 0..2), split exactly
__global__ __launch_bounds__(256) void feature_head_pack_kernel(const float *__restrict__ w, unsigned short *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 3 * 64 * 8) return;
    const int j = i & 7, lane = (i >> 3) & 63, ky = i >> 9;
    const int m = lane & 15, kq = lane >> 4, co = m & 7, kxr = kq - (m >> 3);
    const float x = (kxr >= 0 && kxr < 3) ? w[((co * 8 + j) * 3 + ky) * 3 + kxr] : 0.0f;
    const __bf16 h = (__bf16)x;
    const float r1 = x - (float)h;
    const __bf16 mm = (__bf16)r1;
    const float r2 = r1 - (float)mm;
    const __bf16 l = (__bf16)r2;
    unsigned short *o = out + (size_t)(ky * 3) * 512 + lane * 8 + j;
    o[0] = __builtin_bit_cast(unsigned short, h);
    o[512] = __builtin_bit_cast(unsigned short, mm);
    o[1024] = __builtin_bit_cast(unsigned short, l);
}

}  // namespace mvs

using namespace mvs;

extern "C" int mvs_feature_head_supported(int H, int W) {
    return H > 0 && W > 0 && (W & 3) == 0 && (int64_t)H * W * 12 < 0xffffff00LL;
}

extern "C" size_t mvs_feature_head_packed_bytes(void) { return (size_t)kW1Bytes; }

extern "C" int mvs_feature_head_pack_weights_f32(const float *weight1, void *packed, void *stream) {
    if (!weight1 || !packed) return bare_error(MVS_EINVAL, __func__, __LINE__);
    hipLaunchKernelGGL(feature_head_pack_kernel, dim3(6), dim3(256), 0, (hipStream_t)stream, weight1,
                       static_cast<unsigned short *>(packed));
    return check_launch("mvs_feature_head_pack_weights_f32");
}

extern "C" int mvs_feature_head_f32(const float *img, const float *w0, const float *scale0, const float *shift0,
                                    const void *packed1, const float *scale1, const float *shift1, int N, int H,
                                    int W, float *out, void *stream) {
    return mvs_feature_head_absmax_f32(img, w0, scale0, shift0, packed1, scale1, shift1, N, H, W, out, nullptr, stream);
}

extern "C" int mvs_feature_head_absmax_f32(const float *img, const float *w0, const float *scale0, const float *shift0,
                                           const void *packed1, const float *scale1, const float *shift1, int N, int H,
                                           int W, float *out, void *out_absmax, void *stream) {
    if (!img || !w0 || !packed1 || !out || N <= 0) return bare_error(MVS_EINVAL, __func__, __LINE__);
    if (!mvs_feature_head_supported(H, W)) {
        set_error("mvs_feature_head_f32: needs W %% 4 == 0 and an image below 4 GiB (H=%d W=%d)", H, W);
        return MVS_EUNSUPPORTED;
    }
    HeadArgs a;
    a.img = img; a.w0 = w0; a.scale0 = scale0; a.shift0 = shift0;
    a.wpk1 = static_cast<const unsigned char *>(packed1); a.scale1 = scale1; a.shift1 = shift1; a.out = out;
    a.out_absmax = static_cast<unsigned *>(out_absmax);
    a.N = N; a.H = H; a.W = W;
    a.tiles_x = (W + 31) / 32; a.tiles_y = (H + kTH - 1) / kTH; a.ystrip = 8;
#ifdef MVS_TUNING
    static const int abl = getenv("MVS_HEAD_ABL") ? atoi(getenv("MVS_HEAD_ABL")) : 0;
#else
    constexpr int abl = 0;
#endif
    a.abl = abl;
    const int64_t nt = (int64_t)a.tiles_x * a.tiles_y * N;
    if (nt > 0x7fffffffLL) return bare_error(MVS_EINVAL, __func__, __LINE__);
    const int slots = 2 * device_cu_count();
    hipLaunchKernelGGL(feature_head_kernel, dim3((unsigned)(nt < slots ? nt : slots)), dim3(kHeadThreads), 0,
                       (hipStream_t)stream, a, (int)nt);
    return check_launch("mvs_feature_head_f32");
}
