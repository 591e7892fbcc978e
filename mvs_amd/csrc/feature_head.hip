// FeatureNet's first two layers in one kernel: conv0 (3 -> 8, 3x3) + BN + ReLU and conv1 (8 -> 8, 3x3) + BN + ReLU
// (MVSNet/models/mvsnet.py:11-12,33-34; the same head opens CasMVSNet's FeatureNet, CasMVSNet/models/module.py:318-321).
//
// As two launches the pair moves 0.30 GB of conv0 output out to HBM and back in (1600 x 1184, 5 views) for 0.11 GB of
// image in and 0.30 GB out.  Here a persistent workgroup (4 waves, two workgroups per CU) walks 32 x 16-pixel tiles:
//   copy      the image halo of the NEXT tile, 3 planes x 20 rows x 40 columns, by LDS-DMA (double-buffered; zero
//             fill outside the image by the buffer range check = conv0's zero padding)
//   conv0     on the vector ALU (216 FMAs per pixel cannot fill a matrix tile): a thread owns three consecutive rows of
//             one column of the 18 x 34 region conv1 needs, keeps its 5 x 3 x 3 inputs in registers, reads the 27 x 8
//             weights as broadcast ds_read_b128, applies BN + ReLU, splits every result exactly into three bf16 numbers
//             (conv_split_common.h) and writes them -- zeros outside the image: conv1 pads conv0's OUTPUT -- as the
//             operand planes of conv1's matrix stream
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
constexpr int kImgGran = 3 * kIR * kIP / 4;              // 16-byte pieces: 600
constexpr int kImgDma = (kImgGran + 63) / 64;            // 10 wave instructions
constexpr int kImgFloats = kImgDma * 256;
constexpr int kW1Bytes = 3 * 3 * 1024;                   // conv1 A fragments: [kernel row][part][lane][8 bf16]
constexpr int kW0Floats = 27 * 8 + 16;                   // conv0 [tap][cout], scale, shift
constexpr int kHeadWaves = 4, kHeadThreads = 256, kRPW = kTH / kHeadWaves;
constexpr int kRG = 3, kNRG = kHR / kRG;                 // conv0: rows per thread, row groups
constexpr int kW1Off = 0, kBOff = kW1Off + kW1Bytes / 4, kImgOff = kBOff + 3 * kBPartBytes / 4;
constexpr int kW0Off = kImgOff + 2 * kImgFloats, kHeadLdsFloats = kW0Off + kW0Floats;
static_assert(2 * (kHeadLdsFloats * 4 + 512) <= 160 * 1024 && kNRG * kHT <= kHeadThreads, "two workgroups per CU");
constexpr int kIPW = (kImgDma + kHeadWaves - 1) / kHeadWaves;   // copies per wave
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
    // once: conv1's A fragments, conv0's weights and affine
    for (int i = tid; i < kW1Bytes / 16; i += kHeadThreads)
        reinterpret_cast<float4 *>(lds + kW1Off)[i] = reinterpret_cast<const float4 *>(a.wpk1)[i];
    for (int i = tid; i < kW0Floats; i += kHeadThreads) {
        float v;
        if (i < 216) v = a.w0[(i & 7) * 27 + (i >> 3)];
        else if (i < 224) v = a.scale0 ? a.scale0[i - 216] : 1.0f;
        else v = a.shift0 ? a.shift0[i - 224] : 0.0f;
        lds[kW0Off + i] = v;
    }

    // tile-independent coordinates of this wave's copies: piece q = (channel, row, 4-column group)
    int loc[kIPW];
#pragma unroll
    for (int i = 0; i < kIPW; ++i) {
        const int q = (i * kHeadWaves + wv) * 64 + lane;
        const int qc = min(q, kImgGran - 1);
        const int c = qc / (kIR * 10), r = (qc / 10) % kIR, g = qc % 10;
        loc[i] = g | (r << 8) | (c << 16) | (q < kImgGran ? 0 : (int)0x80000000);
    }
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
    auto issue = [&](int t, int parity) {
        nxt = decode(t);
        const mvs_srd_t srd = make_srd(a.img + (int64_t)nxt.b * 3 * plane, (unsigned)(3 * plane) * 4u);
        const int gx0 = nxt.tx * 32 - 4, gy0 = nxt.ty * kTH - 2;
        const unsigned base = lds_base + (unsigned)(kImgOff + parity * kImgFloats) * 4u;
#pragma unroll
        for (int i = 0; i < kIPW; ++i) {
            if (i * kHeadWaves + wv >= kImgDma) continue;   // wave-uniform
            const int gx = gx0 + (loc[i] & 255) * 4, gy = gy0 + ((loc[i] >> 8) & 255), c = (loc[i] >> 16) & 3;
            const bool ok = loc[i] >= 0 && (unsigned)gx < (unsigned)a.W && (unsigned)gy < (unsigned)a.H;
            const unsigned voff = ok ? (unsigned)((c * plane + gy * a.W + gx) * 4) : 0xffffff00u;
            glds16_buf(voff, srd, 0u, base + (unsigned)(i * kHeadWaves + wv) * 1024u);
        }
    };

    // conv1's BN affine of this lane's four channels
    const int c0 = (kq & 1) * 4;
    const float4 sc1 = a.scale1 ? *reinterpret_cast<const float4 *>(a.scale1 + c0) : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 sh1 = a.shift1 ? *reinterpret_cast<const float4 *>(a.shift1 + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
    // conv1 operand addresses: A = [kernel row][part][lane]; B = this lane's pixel x = 2 n + kq (x de-interleaved) of the
    // wave's first row, every part; rows, kernel rows and parts are compile-time offsets
    const unsigned aA = lds_base + (unsigned)(kW1Off * 4 + lane * 16);
    const unsigned aB = lds_base + (unsigned)(kBOff * 4 + ((wv * kRPW) * kHT + (kq & 1) * kHXH + (kq >> 1) + n) * 16);
    // conv0: this thread's column and first row of the 18 x 34 region
    const int hcol = tid % kHT, hrow0 = (tid / kHT) * kRG;
    const bool conv0_thread = tid < kNRG * kHT;
    const int hxd = (hcol & 1) ? kHXH + (hcol >> 1) : (hcol >> 1);

    int parity = 0;
    bool full_stores = false;   // did this wave issue exactly kRPW stores after its last copies?
    if (t_cur < t_end) issue(t_cur, 0);
    float vmax = 0.0f;
    while (t_cur < t_end) {
        const Tile cur = nxt;
        const int t_next = t_cur + t_step;
        // The image tile was requested before the previous tile's stores; vector memory retires in order, so with
        // a full set of kRPW stores behind the copies a counted wait leaves the HBM write latency out of the tile.
        if (full_stores) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kRPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // the image tile has landed; every wave is done with conv1 of the previous tile
        if (t_next < t_end && !(a.abl & 8)) issue(t_next, parity ^ 1);
        else if (t_next < t_end) nxt = decode(t_next);

        // ---- conv0 + BN + ReLU -> conv1's operand planes
        if (conv0_thread && !(a.abl & 1)) {
            const float *im = lds + kImgOff + parity * kImgFloats + hrow0 * kIP + hcol + 2;
            const float *wl = lds + kW0Off;
            float acc[kRG][8];
#pragma unroll
            for (int r = 0; r < kRG; ++r)
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[r][c] = 0.0f;
            // weights of one (channel, kernel row) = 3 taps x 8 output channels, fetched one group ahead; the
            // scheduler is fenced per group (left alone it hoists all 54 weight reads: 330 registers, one
            // workgroup per CU)
            float4 wq[2][6];
            auto load_w = [&](int slot, int g) {
#pragma unroll
                for (int i = 0; i < 6; ++i) wq[slot][i] = *reinterpret_cast<const float4 *>(wl + g * 24 + i * 4);
            };
            load_w(0, 0);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float in[kRG + 2][3];
#pragma unroll
                for (int r = 0; r < kRG + 2; ++r)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) in[r][kx] = im[(c * kIR + r) * kIP + kx];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int g = c * 3 + ky;
                    if (g + 1 < 9) load_w((g + 1) & 1, g + 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const float4 wa = wq[g & 1][kx * 2], wb = wq[g & 1][kx * 2 + 1];
                        const float w[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
#pragma unroll
                        for (int r = 0; r < kRG; ++r)
#pragma unroll
                            for (int co = 0; co < 8; ++co) acc[r][co] = fmaf(in[r + ky][kx], w[co], acc[r][co]);
                    }
                    // (the FMAs have no side effects: without these pins instruction selection sinks all of them
                    // below the last group's loads)
#pragma unroll
                    for (int r = 0; r < kRG; ++r)
#pragma unroll
                        for (int co = 0; co < 8; ++co) asm volatile("" : "+v"(acc[r][co]));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            const float4 s0a = *reinterpret_cast<const float4 *>(wl + 216), s0b = *reinterpret_cast<const float4 *>(wl + 220);
            const float4 h0a = *reinterpret_cast<const float4 *>(wl + 224), h0b = *reinterpret_cast<const float4 *>(wl + 228);
            const int gx = cur.tx * 32 - 1 + hcol;
            const bool xin = (unsigned)gx < (unsigned)a.W;
            const unsigned bw = lds_base + (unsigned)(kBOff * 4 + (hrow0 * kHT + hxd) * 16);
            static_for<0, kRG>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                const int gy = cur.ty * kTH - 1 + hrow0 + r;
                const bool in_img = xin && (unsigned)gy < (unsigned)a.H;
                f32x4 lo, hi;
                lo[0] = relu_nan(fmaf(acc[r][0], s0a.x, h0a.x)); lo[1] = relu_nan(fmaf(acc[r][1], s0a.y, h0a.y));
                lo[2] = relu_nan(fmaf(acc[r][2], s0a.z, h0a.z)); lo[3] = relu_nan(fmaf(acc[r][3], s0a.w, h0a.w));
                hi[0] = relu_nan(fmaf(acc[r][4], s0b.x, h0b.x)); hi[1] = relu_nan(fmaf(acc[r][5], s0b.y, h0b.y));
                hi[2] = relu_nan(fmaf(acc[r][6], s0b.z, h0b.z)); hi[3] = relu_nan(fmaf(acc[r][7], s0b.w, h0b.w));
                if (!in_img) lo = hi = (f32x4){0.f, 0.f, 0.f, 0.f};
                bf16x8 ph, pm, pl;
                split3_block(lo, hi, ph, pm, pl);
                lds_write_b128<r * kHT * 16>(bw, ph);
                lds_write_b128<r * kHT * 16 + kBPartBytes>(bw, pm);
                lds_write_b128<r * kHT * 16 + 2 * kBPartBytes>(bw, pl);
            });
            lds_wait_n<0>();
        }
        __syncthreads();

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
        full_stores = !(a.abl & 4) && cur.ty * kTH + wv * kRPW + kRPW <= a.H;
        parity ^= 1;
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
