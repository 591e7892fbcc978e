// The full-resolution end of CasMVSNet's FPN in one kernel (CasMVSNet/models/module.py:396-398):
//     intra_feat = F.interpolate(intra_feat, scale_factor=2, mode="nearest") + self.inner2(conv0)     # 1x1, 8 -> 32, bias
//     out3 = self.out3(intra_feat)                                                                     # 3x3, 32 -> 8
// As two launches the 32-channel full-resolution map (1.2 GB at 5 x 1184 x 1600) goes out to HBM and comes back: 0.52 +
// 0.78 ms.  Here a persistent workgroup per CU (8 waves) walks 32 x 8-pixel tiles:
//   copy      conv0's output (8 channels) on the tile's 34 x 10 halo and the half-resolution map (32 channels) on its
//             18 x 6 footprint, by LDS-DMA, double-buffered (zero fill outside by the buffer range check)
//   lateral   on the vector ALU, a thread per halo pixel: the 1x1 convolution (+ bias), plus the nearest-neighbour
//             parent of the half-resolution map; zero outside the image (out3 pads ITS input); every result split exactly
//             into three bf16 numbers and written as out3's operand planes [part][8-channel chunk][pixel]
//   out3      on the BF16 matrix pipe at fp32 accuracy: the Cout = 8 "shifted" form of feature_head.hip (MFMA rows =
//             channel x x-shift, K = 4 x-taps x 8 channels), 3 kernel rows x 4 channel chunks x 6 products per row of
//             32 pixels; a wave owns one row of the tile
// 0.77-0.83 ms against 1.30 for the two launches.  By parts (lateral / out3 / stores removed in turn): copies, stores and
// barriers 0.21, lateral +0.26-0.32, out3 +0.25 -- out3 is bound by LDS reads (every wave reads all 36 A fragments for its
// one row: 72 ds_read_b128 x 8 waves x 8 cycles per tile; a deeper read pipeline changed nothing), the lateral layer by
// instruction issue (340 of 512 threads, ~520 instructions each).
#include "conv_split_common.h"

#include <cstdlib>

namespace mvs {

namespace {
constexpr int kTH = 8;                                    // tile rows (x: 32)
constexpr int kHR = kTH + 2, kHT = 34, kHXH = 17;         // out3's halo of a tile; half of its width (x de-interleaved)
constexpr int kVox = kHR * kHT;                           // 340
constexpr int kPlaneBytes = kVox * 16;                    // one (part, chunk) plane: 8 channels bf16 per pixel
constexpr int kC0Gran = kVox * 2, kC0Dma = (kC0Gran + 63) / 64;       // conv0 halo: 32 bytes per pixel -> 680 pieces, 11 copies
constexpr int kTR = kTH / 2 + 2, kTC = 18;                // half-resolution footprint: 6 rows x 18 columns
constexpr int kTopGran = kTR * kTC * 8, kTopDma = (kTopGran + 63) / 64;   // 128 bytes per pixel -> 864 pieces, 14 copies (13.5)
constexpr int kDma = kC0Dma + kTopDma;                    // 25
constexpr int kInFloats = kDma * 256;                     // one input buffer
constexpr int kWaves = 8, kThreads = 512;
constexpr int kABytes = 3 * 4 * 3 * 1024;                 // out3 A fragments: [kernel row][chunk][part][lane][8 bf16]
constexpr int kWiFloats = 32 * 8 + 32;                    // inner2 [cout][cin], bias
constexpr int kAOff = 0, kBOff = kAOff + kABytes / 4, kInOff = kBOff + 12 * kPlaneBytes / 4;
constexpr int kWiOff = kInOff + 2 * kInFloats, kLdsFloats = kWiOff + kWiFloats;
static_assert(kLdsFloats * 4 <= 160 * 1024 && kVox <= kThreads && kTH == kWaves, "one workgroup per CU, a pixel per thread");
constexpr int kIPW = (kDma + kWaves - 1) / kWaves;        // copies per wave
}  // namespace

struct FpnTailArgs {
    const float *c0;       // [N,H,W,8]
    const float *top;      // [N,H/2,W/2,32]
    const float *wi, *bi;  // inner2: PyTorch layout (32,8[,1,1]); bias (or NULL)
    const unsigned char *wpk;   // out3: mvs_fpn_tail_pack_weights_f32
    const float *bo;       // out3 bias (or NULL)
    float *out;            // [N,H,W,8]
    int N, H, W, tiles_x, tiles_y, ystrip;
};

__global__ __launch_bounds__(kThreads) void fpn_tail_kernel(FpnTailArgs a, int ntiles) {
    __shared__ __attribute__((aligned(16))) float lds[kLdsFloats];
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
    // once: out3's A fragments, inner2's weights and bias
    for (int i = tid; i < kABytes / 16; i += kThreads)
        reinterpret_cast<float4 *>(lds + kAOff)[i] = reinterpret_cast<const float4 *>(a.wpk)[i];
    for (int i = tid; i < kWiFloats; i += kThreads) lds[kWiOff + i] = i < 256 ? a.wi[i] : (a.bi ? a.bi[i - 256] : 0.0f);

    // tile-independent coordinates of this wave's copies.  Copies 0 .. kC0Dma-1: piece q = (halo pixel, 16-byte half);
    // the rest: piece q = (footprint pixel, 16-byte eighth)
    int loc[kIPW];
#pragma unroll
    for (int i = 0; i < kIPW; ++i) {
        const int c = i * kWaves + wv;
        if (c < kC0Dma) {
            const int q = c * 64 + lane, qc = min(q, kC0Gran - 1);
            const int vox = qc >> 1;
            loc[i] = (vox % kHT) | ((vox / kHT) << 8) | ((qc & 1) << 16) | (q < kC0Gran ? 0 : (int)0x80000000);
        } else {
            const int q = (c - kC0Dma) * 64 + lane, qc = min(q, kTopGran - 1);
            const int px = qc >> 3;
            loc[i] = (px % kTC) | ((px / kTC) << 8) | ((qc & 7) << 16) | ((q < kTopGran && c < kDma) ? 0 : (int)0x80000000);
        }
    }
    const int Hh = a.H >> 1, Wh = a.W >> 1;
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
        const mvs_srd_t srd0 = make_srd(a.c0 + (int64_t)nxt.b * a.H * a.W * 8, (unsigned)(a.H * a.W) * 32u);
        const mvs_srd_t srd1 = make_srd(a.top + (int64_t)nxt.b * Hh * Wh * 32, (unsigned)(Hh * Wh) * 128u);
        const int gx0 = nxt.tx * 32 - 1, gy0 = nxt.ty * kTH - 1;          // halo origin
        const int hx0 = nxt.tx * 16 - 1, hy0 = nxt.ty * (kTH / 2) - 1;    // footprint origin = (halo origin) >> 1
        const unsigned base = lds_base + (unsigned)(kInOff + parity * kInFloats) * 4u;
#pragma unroll
        for (int i = 0; i < kIPW; ++i) {
            const int c = i * kWaves + wv;
            if (c >= kDma) continue;            // wave-uniform
            if (c < kC0Dma) {
                const int gx = gx0 + (loc[i] & 255), gy = gy0 + ((loc[i] >> 8) & 255), h = (loc[i] >> 16) & 1;
                const bool ok = loc[i] >= 0 && (unsigned)gx < (unsigned)a.W && (unsigned)gy < (unsigned)a.H;
                const unsigned voff = ok ? (unsigned)((gy * a.W + gx) * 32 + h * 16) : 0xffffff00u;
                glds16_buf(voff, srd0, 0u, base + (unsigned)c * 1024u);
            } else {
                const int gx = hx0 + (loc[i] & 255), gy = hy0 + ((loc[i] >> 8) & 255), e = (loc[i] >> 16) & 7;
                const bool ok = loc[i] >= 0 && (unsigned)gx < (unsigned)Wh && (unsigned)gy < (unsigned)Hh;
                const unsigned voff = ok ? (unsigned)((gy * Wh + gx) * 128 + e * 16) : 0xffffff00u;
                glds16_buf(voff, srd1, 0u, base + (unsigned)c * 1024u);
            }
        }
    };

    const int c0lane = (kq & 1) * 4;
    const float4 bo = a.bo ? *reinterpret_cast<const float4 *>(a.bo + c0lane) : make_float4(0.f, 0.f, 0.f, 0.f);
    // out3 operand addresses: A = [kernel row][chunk][part][lane]; B = this lane's pixel x = 2 n + kq (x de-interleaved)
    // of the wave's row; kernel rows, chunks and parts are compile-time offsets
    const unsigned aA = lds_base + (unsigned)(kAOff * 4 + lane * 16);
    const unsigned aB = lds_base + (unsigned)(kBOff * 4 + (wv * kHT + (kq & 1) * kHXH + (kq >> 1) + n) * 16);
    // lateral: this thread's halo pixel
    const int hrow = tid / kHT, hcol = tid - hrow * kHT;
    const bool lat_thread = tid < kVox;
    const int hxd = (hcol & 1) ? kHXH + (hcol >> 1) : (hcol >> 1);

    int parity = 0;
    if (t_cur < t_end) issue(t_cur, 0);
    while (t_cur < t_end) {
        const Tile cur = nxt;
        const int t_next = t_cur + t_step;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // the inputs have landed; every wave is done with out3 of the previous tile
        if (t_next < t_end) issue(t_next, parity ^ 1);

        // ---- lateral 1x1 + bias + nearest parent -> out3's operand planes
        if (lat_thread) {
            const float *in0 = lds + kInOff + parity * kInFloats + tid * 8;      // this pixel's 8 channels of conv0
            // the parent pixel: (halo origin + offset) >> 1 - footprint origin, with halo origin odd (= 2 k - 1)
            const int py = (hrow + 1) >> 1, px = (hcol + 1) >> 1;
            const float *in1 = lds + kInOff + parity * kInFloats + kC0Dma * 256 + (py * kTC + px) * 32;
            const float *wl = lds + kWiOff;
            const float4 xa = *reinterpret_cast<const float4 *>(in0), xb = *reinterpret_cast<const float4 *>(in0 + 4);
            const int gx = cur.tx * 32 - 1 + hcol, gy = cur.ty * kTH - 1 + hrow;
            const bool in_img = (unsigned)gx < (unsigned)a.W && (unsigned)gy < (unsigned)a.H;
            const unsigned bw = lds_base + (unsigned)(kBOff * 4 + (hrow * kHT + hxd) * 16);
            static_for<0, 4>([&](auto cc) {
                constexpr int ch = decltype(cc)::value;        // output channels ch * 8 .. ch * 8 + 7
                f32x4 lo, hi;
#pragma unroll
                for (int o = 0; o < 8; ++o) {
                    const float4 wa = *reinterpret_cast<const float4 *>(wl + (ch * 8 + o) * 8);
                    const float4 wb = *reinterpret_cast<const float4 *>(wl + (ch * 8 + o) * 8 + 4);
                    float v = xa.x * wa.x;
                    v = fmaf(xa.y, wa.y, v); v = fmaf(xa.z, wa.z, v); v = fmaf(xa.w, wa.w, v);
                    v = fmaf(xb.x, wb.x, v); v = fmaf(xb.y, wb.y, v); v = fmaf(xb.z, wb.z, v); v = fmaf(xb.w, wb.w, v);
                    v = (v + wl[256 + ch * 8 + o]) + in1[ch * 8 + o];
                    if (!in_img) v = 0.0f;
                    if (o < 4) lo[o] = v; else hi[o - 4] = v;
                }
                bf16x8 ph, pm, pl;
                split3_block(lo, hi, ph, pm, pl);
                lds_write_b128<(0 * 4 + ch) * kPlaneBytes>(bw, ph);
                lds_write_b128<(1 * 4 + ch) * kPlaneBytes>(bw, pm);
                lds_write_b128<(2 * 4 + ch) * kPlaneBytes>(bw, pl);
            });
            lds_wait_n<0>();
        }
        __syncthreads();

        // ---- out3: items (kernel row ky, chunk ch): three A and three B reads one item ahead of its six MFMAs
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        {
            bf16x8 A[2][3], Bf[2][3];
            auto fetch = [&](auto ic) {
                constexpr int it = decltype(ic)::value, ky = it / 4, ch = it % 4;
                static_for<0, 3>([&](auto pc) {
                    constexpr int p = decltype(pc)::value;
                    A[it & 1][p] = __builtin_bit_cast(bf16x8, lds_read_b128<((ky * 4 + ch) * 3 + p) * 1024>(aA));
                    Bf[it & 1][p] = __builtin_bit_cast(bf16x8, lds_read_b128<(p * 4 + ch) * kPlaneBytes + ky * kHT * 16>(aB));
                });
            };
            fetch(std::integral_constant<int, 0>{});
            static_for<0, 12>([&](auto ic) {
                constexpr int it = decltype(ic)::value;
                lds_wait_n<0>();
                {
                    bf16x8 &b0 = Bf[it & 1][0], &b1 = Bf[it & 1][1], &b2 = Bf[it & 1][2];
                    bf16x8 &a0 = A[it & 1][0], &a1 = A[it & 1][1], &a2 = A[it & 1][2];
                    asm volatile("" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(a0), "+v"(a1), "+v"(a2));
                }
                if constexpr (it + 1 < 12) fetch(std::integral_constant<int, it + 1>{});
                __builtin_amdgcn_sched_barrier(0);   // the reads go out BEFORE this item's MFMAs
                const bf16x8 ah = A[it & 1][0], am = A[it & 1][1], al = A[it & 1][2];
                const bf16x8 bh = Bf[it & 1][0], bm = Bf[it & 1][1], bl = Bf[it & 1][2];
                // six partial products, small terms first (as conv_split.hip)
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        // ---- epilogue: bias, one 16-byte store per lane (the wave's row)
        {
            const int ox = cur.tx * 32 + 2 * n + (kq >> 1), oy = cur.ty * kTH + wv;
            if (oy < a.H && ox < a.W) {
                const int64_t o = (((int64_t)cur.b * a.H + oy) * a.W + ox) * 8 + c0lane;
                *reinterpret_cast<float4 *>(a.out + o) = make_float4(acc[0] + bo.x, acc[1] + bo.y, acc[2] + bo.z, acc[3] + bo.w);
            }
        }
        parity ^= 1;
        t_cur = t_next;
    }
}

// out3's weight (8, 32, 3, 3) -> [kernel row ky][chunk][part][lane][8 bf16]: lane (m, kq) = MFMA row m = (x-shift m >> 3,
// channel m & 7), K slots kq * 8 + j = (x-tap kq, input channel chunk * 8 + j): w[m & 7][chunk * 8 + j][ky][kq - shift]
__global__ __launch_bounds__(256) void fpn_tail_pack_kernel(const float *__restrict__ w, unsigned short *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 3 * 4 * 64 * 8) return;
    const int j = i & 7, lane = (i >> 3) & 63, ch = (i >> 9) & 3, ky = i >> 11;
    const int m = lane & 15, kq = lane >> 4, co = m & 7, kxr = kq - (m >> 3);
    const float x = (kxr >= 0 && kxr < 3) ? w[((co * 32 + ch * 8 + j) * 3 + ky) * 3 + kxr] : 0.0f;
    const __bf16 h = (__bf16)x;
    const float r1 = x - (float)h;
    const __bf16 mm = (__bf16)r1;
    const float r2 = r1 - (float)mm;
    const __bf16 l = (__bf16)r2;
    unsigned short *o = out + (size_t)((ky * 4 + ch) * 3) * 512 + lane * 8 + j;
    o[0] = __builtin_bit_cast(unsigned short, h);
    o[512] = __builtin_bit_cast(unsigned short, mm);
    o[1024] = __builtin_bit_cast(unsigned short, l);
}

}  // namespace mvs

using namespace mvs;

extern "C" int mvs_fpn_tail_supported(int H, int W) {
    return H > 1 && W > 1 && !(H & 1) && !(W & 1) && (int64_t)H * W * 32 < 0xffffff00LL;
}

extern "C" size_t mvs_fpn_tail_packed_bytes(void) { return (size_t)kABytes; }

extern "C" int mvs_fpn_tail_pack_weights_f32(const float *weight_out, void *packed, void *stream) {
    if (!weight_out || !packed) return bare_error(MVS_EINVAL, __func__, __LINE__);
    hipLaunchKernelGGL(fpn_tail_pack_kernel, dim3(24), dim3(256), 0, (hipStream_t)stream, weight_out,
                       static_cast<unsigned short *>(packed));
    return check_launch("mvs_fpn_tail_pack_weights_f32");
}

extern "C" int mvs_fpn_tail_f32(const float *fine, const float *coarse, const float *weight_inner, const float *bias_inner,
                                const void *packed_out, const float *bias_out, int N, int H, int W, float *out,
                                void *stream) {
    if (!fine || !coarse || !weight_inner || !packed_out || !out || N <= 0) return bare_error(MVS_EINVAL, __func__, __LINE__);
    if (!mvs_fpn_tail_supported(H, W)) {
        set_error("mvs_fpn_tail_f32: needs even H, W and an image below 4 GiB (H=%d W=%d)", H, W);
        return MVS_EUNSUPPORTED;
    }
    FpnTailArgs a;
    a.c0 = fine; a.top = coarse; a.wi = weight_inner; a.bi = bias_inner;
    a.wpk = static_cast<const unsigned char *>(packed_out); a.bo = bias_out; a.out = out;
    a.N = N; a.H = H; a.W = W;
    a.tiles_x = (W + 31) / 32; a.tiles_y = (H + kTH - 1) / kTH; a.ystrip = 16;
    const int64_t nt = (int64_t)a.tiles_x * a.tiles_y * N;
    if (nt > 0x7fffffffLL) return bare_error(MVS_EINVAL, __func__, __LINE__);
    const int slots = device_cu_count();
    hipLaunchKernelGGL(fpn_tail_kernel, dim3((unsigned)(nt < slots ? nt : slots)), dim3(kThreads), 0, (hipStream_t)stream, a,
                       (int)nt);
    return check_launch("mvs_fpn_tail_f32");
}
