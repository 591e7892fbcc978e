// Two consecutive 3x3 stride-1 16 -> 16 layers of FeatureNet as ONE kernel: conv3 + conv4 (MVSNet/models/mvsnet.py:15-16,
// 37-38: ConvBnReLU(16, 16) twice at half resolution).  As two launches of conv_split.hip each layer is bound by its tile
// overheads, not by HBM or the matrix pipe (0.10 ms each for 151 MB in and 151 MB out, 11 GFLOP), and the 151 MB between
// them goes out to HBM and comes back.  Here, by the recipe of tail_fused.hip:
//   tile      14 x 30 output pixels of the second layer <- 16 x 32 pixels of the first layer's output (kept in LDS as two
//             fp16 pieces) <- 18 x 34 input pixels (LDS-DMA, two staging buffers, a tile ahead; split into two fp16 pieces);
//             neighbouring tiles overlap by one pixel of the intermediate map (recomputed 1.22x)
//   layers    M = 16 output channels, N = 16 pixels along x, K = 32 = four (tap, 8-channel chunk) slots: 18 slots in five
//             K-steps, three products (two-piece operands, conv_f16x3.hip); both layers' weight fragments in registers
//   scales    the input's from its absmax block; the intermediate map's from a BOUND (max |scale1| * 9 * 16 * max |in| *
//             max |w1| + max |shift1|): no pass over its values
//   guard     the verdict of conv_guard.h on the input's block, finite weights, a finite bound; a launch that fails it sets
//             a device flag and returns -- the caller (mvs_conv2d_pair_guarded_f16_f32) has the two layers of conv_split.hip
//             enqueued behind it, which run only then
// One 8-wave workgroup per CU (2 waves per SIMD), two barriers per tile: layer 1 + epilogue into LDS | layer 2 + stores, the
// split of the next tile's input, the issue of the copy after next.
#include "conv_split_common.h"
#include "conv_guard.h"

namespace mvs {

namespace pair2d {
constexpr int C = 16, NCH = C / 8;                            // channels; 8-channel chunks
constexpr int OR = 14, OC = 30;                               // output tile
constexpr int MR = OR + 2, MC = OC + 2, MCP = 36;             // intermediate region 16 x 32 (+ zero pad columns)
constexpr int IRW = MR + 2, ICW = MC + 2;                     // input region 18 x 34
constexpr int NVI = IRW * ICW, NVIP = (NVI + 15) / 16 * 16;   // 612 -> 624 voxels
constexpr int NPIECE = NCH * NVIP * 2, NCOPY = (NPIECE + 63) / 64;   // 16-byte pieces (chunk, voxel, half): 2496 -> 39 copies
constexpr int FBYTES = NCOPY * 1024;
constexpr int IPART = NCH * NVIP * 16;                        // one piece plane of the input: [chunk][voxel][8 fp16]
constexpr int MPART = NCH * MR * MCP * 16;                    // ... of the intermediate map
constexpr int NSLOT = 9 * NCH, NG = (NSLOT + 3) / 4;          // 18 slots, 5 K-steps
constexpr int WBYTES = NG * 2 * 1024;                         // one layer's fragments: [K-step][hi, lo][lane][8 fp16]
constexpr int F_OFF = 0, I_OFF = F_OFF + 2 * FBYTES, M_OFF = I_OFF + 2 * IPART, AFF_OFF = M_OFF + 2 * MPART,
              LDS_BYTES = AFF_OFF + 4 * C * 4;
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
constexpr int NTHREADS = 512, NW = 8;
}  // namespace pair2d

struct PairArgs {
    const float *in;              // [N, H, W, 16]
    const unsigned char *wpk;     // [layer][K-step][hi, lo][lane][8 fp16] + trailer {iscale1, wmax1 bits, iscale2, wmax2 bits}
    const float *trailer;
    const float *scale1, *shift1, *scale2, *shift2;   // folded BatchNorm of the two layers (NULL = 1 / 0)
    const unsigned *in_absmax;
    unsigned *out_absmax;         // NULL, or the block the output's largest magnitude is max-ed into
    unsigned *fallback;
    float *out;                   // [N, H, W, 16], or (out_c4) [N, 4, H, W, 4]
    int N, H, W, relu2, out_c4;
    int tiles_x, tiles_y, ntiles;
};

__global__ __launch_bounds__(pair2d::NTHREADS) void conv2d_pair_kernel(PairArgs a) {
    using namespace pair2d;
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4;
    const unsigned lds_base = (unsigned)(uintptr_t)lds;

    // ---------------------------------------------------------------- range guard, operand scales (wave-uniform)
    const AbsmaxVerdict vin = absmax_verdict(a.in_absmax);
    auto uni = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
    const float isw1 = uni(a.trailer[0]), isw2 = uni(a.trailer[2]);
    const unsigned w1max = __float_as_uint(uni(a.trailer[1]));
    float scmax = 0.0f, shmax = 0.0f;
    for (int c = 0; c < C; ++c) {
        scmax = max_nan(scmax, __builtin_fabsf(a.scale1 ? a.scale1[c] : 1.0f));
        shmax = max_nan(shmax, __builtin_fabsf(a.shift1 ? a.shift1[c] : 0.0f));
    }
    // |intermediate| <= max|scale1| * (9 taps x 16 channels) * max|in| * max|w1| + max|shift1|
    const float bound = scmax * (9.0f * C) * __uint_as_float(vin.bits) * __uint_as_float(w1max) * 1.0625f + shmax;
    const unsigned bound_bits = (unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(bound));
    const int xe = absmax_exponent(vin.bits), ce = absmax_exponent(bound_bits) + 1;
    const int E1 = xe - 14 + (int)((__builtin_bit_cast(unsigned, isw1) >> 23) & 255u) - 127;     // undoes layer 1's two scales
    const int E2 = ce - 14 + (int)((__builtin_bit_cast(unsigned, isw2) >> 23) & 255u) - 127;     // ... layer 2's
    const bool decline = vin.code != 0 || isw1 != isw1 || isw2 != isw2 || !(bound_bits < 0x7f800000u) || ce > 126 ||
                         E1 < -120 || E1 > 120 || E2 < -120 || E2 > 120;
    if (decline) {
        if (blockIdx.x == 0 && tid == 0) *a.fallback = 1u;
        return;
    }
    const float sx = pow2f(14 - xe), sm = pow2f(14 - ce), un1 = pow2f(E1), un2 = pow2f(E2);
    if (tid < 4 * C) {
        const int c = tid % C, k = tid / C;
        float v;
        if (k == 0) v = (a.scale1 ? a.scale1[c] : 1.0f) * un1;
        else if (k == 1) v = a.shift1 ? a.shift1[c] : 0.0f;
        else if (k == 2) v = (a.scale2 ? a.scale2[c] : 1.0f) * un2;
        else v = a.shift2 ? a.shift2[c] : 0.0f;
        *reinterpret_cast<float *>(lds + AFF_OFF + tid * 4) = v;
    }
    // zero the intermediate planes once: their pad columns are read against zero weights and must stay finite
    for (int i = tid; i < 2 * MPART / 16; i += NTHREADS) *reinterpret_cast<uint4 *>(lds + M_OFF + i * 16) = make_uint4(0, 0, 0, 0);

    // both layers' weight fragments, in registers for the kernel's lifetime
    f16x8 A1[NG][2], A2[NG][2];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            A1[g][p] = __builtin_bit_cast(f16x8, reinterpret_cast<const uint4 *>(a.wpk)[(g * 2 + p) * 64 + lane]);
            A2[g][p] = __builtin_bit_cast(f16x8, reinterpret_cast<const uint4 *>(a.wpk + WBYTES)[(g * 2 + p) * 64 + lane]);
        }
    // this lane's B voxel of K-step g: slot 4 g + kq = (tap = slot / 2 -> (ky, kx), chunk = slot % 2); byte offsets inside a
    // piece plane relative to (row 0, pixel n) -- slots past the 18 real ones read voxel 0 against zero weights
    unsigned tapI[NG], tapM[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int sl = 4 * g + kq, tap = sl >> 1, ch = sl & 1, ky = tap / 3, kx = tap - ky * 3;
        const bool live = sl < NSLOT;
        tapI[g] = live ? (unsigned)(ch * NVIP * 16 + (ky * ICW + kx) * 16) : 0u;
        tapM[g] = live ? (unsigned)(ch * MR * MCP * 16 + (ky * MCP + kx) * 16) : 0u;
    }

    // 4 values -> hi, lo (fp16 pairs) of v * s (the instruction sequence of split2_block, conv_split_common.h, for one quad)
    auto split2_quad = [](float v0, float v1, float v2, float v3, float s_, unsigned &h0, unsigned &h1, unsigned &l0, unsigned &l1) {
        asm volatile(
            "v_fma_mixlo_f16 %4, %0, %8, 0\n\tv_fma_mixlo_f16 %5, %2, %8, 0\n\t"
            "v_fma_mixhi_f16 %4, %1, %8, 0\n\tv_fma_mixhi_f16 %5, %3, %8, 0\n\t"
            "v_fma_mix_f32 %0, %0, %8, -%4 op_sel_hi:[0,0,1]\n\t"
            "v_fma_mix_f32 %1, %1, %8, -%4 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
            "v_fma_mix_f32 %2, %2, %8, -%5 op_sel_hi:[0,0,1]\n\t"
            "v_fma_mix_f32 %3, %3, %8, -%5 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
            "v_cvt_pk_f16_f32 %6, %0, %1\n\tv_cvt_pk_f16_f32 %7, %2, %3"
            : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "=&v"(h0), "=&v"(h1), "=&v"(l0), "=&v"(l1)
            : "s"(s_));
    };
    // staged input (buffer fsel) -> the two piece planes of the input
    auto split_pass = [&](int fsel) {
        for (int P = tid; P < NPIECE; P += NTHREADS) {
            const f32x4 x = *reinterpret_cast<const f32x4 *>(lds + F_OFF + fsel * FBYTES + P * 16);
            unsigned h0, h1, l0, l1;
            split2_quad(x[0], x[1], x[2], x[3], sx, h0, h1, l0, l1);
            *reinterpret_cast<uint2 *>(lds + I_OFF + P * 8) = make_uint2(h0, h1);
            *reinterpret_cast<uint2 *>(lds + I_OFF + IPART + P * 8) = make_uint2(l0, l1);
        }
    };

    // copies: piece P = (i * 8 + wv) * 64 + lane -> (chunk, voxel, half)
    constexpr int IPW = (NCOPY + NW - 1) / NW;
    int loc[IPW];
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
        const int P = (i * NW + wv) * 64 + lane;
        const int v = (P % (2 * NVIP)) >> 1, c = P / (2 * NVIP);
        loc[i] = (v < NVI && P < NPIECE) ? ((v % ICW) | ((v / ICW) << 8) | ((P & 1) << 16) | (c << 17)) : -1;
    }
    const int64_t img_floats = (int64_t)a.H * a.W * C;
    struct Tile { int tx, ty, b; };
    auto decode = [&](int t) {
        Tile r;
        const int per = a.tiles_x * a.tiles_y;
        r.b = t / per; t -= r.b * per;
        r.ty = t / a.tiles_x; r.tx = t - r.ty * a.tiles_x;
        return r;
    };
    auto issue = [&](int t, int fsel) {
        const Tile tl = decode(t);
        const mvs_srd_t srd = make_srd(a.in + (int64_t)tl.b * img_floats, (unsigned)(img_floats * 4));
        const int gx0 = tl.tx * OC - 2, gy0 = tl.ty * OR - 2;
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            if (i * NW + wv >= NCOPY) continue;   // wave-uniform
            const int gx = gx0 + (loc[i] & 255), gy = gy0 + ((loc[i] >> 8) & 255);
            const int h = (loc[i] >> 16) & 1, c = (loc[i] >> 17) & 1;
            const bool ok = loc[i] >= 0 && (unsigned)gx < (unsigned)a.W && (unsigned)gy < (unsigned)a.H;
            glds16_buf(ok ? (unsigned)(((gy * a.W + gx) * C + c * 8 + h * 4) * 4) : 0xffffff00u, srd, 0u,
                       lds_base + (unsigned)(F_OFF + fsel * FBYTES + (i * NW + wv) * 1024));
        }
    };

    // this CU's tiles: a contiguous range (XCD x: a contiguous share of the list)
    int t_cur, t_end;
    {
        const int nb = gridDim.x;
        int r = blockIdx.x;
        if ((nb & 7) == 0) r = (blockIdx.x & 7) * (nb >> 3) + (blockIdx.x >> 3);
        t_cur = (int)((int64_t)a.ntiles * r / nb);
        t_end = (int)((int64_t)a.ntiles * (r + 1) / nb);
    }
    const int c0 = kq * 4;            // this lane's 4 output channels of an MFMA result (row m = 4 kq + r)
    float vmax = 0.0f;
    if (t_cur < t_end) {
        issue(t_cur, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();                 // affine table, zeroed planes; the first tile's input is staged
    if (t_cur < t_end) {
        split_pass(0);
        if (t_cur + 1 < t_end) issue(t_cur + 1, 1);
    }
    int par = 0;                     // staging buffer of the CURRENT tile's input (already split)
    for (; t_cur < t_end; ++t_cur) {
        const Tile cur = decode(t_cur);
        __syncthreads();             // B: the input pieces of this tile are complete; everybody is done with the intermediate planes
        // ============================================================ phase 1: layer 1 -> intermediate planes
        {
            const float4 sc = *reinterpret_cast<const float4 *>(lds + AFF_OFF + c0 * 4);
            const float4 sh = *reinterpret_cast<const float4 *>(lds + AFF_OFF + (C + c0) * 4);
            // row blocks of this wave: intermediate rows 2 wv, 2 wv + 1, both 16-pixel halves
            static_for<0, 4>([&](auto bc) {
                constexpr int rb = decltype(bc)::value, dr = rb >> 1, xb = rb & 1;
                const int row = 2 * wv + dr;
                const unsigned ab = lds_base + (unsigned)(I_OFF + (row * ICW + xb * 16 + n) * 16);
                f16x8 Bh[NG], Bl[NG];
                static_for<0, NG>([&](auto gc) {
                    constexpr int g = decltype(gc)::value;
                    Bh[g] = __builtin_bit_cast(f16x8, lds_read_b128<0>(ab + tapI[g]));
                    Bl[g] = __builtin_bit_cast(f16x8, lds_read_b128<IPART>(ab + tapI[g]));
                });
                lds_wait_n<0>();
                static_assert(NG == 5, "the pins below name every fragment");
                asm volatile("" : "+v"(Bh[0]), "+v"(Bl[0]), "+v"(Bh[1]), "+v"(Bl[1]), "+v"(Bh[2]), "+v"(Bl[2]));
                asm volatile("" : "+v"(Bh[3]), "+v"(Bl[3]), "+v"(Bh[4]), "+v"(Bl[4]));
                f32x4 e = {0.f, 0.f, 0.f, 0.f}, o = {0.f, 0.f, 0.f, 0.f};
                static_for<0, NG>([&](auto gc) {
                    constexpr int g = decltype(gc)::value;
                    f32x4 &cc = (g & 1) ? o : e;
                    cc = __builtin_amdgcn_mfma_f32_16x16x32_f16(A1[g][1], Bh[g], cc, 0, 0, 0);
                    cc = __builtin_amdgcn_mfma_f32_16x16x32_f16(A1[g][0], Bl[g], cc, 0, 0, 0);
                    cc = __builtin_amdgcn_mfma_f32_16x16x32_f16(A1[g][0], Bh[g], cc, 0, 0, 0);
                });
                // epilogue: affine, ReLU, zero outside the image (layer 2 pads layer 1's OUTPUT), scale, split, into the planes
                const int col = xb * 16 + n;
                const int gy = cur.ty * OR - 1 + row, gx = cur.tx * OC - 1 + col;
                const bool inside = (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
                float v0 = relu_nan((e[0] + o[0]) * sc.x + sh.x), v1 = relu_nan((e[1] + o[1]) * sc.y + sh.y);
                float v2 = relu_nan((e[2] + o[2]) * sc.z + sh.z), v3 = relu_nan((e[3] + o[3]) * sc.w + sh.w);
                if (!inside) v0 = v1 = v2 = v3 = 0.0f;
                unsigned h0, h1, l0, l1;
                split2_quad(v0, v1, v2, v3, sm, h0, h1, l0, l1);
                unsigned char *dst = lds + M_OFF + (kq >> 1) * MR * MCP * 16 + (row * MCP + col) * 16 + (kq & 1) * 8;
                *reinterpret_cast<uint2 *>(dst) = make_uint2(h0, h1);
                *reinterpret_cast<uint2 *>(dst + MPART) = make_uint2(l0, l1);
            });
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the next tile's input has landed (its copies went out a tile ago)
        }
        __syncthreads();             // C: the intermediate planes are complete; the input pieces are free; the next input is staged
        // ============================================================ phase 2: layer 2 -> HBM; the next tile's split; the copy after next
        if (t_cur + 2 < t_end) issue(t_cur + 2, par);            // (buffer `par` held this tile's input: split a tile ago)
        {
            const float4 sc = *reinterpret_cast<const float4 *>(lds + AFF_OFF + (2 * C + c0) * 4);
            const float4 sh = *reinterpret_cast<const float4 *>(lds + AFF_OFF + (3 * C + c0) * 4);
            // output rows of this wave: 2 wv, 2 wv + 1 (waves 0..5), 12 (wave 6), 13 (wave 7); both halves
            static_for<0, 4>([&](auto bc) {
                constexpr int rb = decltype(bc)::value, dr = rb >> 1, xb = rb & 1;
                const int row = wv < 6 ? 2 * wv + dr : 6 + wv;
                if (wv >= 6 && dr == 1) return;                   // wave-uniform
                const unsigned ab = lds_base + (unsigned)(M_OFF + (row * MCP + xb * 16 + n) * 16);
                f16x8 Bh[NG], Bl[NG];
                static_for<0, NG>([&](auto gc) {
                    constexpr int g = decltype(gc)::value;
                    Bh[g] = __builtin_bit_cast(f16x8, lds_read_b128<0>(ab + tapM[g]));
                    Bl[g] = __builtin_bit_cast(f16x8, lds_read_b128<MPART>(ab + tapM[g]));
                });
                lds_wait_n<0>();
                static_assert(NG == 5, "the pins below name every fragment");
                asm volatile("" : "+v"(Bh[0]), "+v"(Bl[0]), "+v"(Bh[1]), "+v"(Bl[1]), "+v"(Bh[2]), "+v"(Bl[2]));
                asm volatile("" : "+v"(Bh[3]), "+v"(Bl[3]), "+v"(Bh[4]), "+v"(Bl[4]));
                f32x4 e = {0.f, 0.f, 0.f, 0.f}, o = {0.f, 0.f, 0.f, 0.f};
                static_for<0, NG>([&](auto gc) {
                    constexpr int g = decltype(gc)::value;
                    f32x4 &cc = (g & 1) ? o : e;
                    cc = __builtin_amdgcn_mfma_f32_16x16x32_f16(A2[g][1], Bh[g], cc, 0, 0, 0);
                    cc = __builtin_amdgcn_mfma_f32_16x16x32_f16(A2[g][0], Bl[g], cc, 0, 0, 0);
                    cc = __builtin_amdgcn_mfma_f32_16x16x32_f16(A2[g][0], Bh[g], cc, 0, 0, 0);
                });
                const int col = xb * 16 + n;
                const int gy = cur.ty * OR + row, gx = cur.tx * OC + col;
                if (col < OC && gy < a.H && gx < a.W) {
                    float v0 = (e[0] + o[0]) * sc.x + sh.x, v1 = (e[1] + o[1]) * sc.y + sh.y;
                    float v2 = (e[2] + o[2]) * sc.z + sh.z, v3 = (e[3] + o[3]) * sc.w + sh.w;
                    if (a.relu2) { v0 = relu_nan(v0); v1 = relu_nan(v1); v2 = relu_nan(v2); v3 = relu_nan(v3); }
                    const int64_t o4 = a.out_c4 ? (((int64_t)cur.b * (C / 4) + kq) * a.H + gy) * (int64_t)a.W * 4 + (int64_t)gx * 4
                                                : (((int64_t)cur.b * a.H + gy) * a.W + gx) * C + c0;
                    *reinterpret_cast<float4 *>(a.out + o4) = make_float4(v0, v1, v2, v3);
                    vmax = amax4_nan(vmax, v0, v1, v2, v3);
                }
            });
        }
        if (t_cur + 1 < t_end) split_pass(par ^ 1);               // the next tile's input -> the piece planes
        par ^= 1;
    }
    publish_absmax(a.out_absmax, vmax);
}

// layer weights (16, 16, 3, 3) x 2 -> [layer][K-step g][hi, lo][lane][8 fp16] of w * 2^(14 - exponent(max |w|)); lane (m, kq):
// output channel m, slot 4 g + kq = (tap = slot / 2, chunk = slot % 2), input channel chunk * 8 + i.  Trailer per layer:
// {what undoes the scale (NaN: weights not finite), max |w| bits}
__global__ __launch_bounds__(256) void pack_pair_kernel(const float *__restrict__ w1, const float *__restrict__ w2,
                                                        unsigned short *__restrict__ out, const unsigned *__restrict__ wmax,
                                                        float *__restrict__ trailer) {
    using namespace pair2d;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < 2) {
        const unsigned m = wmax[i];
        trailer[2 * i] = m >= 0x7f800000u ? __builtin_nanf("") : pow2f(absmax_exponent(m) - 14);
        trailer[2 * i + 1] = __uint_as_float(m);
    }
    if (i >= 2 * NG * 512) return;
    const int layer = i / (NG * 512), r = i - layer * (NG * 512);
    const int jj = r & 7, lane = (r >> 3) & 63, g = r >> 9;
    const int m = lane & 15, kq = lane >> 4, sl = 4 * g + kq, tap = sl >> 1, ch = sl & 1;
    const float *w = layer ? w2 : w1;
    const int we = absmax_exponent(wmax[layer]);
    float x = 0.0f;
    if (sl < NSLOT) x = w[((int64_t)m * C + ch * 8 + jj) * 9 + tap];
    x *= pow2f(14 - we);
    const _Float16 h = (_Float16)x;
    const _Float16 l = (_Float16)(x - (float)h);
    unsigned short *o = out + (size_t)layer * (WBYTES / 2) + ((size_t)g * 2) * 512 + lane * 8 + jj;
    o[0] = __builtin_bit_cast(unsigned short, h);
    o[512] = __builtin_bit_cast(unsigned short, l);
}

int launch_absmax_word(const float *x, int64_t n, unsigned *word, hipStream_t st);   // conv_f16x3.hip

}  // namespace mvs

using namespace mvs;

extern "C" size_t mvs_conv2d_pair_packed_bytes(int C) { return C == 16 ? (size_t)2 * pair2d::WBYTES + 32 : 0; }

extern "C" int mvs_conv2d_pair_supported(int C, int N, int H, int W) {
    if (C != 16 || N <= 0 || H <= 0 || W <= 0) return 0;
    if ((int64_t)H * W * C * 4 >= 0xffffff00LL) return 0;              // 32-bit byte offsets inside an image
    return (int64_t)N * ((H + 13) / 14) * ((W + 29) / 30) < (1LL << 30) ? 1 : 0;
}

extern "C" int mvs_conv2d_pair_pack_weights_f32(const float *w1, const float *w2, int C, void *packed, void *stream) {
    if (!w1 || !w2 || !packed || C != 16) {
        set_error("mvs_conv2d_pair_pack_weights_f32: needs two (16, 16, 3, 3) weights and a packed buffer");
        return MVS_EINVAL;
    }
    unsigned char *pk = static_cast<unsigned char *>(packed);
    float *trailer = reinterpret_cast<float *>(pk + 2 * pair2d::WBYTES);
    unsigned *wmax = reinterpret_cast<unsigned *>(pk + 2 * pair2d::WBYTES + 16);
    int rc = launch_absmax_word(w1, (int64_t)C * C * 9, wmax, as_stream(stream));
    if (rc != MVS_OK) return rc;
    rc = launch_absmax_word(w2, (int64_t)C * C * 9, wmax + 1, as_stream(stream));
    if (rc != MVS_OK) return rc;
    hipLaunchKernelGGL(pack_pair_kernel, dim3((2 * pair2d::NG * 512 + 255) / 256), dim3(256), 0, as_stream(stream), w1, w2,
                       reinterpret_cast<unsigned short *>(pk), wmax, trailer);
    return check_launch("mvs_conv2d_pair_pack_weights_f32");
}

extern "C" int mvs_conv2d_pair_f16_f32(const float *in, const void *in_absmax, const void *packed_pair, const float *scale1,
                                       const float *shift1, const float *scale2, const float *shift2, int relu2, int N, int C,
                                       int H, int W, int out_c4, float *out, void *out_absmax, void *fallback_flag, void *stream) {
    if (!in || !in_absmax || !packed_pair || !out || !fallback_flag || !mvs_conv2d_pair_supported(C, N, H, W)) {
        set_error("mvs_conv2d_pair_f16_f32: invalid argument ([N, H, W, 16] channels-last input with its absmax block, the pair pack, "
                  "a flag word)");
        return MVS_EINVAL;
    }
    PairArgs a;
    a.in = in; a.wpk = static_cast<const unsigned char *>(packed_pair);
    a.trailer = reinterpret_cast<const float *>(a.wpk + 2 * pair2d::WBYTES);
    a.scale1 = scale1; a.shift1 = shift1; a.scale2 = scale2; a.shift2 = shift2;
    a.in_absmax = static_cast<const unsigned *>(in_absmax);
    a.out_absmax = static_cast<unsigned *>(out_absmax);
    a.fallback = static_cast<unsigned *>(fallback_flag);
    a.out = out; a.N = N; a.H = H; a.W = W; a.relu2 = relu2; a.out_c4 = out_c4;
    a.tiles_x = (W + pair2d::OC - 1) / pair2d::OC; a.tiles_y = (H + pair2d::OR - 1) / pair2d::OR;
    a.ntiles = N * a.tiles_x * a.tiles_y;
    const int n_cu = device_cu_count();
    hipLaunchKernelGGL(conv2d_pair_kernel, dim3((unsigned)(a.ntiles < n_cu ? a.ntiles : n_cu)), dim3(pair2d::NTHREADS), 0,
                       as_stream(stream), a);
    return check_launch("mvs_conv2d_pair_f16_f32");
}

// The fused kernel, then the two layers of conv_split.hip enqueued behind it with the "run only if" word (mvs_common.h:
// conv_run_flag) = the flag the fused kernel's range guard sets when it declines.  No host synchronisation.  *flag = 0 on entry;
// mid_scratch: [N, H, W, 16] floats, touched only by the unfused path; packed1_f16 / packed2_f16: the layers' own
// mvs_conv_split_pack_weights_f16_f32 packs.
extern "C" int mvs_conv2d_pair_guarded_f16_f32(const float *in, const void *in_absmax, const void *packed_pair, const void *packed1_f16,
                                               const void *packed2_f16, const float *scale1, const float *shift1, const float *scale2,
                                               const float *shift2, int relu2, int N, int C, int H, int W, int out_c4, float *mid_scratch,
                                               float *out, void *out_absmax, void *flag, void *stream) {
    if (!packed1_f16 || !packed2_f16 || !mid_scratch || !flag) {
        set_error("mvs_conv2d_pair_guarded_f16_f32: needs the two layers' two-piece packs, a [N, H, W, C] scratch map and a zeroed flag word");
        return MVS_EINVAL;
    }
    int rc = mvs_conv2d_pair_f16_f32(in, in_absmax, packed_pair, scale1, shift1, scale2, shift2, relu2, N, C, H, W, out_c4, out, out_absmax,
                                     flag, stream);
    if (rc != MVS_OK) return rc;
    struct FlagScope {
        explicit FlagScope(const void *f) { conv_run_flag() = static_cast<const unsigned *>(f); }
        ~FlagScope() { conv_run_flag() = nullptr; }
    } scope(flag);
    // the intermediate map's block: behind the flag words (the caller's flag buffer is MVS_ABSMAX_WORDS + 64 words: [flag .. | block])
    void *mid_absmax = static_cast<unsigned *>(flag) + 64;
    rc = mvs_conv_split_f16_f32(in, in_absmax, packed1_f16, scale1, shift1, nullptr, 1, 1, 1, 1, C, C, N, H, W, 0, mid_scratch, mid_absmax, stream);
    if (rc != MVS_OK) return rc;
    return mvs_conv_split_f16_f32(mid_scratch, mid_absmax, packed2_f16, scale2, shift2, nullptr, relu2, 1, 1, 1, C, C, N, H, W, out_c4, out,
                                  out_absmax, stream);
}
