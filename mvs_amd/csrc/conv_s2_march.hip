// CostRegNet's conv1 (ConvBnReLU3D 8 -> 16, k3 s2 p1; MVSNet/models/mvsnet.py:53,84) as a z-MARCHING kernel.
//
// The per-tile kernel of conv_split.hip stages the 5 x 9 x 33-voxel halo of a (2, 4, 16) output tile: 1.45 input voxels
// fetched (L2 -> LDS) per voxel used, 47 copies for 42 MFMAs per wave -- the layer is bound by its copies (0.23-0.26 ms for
// 727 MB in, 182 MB out).  Here a workgroup owns a column of 8 x 16 output pixels and marches along z: an output plane needs
// input planes 2z - 1, 2z, 2z + 1, so a step brings TWO new input planes of 17 x 33 voxels (1.10 fetched per voxel used) into
// a ring of three piece planes; every input voxel is copied and split once per column.
//   step      phase 1: 21 MFMAs per wave (wave = output row; M = 16 channels, N = 16 voxels along x, K = 32 = four taps x 8
//             channels, 27 taps in seven K-steps, three products of two-piece fp16 operands -- the fragments of
//             mvs_conv_split_pack_weights_f16_f32 as they are, in registers), affine, ReLU, one 16-byte store per lane;
//             the copies of the pair after next go out behind the MFMAs;
//             phase 2: the next pair (landed a step ago) is split into the ring slots the two oldest planes leave.  Two
//             barriers per step, two staging buffers.
//   ranges    every CU takes a contiguous range of (column, step) pairs, as tail_fused.hip: a range that begins inside a
//             column stages one extra plane.
//   guard     as conv_split.hip: the verdict on the input's absmax block decides at launch; a launch that fails it computes
//             the layer with guard_direct_conv on the fp32 weights behind the pack (same kernel, no flag).
// 8 waves per workgroup, one workgroup per CU.  0.21-0.22 ms at config 2 against 0.235-0.24 for the per-tile kernel on the same
// box (scripts/exp_conv1_march.py; tuning build MVS_S2M_ABL: without the copies 0.17, the split pass 0.20, the stores 0.19,
// copies and split 0.12, everything 0.07).  Tried and not kept: a ring of seven fp32 planes written by the copies directly
// (no staging, no split pass, one barrier, copies two steps ahead under counted waits) with the split at fragment-read time --
// 140 vector instructions and 14 conflicted 32-byte reads per wave and step cost more than the pass they replace: 0.235 ms.
#include "conv_split_common.h"
#include "conv_guard.h"

#include <cstdlib>

namespace mvs {

namespace s2m {
constexpr int CIN = 8, COUT = 16;
constexpr int TY = 8, TX = 16;                                // output tile of a step (one plane)
constexpr int IR = 2 * TY + 1, IC = 2 * TX + 1;               // input rows / columns of a plane: 17 x 33
constexpr int NV = IR * IC, NVP = (NV + 15) / 16 * 16;        // 561 -> 576 voxels
constexpr int PPIECE = NVP * 2, PCOPY = PPIECE / 64;          // 16-byte pieces of ONE plane (voxel, half): 1152 -> 18 copies
constexpr int FPLANE = PCOPY * 1024;                          // staging bytes of one plane
constexpr int IPART = NVP * 16;                               // one piece plane of one input plane: [voxel][8 fp16]
constexpr int NG = 7;                                         // K-steps: 27 taps, four per step
constexpr int F_OFF = 0, R_OFF = F_OFF + 2 * 2 * FPLANE, AFF_OFF = R_OFF + 3 * 2 * IPART, LDS_BYTES = AFF_OFF + 2 * COUT * 4;
static_assert(LDS_BYTES <= 160 * 1024 && PPIECE % 64 == 0, "LDS budget");
constexpr int NTHREADS = 512, NW = 8;
}  // namespace s2m

struct S2MArgs {
    const float *in;            // [B, D, H, W, 8]
    const unsigned char *wpk;   // [K-step][hi, lo][lane][8 fp16]
    const float *w_iscale;      // behind the fragments: what undoes the weights' scale; + 4 floats: the fp32 weights [tap][Cin][Cout]
    const float *scale, *shift;
    const unsigned *in_absmax;
    unsigned *out_absmax;
    unsigned long long *guard_cnt;
    const unsigned *run_flag;
    float *out;                 // [B, Do, Ho, Wo, 16]
    int B, D, H, W, Do, Ho, Wo, relu;
    int tiles_x, tiles_y, ncols;
    int abl;                    // tuning builds (MVS_S2M_ABL; wrong results): 1 no copies, 2 no split, 4 no MFMAs, 8 no stores
};

__global__ __launch_bounds__(s2m::NTHREADS) void conv_s2_march_kernel(S2MArgs a) {
    using namespace s2m;
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
    if (a.run_flag && *a.run_flag == 0u) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4;
    const unsigned lds_base = (unsigned)(uintptr_t)lds;

    // ---------------------------------------------------------------- operand scales, range guard (conv_split.hip's)
    const AbsmaxVerdict verdict = absmax_verdict(a.in_absmax);
    const int xe = absmax_exponent(verdict.bits);
    const float isw = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, *a.w_iscale)));
    const float sx = pow2f(14 - xe);
    const int E = xe - 14 + (int)((__builtin_bit_cast(unsigned, isw) >> 23) & 255u) - 127;
    const int e1 = E < -60 ? -60 : (E > 60 ? 60 : E), e2 = E - e1 < -126 ? -126 : (E - e1 > 127 ? 127 : E - e1);
    const float unscale = pow2f(e1), unscale2 = pow2f(e2);
    if (verdict.code != 0 || isw != isw) {
        GuardConv g;
        g.in = a.in; g.w = a.w_iscale + 4; g.scale = a.scale; g.shift = a.shift; g.residual = nullptr; g.out = a.out;
        g.out_absmax = a.out_absmax; g.counter = a.guard_cnt;
        g.B = a.B; g.D = a.D; g.H = a.H; g.W = a.W; g.Cin = CIN; g.Do = a.Do; g.Ho = a.Ho; g.Wo = a.Wo;
        g.ldc = COUT; g.co0 = 0; g.nco = COUT; g.kd = 3; g.kh = 3; g.stride = 2; g.transposed = 0;
        g.relu = a.relu; g.in_c8 = 0; g.out_c4 = 0;
        guard_direct_conv(g);
        return;
    }
    if (tid < 2 * COUT) {
        const int c = tid % COUT;
        *reinterpret_cast<float *>(lds + AFF_OFF + tid * 4) = tid < COUT ? (a.scale ? a.scale[c] : 1.0f) * unscale : (a.shift ? a.shift[c] : 0.0f);
    }
    // the weight fragments, in registers for the kernel's lifetime: slot 4 g + kq = tap (kz, ky, kx), 8 input channels
    f16x8 A[NG][2];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int p = 0; p < 2; ++p) A[g][p] = __builtin_bit_cast(f16x8, reinterpret_cast<const uint4 *>(a.wpk)[(g * 2 + p) * 64 + lane]);
    // this lane's B voxel of K-step g: tap t = 4 g + kq -> plane kz (ring slot chosen per step), byte offset of (row 2 wv + ky,
    // column 2 n + kx) inside a piece plane; the 28th slot reads voxel 0 against zero weights
    int tkz[NG];
    unsigned toff[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int t = 4 * g + kq, kz = t / 9, ky = (t / 3) % 3, kx = t % 3;
        const bool live = t < 27;
        tkz[g] = live ? kz : 0;
        toff[g] = live ? (unsigned)(((2 * wv + ky) * IC + 2 * n + kx) * 16) : 0u;
    }

    // 4 values -> hi, lo (fp16 pairs) of v * s
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
    // staged plane `fslot` (0..3: buffer x plane of the pair) -> ring slot `rslot`
    auto split_plane = [&](int fslot, int rslot) {
        if (a.abl & 2) return;
        for (int P = tid; P < PPIECE; P += NTHREADS) {
            const f32x4 x = *reinterpret_cast<const f32x4 *>(lds + F_OFF + fslot * FPLANE + P * 16);
            unsigned h0, h1, l0, l1;
            split2_quad(x[0], x[1], x[2], x[3], sx, h0, h1, l0, l1);
            *reinterpret_cast<uint2 *>(lds + R_OFF + (rslot * 2 + 0) * IPART + P * 8) = make_uint2(h0, h1);
            *reinterpret_cast<uint2 *>(lds + R_OFF + (rslot * 2 + 1) * IPART + P * 8) = make_uint2(l0, l1);
        }
    };

    // copies of one plane: piece P = (i * 8 + wv) * 64 + lane -> (voxel, half)
    constexpr int IPW = (PCOPY + NW - 1) / NW;
    int loc[IPW];
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
        const int P = (i * NW + wv) * 64 + lane;
        const int v = P >> 1;
        loc[i] = (v < NV && P < PPIECE) ? ((v % IC) | ((v / IC) << 8) | ((P & 1) << 16)) : -1;
    }
    const int64_t plane_in = (int64_t)a.H * a.W * CIN;
    const unsigned plane_bytes = (unsigned)(plane_in * 4);
    const int64_t G = (int64_t)a.ncols * a.Do;
    int64_t g0, g1;
    {
        const int nb = gridDim.x;
        int r = blockIdx.x;
        if ((nb & 7) == 0) r = (blockIdx.x & 7) * (nb >> 3) + (blockIdx.x >> 3);
        g0 = G * r / nb; g1 = G * (r + 1) / nb;
    }
    const int c0 = kq * 4;
    float vmax = 0.0f;
    __syncthreads();                     // the affine table

    for (int64_t g = g0; g < g1;) {
        const int col = (int)(g / a.Do);
        const int j0 = (int)(g - (int64_t)col * a.Do);
        const int64_t gend = min(g1, (int64_t)(col + 1) * a.Do);
        const int j1 = (int)(gend - (int64_t)col * a.Do) - 1;
        g = gend;
        const int tx = col % a.tiles_x, ty = (col / a.tiles_x) % a.tiles_y, b = col / (a.tiles_x * a.tiles_y);
        const int ix0 = 2 * tx * TX - 1, iy0 = 2 * ty * TY - 1;       // input column / row of local (0, 0)
        unsigned voff[IPW];
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            const int gx = ix0 + (loc[i] & 255), gy = iy0 + ((loc[i] >> 8) & 255), h = (loc[i] >> 16) & 1;
            const bool ok = loc[i] >= 0 && (unsigned)gx < (unsigned)a.W && (unsigned)gy < (unsigned)a.H;
            voff[i] = ok ? (unsigned)(((gy * a.W + gx) * CIN + h * 4) * 4) : 0xffffff00u;
        }
        // input plane p (zeros outside the volume) -> staging slot fslot
        auto issue_plane = [&](int p, int fslot) {
            const bool ok = p >= 0 && p < a.D;
            const mvs_srd_t srd = make_srd(a.in + ((int64_t)b * a.D + (ok ? p : 0)) * plane_in, ok ? plane_bytes : 0u);
#pragma unroll
            for (int i = 0; i < IPW; ++i) {
                if (i * NW + wv >= PCOPY) continue;   // wave-uniform
                if (a.abl & 1) continue;
                glds16_buf(voff[i], srd, 0u, lds_base + (unsigned)(F_OFF + fslot * FPLANE + (i * NW + wv) * 1024));
            }
        };
        // pair of step j = input planes 2 j, 2 j + 1 -> staging buffer j & 1
        auto issue_pair = [&](int j) { issue_plane(2 * j, (j & 1) * 2); issue_plane(2 * j + 1, (j & 1) * 2 + 1); };
        auto ring = [](int p) { return (p + 3) % 3; };               // plane p -> ring slot (p >= -1)

        // prologue: plane 2 j0 - 1 (through the other buffer) and the pair of step j0
        issue_plane(2 * j0 - 1, ((j0 + 1) & 1) * 2);
        issue_pair(j0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                 // P1 (also: everybody is done with the previous run's ring)
        split_plane(((j0 + 1) & 1) * 2, ring(2 * j0 - 1));
        split_plane((j0 & 1) * 2, ring(2 * j0));
        split_plane((j0 & 1) * 2 + 1, ring(2 * j0 + 1));
        __syncthreads();                 // P2 = B(j0)
        if (j0 < j1) issue_pair(j0 + 1);

        for (int j = j0; j <= j1; ++j) {
            // ============================================================ phase 1: the output plane j
            {
                unsigned rb[3];
#pragma unroll
                for (int kz = 0; kz < 3; ++kz) rb[kz] = lds_base + (unsigned)(R_OFF + ring(2 * j - 1 + kz) * 2 * IPART);
                f16x8 Bh[NG], Bl[NG];
                static_for<0, NG>([&](auto gc) {
                    constexpr int gk = decltype(gc)::value;
                    const unsigned ad = (tkz[gk] == 0 ? rb[0] : (tkz[gk] == 1 ? rb[1] : rb[2])) + toff[gk];
                    Bh[gk] = __builtin_bit_cast(f16x8, lds_read_b128<0>(ad));
                    Bl[gk] = __builtin_bit_cast(f16x8, lds_read_b128<IPART>(ad));
                });
                lds_wait_n<0>();
                static_assert(NG == 7, "the pins below name every fragment");
                asm volatile("" : "+v"(Bh[0]), "+v"(Bl[0]), "+v"(Bh[1]), "+v"(Bl[1]), "+v"(Bh[2]), "+v"(Bl[2]), "+v"(Bh[3]), "+v"(Bl[3]));
                asm volatile("" : "+v"(Bh[4]), "+v"(Bl[4]), "+v"(Bh[5]), "+v"(Bl[5]), "+v"(Bh[6]), "+v"(Bl[6]));
                f32x4 e = {0.f, 0.f, 0.f, 0.f}, o = {0.f, 0.f, 0.f, 0.f};
                if (!(a.abl & 4))
                static_for<0, NG>([&](auto gc) {
                    constexpr int gk = decltype(gc)::value;
                    f32x4 &cc = (gk & 1) ? o : e;
                    cc = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[gk][1], Bh[gk], cc, 0, 0, 0);
                    cc = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[gk][0], Bl[gk], cc, 0, 0, 0);
                    cc = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[gk][0], Bh[gk], cc, 0, 0, 0);
                });
                // with the MFMAs in flight: the next pair (issued a step ago) has landed; the copies of the pair after next go out
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (j + 2 <= j1) issue_pair(j + 2);        // (buffer j & 1: split a step ago)
                const int oy = ty * TY + wv, ox = tx * TX + n;
                if (oy < a.Ho && ox < a.Wo && !(a.abl & 8)) {
                    const float4 sc = *reinterpret_cast<const float4 *>(lds + AFF_OFF + c0 * 4);
                    const float4 sh = *reinterpret_cast<const float4 *>(lds + AFF_OFF + (COUT + c0) * 4);
                    f32x4 v = {e[0] + o[0], e[1] + o[1], e[2] + o[2], e[3] + o[3]};
                    if (unscale2 != 1.0f) { v[0] *= unscale2; v[1] *= unscale2; v[2] *= unscale2; v[3] *= unscale2; }
                    v[0] = v[0] * sc.x + sh.x; v[1] = v[1] * sc.y + sh.y; v[2] = v[2] * sc.z + sh.z; v[3] = v[3] * sc.w + sh.w;
                    if (a.relu == 1) { v[0] = relu_nan(v[0]); v[1] = relu_nan(v[1]); v[2] = relu_nan(v[2]); v[3] = relu_nan(v[3]); }
                    else if (a.relu == 2) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] = v[q] > 0.f ? v[q] : v[q] * 0.1f;
                    }
                    const int64_t oo = ((((int64_t)b * a.Do + j) * a.Ho + oy) * a.Wo + ox) * COUT + c0;
                    *reinterpret_cast<float4 *>(a.out + oo) = make_float4(v[0], v[1], v[2], v[3]);
                    vmax = amax4_nan(vmax, v[0], v[1], v[2], v[3]);
                }
            }
            __syncthreads();             // C: planes 2j - 1, 2j are dead; the next pair is staged (every wave has waited for its copies)
            // ============================================================ phase 2
            if (j < j1) {
                split_plane(((j + 1) & 1) * 2, ring(2 * j + 2));
                split_plane(((j + 1) & 1) * 2 + 1, ring(2 * j + 3));
            }
            __syncthreads();             // B of the next step
        }
    }
    publish_absmax(a.out_absmax, vmax);
}

// the launcher conv_split.hip calls for (kd 3, stride 2, 8 -> 16, two-piece form): same packed weights
int launch_conv_s2_march(const float *in, const void *in_absmax, const void *packed, const float *w_iscale, const float *scale,
                         const float *shift, int relu, int B, int D, int H, int W, float *out, void *out_absmax,
                         unsigned long long *guard_cnt, const unsigned *run_flag, hipStream_t st) {
    S2MArgs a;
    a.in = in; a.wpk = static_cast<const unsigned char *>(packed); a.w_iscale = w_iscale; a.scale = scale; a.shift = shift;
    a.in_absmax = static_cast<const unsigned *>(in_absmax); a.out_absmax = static_cast<unsigned *>(out_absmax);
    a.guard_cnt = guard_cnt; a.run_flag = run_flag; a.out = out;
    a.B = B; a.D = D; a.H = H; a.W = W; a.relu = relu;
    a.Do = (D - 1) / 2 + 1; a.Ho = (H - 1) / 2 + 1; a.Wo = (W - 1) / 2 + 1;
    a.tiles_x = (a.Wo + s2m::TX - 1) / s2m::TX; a.tiles_y = (a.Ho + s2m::TY - 1) / s2m::TY;
    const int64_t ncols = (int64_t)B * a.tiles_x * a.tiles_y;
    if (ncols <= 0 || ncols >= (1 << 30) || (int64_t)H * W * s2m::CIN * 4 >= 0xffffff00LL) return MVS_EUNSUPPORTED;
    a.ncols = (int)ncols;
#ifdef MVS_TUNING
    a.abl = getenv("MVS_S2M_ABL") ? atoi(getenv("MVS_S2M_ABL")) : 0;
#else
    a.abl = 0;
#endif
    const int64_t Gs = ncols * a.Do;
    const int n_cu = device_cu_count();
    hipLaunchKernelGGL(conv_s2_march_kernel, dim3((unsigned)(Gs < n_cu ? Gs : n_cu)), dim3(s2m::NTHREADS), 0, st, a);
    return check_launch("mvs_conv_split_f16_f32(stride 2, marching)");
}

}  // namespace mvs
