// The tail of CostRegNet as ONE kernel: x = conv0 + conv11(x) followed by `prob` (MVSNet/models/mvsnet.py:79-81,
// 89-93: ConvTranspose3d 16 -> 8, k3 s2 p1 op1 + BN + ReLU, the skip add AFTER the ReLU, then Conv3d 8 -> 1, k3 p1,
// bias).  As three launches the full-resolution 8-channel volume d11 goes to HBM and comes back (727 MB each way at
// config 2: conv11 0.35 ms at 4.7 TB/s, `prob` 0.25 ms); here it lives in LDS for the two z planes a step produces.
//
// Work unit = a column of the volume, marched along z: 9 x 17 voxels of conv11's INPUT (rows 7 ty - 1 .. 7 ty + 7,
// x 15 tx - 1 .. 15 tx + 15) give a 16-row x 32-voxel window of its output d11 (rows 14 ty - 1 .. , x 30 tx - 1 ..),
// which gives 14 x 30 `prob` outputs: neighbouring columns overlap by one input voxel (conv11 is recomputed 1.22x,
// the skip volume is read 1.22x -- mostly from the L2 of the XCD that owns the neighbouring columns).  A step =
// one input plane j: d11 planes 2 j and 2 j + 1, `prob` planes 2 j - 1 and 2 j.
//
//   conv11: the parity-class form of deconv_split.hip (out[o] += in[i] w[k], o = 2 i - 1 + k) on the fp16 matrix
//     pipe with two-piece operands (conv_f16x3.hip), shifted by one voxel so that the window starts on an ODD output
//     row / column: wave w owns the odd row 2 (J + w) + 1 (taps k = 2 on input row w, k = 0 on row w + 1) and the
//     even row 2 (J + w + 1) (k = 1 on row w + 1); MFMA rows = (x parity, 8 channels) as there.
//   epilogue: affine, ReLU, skip add, ZERO outside the volume (`prob` pads with zeros), then scaled and split into
//     two fp16 pieces straight into the LDS planes `prob` reads.  The scale is a power of two from a BOUND on |d11|
//     (absmax blocks of conv11's input and of the skip volume, the weights' largest magnitude, the affine): no
//     reduction over the values themselves; a bound 2^k too high costs k of the 18 bits of headroom the two-piece
//     form has before its relative bound turns into the absolute one (2^-40 of the scale's range).
//   prob: MFMA rows = (kz, ky) tap rows (11 of 16: row 4 kz + ky), K = (kx slot, 8 channels), N = 16 voxels along x:
//     one product triple per d11 row and 16 voxels gives D[kz][ky][x]; a wave walks 9 consecutive rows and forms
//     S_kz[y] = D_y[kz][0] + D_{y+1}[kz][1] + D_{y+2}[kz][2] in registers (lane quad = kz), leaves it in LDS, and
//     out[z] = S_0[z - 1] + S_1[z] + S_2[z + 1] is summed one step later (it needs the next plane anyway).
//
// Every CU takes a contiguous range of (column, step) pairs -- 115.5 steps each at config 2 -- so the 308 columns of
// that volume spread evenly over 256 CUs; a range that begins inside a column recomputes one step (the S sums of the
// plane before its first output).  8 multiplying + 4 copy waves, three barriers per step.
//
// Range guard: conv11's verdict on its input block (conv_guard.h), the same verdict on the skip volume's block, the
// weights' and the bound's finiteness.  On any failure the launch writes `fallback` = 1 and returns; the caller has
// enqueued the unfused layers behind it, which run only then (their `run_flag`).
#include "conv_split_common.h"
#include "conv_guard.h"

namespace mvs {

namespace tail {
constexpr int CIN = 16, COUT = 8;
constexpr int RI = 9, XI = 17, NVI = RI * XI, NVIP = 160;     // staged input voxels of a plane (153, padded)
constexpr int NPIECE = 2 * NVIP * 2;                          // 16-byte pieces: (chunk, voxel, half)
constexpr int NCOPY = NPIECE / 64;                            // 10 wave copies
constexpr int NK = 9;                                         // K-steps of conv11: classes (pz, row kind) of 1, 2, 2, 4
constexpr int WBYTES = NK * 2 * 1024;
constexpr int FBYTES = NCOPY * 1024;
constexpr int IPART = 2 * NVIP * 16;                          // one piece plane of one input plane: [chunk][voxel][8 fp16]
constexpr int CR = 16, CX = 36, CPART = CR * CX * 16;         // d11 piece plane: [row][x (32 + zero pad)][8 fp16]
constexpr int PR = 14, PX = 30;                               // `prob` outputs of a column and step: rows, voxels
constexpr int SPLANE = 3 * PR * 32;                           // floats: [kz][row][x]
constexpr int W_OFF = 0, F_OFF = W_OFF + WBYTES, I_OFF = F_OFF + FBYTES, C_OFF = I_OFF + 4 * IPART,
              S_OFF = C_OFF + 4 * CPART, AFF_OFF = S_OFF + 4 * SPLANE * 4, LDS_BYTES = AFF_OFF + 2 * COUT * 4;
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
constexpr int NTHREADS = 768, NCW = 4;

// conv11 class c = (pz, kind): kind 0 = the wave's odd output row (two y taps), kind 1 = its even row (one)
__host__ __device__ constexpr int pz(int c) { return c >> 1; }
__host__ __device__ constexpr int kind(int c) { return c & 1; }
__host__ __device__ constexpr int nty(int c) { return kind(c) == 0 ? 2 : 1; }
__host__ __device__ constexpr int ntap(int c) { return (1 + pz(c)) * nty(c) * 2; }
__host__ __device__ constexpr int nk(int c) { return ntap(c) / 2; }          // slots = taps x 2 chunks, 4 slots per K-step
__host__ __device__ constexpr int kbase(int c) { int s = 0; for (int i = 0; i < c; ++i) s += nk(i); return s; }
__host__ __device__ constexpr int cls_of(int g) { int c = 0; while (g >= kbase(c + 1)) ++c; return c; }
static_assert(kbase(4) == NK, "K-steps");
}  // namespace tail

struct TailArgs {
    const float *in;            // conv11's input [B, Di, Hi, Wi, 16]
    const float *skip;          // [B, 2 Di, 2 Hi, 2 Wi, 8]
    const unsigned char *wpk;   // [K-step][hi, lo][lane][8 fp16] (mvs_costreg_tail_pack_weights_f32)
    const float *w_iscale;      // behind the fragments: what undoes the weights' scale (NaN: weights not finite), their absmax bits
    const float *scale, *shift; // conv11's folded BatchNorm (8 each; NULL = 1 / 0)
    const float *pw;            // prob weight (1, 8, 3, 3, 3)
    const float *pscale, *pshift;   // prob's affine (NULL = 1 / 0; shift = the bias)
    const unsigned *in_absmax, *skip_absmax;
    unsigned *fallback;         // set to 1 when the launch declines (range guard)
    float *out;                 // [B, 2 Di, 2 Hi, 2 Wi]
    int B, Di, Hi, Wi;
    int tiles_x, tiles_y, ncols;
};

__global__ __launch_bounds__(tail::NTHREADS) void costreg_tail_kernel(TailArgs a) {
    using namespace tail;
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4;
    const unsigned lds_base = (unsigned)(uintptr_t)lds;
    const bool copier = wv >= 8;
    const int cw = wv - 8;
    const int D = 2 * a.Di, H = 2 * a.Hi, W = 2 * a.Wi;

    // ---------------------------------------------------------------- range guard, operand scales (wave-uniform)
    const AbsmaxVerdict vin = absmax_verdict(a.in_absmax), vsk = absmax_verdict(a.skip_absmax);
    const float isw = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a.w_iscale[0])));
    const unsigned wmax_bits = (unsigned)__builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a.w_iscale[1]));
    // prob weights: largest magnitude (216 values, every wave for itself)
    float pwm = 0.0f;
    for (int i = lane; i < 27 * COUT; i += 64) pwm = max_nan(pwm, __builtin_fabsf(a.pw[i]));
#pragma unroll
    for (int o = 32; o; o >>= 1) pwm = max_nan(pwm, __shfl_xor(pwm, o));
    const unsigned pw_bits = (unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(pwm));
    float scmax = 0.0f, shmax = 0.0f;
    for (int c = 0; c < COUT; ++c) {
        scmax = max_nan(scmax, __builtin_fabsf(a.scale ? a.scale[c] : 1.0f));
        shmax = max_nan(shmax, __builtin_fabsf(a.shift ? a.shift[c] : 0.0f));
    }
    // |d11| <= max|scale| * (8 taps x 16 channels) * max|in| * max|w| + max|shift| + max|skip|
    const float bound = scmax * 128.0f * __uint_as_float(vin.bits) * __uint_as_float(wmax_bits) * 1.0625f + shmax + __uint_as_float(vsk.bits);
    const unsigned bound_bits = (unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(bound));
    const int xe = absmax_exponent(vin.bits);
    const int ce = absmax_exponent(bound_bits) + 1;                 // d11's operand scale: 2^(14 - ce)
    const int pe = absmax_exponent(pw_bits);                        // prob weights': 2^(14 - pe)
    const int E11 = xe - 14 + (int)((__builtin_bit_cast(unsigned, isw) >> 23) & 255u) - 127;   // undoes conv11's two scales
    const int EP = ce + pe - 28;                                     // undoes prob's
    const bool decline = vin.code != 0 || vsk.code != 0 || isw != isw || !(bound_bits < 0x7f800000u) || !(pw_bits < 0x7f800000u) ||
                         EP < -120 || EP > 120 || ce > 126;
    if (decline) {
        if (blockIdx.x == 0 && tid == 0) *a.fallback = 1u;
        return;
    }
    const float sx = pow2f(14 - xe), sc11 = pow2f(14 - ce), spw = pow2f(14 - pe), unp = pow2f(EP);
    const int e1 = E11 < -60 ? -60 : (E11 > 60 ? 60 : E11), e2 = E11 - e1 < -126 ? -126 : (E11 - e1 > 127 ? 127 : E11 - e1);
    const float unscale = pow2f(e1), unscale2 = pow2f(e2);
    if (tid < 2 * COUT) {
        const int c = tid % COUT;
        const float v = tid < COUT ? (a.scale ? a.scale[c] : 1.0f) * unscale : (a.shift ? a.shift[c] : 0.0f);
        *reinterpret_cast<float *>(lds + AFF_OFF + tid * 4) = v;
    }
    for (int i = tid; i < WBYTES / 16; i += NTHREADS)      // conv11's fragments stay in LDS for the kernel's lifetime
        *reinterpret_cast<uint4 *>(lds + W_OFF + i * 16) = reinterpret_cast<const uint4 *>(a.wpk)[i];
    // zero the d11 planes once: their pad columns (x 32..35) are read against zero weights and must stay finite
    for (int i = tid; i < 4 * CPART / 16; i += NTHREADS) *reinterpret_cast<uint4 *>(lds + C_OFF + i * 16) = make_uint4(0, 0, 0, 0);
    const float psc = a.pscale ? a.pscale[0] : 1.0f, psh = a.pshift ? a.pshift[0] : 0.0f;

    // ---------------------------------------------------------------- this CU's range of (column, step) pairs
    const int64_t G = (int64_t)a.ncols * a.Di;
    int64_t g0, g1;
    {
        const int nb = gridDim.x;
        int r = blockIdx.x;
        if ((nb & 7) == 0) r = (blockIdx.x & 7) * (nb >> 3) + (blockIdx.x >> 3);     // XCD x: a contiguous share of the columns
        g0 = G * r / nb; g1 = G * (r + 1) / nb;
    }
    const int64_t plane_in = (int64_t)a.Hi * a.Wi * CIN;
    const unsigned plane_bytes = (unsigned)(plane_in * 4);

    // split pass (all 12 waves): piece P = tid < 640: 4 floats of the staging buffer -> 4 + 4 fp16 of the plane slot `sel`
    auto split_pass = [&](int sel) {
        if (tid < NPIECE) {
            const f32x4 x = *reinterpret_cast<const f32x4 *>(lds + F_OFF + tid * 16);
            typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
            f16x4 h, l;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float xs = x[i] * sx;
                h[i] = (_Float16)xs;
                l[i] = (_Float16)(xs - (float)h[i]);
            }
            *reinterpret_cast<f16x4 *>(lds + I_OFF + (sel * 2 + 0) * IPART + tid * 8) = h;
            *reinterpret_cast<f16x4 *>(lds + I_OFF + (sel * 2 + 1) * IPART + tid * 8) = l;
        }
    };

    // output phase (all 12 waves): the `prob` planes whose three S sums are complete.  cur / prev: slots of the step
    // that just finished and of the one before it; first: prev holds nothing (z = 0: S of plane -1 is zero);
    // emit: this CU owns the outputs (not a warm-up step); last: the column's last step -- also plane 2 Di - 1.
    auto output_phase = [&](int b, int ty, int tx, int j, int cur, bool first, bool emit, bool last) {
        if (!emit) return;
        const float *Sc = reinterpret_cast<const float *>(lds + S_OFF) + cur * 2 * SPLANE;
        const float *Sp = reinterpret_cast<const float *>(lds + S_OFF) + (cur ^ 1) * 2 * SPLANE;
        const int nplanes = last ? 3 : 2;
        for (int o = tid; o < nplanes * PR * PX; o += NTHREADS) {
            const int pl = o / (PR * PX), rem = o - pl * (PR * PX), row = rem / PX, xx = rem - row * PX;
            const int so = row * 32 + xx;
            float v;
            int z;
            if (pl == 0) {          // z = 2 j - 1 = S0[2j-2] + S1[2j-1] + S2[2j]
                if (first) continue;
                z = 2 * j - 1;
                v = (Sp[0 * SPLANE + 0 * PR * 32 + so] + Sp[1 * SPLANE + 1 * PR * 32 + so]) + Sc[0 * SPLANE + 2 * PR * 32 + so];
            } else if (pl == 1) {   // z = 2 j = S0[2j-1] + S1[2j] + S2[2j+1]
                z = 2 * j;
                const float s0 = first ? 0.0f : Sp[1 * SPLANE + 0 * PR * 32 + so];
                v = (s0 + Sc[0 * SPLANE + 1 * PR * 32 + so]) + Sc[1 * SPLANE + 2 * PR * 32 + so];
            } else {                // z = 2 j + 1 = D - 1: S0[2j] + S1[2j+1]
                z = 2 * j + 1;
                v = Sc[0 * SPLANE + 0 * PR * 32 + so] + Sc[1 * SPLANE + 1 * PR * 32 + so];
            }
            const int oy = 14 * ty + row, ox = 30 * tx + xx;
            if (oy < H && ox < W) a.out[(((int64_t)b * D + z) * H + oy) * W + ox] = (v * unp) * psc + psh;
        }
    };

    auto decode_col = [&](int col, int &b, int &ty, int &tx) {
        tx = col % a.tiles_x;
        const int r = col / a.tiles_x;
        ty = r % a.tiles_y;
        b = r / a.tiles_y;
    };

    // ---------------------------------------------------------------- per-wave constants
    // copy waves: piece P = (i * 4 + cw) * 64 + lane -> (chunk, voxel, half)
    constexpr int IPW = (NCOPY + NCW - 1) / NCW;
    int loc[IPW];
    unsigned voff[IPW];
    if (copier) {
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            const int P = (i * NCW + cw) * 64 + lane;
            const int v = (P % (2 * NVIP)) >> 1, c = P / (2 * NVIP);
            const bool okv = v < NVI;
            loc[i] = okv ? ((v % XI) | ((v / XI) << 8) | ((P & 1) << 16) | (c << 17)) : -1;
        }
    }
    // multiplying waves, conv11: B voxel of this lane for K-step g: (plane j or j + 1, row wv + dy, x n + dx), chunk cc
    unsigned tapo[NK];      // byte offset inside an input plane's piece plane; bit 31: plane j + 1
    static_for<0, NK>([&](auto gc) {
        constexpr int g = decltype(gc)::value, c = cls_of(g), ks = g - kbase(c);
        const int sl = 4 * ks + kq, ti = sl >> 1, cc = sl & 1;
        const int dx = ti & 1, ty_ = (ti >> 1) % nty(c), tz = (ti >> 1) / nty(c);
        const int dy = kind(c) == 0 ? ty_ : 1;
        const int dz = pz(c) == 1 && tz == 1 ? 1 : 0;
        const bool live = sl < ntap(c) * 2;
        tapo[g] = live ? (unsigned)(cc * NVIP * 16 + ((wv + dy) * XI + n + dx) * 16) | (dz ? 0x80000000u : 0u) : 0u;
    });
    // prob: this lane's A fragment = w[kz = m >> 2][ky = m & 3][kx = kq][8 channels], two fp16 pieces of w * 2^(14 - pe)
    f16x8 pah, pal;
    {
        const int m = lane & 15, kz = m >> 2, ky = m & 3;
        const bool live = kz < 3 && ky < 3 && kq < 3;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float x = live ? a.pw[c * 27 + kz * 9 + ky * 3 + kq] * spw : 0.0f;
            const _Float16 h = (_Float16)x;
            pah[c] = h;
            pal[c] = (_Float16)(x - (float)h);
        }
    }
    const int p_pl = wv & 1, p_xb = (wv >> 1) & 1, p_hf = (wv >> 2) & 1;     // prob phase: d11 plane, x block, row half
    const int c0 = (kq & 1) * 4, pxr = kq >> 1;                              // conv11 epilogue: channels, x parity row group

    __syncthreads();      // affine table, zeroed d11 planes

    // ================================================================ runs: (column, j0 .. j1) pieces of [g0, g1)
    for (int64_t g = g0; g < g1;) {
        const int col = (int)(g / a.Di);
        const int j0 = (int)(g - (int64_t)col * a.Di);
        const int64_t gend = min(g1, (int64_t)(col + 1) * a.Di);
        const int j1 = (int)(gend - (int64_t)col * a.Di) - 1;          // last step of the run (inclusive)
        const int js = j0 > 0 ? j0 - 1 : 0;                            // a run inside a column recomputes one step
        g = gend;
        int b, ty, tx;
        decode_col(col, b, ty, tx);
        const int Jb = 7 * ty - 1, Ib = 15 * tx - 1;                   // input row / x of local (0, 0)

        mvs_srd_t srd = make_srd(a.in, 0);
        auto plane_srd = [&](int j) {
            const bool ok = j < a.Di;
            srd = make_srd(a.in + ((int64_t)b * a.Di + (ok ? j : 0)) * plane_in, ok ? plane_bytes : 0u);
        };
        auto issue_plane = [&]() {
#pragma unroll
            for (int i = 0; i < IPW; ++i) {
                if (i * NCW + cw >= NCOPY) continue;   // wave-uniform
                glds16_buf(voff[i], srd, 0u, lds_base + (unsigned)(F_OFF + (i * NCW + cw) * 1024));
            }
        };
        if (copier) {
#pragma unroll
            for (int i = 0; i < IPW; ++i) {
                const int gx = Ib + (loc[i] & 255), gy = Jb + ((loc[i] >> 8) & 255);
                const int h = (loc[i] >> 16) & 1, c = (loc[i] >> 17) & 1;
                const bool ok = loc[i] >= 0 && gx >= 0 && gx < a.Wi && gy >= 0 && gy < a.Hi;
                voff[i] = ok ? (unsigned)((((int64_t)gy * a.Wi + gx) * CIN + c * 8 + h * 4) * 4) : 0xffffff00u;
            }
            plane_srd(js);
            issue_plane();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();                 // P1: plane js is in the staging buffer
        split_pass(js & 1);
        __syncthreads();                 // P2: ... and split; the staging buffer is free
        if (copier) {
            plane_srd(js + 1);
            issue_plane();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }

        for (int j = js; j <= j1; ++j) {
            // skip-connection values of the four outputs this lane will write: requested here, consumed in the epilogue
            float4 res[4];
            bool inside[4];
            if (!copier) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int gy = 14 * ty - 1 + 2 * wv + kind(c), gx = 30 * tx - 1 + 2 * n + pxr;
                    inside[c] = gy >= 0 && gy < H && gx >= 0 && gx < W;
                    const int64_t o = ((((int64_t)b * D + 2 * j + pz(c)) * H + gy) * W + gx) * COUT + c0;
                    res[c] = inside[c] ? *reinterpret_cast<const float4 *>(a.skip + o) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            __syncthreads();             // A: plane j + 1 is staged; the S sums of step j - 1 are in LDS
            if (j > js) output_phase(b, ty, tx, j - 1, (j - 1) & 1, j - 1 == 0, j - 1 >= j0, false);
            split_pass((j + 1) & 1);
            __syncthreads();             // B: both input planes are split; the staging buffer is free
            if (copier) {
                if (j < j1) {
                    plane_srd(j + 2);
                    issue_plane();
                }
                __syncthreads();         // C
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (before barrier A of the next step)
                continue;
            }
            // ---------------------------------------------------------------- conv11: 9 K-steps x 3 products
            {
                const unsigned aA = lds_base + (unsigned)(W_OFF + lane * 16);
                const unsigned pj = lds_base + (unsigned)(I_OFF + ((j & 1) * 2) * IPART);
                const unsigned pj1 = lds_base + (unsigned)(I_OFF + (((j + 1) & 1) * 2) * IPART);
                f32x4 acc[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
                f16x8 A[2][2], Bf[2][2];
                auto read_ab = [&](auto gc) {
                    constexpr int gk = decltype(gc)::value;
                    const unsigned ad = ((tapo[gk] & 0x80000000u) ? pj1 : pj) + (tapo[gk] & 0x7fffffffu);
                    A[gk & 1][0] = __builtin_bit_cast(f16x8, lds_read_b128<(gk * 2 + 0) * 1024>(aA));
                    A[gk & 1][1] = __builtin_bit_cast(f16x8, lds_read_b128<(gk * 2 + 1) * 1024>(aA));
                    Bf[gk & 1][0] = __builtin_bit_cast(f16x8, lds_read_b128<0>(ad));
                    Bf[gk & 1][1] = __builtin_bit_cast(f16x8, lds_read_b128<IPART>(ad));
                };
                read_ab(std::integral_constant<int, 0>{});
                static_for<0, NK>([&](auto gc) {
                    constexpr int gk = decltype(gc)::value, c = cls_of(gk);
                    lds_wait_n<0>();
                    asm volatile("" : "+v"(A[gk & 1][0]), "+v"(A[gk & 1][1]), "+v"(Bf[gk & 1][0]), "+v"(Bf[gk & 1][1]));
                    if constexpr (gk + 1 < NK) read_ab(std::integral_constant<int, gk + 1>{});
                    __builtin_amdgcn_sched_barrier(0);
                    f32x4 &cc = acc[c];
                    cc = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[gk & 1][1], Bf[gk & 1][0], cc, 0, 0, 0);
                    cc = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[gk & 1][0], Bf[gk & 1][1], cc, 0, 0, 0);
                    cc = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[gk & 1][0], Bf[gk & 1][0], cc, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                });
                // epilogue: affine, ReLU, skip add; zero outside the volume; scale, split, into the d11 planes
                const float4 sc = *reinterpret_cast<const float4 *>(lds + AFF_OFF + c0 * 4);
                const float4 sh = *reinterpret_cast<const float4 *>(lds + AFF_OFF + (COUT + c0) * 4);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    f32x4 v = acc[c];
                    if (unscale2 != 1.0f) { v[0] *= unscale2; v[1] *= unscale2; v[2] *= unscale2; v[3] *= unscale2; }
                    v[0] = v[0] * sc.x + sh.x; v[1] = v[1] * sc.y + sh.y;
                    v[2] = v[2] * sc.z + sh.z; v[3] = v[3] * sc.w + sh.w;
                    v[0] = relu_nan(v[0]); v[1] = relu_nan(v[1]); v[2] = relu_nan(v[2]); v[3] = relu_nan(v[3]);
                    v[0] += res[c].x; v[1] += res[c].y; v[2] += res[c].z; v[3] += res[c].w;
                    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
                    f16x4 h, l;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float xs = inside[c] ? v[i] * sc11 : 0.0f;
                        h[i] = (_Float16)xs;
                        l[i] = (_Float16)(xs - (float)h[i]);
                    }
                    const int lr = 2 * wv + kind(c), xl = 2 * n + pxr;
                    unsigned char *dst = lds + C_OFF + (pz(c) * 2) * CPART + (lr * CX + xl) * 16 + (kq & 1) * 8;
                    *reinterpret_cast<f16x4 *>(dst) = h;
                    *reinterpret_cast<f16x4 *>(dst + CPART) = l;
                }
            }
            __syncthreads();             // C: d11 planes 2 j, 2 j + 1 are in LDS
            // ---------------------------------------------------------------- prob: 9 d11 rows, S sums of 7 output rows
            {
                const unsigned cb = lds_base + (unsigned)(C_OFF + (p_pl * 2) * CPART + ((7 * p_hf) * CX + p_xb * 16 + n + kq) * 16);
                f16x8 Bh[2], Bl[2];
                auto read_row = [&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    Bh[i & 1] = __builtin_bit_cast(f16x8, lds_read_b128<i * CX * 16>(cb));
                    Bl[i & 1] = __builtin_bit_cast(f16x8, lds_read_b128<i * CX * 16 + CPART>(cb));
                };
                float s[7];
                f32x4 Dm1 = (f32x4){0.f, 0.f, 0.f, 0.f}, Dm2 = (f32x4){0.f, 0.f, 0.f, 0.f};     // D of the two rows before
                read_row(std::integral_constant<int, 0>{});
                static_for<0, 9>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    lds_wait_n<0>();
                    asm volatile("" : "+v"(Bh[i & 1]), "+v"(Bl[i & 1]));
                    if constexpr (i + 1 < 9) read_row(std::integral_constant<int, i + 1>{});
                    __builtin_amdgcn_sched_barrier(0);
                    f32x4 d = (f32x4){0.f, 0.f, 0.f, 0.f};
                    d = __builtin_amdgcn_mfma_f32_16x16x32_f16(pal, Bh[i & 1], d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_16x16x32_f16(pah, Bl[i & 1], d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_16x16x32_f16(pah, Bh[i & 1], d, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    // S[i - 2] = D_{i-2}[0] + D_{i-1}[1] + D_i[2]
                    if constexpr (i >= 2) s[i - 2] = (Dm2[0] + Dm1[1]) + d[2];
                    Dm2 = Dm1;
                    Dm1 = d;
                });
                if (kq < 3) {
                    float *Sd = reinterpret_cast<float *>(lds + S_OFF) + ((j & 1) * 2 + p_pl) * SPLANE + kq * PR * 32 + (7 * p_hf) * 32 + p_xb * 16 + n;
#pragma unroll
                    for (int i = 0; i < 7; ++i) Sd[i * 32] = s[i];
                }
            }
        }
        __syncthreads();                 // E: the S sums of the run's last step are in LDS
        output_phase(b, ty, tx, j1, j1 & 1, j1 == 0, j1 >= j0, j1 == a.Di - 1);
        __syncthreads();                 // (the next run's first prob phase lies three barriers ahead; its split pass does not touch S)
    }
}

// PyTorch ConvTranspose3d weight (16, 8, 3, 3, 3) -> [K-step g][hi, lo][lane][8 fp16] of w * 2^(14 - exponent(max |w|));
// lane (mrow, kq): slot 4 ks + kq of g's class = (tap (tz, ty, dx), chunk cc), input channel cc * 8 + i; MFMA row
// mrow = (x parity row group, output channel): group 0 = the ODD output x of the pair (taps k = 2 on dx = 0, k = 0 on
// dx = 1), group 1 = the even one (k = 1 on dx = 1).  Thread 0 writes the trailer {what undoes the scale, max |w| bits}.
__global__ __launch_bounds__(256) void pack_tail_kernel(const float *__restrict__ w, unsigned short *__restrict__ out,
                                                        const unsigned *__restrict__ wmax, float *__restrict__ trailer) {
    using namespace tail;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int we = absmax_exponent(*wmax);
    if (i == 0) {
        trailer[0] = *wmax >= 0x7f800000u ? __builtin_nanf("") : pow2f(we - 14);
        trailer[1] = __uint_as_float(*wmax);
    }
    if (i >= NK * 512) return;
    const int jj = i & 7, lane = (i >> 3) & 63, g = i >> 9;
    const int c = cls_of(g), ks = g - kbase(c);
    const int mrow = lane & 15, kq = lane >> 4, sl = 4 * ks + kq, ti = sl >> 1, cc = sl & 1;
    float x = 0.0f;
    if (sl < ntap(c) * 2) {
        const int dx = ti & 1, ty = (ti >> 1) % nty(c), tz = (ti >> 1) / nty(c);
        const int kz = pz(c) == 0 ? 1 : (tz == 0 ? 2 : 0);
        const int ky = kind(c) == 0 ? (ty == 0 ? 2 : 0) : 1;
        const int grp = mrow >> 3, co = mrow & 7;
        int kx = -1;
        if (grp == 0) kx = dx == 0 ? 2 : 0; else if (dx == 1) kx = 1;
        const int ci = cc * 8 + jj;
        if (kx >= 0) x = w[(((int64_t)ci * COUT + co) * 3 + kz) * 9 + ky * 3 + kx];
    }
    x *= pow2f(14 - we);
    const _Float16 h = (_Float16)x;
    const _Float16 l = (_Float16)(x - (float)h);
    unsigned short *o = out + ((size_t)g * 2) * 512 + lane * 8 + jj;
    o[0] = __builtin_bit_cast(unsigned short, h);
    o[512] = __builtin_bit_cast(unsigned short, l);
}

int launch_absmax_word(const float *x, int64_t n, unsigned *word, hipStream_t st);   // conv_f16x3.hip

}  // namespace mvs

using namespace mvs;

extern "C" size_t mvs_costreg_tail_packed_bytes(void) { return (size_t)tail::WBYTES + 16; }

extern "C" int mvs_costreg_tail_pack_weights_f32(const float *conv11_weight, void *packed, void *stream) {
    if (!conv11_weight || !packed) {
        set_error("mvs_costreg_tail_pack_weights_f32: needs the (16, 8, 3, 3, 3) weight of conv11 and a packed buffer");
        return MVS_EINVAL;
    }
    unsigned char *pk = static_cast<unsigned char *>(packed);
    float *trailer = reinterpret_cast<float *>(pk + tail::WBYTES);
    unsigned *wmax = reinterpret_cast<unsigned *>(pk + tail::WBYTES + 8);
    const int rc = launch_absmax_word(conv11_weight, (int64_t)16 * 8 * 27, wmax, as_stream(stream));
    if (rc != MVS_OK) return rc;
    hipLaunchKernelGGL(pack_tail_kernel, dim3((tail::NK * 512 + 255) / 256), dim3(256), 0, as_stream(stream), conv11_weight,
                       reinterpret_cast<unsigned short *>(pk), wmax, trailer);
    return check_launch("mvs_costreg_tail_pack_weights_f32");
}

extern "C" int mvs_costreg_tail_supported(int B, int Di, int Hi, int Wi) {
    if (B <= 0 || Di <= 0 || Hi <= 0 || Wi <= 0) return 0;
    if ((int64_t)Hi * Wi * 16 * 4 >= 0xffffff00LL) return 0;                 // 32-bit offsets inside an input plane
    const int64_t cols = (int64_t)B * ((2 * Hi + 13) / 14) * ((2 * Wi + 29) / 30);
    return cols * Di < (1LL << 40) && cols < (1 << 30) ? 1 : 0;
}

extern "C" int mvs_costreg_tail_f16_f32(const float *in, const void *in_absmax, const float *skip, const void *skip_absmax,
                                        const void *packed_tail, const float *scale, const float *shift, const float *prob_weight,
                                        const float *prob_scale, const float *prob_shift, int B, int Di, int Hi, int Wi,
                                        float *out_cost, void *fallback_flag, void *stream) {
    if (!in || !in_absmax || !skip || !skip_absmax || !packed_tail || !prob_weight || !out_cost || !fallback_flag ||
        !mvs_costreg_tail_supported(B, Di, Hi, Wi)) {
        set_error("mvs_costreg_tail_f16_f32: invalid argument (conv11 input [B, Di, Hi, Wi, 16] channels-last with its absmax block, "
                  "skip volume [B, 2Di, 2Hi, 2Wi, 8] with its block, the tail pack, prob weight (1, 8, 3, 3, 3), a flag word)");
        return MVS_EINVAL;
    }
    TailArgs a;
    a.in = in; a.skip = skip;
    a.wpk = static_cast<const unsigned char *>(packed_tail);
    a.w_iscale = reinterpret_cast<const float *>(a.wpk + tail::WBYTES);
    a.scale = scale; a.shift = shift; a.pw = prob_weight; a.pscale = prob_scale; a.pshift = prob_shift;
    a.in_absmax = static_cast<const unsigned *>(in_absmax);
    a.skip_absmax = static_cast<const unsigned *>(skip_absmax);
    a.fallback = static_cast<unsigned *>(fallback_flag);
    a.out = out_cost;
    a.B = B; a.Di = Di; a.Hi = Hi; a.Wi = Wi;
    a.tiles_x = (2 * Wi + 29) / 30; a.tiles_y = (2 * Hi + 13) / 14;
    a.ncols = B * a.tiles_x * a.tiles_y;
    const int64_t G = (int64_t)a.ncols * Di;
    const int n_cu = device_cu_count();
    const unsigned grid = (unsigned)(G < n_cu ? G : n_cu);
    hipLaunchKernelGGL(costreg_tail_kernel, dim3(grid), dim3(tail::NTHREADS), 0, as_stream(stream), a);
    return check_launch("mvs_costreg_tail_f16_f32");
}
