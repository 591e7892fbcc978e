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
// plane before its first output).  One 8-wave workgroup per CU (2 waves per SIMD: 228 registers -- conv11's nine weight
// fragments live in registers, its four B fragments serve all nine K-steps); every wave also issues its share of the
// input copies (10 KB per step: no copy waves needed), splits, and writes outputs.  Two barriers per step: conv11 +
// epilogue | prob + split of the next plane + outputs of the step before (three S slots, two staging buffers).
// Measured at config 2 (scripts/exp_tail_fused.py, tuning build MVS_TAIL_ABL): 0.40 ms against 0.59-0.61 for the two
// launches on the same box; with the skip loads removed 0.33, the epilogue 0.32, every MFMA 0.33, all of it 0.14 --
// the phases of a step do not overlap (two waves per SIMD in lockstep), which is what is left.
//
// Range guard: conv11's verdict on its input block (conv_guard.h), the same verdict on the skip volume's block, the
// weights' and the bound's finiteness.  On any failure the launch writes `fallback` = 1 and returns; the caller has
// enqueued the unfused layers behind it, which run only then (their `run_flag`).
#include "conv_split_common.h"
#include "conv_guard.h"

#include <cstdlib>

namespace mvs {

namespace tail {
constexpr int CIN = 16, COUT = 8;
constexpr int RI = 9, XI = 17, NVI = RI * XI, NVIP = 160;     // staged input voxels of a plane (153, padded)
constexpr int NPIECE = 2 * NVIP * 2;                          // 16-byte pieces: (chunk, voxel, half)
constexpr int NCOPY = NPIECE / 64;                            // 10 wave copies
constexpr int NK = 9;                                         // K-steps of conv11: classes (pz, row kind) of 2, 1, 4, 2
constexpr int WBYTES = NK * 2 * 1024;
constexpr int FBYTES = NCOPY * 1024;
constexpr int IPART = 2 * NVIP * 16;                          // one piece plane of one input plane: [chunk][voxel][8 fp16]
constexpr int CR = 16, CX = 36, CPART = CR * CX * 16;         // d11 piece plane: [row][x (32 + zero pad)][8 fp16]
constexpr int PR = 14, PX = 30;                               // `prob` outputs of a column and step: rows, voxels
constexpr int SPLANE = 3 * PR * 32;                           // floats: [kz][row][x]
constexpr int NSS = 3;                                        // S slots (steps j, j - 1, j - 2 are alive at once)
constexpr int F_OFF = 0, I_OFF = F_OFF + 2 * FBYTES, C_OFF = I_OFF + 4 * IPART,
              S_OFF = C_OFF + 4 * CPART, AFF_OFF = S_OFF + NSS * 2 * SPLANE * 4, LDS_BYTES = AFF_OFF + 2 * COUT * 4;
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
constexpr int NTHREADS = 512, NCW = 8;       // eight waves; every wave also issues its share of the input copies

// conv11 class c = (pz, kind): kind 0 = the wave's odd output row (two y taps), kind 1 = its even row (one).
// K-step g = (class, B fragment): a B fragment is the (input plane p = j + p, row wv + dy) pair, its four K slots kq =
// (dx = kq >> 1, chunk = kq & 1) -- the SAME four fragments serve all nine K-steps (a wave reads 8 x 16 bytes of operands per
// step instead of 36; the weight fragments live in registers).
__host__ __device__ constexpr int pz(int c) { return c >> 1; }
__host__ __device__ constexpr int kind(int c) { return c & 1; }
//                                         g:  0  1  2  3  4  5  6  7  8
__host__ __device__ constexpr int g_cls(int g) { constexpr int t[NK] = {0, 0, 1, 2, 2, 2, 2, 3, 3}; return t[g]; }
__host__ __device__ constexpr int g_p(int g)   { constexpr int t[NK] = {0, 0, 0, 0, 0, 1, 1, 0, 1}; return t[g]; }   // input plane j + p
__host__ __device__ constexpr int g_dy(int g)  { constexpr int t[NK] = {0, 1, 1, 0, 1, 0, 1, 1, 1}; return t[g]; }   // input row wv + dy
__host__ __device__ constexpr int g_kz(int g)  { constexpr int t[NK] = {1, 1, 1, 2, 2, 0, 0, 2, 0}; return t[g]; }   // kernel taps: o = 2 i - 1 + k
__host__ __device__ constexpr int g_ky(int g)  { constexpr int t[NK] = {2, 0, 1, 2, 0, 2, 0, 1, 1}; return t[g]; }
__host__ __device__ constexpr int g_frag(int g) { return g_p(g) * 2 + g_dy(g); }                                      // B fragment index 0..3
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
    int abl;                    // tuning builds (MVS_TAIL_ABL, wrong results by design): 1 no skip loads, 2 no prob MFMAs, 4 no conv11 MFMAs,
                                // 8 no output stores, 16 no epilogue arithmetic / d11 writes, 32 no input copies; 0 in the release build
};

__global__ __launch_bounds__(tail::NTHREADS) void costreg_tail_kernel(TailArgs a) {
    using namespace tail;
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4;
    const unsigned lds_base = (unsigned)(uintptr_t)lds;
    const int cw = wv;
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

    // 4 values -> hi, lo (fp16 pairs) of v * s: the instruction sequence of split2_block (conv_split_common.h) for one quad
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

    // split pass: `nthr` threads (index t) turn the staged plane in buffer `fsel` into the two piece planes of slot `isel`:
    // piece P: 4 floats -> 4 + 4 fp16
    auto split_pass = [&](int fsel, int isel, int t, int nthr) {
        for (int P = t; P < NPIECE; P += nthr) {
            const f32x4 x = *reinterpret_cast<const f32x4 *>(lds + F_OFF + fsel * FBYTES + P * 16);
            unsigned h0, h1, l0, l1;
            split2_quad(x[0], x[1], x[2], x[3], sx, h0, h1, l0, l1);
            *reinterpret_cast<uint2 *>(lds + I_OFF + (isel * 2 + 0) * IPART + P * 8) = make_uint2(h0, h1);
            *reinterpret_cast<uint2 *>(lds + I_OFF + (isel * 2 + 1) * IPART + P * 8) = make_uint2(l0, l1);
        }
    };

    // output phase: the `prob` planes whose three S sums are complete.  Thread t owns outputs o = t, t + 512, t + 1024 of the
    // (plane, row, x) list of a step (2 planes x 14 x 30, a third plane at the column's end): its S offsets are constants, its
    // output offsets change with the column only.  j: the step that just finished (its S sums in slot j % 3, those of step
    // j - 1 in slot (j - 1) % 3); first: nothing before it (z = 0: S of plane -1 is zero); last: the column's last step -- also
    // plane 2 Di - 1.
    int o_so[3], o_pl[3], o_row[3], o_xx[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int o = tid + k * NTHREADS;
        o_pl[k] = o / (PR * PX);
        const int rem = o - o_pl[k] * (PR * PX);
        o_row[k] = rem / PX;
        o_xx[k] = rem - o_row[k] * PX;
        o_so[k] = o_row[k] * 32 + o_xx[k];
        if (o >= 3 * PR * PX) o_pl[k] = 3;      // nothing
    }
    unsigned o_off[3];       // (row, x) offset inside an output plane of the current column, or ~0 outside the volume
    auto output_phase = [&](int b, int j, bool first, bool last) {
        const float *Sc = reinterpret_cast<const float *>(lds + S_OFF) + (j % NSS) * 2 * SPLANE;
        const float *Sp = reinterpret_cast<const float *>(lds + S_OFF) + ((j + NSS - 1) % NSS) * 2 * SPLANE;
        float *ob = a.out + ((int64_t)b * D + 2 * j - 1) * (int64_t)H * W;      // plane 2 j - 1
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int pl = o_pl[k], so = o_so[k];
            if (pl > (last ? 2 : 1) || o_off[k] == ~0u || (a.abl & 8)) continue;
            float v;
            if (pl == 0) {          // z = 2 j - 1 = S0[2j-2] + S1[2j-1] + S2[2j]
                if (first) continue;
                v = (Sp[0 * SPLANE + 0 * PR * 32 + so] + Sp[1 * SPLANE + 1 * PR * 32 + so]) + Sc[0 * SPLANE + 2 * PR * 32 + so];
            } else if (pl == 1) {   // z = 2 j = S0[2j-1] + S1[2j] + S2[2j+1]
                const float s0 = first ? 0.0f : Sp[1 * SPLANE + 0 * PR * 32 + so];
                v = (s0 + Sc[0 * SPLANE + 1 * PR * 32 + so]) + Sc[1 * SPLANE + 2 * PR * 32 + so];
            } else {                // z = 2 j + 1 = D - 1: S0[2j] + S1[2j+1]
                v = Sc[0 * SPLANE + 0 * PR * 32 + so] + Sc[1 * SPLANE + 1 * PR * 32 + so];
            }
            ob[(int64_t)pl * H * W + o_off[k]] = (v * unp) * psc + psh;
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
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
        const int P = (i * NCW + cw) * 64 + lane;
        const int v = (P % (2 * NVIP)) >> 1, c = P / (2 * NVIP);
        const bool okv = v < NVI && P < NPIECE;
        loc[i] = okv ? ((v % XI) | ((v / XI) << 8) | ((P & 1) << 16) | (c << 17)) : -1;
    }
    // conv11: this lane's voxel of B fragment (p, dy): (input plane j + p, row wv + dy, x n + dx), chunk cc; kq = (dx, cc)
    const unsigned frag_off = (unsigned)((kq & 1) * NVIP * 16 + (wv * XI + n + (kq >> 1)) * 16);      // + dy * XI * 16
    // ... and its weight fragments, in registers for the kernel's lifetime: [K-step][hi, lo]
    f16x8 Aw[NK][2];
#pragma unroll
    for (int gk = 0; gk < NK; ++gk)
#pragma unroll
        for (int pp = 0; pp < 2; ++pp)
            Aw[gk][pp] = __builtin_bit_cast(f16x8, reinterpret_cast<const uint4 *>(a.wpk)[(gk * 2 + pp) * 64 + lane]);
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

    __syncthreads();      // affine table, zeroed d11 planes, conv11's fragments

    // ================================================================ runs: (column, j0 .. j1) pieces of [g0, g1)
    // Schedule of a step j (two barriers):
    //   phase 1 (behind barrier B): conv11 of step j from input planes j, j + 1 -> epilogue (skip values requested a step ago)
    //            -> d11 planes 2j, 2j + 1 in LDS; wait for the copy of input plane j + 2 (issued a step ago); request the skip
    //            values of step j + 1.
    //   phase 2 (behind barrier C): issue the copy of plane j + 3; prob of step j -> S slot j % 3; split plane j + 2 into the
    //            slot plane j leaves; write the outputs of step j - 1 (S slots (j - 1) % 3, (j - 2) % 3).
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
        // plane p -> staging buffer p & 1 (planes beyond the volume: zeros)
        auto issue_plane = [&](int p) {
            const bool ok = p < a.Di;
            srd = make_srd(a.in + ((int64_t)b * a.Di + (ok ? p : 0)) * plane_in, ok ? plane_bytes : 0u);
#pragma unroll
            for (int i = 0; i < IPW; ++i) {
                if (i * NCW + cw >= NCOPY) continue;   // wave-uniform
                if (a.abl & 32) continue;
                glds16_buf(voff[i], srd, 0u, lds_base + (unsigned)(F_OFF + (p & 1) * FBYTES + (i * NCW + cw) * 1024));
            }
        };
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int oy = 14 * ty + o_row[k], ox = 30 * tx + o_xx[k];
            o_off[k] = (oy < H && ox < W) ? (unsigned)(oy * W + ox) : ~0u;
        }
        // skip-connection values of the four outputs this lane writes in step j: the two rows (kinds) of the wave, two planes
        float4 res[4];            // (two sets, requested two steps ahead: 256 registers and spills -- 0.40 -> 0.43 ms)
        bool inside[4];
        unsigned sk_off[2];       // float offset of (row of kind k, x, channel quad) inside a plane of the skip volume
        const float *skb = a.skip + (int64_t)b * D * H * W * COUT;
        auto request_skip = [&](int j) {
            const float *pl0 = skb + (int64_t)(2 * j) * H * W * COUT;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                res[c] = inside[c] ? *reinterpret_cast<const float4 *>(pl0 + (int64_t)pz(c) * H * W * COUT + sk_off[kind(c)])
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
        };
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            const int gx = Ib + (loc[i] & 255), gy = Jb + ((loc[i] >> 8) & 255);
            const int h = (loc[i] >> 16) & 1, c = (loc[i] >> 17) & 1;
            const bool ok = loc[i] >= 0 && gx >= 0 && gx < a.Wi && gy >= 0 && gy < a.Hi;
            voff[i] = ok ? (unsigned)((((int64_t)gy * a.Wi + gx) * CIN + c * 8 + h * 4) * 4) : 0xffffff00u;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int gy = 14 * ty - 1 + 2 * wv + kind(c), gx = 30 * tx - 1 + 2 * n + pxr;
            inside[c] = gy >= 0 && gy < H && gx >= 0 && gx < W && !(a.abl & 1);
            if (c < 2) sk_off[c] = inside[c] ? (unsigned)((gy * W + gx) * COUT + c0) : 0u;
        }
        issue_plane(js);
        issue_plane(js + 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        request_skip(js);
        __syncthreads();                 // P1: planes js, js + 1 are staged
        split_pass(js & 1, js & 1, tid, NTHREADS);
        split_pass((js + 1) & 1, (js + 1) & 1, tid, NTHREADS);
        __syncthreads();                 // P2 = B(js): both are split; the staging buffers are free
        issue_plane(js + 2);

        for (int j = js; j <= j1; ++j) {
            // ============================================================ phase 1
            {
                const unsigned pj = lds_base + (unsigned)(I_OFF + ((j & 1) * 2) * IPART) + frag_off;
                const unsigned pj1 = lds_base + (unsigned)(I_OFF + (((j + 1) & 1) * 2) * IPART) + frag_off;
                f32x4 acc[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
                // the four B fragments (plane, dy), two pieces each: 8 reads serve the 27 MFMAs of the step
                f16x8 Bf[4][2];
                Bf[0][0] = __builtin_bit_cast(f16x8, lds_read_b128<0>(pj));
                Bf[0][1] = __builtin_bit_cast(f16x8, lds_read_b128<IPART>(pj));
                Bf[1][0] = __builtin_bit_cast(f16x8, lds_read_b128<XI * 16>(pj));
                Bf[1][1] = __builtin_bit_cast(f16x8, lds_read_b128<XI * 16 + IPART>(pj));
                Bf[2][0] = __builtin_bit_cast(f16x8, lds_read_b128<0>(pj1));
                Bf[2][1] = __builtin_bit_cast(f16x8, lds_read_b128<IPART>(pj1));
                Bf[3][0] = __builtin_bit_cast(f16x8, lds_read_b128<XI * 16>(pj1));
                Bf[3][1] = __builtin_bit_cast(f16x8, lds_read_b128<XI * 16 + IPART>(pj1));
                lds_wait_n<0>();
                asm volatile("" : "+v"(Bf[0][0]), "+v"(Bf[0][1]), "+v"(Bf[1][0]), "+v"(Bf[1][1]));
                asm volatile("" : "+v"(Bf[2][0]), "+v"(Bf[2][1]), "+v"(Bf[3][0]), "+v"(Bf[3][1]));
                // K-steps in pairs of DIFFERENT classes (accumulators): product k of both, then product k + 1 -- a dependent
                // MFMA never follows its producer directly
                constexpr int ORD[NK + 1] = {0, 3, 1, 4, 2, 5, 7, 6, 8, -1};
                if (!(a.abl & 4))
                static_for<0, (NK + 1) / 2>([&](auto qc) {
                    constexpr int q = decltype(qc)::value, g1_ = ORD[2 * q], g2_ = ORD[2 * q + 1];
                    constexpr int C1 = g_cls(g1_), F1 = g_frag(g1_);
                    f32x4 &c1 = acc[C1];
                    if constexpr (g2_ >= 0) {
                        constexpr int C2 = g_cls(g2_ >= 0 ? g2_ : 0), F2 = g_frag(g2_ >= 0 ? g2_ : 0), G2 = g2_ >= 0 ? g2_ : 0;
                        f32x4 &c2 = acc[C2];
                        static_assert(C1 != C2, "a pair shares no accumulator");
                        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Aw[g1_][1], Bf[F1][0], c1, 0, 0, 0);
                        c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Aw[G2][1], Bf[F2][0], c2, 0, 0, 0);
                        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Aw[g1_][0], Bf[F1][1], c1, 0, 0, 0);
                        c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Aw[G2][0], Bf[F2][1], c2, 0, 0, 0);
                        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Aw[g1_][0], Bf[F1][0], c1, 0, 0, 0);
                        c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Aw[G2][0], Bf[F2][0], c2, 0, 0, 0);
                    } else {
                        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Aw[g1_][1], Bf[F1][0], c1, 0, 0, 0);
                        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Aw[g1_][0], Bf[F1][1], c1, 0, 0, 0);
                        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Aw[g1_][0], Bf[F1][0], c1, 0, 0, 0);
                    }
                });
                // epilogue: affine, ReLU, skip add; zero outside the volume; scale, split, into the d11 planes
                const float4 sc = *reinterpret_cast<const float4 *>(lds + AFF_OFF + c0 * 4);
                const float4 sh = *reinterpret_cast<const float4 *>(lds + AFF_OFF + (COUT + c0) * 4);
                if (!(a.abl & 16))
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    f32x4 v = acc[c];
                    if (unscale2 != 1.0f) { v[0] *= unscale2; v[1] *= unscale2; v[2] *= unscale2; v[3] *= unscale2; }
                    v[0] = v[0] * sc.x + sh.x; v[1] = v[1] * sc.y + sh.y;
                    v[2] = v[2] * sc.z + sh.z; v[3] = v[3] * sc.w + sh.w;
                    v[0] = relu_nan(v[0]); v[1] = relu_nan(v[1]); v[2] = relu_nan(v[2]); v[3] = relu_nan(v[3]);
                    v[0] += res[c].x; v[1] += res[c].y; v[2] += res[c].z; v[3] += res[c].w;
                    if (!inside[c]) v = (f32x4){0.f, 0.f, 0.f, 0.f};
                    unsigned h0, h1, l0, l1;
                    split2_quad(v[0], v[1], v[2], v[3], sc11, h0, h1, l0, l1);
                    const int lr = 2 * wv + kind(c), xl = 2 * n + pxr;
                    unsigned char *dst = lds + C_OFF + (pz(c) * 2) * CPART + (lr * CX + xl) * 16 + (kq & 1) * 8;
                    *reinterpret_cast<uint2 *>(dst) = make_uint2(h0, h1);
                    *reinterpret_cast<uint2 *>(dst + CPART) = make_uint2(l0, l1);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // plane j + 2 has landed (nothing else is outstanding here)
                if (j < j1) request_skip(j + 1);
            }
            __syncthreads();             // C: d11 planes 2 j, 2 j + 1 are in LDS; plane j + 2 is staged
            // ============================================================ phase 2
            if (j + 3 <= j1 + 1) issue_plane(j + 3);
            {
                // prob: 9 d11 rows in groups of three (three accumulators in flight), then the S sums of 7 output rows
                const unsigned cb = lds_base + (unsigned)(C_OFF + (p_pl * 2) * CPART + ((7 * p_hf) * CX + p_xb * 16 + n + kq) * 16);
                f16x8 Bh[2][3], Bl[2][3];
                auto read_rows = [&](auto gc) {
                    constexpr int gg = decltype(gc)::value;
                    static_for<0, 3>([&](auto rc) {
                        constexpr int r = decltype(rc)::value, i = gg * 3 + r;
                        Bh[gg & 1][r] = __builtin_bit_cast(f16x8, lds_read_b128<i * CX * 16>(cb));
                        Bl[gg & 1][r] = __builtin_bit_cast(f16x8, lds_read_b128<i * CX * 16 + CPART>(cb));
                    });
                };
                f32x4 d[9];
#pragma unroll
                for (int i = 0; i < 9; ++i) d[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                read_rows(std::integral_constant<int, 0>{});
                if (!(a.abl & 2))
                static_for<0, 3>([&](auto gc) {
                    constexpr int gg = decltype(gc)::value, pb = gg & 1;
                    lds_wait_n<0>();
                    asm volatile("" : "+v"(Bh[pb][0]), "+v"(Bh[pb][1]), "+v"(Bh[pb][2]), "+v"(Bl[pb][0]), "+v"(Bl[pb][1]), "+v"(Bl[pb][2]));
                    if constexpr (gg + 1 < 3) read_rows(std::integral_constant<int, gg + 1>{});
                    __builtin_amdgcn_sched_barrier(0);
                    static_for<0, 3>([&](auto rc) { constexpr int r = decltype(rc)::value;
                        d[gg * 3 + r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pal, Bh[pb][r], d[gg * 3 + r], 0, 0, 0); });
                    static_for<0, 3>([&](auto rc) { constexpr int r = decltype(rc)::value;
                        d[gg * 3 + r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pah, Bl[pb][r], d[gg * 3 + r], 0, 0, 0); });
                    static_for<0, 3>([&](auto rc) { constexpr int r = decltype(rc)::value;
                        d[gg * 3 + r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pah, Bh[pb][r], d[gg * 3 + r], 0, 0, 0); });
                    __builtin_amdgcn_sched_barrier(0);
                });
                if (kq < 3) {
                    float *Sd = reinterpret_cast<float *>(lds + S_OFF) + ((j % NSS) * 2 + p_pl) * SPLANE + kq * PR * 32 + (7 * p_hf) * 32 + p_xb * 16 + n;
#pragma unroll
                    for (int i = 0; i < 7; ++i) Sd[i * 32] = (d[i][0] + d[i + 1][1]) + d[i + 2][2];      // S[y] = D_y[0] + D_{y+1}[1] + D_{y+2}[2]
                }
            }
            if (j + 2 <= j1 + 1) split_pass(j & 1, j & 1, tid, NTHREADS);            // plane j + 2 (buffer (j + 2) & 1) -> the slot of plane j
            if (j > js && j - 1 >= j0) output_phase(b, j - 1, j - 1 == 0, false);
            __syncthreads();             // B of the next step: input planes j + 1, j + 2 are split, the S sums of step j are in LDS
        }
        output_phase(b, j1, j1 == 0, j1 == a.Di - 1);
        __syncthreads();                 // (the next run stages, splits and -- three barriers on -- writes S again)
    }
}

// PyTorch ConvTranspose3d weight (16, 8, 3, 3, 3) -> [K-step g][hi, lo][lane][8 fp16] of w * 2^(14 - exponent(max |w|));
// lane (mrow, kq): K slot kq = (dx = kq >> 1, chunk cc = kq & 1), input channel cc * 8 + i, kernel taps (kz, ky) of K-step g;
// MFMA row mrow = (x parity row group, output channel): group 0 = the ODD output x of the pair (taps k = 2 on dx = 0, k = 0 on
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
    const int mrow = lane & 15, kq = lane >> 4, dx = kq >> 1, cc = kq & 1;
    const int grp = mrow >> 3, co = mrow & 7;
    int kx = -1;
    if (grp == 0) kx = dx == 0 ? 2 : 0; else if (dx == 1) kx = 1;
    const int ci = cc * 8 + jj;
    float x = 0.0f;
    if (kx >= 0) x = w[(((int64_t)ci * COUT + co) * 3 + g_kz(g)) * 9 + g_ky(g) * 3 + kx];
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
#ifdef MVS_TUNING
    a.abl = getenv("MVS_TAIL_ABL") ? atoi(getenv("MVS_TAIL_ABL")) : 0;
#else
    a.abl = 0;
#endif
    const int64_t G = (int64_t)a.ncols * Di;
    const int n_cu = device_cu_count();
    const unsigned grid = (unsigned)(G < n_cu ? G : n_cu);
    hipLaunchKernelGGL(costreg_tail_kernel, dim3(grid), dim3(tail::NTHREADS), 0, as_stream(stream), a);
    return check_launch("mvs_costreg_tail_f16_f32");
}
