// Range guard of the two-piece fp16 convolutions (conv_f16x3.hip, conv_split.hip / deconv_split.hip with NP = 2).
//
// Those kernels scale their input by ONE power of two per tensor, taken from the input's absmax block, and carry every
// operand as two fp16 pieces: |error| <= 2^-22 |a b| per product for operands within 2^-18 of their tensor's maximum,
// and <= 2^-40 max|x| |w| absolute for smaller ones.  That contract breaks in two ways, and both are decided here, on
// the device, per launch, from the block the producer filled (no host synchronisation):
//   code 2  the maximum is not finite (an Inf or NaN voxel; the reference -- MVSNet/models/module.py:83-84 feeding
//           mvsnet.py:83-93 -- keeps its damage inside the receptive field, a global scale of 2^-113 would flush every
//           finite voxel to zero), or the layer's weights are not finite;
//   code 1  the maximum is carried by a few outliers: fewer than one in eight of the block's non-zero words (each the
//           maximum over a share of the producer's workgroups) lie within 2^-16 of it -- the rest of the tensor would
//           sit in fp16's subnormals.
// In either case the launch computes the layer with guard_direct_conv below instead: plain fp32 FMAs on the ORIGINAL
// fp32 weights (kept behind the packed fragments), real taps only (zero padding is skipped, as it contributes +0 in the
// reference), IEEE semantics for Inf / NaN, the reference's NaN-propagating ReLU.  Slow (no LDS, no matrix pipe) and
// meant to be: it is the path of broken cameras and corrupt inputs, never of a sane volume; every launch that takes it
// bumps a device counter (mvs_guard_fallback_count) so a run can assert it never happened.
#ifndef MVS_CONV_GUARD_H
#define MVS_CONV_GUARD_H
#include "mvs_common.h"

namespace mvs {

// PyTorch-layout fp32 weights -- (Cout, Cin, taps), or transposed (Cin, Cout, taps) -- into the guard's layout [tap][Cin][Cout]
// behind a two-piece pack (conv_f16x3.hip; called by every *_pack_weights_f16* entry)
int launch_guard_weights(const float *w, int transposed, int Cin, int Cout, int ntap, float *dst, hipStream_t st);

// the device counter of launches that took the fallback (capi.hip; one per device, the launchers pass its address)
unsigned long long *guard_counter();

// verdict of an absmax block: wave-uniform maximum bits + the code above (0 = the two-piece arithmetic holds)
struct AbsmaxVerdict { unsigned bits; int code; };
__device__ __forceinline__ AbsmaxVerdict absmax_verdict(const unsigned *absmax) {
    const uint4 v = reinterpret_cast<const uint4 *>(absmax)[threadIdx.x & 63];
    unsigned m = max(max(v.x, v.y), max(v.z, v.w));
#pragma unroll
    for (int o = 32; o; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    m = (unsigned)__builtin_amdgcn_readfirstlane((int)m);
    // bit patterns of non-negative floats order like the numbers: "within 2^-16" = exponent field at most 16 lower
    const unsigned thr = m > (16u << 23) ? m - (16u << 23) : 1u;
    int near = 0, nz = 0;
    near += __popcll(__ballot(v.x >= thr)) + __popcll(__ballot(v.y >= thr)) + __popcll(__ballot(v.z >= thr)) + __popcll(__ballot(v.w >= thr));
    nz += __popcll(__ballot(v.x != 0)) + __popcll(__ballot(v.y != 0)) + __popcll(__ballot(v.z != 0)) + __popcll(__ballot(v.w != 0));
    const int need = min(nz, max(2, nz >> 3));
    AbsmaxVerdict r;
    r.bits = m;
    r.code = m >= 0x7f800000u ? 2 : (near < need ? 1 : 0);
    return r;
}

struct GuardConv {
    const float *in;      // channels-last [B, D, H, W, Cin], or (in_c8) 8-channel blocks [B, D, H, Cin/8, W, 8]
    const float *w;       // the layer's fp32 weights as launch_guard_weights() lays them out: [tap][Cin][Cout_total]
    const float *scale, *shift, *residual;   // as the fast kernel's arguments (already offset to the launch's first channel)
    float *out;
    unsigned *out_absmax;
    unsigned long long *counter;             // guard_counter()
    int B, D, H, W, Cin;  // input dims (a 2D layer: D = images, kd = 1)
    int Do, Ho, Wo;       // output dims
    int ldc, co0, nco;    // channels of the whole output tensor, first channel / channel count of this launch
    int kd, kh, stride, transposed, relu, in_c8, out_c4;
};

// Every thread of the launch (copy waves included) takes output VOXELS in a grid-stride loop and computes all NCO output
// channels of the launch for its voxel: the weight addresses are then the same for every lane (scalar loads, the weight rides
// in an SGPR operand of the FMA) and the input comes in as one 16-byte load per four input channels -- per (tap, input channel)
// a quarter of a vector load and NCO FMAs, where the first version (one 4-channel block per thread, scalar loads of everything)
// issued five loads per four FMAs: one NaN pixel in a source image cost 178 ms per forward at config 2, now 43 ms
// (scripts/exp_guard_cost.py).  The accumulation order per output is unchanged: taps (z, y, x), then input channels.
template <int NCO>
__device__ __forceinline__ void guard_direct_conv_n(const GuardConv &g) {
    const int64_t total = (int64_t)g.B * g.Do * g.Ho * g.Wo;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int pz = g.kd / 2, ph = g.kh / 2;
    const int sz = g.kd == 1 ? 1 : g.stride;          // images are not strided in z
    const int c0 = g.co0;                              // first output channel of the launch in the whole tensor
    // weight of (tap t, input channel ci, output channel c0 + j): g.w[(t * Cin + ci) * ldc + c0 + j] -- the NCO weights of one
    // (tap, input channel) are contiguous: one wide scalar load
    unsigned vmax = 0;
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += stride) {
        int64_t vox = it;
        const int x = (int)(vox % g.Wo); vox /= g.Wo;
        const int y = (int)(vox % g.Ho); vox /= g.Ho;
        const int z = (int)(vox % g.Do);
        const int b = (int)(vox / g.Do);
        float acc[NCO];
#pragma unroll
        for (int j = 0; j < NCO; ++j) acc[j] = 0.f;
        for (int tz = 0; tz < g.kd; ++tz) {
            int iz;
            if (g.transposed) { const int n = z + 1 - tz; if (n & 1) continue; iz = n >> 1; }
            else iz = z * sz - pz + tz;
            if ((unsigned)iz >= (unsigned)g.D) continue;
            for (int ty = 0; ty < g.kh; ++ty) {
                int iy;
                if (g.transposed) { const int n = y + 1 - ty; if (n & 1) continue; iy = n >> 1; }
                else iy = y * g.stride - ph + ty;
                if ((unsigned)iy >= (unsigned)g.H) continue;
                for (int tx = 0; tx < g.kh; ++tx) {
                    int ix;
                    if (g.transposed) { const int n = x + 1 - tx; if (n & 1) continue; ix = n >> 1; }
                    else ix = x * g.stride - ph + tx;
                    if ((unsigned)ix >= (unsigned)g.W) continue;
                    const int t = (tz * g.kh + ty) * g.kh + tx;
                    const int64_t row = ((int64_t)b * g.D + iz) * g.H + iy;
                    const float *const wt = g.w + (int64_t)t * g.Cin * g.ldc + c0;
                    for (int ci = 0; ci < g.Cin; ci += 4) {
                        const float4 xq = g.in_c8 ? *reinterpret_cast<const float4 *>(g.in + ((row * (g.Cin >> 3) + (ci >> 3)) * g.W + ix) * 8 + (ci & 7))
                                                  : *reinterpret_cast<const float4 *>(g.in + (row * g.W + ix) * g.Cin + ci);
                        const float xs[4] = {xq.x, xq.y, xq.z, xq.w};
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float *const wk = wt + (ci + k) * g.ldc;
#pragma unroll
                            for (int j = 0; j < NCO; ++j) acc[j] = __fmaf_rn(wk[j], xs[k], acc[j]);
                        }
                    }
                }
            }
        }
        const int64_t ovox = (((int64_t)b * g.Do + z) * g.Ho + y) * g.Wo + x;
#pragma unroll
        for (int q = 0; q < NCO / 4; ++q) {
            const int64_t o = g.out_c4 ? (((int64_t)b * g.Do + z) * (g.ldc >> 2) + ((c0 >> 2) + q)) * ((int64_t)g.Ho * g.Wo * 4) + ((int64_t)y * g.Wo + x) * 4
                                       : ovox * g.ldc + 4 * q;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = acc[4 * q + j] * (g.scale ? g.scale[4 * q + j] : 1.0f) + (g.shift ? g.shift[4 * q + j] : 0.0f);
                if (g.relu == 1) v[j] = relu_nan(v[j]);
                else if (g.relu == 2) v[j] = v[j] > 0.f ? v[j] : v[j] * 0.1f;
                if (g.residual) v[j] += g.residual[ovox * g.ldc + 4 * q + j];
                vmax = max(vmax, __float_as_uint(v[j]) & 0x7fffffffu);
            }
            *reinterpret_cast<float4 *>(g.out + o) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
    if (g.out_absmax && vmax) atomicMax(g.out_absmax + (blockIdx.x & (kAbsmaxWords - 1)), vmax);
}

__device__ __forceinline__ void guard_direct_conv(const GuardConv &g) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && g.counter) atomicAdd(g.counter, 1ull);
    if (g.nco == 8) guard_direct_conv_n<8>(g);
    else if (g.nco == 16) guard_direct_conv_n<16>(g);
    else if (g.nco == 32) guard_direct_conv_n<32>(g);
    else {      // (no launcher passes another count; keep the data path defined: 4 channels at a time)
        GuardConv h = g;
        for (int q = 0; q < g.nco / 4; ++q) {
            h.co0 = g.co0 + 4 * q; h.nco = 4;
            h.scale = g.scale ? g.scale + 4 * q : nullptr; h.shift = g.shift ? g.shift + 4 * q : nullptr;
            h.residual = g.residual ? g.residual + 4 * q : nullptr; h.out = g.out + (g.out_c4 ? 0 : 4 * q);
            guard_direct_conv_n<4>(h);
        }
    }
}

}  // namespace mvs
#endif  // MVS_CONV_GUARD_H
