// Two fp16 pieces of an fp32 value scaled by a power of two (conv_f16x3.hip has the arithmetic and its error bound): shared by the
// convolution kernels that split their operands and by the plane-sweep kernel that hands its volume over already split.
#ifndef MVS_SPLIT2_H
#define MVS_SPLIT2_H
#include "mvs_common.h"

namespace mvs {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// scale = 2^(14 - e), e = exponent of the largest magnitude (clamped below: an all-zero input keeps the arithmetic defined;
// every finite fp32 maximum is covered: 2^127 * 2^(14 - 127) < 2^15)
__device__ __host__ __forceinline__ int absmax_exponent(unsigned bits) {
    int e = (int)((bits >> 23) & 255u) - 127;
    return e < -100 ? -100 : (e > 127 ? 127 : e);
}
__device__ __forceinline__ float pow2f(int e) { return __builtin_bit_cast(float, (unsigned)(e + 127) << 23); }

// 8 fp32 values -> hi, lo (fp16 pairs) of x * s
__device__ __forceinline__ void split2_block(f32x4 &a, f32x4 &b, float s, u32x4 &h, u32x4 &l) {
    unsigned h0, h1, h2, h3, l0, l1, l2, l3;
    float x0 = a[0], x1 = a[1], x2 = a[2], x3 = a[3], x4 = b[0], x5 = b[1], x6 = b[2], x7 = b[3];
    asm volatile(
        "v_fma_mixlo_f16 %8, %0, %16, 0\n\tv_fma_mixlo_f16 %9, %2, %16, 0\n\t"
        "v_fma_mixlo_f16 %10, %4, %16, 0\n\tv_fma_mixlo_f16 %11, %6, %16, 0\n\t"
        "v_fma_mixhi_f16 %8, %1, %16, 0\n\tv_fma_mixhi_f16 %9, %3, %16, 0\n\t"
        "v_fma_mixhi_f16 %10, %5, %16, 0\n\tv_fma_mixhi_f16 %11, %7, %16, 0\n\t"
        "v_fma_mix_f32 %0, %0, %16, -%8 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mix_f32 %1, %1, %16, -%8 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mix_f32 %2, %2, %16, -%9 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mix_f32 %3, %3, %16, -%9 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mix_f32 %4, %4, %16, -%10 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mix_f32 %5, %5, %16, -%10 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mix_f32 %6, %6, %16, -%11 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mix_f32 %7, %7, %16, -%11 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_cvt_pk_f16_f32 %12, %0, %1\n\tv_cvt_pk_f16_f32 %13, %2, %3\n\t"
        "v_cvt_pk_f16_f32 %14, %4, %5\n\tv_cvt_pk_f16_f32 %15, %6, %7"
        : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7),
          "=&v"(h0), "=&v"(h1), "=&v"(h2), "=&v"(h3), "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3)
        : "s"(s));
    h = (u32x4){h0, h1, h2, h3};
    l = (u32x4){l0, l1, l2, l3};
}


}  // namespace mvs
#endif  // MVS_SPLIT2_H
