// Shared device helpers of the split-operand (bf16x6) convolution kernels: conv_bf16x6.hip (conv0 class, Cout = 8)
// and conv_split.hip (Cout = 16 / 32 layers).  Every fp32 number is EXACTLY the sum of three bf16 numbers
// (8 + 8 + 8 significand bits: hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid), each subtraction exact).
#ifndef MVS_CONV_SPLIT_COMMON_H
#define MVS_CONV_SPLIT_COMMON_H
#include "conv_persistent.h"
#include "split2.h"

namespace mvs {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// c - (a.lo * b.lo + a.hi * b.hi) with a, b pairs of bf16: v_dot2c_f32_bf16 (gfx950).  With b = (-1, 0) or (0, -1)
// it subtracts one bf16 of a packed pair from an fp32 number in ONE instruction (no unpack); the difference
// of a number and its own bf16 rounding is exactly representable, so the result is exact whatever the
// instruction's internal rounding.  (The compiler cannot select the builtin on this target; the assembler
// knows the instruction.)
__device__ __forceinline__ float sub_bf16_half(float c, unsigned pair, unsigned sel) {
    asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(c) : "s"(sel), "v"(pair));
    return c;
}

// x (8 fp32 channels of one voxel) -> hi, mid, lo with x = hi + mid + lo exactly.
// DOT2: 7 VALU per pair of values (3 cvt_pk + 4 dot2c) instead of 11 (3 cvt_pk + 4 unpack + 4 sub).
template <bool DOT2>
__device__ __forceinline__ void split3(const f32x4 &a, const f32x4 &b, bf16x8 &h, bf16x8 &m, bf16x8 &l) {
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        const f32x2 x = (i < 4) ? (f32x2){a[i], a[i + 1]} : (f32x2){b[i - 4], b[i - 3]};
        const bf16x2 hh = __builtin_convertvector(x, bf16x2);
        f32x2 r1, r2;
        if constexpr (DOT2) {
            const unsigned hu = __builtin_bit_cast(unsigned, hh);
            r1[0] = sub_bf16_half(x[0], hu, 0x0000bf80u);
            r1[1] = sub_bf16_half(x[1], hu, 0xbf800000u);
        } else {
            r1 = x - __builtin_convertvector(hh, f32x2);
        }
        const bf16x2 mm = __builtin_convertvector(r1, bf16x2);
        if constexpr (DOT2) {
            const unsigned mu = __builtin_bit_cast(unsigned, mm);
            r2[0] = sub_bf16_half(r1[0], mu, 0x0000bf80u);
            r2[1] = sub_bf16_half(r1[1], mu, 0xbf800000u);
        } else {
            r2 = r1 - __builtin_convertvector(mm, f32x2);
        }
        const bf16x2 ll = __builtin_convertvector(r2, bf16x2);
        h[i] = hh[0]; h[i + 1] = hh[1];
        m[i] = mm[0]; m[i + 1] = mm[1];
        l[i] = ll[0]; l[i + 1] = ll[1];
    }
}


// the same split of one fragment as ONE scheduled block.  A dot instruction's result may be read by a
// different VALU instruction only 3 issue slots later (the compiler's hazard recogniser inserts the nops for
// code it generates, it does not look inside inline assembly): all eight subtractions of a level are issued
// before the first conversion of the next.
__device__ __forceinline__ void split3_block(f32x4 &a, f32x4 &b, bf16x8 &h, bf16x8 &m, bf16x8 &l) {
    unsigned h0, h1, h2, h3, m0, m1, m2, m3, l0, l1, l2, l3;
    float x0 = a[0], x1 = a[1], x2 = a[2], x3 = a[3], x4 = b[0], x5 = b[1], x6 = b[2], x7 = b[3];
    asm volatile(
        "v_cvt_pk_bf16_f32 %8, %0, %1\n\tv_cvt_pk_bf16_f32 %9, %2, %3\n\t"
        "v_cvt_pk_bf16_f32 %10, %4, %5\n\tv_cvt_pk_bf16_f32 %11, %6, %7\n\t"
        "v_dot2c_f32_bf16 %0, %20, %8\n\tv_dot2c_f32_bf16 %1, %21, %8\n\t"
        "v_dot2c_f32_bf16 %2, %20, %9\n\tv_dot2c_f32_bf16 %3, %21, %9\n\t"
        "v_dot2c_f32_bf16 %4, %20, %10\n\tv_dot2c_f32_bf16 %5, %21, %10\n\t"
        "v_dot2c_f32_bf16 %6, %20, %11\n\tv_dot2c_f32_bf16 %7, %21, %11\n\t"
        "v_cvt_pk_bf16_f32 %12, %0, %1\n\tv_cvt_pk_bf16_f32 %13, %2, %3\n\t"
        "v_cvt_pk_bf16_f32 %14, %4, %5\n\ts_nop 0\n\tv_cvt_pk_bf16_f32 %15, %6, %7\n\t"
        "v_dot2c_f32_bf16 %0, %20, %12\n\tv_dot2c_f32_bf16 %1, %21, %12\n\t"
        "v_dot2c_f32_bf16 %2, %20, %13\n\tv_dot2c_f32_bf16 %3, %21, %13\n\t"
        "v_dot2c_f32_bf16 %4, %20, %14\n\tv_dot2c_f32_bf16 %5, %21, %14\n\t"
        "v_dot2c_f32_bf16 %6, %20, %15\n\tv_dot2c_f32_bf16 %7, %21, %15\n\t"
        "v_cvt_pk_bf16_f32 %16, %0, %1\n\tv_cvt_pk_bf16_f32 %17, %2, %3\n\t"
        "v_cvt_pk_bf16_f32 %18, %4, %5\n\ts_nop 0\n\tv_cvt_pk_bf16_f32 %19, %6, %7\n\ts_nop 1"
        : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7),
          "=&v"(h0), "=&v"(h1), "=&v"(h2), "=&v"(h3), "=&v"(m0), "=&v"(m1), "=&v"(m2), "=&v"(m3),
          "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3)
        : "s"(0x0000bf80u), "s"(0xbf800000u));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    h = __builtin_bit_cast(bf16x8, (u32x4){h0, h1, h2, h3});
    m = __builtin_bit_cast(bf16x8, (u32x4){m0, m1, m2, m3});
    l = __builtin_bit_cast(bf16x8, (u32x4){l0, l1, l2, l3});
}

// ---- two-piece fp16 operands: split2.h (shared with the plane-sweep kernel, which hands its volume over already split)
template <int OFF>
__device__ __forceinline__ f32x4 lds_read_b128(unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536 && OFF % 16 == 0, "ds_read_b128 offset field");
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}

template <int OFF>
__device__ __forceinline__ void lds_write_b128(unsigned addr, const bf16x8 &v) {
    static_assert(OFF >= 0 && OFF < 65536 && OFF % 16 == 0, "ds_write_b128 offset field");
    asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}

template <int OFF>
__device__ __forceinline__ void lds_write_b64(unsigned addr, unsigned lo, unsigned hi) {
    static_assert(OFF >= 0 && OFF < 65536 && OFF % 8 == 0, "ds_write_b64 offset field");
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 v = {lo, hi};
    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}

}  // namespace mvs
#endif  // MVS_CONV_SPLIT_COMMON_H
