/* Entry points that exist only in the TUNING build of the library (libmvs_hip_tuning.so: `python -m mvs_amd.build --tuning`,
 * -DMVS_TUNING): measured experiments that are not part of the product.  The release library does not export them. */
#ifndef MVS_HIP_TUNING_H
#define MVS_HIP_TUNING_H
#include "mvs_hip.h"

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: what is declared between this push and its pop is the exported C ABI, nothing else */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

/* conv0 on a volume that ARRIVES as fp16 pairs (round 3: built, bit-identical to mvs_conv3d_c8_f16x3_f32, worth 0.06 ms, not wired:
 * DESIGN.md section 6; mvs_amd/csrc/conv_f16x3_pairs.hip). */
/* A volume that arrives as fp16 pairs (MVS_LAYOUT_C8H: [B,D,H,C/8, part (hi, lo), parity (x & 1), ceil(W/2), 8 fp16],
 * mvs_c8h_bytes(B, C, D, H, W) bytes -- 4 per element like the fp32 volume): the producer has scaled every value by
 * 2^(14 - exponent(max of the absmax block)) and split it, so the kernel has no fp32 staging buffer, no split pass and one
 * barrier per step instead of two (mvs_amd/csrc/conv_f16x3_pairs.hip); results are bit-identical to mvs_conv3d_c8_f16x3_f32 on
 * the fp32 volume with the same block.  in_absmax = the block the producer scaled by -- it must BOUND the data (a larger bound
 * costs range in the lo piece only: an element below 2^-18 of the bound keeps an absolute error below 2^-40 of the bound).
 * Producers: mvs_costvol_variance_fwd_ws2_f32 with out_layout MVS_LAYOUT_C8H (the block is then an INPUT: the bound of the
 * variance, e.g. the square of the feature maps' largest magnitude), or mvs_c8_to_c8h_f32 from an fp32 MVS_LAYOUT_C8 volume. */
size_t mvs_c8h_bytes(int B, int C, int D, int H, int W);
int mvs_c8_to_c8h_f32(const float *in_c8, const void *absmax, int B, int C, int D, int H, int W, void *out_pairs, void *stream);
int mvs_conv3d_c8h_f16x3_f32(const void *in_pairs, const void *in_absmax, const void *packed, const float *scale,
                             const float *shift, const float *residual, int relu, int B, int Cin, int D, int H, int W,
                             float *out, void *out_absmax, void *stream);

/* (round 6: the hand-over with eight-row tiles is in the release library -- include/mvs_hip.h, mvs_conv3d_c8p_f16x3_f32; the tuning
 * build also takes npair = 5 there and MVS_CONV0_Y8 = 1 | 2 selects the eight-row kernels that keep the staging buffer,
 * mvs_amd/csrc/conv_f16x3_y8.hip) */

/* Cycles of wave 0 of every block of the training path's variance backward (mvs_costvol_variance_bwd_f32, C16 form), summed since the
 * last reset: [0] set-up, [1] clear + staging, [2] bound, [3] accumulate, [4] flush, [5] passes counted (scripts/exp_varbwd_laps.py). */
int mvs_tuning_varbwd_laps(unsigned long long *out8, int reset);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* MVS_HIP_TUNING_H */
