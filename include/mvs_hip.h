/*
 * mvs_hip.h -- C ABI of libmvs_hip.so: the MI355X (gfx950) implementation of
 * the MVSNet cost-volume hot path.
 *
 * The reference (doubleZ0108/MVS) has no FFI layer: its boundary for this path
 * is the Python model API (MVSNet/models/__init__.py:1, module.py, mvsnet.py).
 * mvs_amd/models mirrors that API and calls the entry points below through
 * ctypes on tensor.data_ptr(); INTEGRATION.md shows the binding.  Each entry
 * point cites the reference lines whose arithmetic it replaces.
 *
 * Conventions (all entry points)
 *   - extern "C", plain pointers and ints; no torch types.
 *   - every pointer is a DEVICE pointer on the current HIP device, float32,
 *     contiguous in the stated layout; the library never allocates.
 *   - asynchronous on `stream` (a hipStream_t passed as void*; NULL = default).
 *   - returns MVS_OK (0) or a negative MVS_E* code and never throws;
 *     mvs_last_error_string() describes the last failure on this thread.
 *   - re-entrant: no global mutable state besides the thread-local error text.
 *
 * Layout codes
 *   MVS_LAYOUT_NCHW (0): [B,C,H,W] / [B,C,D,H,W]   (the reference's layout)
 *   MVS_LAYOUT_NHWC (1): [B,H,W,C] / [B,D,H,W,C]   (channels-last; what the
 *                        MFMA convolution kernels consume and produce)
 *   MVS_LAYOUT_C8   (2): [B,D,H,C/8,W,8]  8-channel blocked volume.  Only as the
 *                        out_layout of mvs_costvol_variance_fwd_f32 (with NHWC
 *                        features) and the input layout of an MFMA
 *                        mvs_conv3d_f32 (whose output is then NHWC): the 32-channel
 *                        variance volume is consumed 8 channels at a time, and this
 *                        keeps every pass on whole, distinct cache lines.
 *   MVS_LAYOUT_C16  (3): [B,C/16,H,W,16]  16-channel blocked FEATURE maps, the
 *                        fea_layout of the LDS-staged variance kernel (a source
 *                        footprint row is one contiguous run); build it with
 *                        mvs_nchw_to_nhwc_f32(in, out, B*C/16, 16, H*W).
 *   MVS_LAYOUT_C4   (4): [B,C/4,H,W,4]    4-channel blocked FEATURE maps, the layout the
 *                        persistent variance kernel copies fastest (a footprint row of one
 *                        channel quad is one contiguous run, so an LDS-DMA instruction touches
 *                        a quarter of the cache lines it touches with C16); only for
 *                        mvs_costvol_variance_fwd_ws_f32 where
 *                        mvs_costvol_variance_workspace_bytes2(..., MVS_LAYOUT_C4, alias_quirk) > 0.
 * depth_mode
 *   0: depth_values [B,D]        (MVSNet, module.py:74)
 *   1: depth_values [B,D,H,W]    (CasMVSNet/models/module.py:249,267;
 *                                 CVP-MVSNet/models/modules.py:253)
 */
#ifndef MVS_HIP_H
#define MVS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: what is declared between this push and its pop is the exported C ABI, nothing else */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define MVS_OK 0
#define MVS_EINVAL (-1)      /* bad argument (null pointer, non-positive size, unsupported shape) */
#define MVS_EUNSUPPORTED (-2) /* combination not implemented by this build */
#define MVS_ELAUNCH (-3)     /* HIP launch / runtime error (see error string) */
#define MVS_EWORKSPACE (-4)  /* workspace too small */

#define MVS_LAYOUT_NCHW 0
#define MVS_LAYOUT_NHWC 1
#define MVS_LAYOUT_C8 2
#define MVS_LAYOUT_C16 3
#define MVS_LAYOUT_C4 4
#define MVS_LAYOUT_C8H 5 /* fp16 pairs of an 8-channel-blocked volume: tuning builds only, include/mvs_hip_tuning.h */
/* An 8-channel-blocked VOLUME as two fp16 pieces per value, hi = fp16(x s), lo = fp16(x s - hi), s a power of two taken from an
 * absmax block (4 bytes per element like fp32; mvs_amd/csrc/conv_f16x3_y8p.hip has the byte layouts).  16-byte piece = the 8
 * channels of one voxel and part; parity = of the voxel's x.
 *   MVS_LAYOUT_C8P  (6): [B, D, C/8, part (hi, lo), parity, H, ceil(W/2)] pieces
 *   MVS_LAYOUT_C8PT (7): [B, D, C/8, ceil(W/32), part, local parity, H, 17] pieces -- x-tiled, each 32-voxel tile with its two halo
 *                        columns (+6 % bytes): conv0 reads ten rows at a time as one run, but a sweep tile's stores straddle cache
 *                        lines (rows of 272 bytes): the sweep is 0.5 ms slower at configs[1] (profiles/r06_handover_sweep.json)
 *   MVS_LAYOUT_C8PH (8): MVS_LAYOUT_C8P per 8-channel chunk, followed by that chunk's halo strips [part, ceil(W/32), side, H] pieces:
 *                        the columns x = 32 t - 1 (side 0) and x = 32 t + 32 (side 1) of every 32-voxel tile t once more, H pieces in
 *                        a row (+6 % bytes; the chunk's block padded to 256 bytes).  What a hand-over sweep writes -- whole aligned
 *                        cache lines per store -- and conv0 reads: a tile's halo column costs it two cache lines per plane, not ten */
#define MVS_LAYOUT_C8P 6
#define MVS_LAYOUT_C8PT 7
#define MVS_LAYOUT_C8PH 8

/* Library version: major*10000 + minor*100 + patch.  101 (0.1.1): every *_f16*_packed_bytes size grew -- the fp32 weights
 * ride behind the packed fragments for the range guard -- so buffers sized by 100 are too small: re-query the sizes.
 * 102 (0.1.2): mvs_conv3d_wgrad_c8_f16_* and mvs_pack_batch_begin / _end added; a planar input of mvs_conv2d_wgrad_f32 is the image (Cin <= 4, else
 * MVS_EUNSUPPORTED); mvs_costreg_fwd3_f32, the fused tail entry points and mvs_conv2d_pair_* added; mvs_costreg_workspace_bytes grew
 * by 64 flag words (re-query it).
 * 103 (0.1.3): the hand-over of the variance volume as fp16 pieces -- mvs_costvol_variance_fwd_ws3_f32 / _handover_bytes,
 * mvs_conv3d_c8p_f16x3_f32, mvs_conv3d_c8_handed_f16x3_f32, mvs_c8p_bytes, mvs_c8_to_c8p_f32, mvs_conv3d_f16x3_pack_veto_word, mvs_costreg_fwd4_f32; a launch
 * under a run-only-if word that has no kernel honouring it now fails with MVS_EUNSUPPORTED instead of running unconditionally;
 * mvs_costreg_tail_guarded_f16_f32 needs prob's packed weights.  Nothing a 102 caller sized or packed changes. */
int mvs_version(void);
/* Text of the last error on the calling thread ("" if none). */
const char *mvs_last_error_string(void);
/* Name of the compiled GPU architecture ("gfx950"). */
const char *mvs_arch(void);

/* ---- layout helpers ------------------------------------------------- */
/* [B,C,S] -> [B,S,C] and back (S = H*W or D*H*W). */
int mvs_nchw_to_nhwc_f32(const float *in, float *out, int B, int C, int64_t S, void *stream);
int mvs_nhwc_to_nchw_f32(const float *in, float *out, int B, int C, int64_t S, void *stream);

/* ---- input pipeline on the device ----------------- */
/* What the reference's loader does to a decoded image, MVSNet/datasets/dtu_yao_eval.py:60-67,102:
 * np.array(img, float32) / 255. (IEEE division), crop to the top-left H x W (the loader drops the
 * bottom 16 rows of 1600x1200), HWC -> CHW.  in: N decoded images [N,Hs,Ws,3] uint8 on the device;
 * out [N,3,H,W] float32 -- bit-identical to the loader's tensors. */
int mvs_images_u8_to_planar_f32(const unsigned char *in, int N, int Hs, int Ws, int H, int W, float *out,
                                void *stream);
/* dtu_yao_eval.py:54,93-95: out = E with its top 3x4 block replaced by (K with rows 0 and 1 divided by
 * intrinsics_div) @ E[:3,:4]; K [N,3,3], E [N,4,4], out [N,4,4].  The dot products are evaluated as
 * numpy's float32 matmul evaluates them (rounded first product, two FMAs). */
int mvs_proj_matrices_f32(const float *K, const float *E, float intrinsics_div, int N, float *out,
                          void *stream);

/* ---- K1: plane-sweep warp -- MVSNet/models/module.py:46-87 ---------- */
/* src_fea [B,C,H,W]; rot_trans [B,12] = rows of (src_proj @ inverse(ref_proj))[:3,:4]
 * (module.py:63-65, evaluated by the caller with torch); out [B,C,D,H,W].
 * align_corners: 0 = what the reference's grid_sample call runs with on
 * torch >= 1.3 (the parity target), 1 = torch 1.2 / MVSNet_pl behaviour. */
int mvs_warp_fwd_f32(const float *src_fea, const float *rot_trans, const float *depth_values,
                     int depth_mode, int B, int C, int D, int H, int W, int align_corners,
                     float *out, void *stream);
/* grad_src [B,C,H,W] (overwritten) = d/d src_fea of <grad_out, warp(src)>;
 * the grid carries no gradient (module.py:62). */
int mvs_warp_bwd_f32(const float *grad_out, const float *rot_trans, const float *depth_values,
                     int depth_mode, int B, int C, int D, int H, int W, int align_corners,
                     float *grad_src, void *stream);

/* ---- K1+K2: fused warp + variance -- mvsnet.py:152-170 -------------- */
/* ref_fea [B,C,H,W]|[B,H,W,C]; src_feas [V-1,B,...] same layout; rot_trans
 * [V-1,B,12]; out_var [B,C,D,H,W] | [B,D,H,W,C].  Per-view warped volumes are
 * never materialised.  fea_layout C16 (with out_layout NHWC or C8) selects the
 * LDS-staged kernel: per-view source footprints of a 4x16x4-voxel block are
 * staged in LDS and sampled from there.  alias_quirk=1 reproduces CVP-MVSNet/models/modules.py:
 * 228-229 (S0 = Q0 = ref^2).  C must be a multiple of 4 and <= 64. */
int mvs_costvol_variance_fwd_f32(const float *ref_fea, const float *src_feas,
                                 const float *rot_trans, const float *depth_values,
                                 int depth_mode, int B, int V, int C, int D, int H, int W,
                                 int align_corners, int alias_quirk, int fea_layout,
                                 int out_layout, float *out_var, void *stream);
/* The same operation with a caller workspace: for shared depth planes (depth_mode 0, no alias quirk, C % 16 == 0)
 * this runs the persistent kernel -- one workgroup per CU walks (16x4 pixels x 16 or 8 depth planes) tiles, the source
 * footprints of the next tile land in LDS by DMA while the current one is sampled -- or, when the footprints of such
 * tiles outgrow LDS (wide baselines against the depth range of a tile), the per-tile kernel of
 * mvs_costvol_variance_fwd_f32.  WHICH of the three is decided on the device, per call, from the cameras and depth
 * planes themselves: a one-workgroup kernel projects the corner voxels of sample tiles and writes the choice into the
 * workspace header; all candidates are enqueued behind it and begin by reading that word (MVS_SWEEP_PERSIST=16|8|0 in
 * the environment forces one).  The workspace holds that header, the queue of the persistent kernel's cold path
 * (waves whose footprint does not fit LDS are served by a gather-based kernel) and, for C4 features, room for their
 * 16-channel-blocked copy (made only if the per-tile kernel is chosen).  Every other shape takes the kernels of
 * mvs_costvol_variance_fwd_f32 and ignores the workspace (which may then be NULL / 0 bytes;
 * mvs_costvol_variance_workspace_bytes2 returns 0 -- exactly when the persistent kernel does not take the shape: the
 * query and the launcher share one predicate).  The persistent kernel also reads channels-last maps [B,H,W,C]
 * (MVS_LAYOUT_NHWC) in place.  Results are bit-identical to
 * mvs_costvol_variance_fwd_f32 unless flags has MVS_SWEEP_FAST: coordinates from one refined
 * reciprocal per voxel and view with the reference's normalise / un-normalise pair
 * (module.py:78-79 + grid_sample) folded into one FMA, Q += w*w as an FMA, multiplication by
 * 1/V -- sampling positions within ~1e-4 texel of the reference's. */
#define MVS_SWEEP_FAST 1
/* mvs_costvol_variance_fwd_ws*_f32, device-selected kernel: launch the per-tile CANDIDATE as one block per tile (fastest when the
 * chooser picks it: wide footprints) instead of the looping form (cheapest when it does not: the usual case).  A speed hint only:
 * the result is the same.  The word the chooser writes -- workspace word 1: 16 / 8 = the persistent kernel's tile depth, 0 = the
 * per-tile kernel -- tells a caller which way its geometry falls. */
#define MVS_SWEEP_TILE_CANDIDATE_PER_TILE 2
size_t mvs_costvol_variance_workspace_bytes(int depth_mode, int B, int V, int C, int D, int H, int W,
                                            int fea_layout);   /* = ..._bytes2(..., alias_quirk 0) */
size_t mvs_costvol_variance_workspace_bytes2(int depth_mode, int B, int V, int C, int D, int H, int W,
                                             int fea_layout, int alias_quirk);
int mvs_costvol_variance_fwd_ws_f32(const float *ref_fea, const float *src_feas,
                                    const float *rot_trans, const float *depth_values,
                                    int depth_mode, int B, int V, int C, int D, int H, int W,
                                    int align_corners, int alias_quirk, int fea_layout,
                                    int out_layout, int flags, float *out_var, void *workspace,
                                    size_t workspace_bytes, void *stream);
/* The same, and the largest magnitude of the volume as a by-product: absmax = NULL, or an ABSMAX BLOCK -- MVS_ABSMAX_WORDS
 * 32-bit device words, 16-byte aligned, that together hold max |x| of an array as the bit pattern of a non-negative float:
 * word i collects the workgroups with index i mod MVS_ABSMAX_WORDS (one shared word made the atomics of thousands of waves
 * queue up at the end of a kernel), the reader takes the maximum of all words.  It is the operand scale
 * mvs_conv3d_c8_f16x3_f32 / mvs_costreg_fwd2_f32 / mvs_conv_split_f16_f32 need.  This call RESETS the block; the LDS-staged
 * kernels then collect as they store (one atomic max per wave); behind the gather kernels it costs one more pass over the
 * volume (mvs_absmax_f32). */
#define MVS_ABSMAX_WORDS 256
int mvs_costvol_variance_fwd_ws2_f32(const float *ref_fea, const float *src_feas,
                                     const float *rot_trans, const float *depth_values,
                                     int depth_mode, int B, int V, int C, int D, int H, int W,
                                     int align_corners, int alias_quirk, int fea_layout,
                                     int out_layout, int flags, float *out_var, void *workspace,
                                     size_t workspace_bytes, void *absmax, void *stream);
/* Self-test: the variance kernel divides by the view count V with a 3-op
 * multiply/FMA sequence instead of an IEEE division; this checks it against
 * x / V for EVERY float bit pattern on the device and writes the number of
 * mismatching patterns (expected 0) to *mismatch_count (device pointer). */
int mvs_selftest_div_by_views_f32(int V, unsigned long long *mismatch_count, void *stream);
/* grad_var in out_layout; grad_ref / grad_srcs in fea_layout, overwritten. */
int mvs_costvol_variance_bwd_f32(const float *grad_var, const float *ref_fea,
                                 const float *src_feas, const float *rot_trans,
                                 const float *depth_values, int depth_mode, int B, int V, int C,
                                 int D, int H, int W, int align_corners, int fea_layout,
                                 int out_layout, float *grad_ref, float *grad_srcs, void *stream);

/* ---- K3: CostRegNet layers -- mvsnet.py:48-93, module.py:26-33 ------ */
/* One 3x3x3 convolution (pad 1, stride 1|2, no bias) or transposed
 * convolution (stride 2, pad 1, output_padding 1 -> exactly 2x; stride 1 for
 * CVP net.py:66-69), followed by y = acc*scale[co] + shift[co] (eval-mode
 * BatchNorm3d folded to an affine; NULL = identity / conv bias in `shift`),
 * optional ReLU, optional residual add AFTER the ReLU (mvsnet.py:89-91).
 * weight: PyTorch layout -- conv (Cout,Cin,3,3,3), transposed (Cin,Cout,3,3,3).
 * in [B,Cin,D,H,W]|[B,D,H,W,Cin]; out/residual in the same layout family with
 * Cout channels at the output resolution.
 * impl: 0 = auto, 1 = direct (VALU) kernel, 2 = MFMA implicit-GEMM kernel
 * (channels-last only).  `packed_weight` (from mvs_conv3d_pack_weights_f32) is
 * required by impl 2 and ignored by impl 1; with impl 0 it is used if given. */
int mvs_conv3d_f32(const float *in, const float *weight, const float *packed_weight,
                   const float *scale, const float *shift, const float *residual, int relu,
                   int transposed, int B, int Cin, int Cout, int D, int H, int W, int stride,
                   int layout, int impl, float *out, void *stream);
/* The same, and the largest magnitude of `out` into the absmax block out_absmax (MVS_ABSMAX_WORDS words, see
 * mvs_costvol_variance_fwd_ws2_f32; NULL = none): the MFMA convolutions collect it in their epilogue (max-ed INTO the block:
 * the caller clears it); behind the other kernels it costs one more pass over `out` (mvs_absmax_f32).  For a layer whose
 * output feeds a two-piece fp16 layer (mvs_conv_split_f16_f32 ...). */
int mvs_conv3d_absmax_f32(const float *in, const float *weight, const float *packed_weight,
                          const float *scale, const float *shift, const float *residual, int relu,
                          int transposed, int B, int Cin, int Cout, int D, int H, int W, int stride,
                          int layout, int impl, float *out, void *out_absmax, void *stream);
/* Number of floats mvs_conv3d_pack_weights_f32 writes for this layer. */
int64_t mvs_conv3d_packed_weight_floats(int transposed, int Cin, int Cout, int stride);
/* Re-order a PyTorch-layout weight tensor into MFMA A-fragment order. */
int mvs_conv3d_pack_weights_f32(const float *weight, int transposed, int Cin, int Cout,
                                int stride, float *packed, void *stream);
/* 1 if impl 2 (MFMA) supports this layer shape, else 0. */
int mvs_conv3d_mfma_supported(int transposed, int Cin, int Cout, int stride);

/* Cout = 8, stride-1 layers over an 8-channel-blocked input (conv0 of every CostRegNet: mvsnet.py:66,
 * CasMVSNet/models/module.py:411) on the BF16 matrix pipe at fp32 accuracy: every fp32 operand is split
 * exactly into three bf16 numbers (hi + mid + lo) and the six largest partial products are accumulated in
 * fp32 -- error <= the fp32 kernels' own (mvs_amd/csrc/conv_bf16x6.hip), 2.7x less matrix time.
 * `packed` = mvs_conv3d_pack_weights_bf16x6_f32 of the PyTorch-layout (8, Cin, 3, 3, 3) weight,
 * mvs_conv3d_bf16x6_packed_bytes(Cin) bytes (0 = unsupported Cin; supported: 8, 16, 32).
 * in [B,D,H,Cin/8,W,8]; out / residual [B,D,H,W,8]; scale/shift/relu/residual as mvs_conv3d_f32. */
size_t mvs_conv3d_bf16x6_packed_bytes(int Cin);
int mvs_conv3d_pack_weights_bf16x6_f32(const float *weight, int Cin, void *packed, void *stream);
int mvs_conv3d_c8_bf16x6_f32(const float *in, const void *packed, const float *scale, const float *shift,
                             const float *residual, int relu, int B, int Cin, int D, int H, int W,
                             float *out, void *stream);

/* The same layers on the FP16 matrix pipe with TWO-piece operands and three products per fp32 product (half the
 * matrix work of the bf16 form; mvs_amd/csrc/conv_f16x3.hip): x * 2^k = hi + lo + e, hi and lo fp16, |e| <= 2^-23 |x 2^k|;
 * a b ~ ah bh + ah bl + al bh, each product exact in the fp32 accumulator.  Against a float64 convolution the result is
 * as close as the bf16 form's and ATen's fp32 convolution (the fp32 accumulation, common to all three, is what sets the
 * distance; tests/test_gpu_parity.py::test_conv3d_f16x3_*).  fp16 has 5 exponent bits, so the kernel scales its input
 * by 2^(14 - exponent(max |in|)): in_absmax = the absmax block (MVS_ABSMAX_WORDS words, see
 * mvs_costvol_variance_fwd_ws2_f32) of `in`, filled by the producer of the volume or by mvs_absmax_f32 (one pass over the
 * array: resets the block, then collects; n floats, 16-byte aligned).  An element below 2^-18 of the maximum keeps an
 * absolute error below 2^-40 of the maximum.  The weights are scaled the same way when packed:
 * mvs_conv3d_f16x3_packed_bytes(Cin) bytes (0 = unsupported Cin; supported: 8, 16, 32).  out_absmax: NULL, or the
 * absmax block the largest magnitude of `out` is atomically max-ed INTO as it is stored (the next layer's operand scale;
 * the caller clears it -- one memset serves the blocks of a whole network).  Other arguments as mvs_conv3d_c8_bf16x6_f32.
 *
 * RANGE GUARD (mvs_amd/csrc/conv_guard.h; the same in mvs_conv_split_f16_f32 / mvs_deconv_split_f16_f32 and, through them,
 * mvs_costreg_fwd2_f32).  One scale per tensor cannot serve every input: an Inf or NaN voxel (the reference keeps its damage
 * inside the receptive field, module.py:83-84 -> mvsnet.py:83-93) or a few outliers far above the rest would push everything
 * else out of fp16's range.  Every launch therefore first judges in_absmax ON THE DEVICE: maximum not finite, or fewer than one
 * in eight of the block's non-zero words within 2^-16 of it (the maximum is carried by outliers), or weights not finite -> the
 * launch computes the layer in plain fp32 from the ORIGINAL weights (kept behind the packed fragments: packed_bytes includes
 * them), real taps only, IEEE Inf / NaN semantics and torch's NaN-propagating ReLU -- the reference's arithmetic, at the speed
 * of a direct convolution.  Otherwise the bound above holds: per product 2^-22 relative for operands within 2^-18 of their
 * tensor's maximum, 2^-40 of the maximum absolute for smaller ones.  mvs_guard_fallback_count reports how many launches of the
 * current device took the fp32 path since the library was loaded (it synchronises: a diagnostic, not a data-path call; call it
 * once before capturing HIP graphs of these layers -- the first call resolves the counter's device address).
 *
 * LIMITS OF THE VERDICT (it reads only the 256 words of the block, each the maximum over a share of the producer's
 * workgroups): a block with ONE non-zero word -- a producer that ran a single workgroup, or mvs_absmax_f32 over an array
 * small enough for one -- has nothing to compare and always passes; outliers spread over an eighth or more of the words are
 * the tensor's range as far as the verdict can tell.  In both cases elements more than 2^18 below the maximum keep the
 * ABSOLUTE bound (2^-40 of the maximum), not the relative one.  A caller that cannot rule such inputs out runs the exact
 * three-piece kernels (mvs_conv3d_c8_bf16x6_f32 / mvs_conv_split_f32 / mvs_deconv_split_f32), which need no scale.
 * mvs_guard_resolve_all_devices(): resolve the counter on every visible device (call once, before any HIP-graph capture, in a
 * multi-device process). */
int mvs_guard_fallback_count(unsigned long long *count);
int mvs_guard_resolve_all_devices(void);
/* 0 when `stream` is not capturing a HIP graph, else an id that differs between captures (hipStreamGetCaptureInfo). */
int mvs_stream_capture_id(void *stream, unsigned long long *id);
size_t mvs_conv3d_f16x3_packed_bytes(int Cin);
int mvs_conv3d_pack_weights_f16x3_f32(const float *weight, int Cin, void *packed, void *stream);
int mvs_absmax_f32(const float *x, int64_t n, void *absmax, void *stream);
int mvs_conv3d_c8_f16x3_f32(const float *in, const void *in_absmax, const void *packed, const float *scale,
                            const float *shift, const float *residual, int relu, int B, int Cin, int D, int H, int W,
                            float *out, void *out_absmax, void *stream);

/* The same split-operand arithmetic for the 3x3(x3), stride-1, pad-1 layers with 16 / 32 / 64 input and output
 * channels (CostRegNet conv2 / conv4 / conv6, mvsnet.py:68-72; FeatureNet's 16 -> 16 and 32 -> 32 layers,
 * mvsnet.py:21-27; the same shapes in CasMVSNet/models/module.py and CVP-MVSNet/models/net.py:22-97):
 * mvs_amd/csrc/conv_split.hip.  kd = 3: volume [B,D,H,W,Cin] -> [B,D,H,W,Cout]; kd = 1: B*D images
 * [B,D,H,W,Cin] convolved plane by plane (pass D = number of images, B = 1).  weight: PyTorch layout
 * (Cout, Cin, [kd,] 3, 3); relu: 0 none, 1 ReLU, 2 LeakyReLU(0.1); scale / shift / residual as mvs_conv3d_f32;
 * out_c4 = 1: the output is written as 4-channel blocks [B*D, Cout/4, H, W, 4] (MVS_LAYOUT_C4; no residual).
 * Also kd = 3, stride 1, (Cin, Cout) = (8, 32): the input gradient of a 32 -> 8 layer (conv0) in training.
 * stride = 2 with kd = 3 (Cin in {8, 16, 32}: CostRegNet conv1 / conv3 / conv5, mvsnet.py:67-71): output
 * [B, (D-1)/2+1, (H-1)/2+1, (W-1)/2+1, Cout].  stride = 2 with kd = 1 is FeatureNet's 5x5, pad-2 form
 * (mvsnet.py:13,16 `ConvBnReLU(8, 16, 5, 2, 2)`, `ConvBnReLU(16, 32, 5, 2, 2)`; CasMVSNet/models/module.py:323,329):
 * weight (Cout, Cin, 5, 5) with (Cin, Cout) = (8, 16) or (16, 32), output [B*D, (H-1)/2+1, (W-1)/2+1, Cout]. */
int mvs_conv_split_supported(int kd, int Cin, int Cout, int stride);
size_t mvs_conv_split_packed_bytes(int kd, int Cin, int Cout, int stride);
int mvs_conv_split_pack_weights_f32(const float *weight, int kd, int Cin, int Cout, int stride, void *packed, void *stream);
/* Several packs as ONE launch: between mvs_pack_batch_begin() and mvs_pack_batch_end(stream) on a thread, every
 * mvs_conv_split_pack_weights_f32 call records its job (weight and packed pointers must stay valid) instead of launching;
 * _end launches them together on `stream` (a training step re-packs every layer: 26 launches of microseconds of work).
 * Other entry points are unaffected.  Errors: a second _begin, an _end without _begin (MVS_EINVAL). */
int mvs_pack_batch_begin(void);
int mvs_pack_batch_end(void *stream);
int mvs_conv_split_f32(const float *in, const void *packed, const float *scale, const float *shift,
                       const float *residual, int relu, int kd, int stride, int B, int Cin, int Cout, int D, int H, int W,
                       int out_c4, float *out, void *stream);

/* ... and for the transposed layers (3x3x3, stride 2, pad 1, output_padding 1: CostRegNet conv7 / conv9 / conv11,
 * mvsnet.py:76-81,89-91): mvs_amd/csrc/deconv_split.hip.  in [B,D,H,W,Cin] -> out [B,2D,2H,2W,Cout];
 * weight: PyTorch layout (Cin, Cout, 3, 3, 3); residual (same shape as out) is added AFTER the ReLU.
 * Cin in {16, 32, 64}, Cout in {8, 16, 32}. */
int mvs_deconv_split_supported(int Cin, int Cout);
size_t mvs_deconv_split_packed_bytes(int Cin, int Cout);
int mvs_deconv_split_pack_weights_f32(const float *weight, int Cin, int Cout, void *packed, void *stream);
int mvs_deconv_split_f32(const float *in, const void *packed, const float *scale, const float *shift,
                         const float *residual, int relu, int B, int Cin, int Cout, int D, int H, int W,
                         float *out, void *stream);

/* The two-piece fp16 form of both (mvs_conv3d_c8_f16x3_f32 has the arithmetic and the error bound; same shapes, same
 * supported-predicates): weights scaled and packed by the *_pack_weights_f16_f32 functions (packed_bytes includes a 16-byte
 * trailer); in_absmax = the absmax block of `in` (required); out_absmax = NULL, or the absmax block the largest magnitude
 * of `out` is max-ed INTO in the epilogue (the caller clears it; the next two-piece layer's in_absmax). */
size_t mvs_conv_split_f16_packed_bytes(int kd, int Cin, int Cout, int stride);
int mvs_conv_split_pack_weights_f16_f32(const float *weight, int kd, int Cin, int Cout, int stride, void *packed, void *stream);
int mvs_conv_split_f16_f32(const float *in, const void *in_absmax, const void *packed, const float *scale,
                           const float *shift, const float *residual, int relu, int kd, int stride, int B, int Cin,
                           int Cout, int D, int H, int W, int out_c4, float *out, void *out_absmax, void *stream);
/* ---- two consecutive 3x3 stride-1 C -> C layers of a 2D net as one kernel (mvs_amd/csrc/conv2d_pair.hip; C = 16: conv3 + conv4 of
 * FeatureNet, MVSNet/models/mvsnet.py:15-16,37-38) -- the map between them stays in LDS.  Two-piece fp16 operands; the
 * intermediate map is scaled by a power of two from a BOUND on its magnitude (the input's absmax block, the first layer's largest
 * weight and affine).  in [N, H, W, C] channels-last with its absmax block; packed_pair: mvs_conv2d_pair_packed_bytes(C) bytes from
 * mvs_conv2d_pair_pack_weights_f32(w1, w2: (C, C, 3, 3) as PyTorch stores them); scale / shift per layer (NULL = 1 / 0); the first
 * layer ends in a ReLU, the second in one if relu2; out [N, H, W, C] or (out_c4) [N, C/4, H, W, 4]; out_absmax: NULL or a zeroed block.
 * RANGE GUARD as mvs_costreg_tail_f16_f32: a launch that fails it writes 1 to *fallback_flag and computes nothing.
 * mvs_conv2d_pair_guarded_f16_f32 = that launch with the two layers of mvs_conv_split_f16_f32 (packed1_f16 / packed2_f16: their own
 * packs) enqueued behind it so that they run only if the flag was set; flag: MVS_ABSMAX_WORDS + 64 zeroed words (the flag, then the
 * block of the unfused path's intermediate map); mid_scratch: [N, H, W, C] floats that only the unfused path touches. */
size_t mvs_conv2d_pair_packed_bytes(int C);
int mvs_conv2d_pair_supported(int C, int N, int H, int W);
int mvs_conv2d_pair_pack_weights_f32(const float *w1, const float *w2, int C, void *packed, void *stream);
int mvs_conv2d_pair_f16_f32(const float *in, const void *in_absmax, const void *packed_pair, const float *scale1,
                            const float *shift1, const float *scale2, const float *shift2, int relu2, int N, int C,
                            int H, int W, int out_c4, float *out, void *out_absmax, void *fallback_flag, void *stream);
int mvs_conv2d_pair_guarded_f16_f32(const float *in, const void *in_absmax, const void *packed_pair, const void *packed1_f16,
                                    const void *packed2_f16, const float *scale1, const float *shift1, const float *scale2,
                                    const float *shift2, int relu2, int N, int C, int H, int W, int out_c4, float *mid_scratch,
                                    float *out, void *out_absmax, void *flag, void *stream);
size_t mvs_deconv_split_f16_packed_bytes(int Cin, int Cout);
int mvs_deconv_split_pack_weights_f16_f32(const float *weight, int Cin, int Cout, void *packed, void *stream);
int mvs_deconv_split_f16_f32(const float *in, const void *in_absmax, const void *packed, const float *scale,
                             const float *shift, const float *residual, int relu, int B, int Cin, int Cout, int D,
                             int H, int W, float *out, void *out_absmax, void *stream);

/* The whole 3D U-Net in one call -- CostRegNet.forward, mvsnet.py:83-93 (also the cascade's
 * CostRegNet, CasMVSNet/models/module.py:407-438): conv0 .. conv6 (3x3x3 + folded BN + ReLU, strides
 * 1 2 1 2 1 2 1), conv7 / conv9 / conv11 (transposed, stride 2, + BN + ReLU, skip-add of conv4 /
 * conv2 / conv0 after the ReLU), prob (3x3x3 to one channel, `shift` = its bias or NULL, no ReLU).
 * layers[11] in that order; widths Cin -> base, 2 base, 2 base, 4 base, 4 base, 8 base, 8 base,
 * 4 base, 2 base, base, 1 (base = 8 in both networks).  in: the variance volume, channels-last
 * [B,D,H,W,Cin] (MVS_LAYOUT_NHWC) or 8-channel blocked [B,D,H,Cin/8,W,8] (MVS_LAYOUT_C8);
 * out_cost [B,D,H,W].  D, H, W multiples of 8.  Every intermediate activation lives in
 * `workspace` (mvs_costreg_workspace_bytes; nothing is allocated), the layers are the kernels
 * of mvs_conv3d_f32 enqueued back to back on `stream`. */
typedef struct mvs_conv_layer {
    const float *weight; /* PyTorch layout: conv (Cout,Cin,3,3,3), transposed (Cin,Cout,3,3,3) */
    const float *packed; /* mvs_conv3d_pack_weights_f32 output, or NULL (direct kernels) */
    const float *scale;  /* folded BatchNorm scale [Cout], or NULL */
    const float *shift;  /* folded BatchNorm shift / conv bias [Cout], or NULL */
    const void *packed_split; /* conv0: mvs_conv3d_pack_weights_bf16x6_f32 output -- with an MVS_LAYOUT_C8
                               * input the layer then runs as mvs_conv3d_c8_bf16x6_f32; conv2 / conv4 / conv6:
                               * mvs_conv_split_pack_weights_f32 output -> mvs_conv_split_f32; conv7 / conv9 / conv11:
                               * mvs_deconv_split_pack_weights_f32 output -> mvs_deconv_split_f32;
                               * NULL = the fp32 MFMA kernel */
} mvs_conv_layer;
size_t mvs_costreg_workspace_bytes(int B, int base, int D, int H, int W);
int mvs_costreg_fwd_f32(const float *in, int in_layout, const mvs_conv_layer *layers, int B, int Cin,
                        int base, int D, int H, int W, int impl, void *workspace,
                        size_t workspace_bytes, float *out_cost, void *stream);
/* The same with layers on the two-piece fp16 kernels (three products per fp32 product instead of six): packed_f16[11] =
 * per layer, in the order of `layers`, its two-piece pack or NULL -- conv0: mvs_conv3d_pack_weights_f16x3_f32 (taken for an
 * MVS_LAYOUT_C8 input); conv1 .. conv6: mvs_conv_split_pack_weights_f16_f32; conv7 / conv9 / conv11:
 * mvs_deconv_split_pack_weights_f16_f32; a NULL entry (and prob) runs as in mvs_costreg_fwd_f32.  The operand scale of such
 * a layer is the absmax block of its input, which the layer in front of it collects in its epilogue into the workspace
 * (behind conv3 / conv5 on the fp32 kernels: one more pass over their small outputs).  in_absmax = the absmax block the
 * producer of `in` filled (mvs_costvol_variance_fwd_ws2_f32), or NULL: collected here by one more pass over `in`. */
int mvs_costreg_fwd2_f32(const float *in, int in_layout, const mvs_conv_layer *layers, const void *const *packed_f16,
                         int B, int Cin, int base, int D, int H, int W, int impl, void *workspace, size_t workspace_bytes,
                         const void *in_absmax, float *out_cost, void *stream);

/* ---- the tail of CostRegNet as one kernel (mvs_amd/csrc/tail_fused.hip; MVSNet/models/mvsnet.py:79-81,89-93) -------
 * x = conv0 + conv11(x) (ConvTranspose3d 16 -> 8, k3 s2 p1 op1, folded BatchNorm, ReLU, skip add after the ReLU) followed by
 * prob (Conv3d 8 -> 1, k3 p1, bias): the full-resolution 8-channel volume between them never reaches HBM.  Two-piece fp16
 * operands as mvs_conv3d_c8_f16x3_f32 (same bound per product; the intermediate volume is scaled by a power of two derived from
 * a BOUND on its magnitude -- the two absmax blocks, the weights' largest magnitude, the affine -- so no pass over it is needed).
 *   in [B, Di, Hi, Wi, 16] channels-last with its absmax block; skip [B, 2Di, 2Hi, 2Wi, 8] with its block;
 *   packed_tail: mvs_costreg_tail_packed_bytes() bytes written by mvs_costreg_tail_pack_weights_f32 from conv11's weight
 *   (16, 8, 3, 3, 3); scale / shift: conv11's 8 + 8 affine (NULL = 1 / 0); prob_weight (1, 8, 3, 3, 3) as PyTorch stores it,
 *   prob_scale / prob_shift: 1 + 1 (NULL = 1 / 0; shift = the bias); out_cost [B, 2Di, 2Hi, 2Wi].
 * RANGE GUARD: the verdict of mvs_conv3d_c8_f16x3_f32 on BOTH blocks, finite weights, a finite bound.  A launch that fails it
 * computes nothing and writes 1 to *fallback_flag (a device word the caller zeroed); the caller then runs the unfused layers
 * (mvs_deconv_split_f16_f32 + mvs_conv3d_f32), whose own guard reproduces the reference's Inf / NaN semantics.
 * mvs_costreg_fwd3_f32 = mvs_costreg_fwd2_f32 with that arrangement inside: packed_tail non-NULL (and base = 8) routes conv11 + prob
 * through the fused kernel and enqueues the unfused layers behind it so that they run only if the flag was set -- no host
 * synchronisation.  mvs_costreg_workspace_bytes covers the flag. */
size_t mvs_costreg_tail_packed_bytes(void);
int mvs_costreg_tail_pack_weights_f32(const float *conv11_weight, void *packed, void *stream);
int mvs_costreg_tail_supported(int B, int Di, int Hi, int Wi);
int mvs_costreg_tail_f16_f32(const float *in, const void *in_absmax, const float *skip, const void *skip_absmax,
                             const void *packed_tail, const float *scale, const float *shift, const float *prob_weight,
                             const float *prob_scale, const float *prob_shift, int B, int Di, int Hi, int Wi,
                             float *out_cost, void *fallback_flag, void *stream);
/* The fused kernel with the unfused layers enqueued behind it under its flag (what mvs_costreg_fwd3_f32 does for conv11 + prob):
 * conv11 / prob = the layers' descriptors (weight, packed, scale, shift), conv11_f16 = conv11's mvs_deconv_split_f16 pack,
 * d11_scratch = [B, 2Di, 2Hi, 2Wi, 8] floats the unfused path writes conv11's output to, *flag = 0 on entry. */
int mvs_costreg_tail_guarded_f16_f32(const float *in, const void *in_absmax, const float *skip, const void *skip_absmax,
                                     const void *packed_tail, const mvs_conv_layer *conv11, const void *conv11_f16,
                                     const mvs_conv_layer *prob, int B, int Di, int Hi, int Wi, float *d11_scratch,
                                     float *out_cost, void *flag, void *stream);
int mvs_costreg_fwd3_f32(const float *in, int in_layout, const mvs_conv_layer *layers, const void *const *packed_f16,
                         const void *packed_tail, int B, int Cin, int base, int D, int H, int W, int impl, void *workspace,
                         size_t workspace_bytes, const void *in_absmax, float *out_cost, void *stream);

/* ---- hand-over of the variance volume as fp16 pieces (round 6) ------------------------------------------------------------
 * conv0 on the two-piece fp16 kernel spent a third of its time turning fp32 planes into pieces between two barriers.  The sweep
 * holds every variance in a register anyway: it can store the pieces, under a scale known BEFORE it runs -- var = E[x^2] - E[x]^2
 * <= max|f|^2 over all feature maps, whose absmax block FeatureNet's last layer collects.  No host synchronisation anywhere:
 *
 *   mvs_costvol_variance_fwd_ws3_f32   = ..._ws2_f32 for shared depth planes on the device-selected persistent kernels (else
 *       MVS_EUNSUPPORTED, nothing launched), fea_layout C16 / C4 / NHWC.  out_volume (mvs_costvol_variance_handover_bytes) receives
 *       MVS_LAYOUT_C8PH pieces scaled by 2^(14 - exponent(bound)), hand (MVS_ABSMAX_WORDS words) the bits of the bound, var_absmax the
 *       volume's TRUE largest magnitude, *redo = 0 -- or, *redo = 1, the plain fp32 MVS_LAYOUT_C8 volume: when the chooser took the
 *       per-tile kernel, when the bound is not finite, when *reader_veto (optional: one float of the reader, e.g.
 *       mvs_conv3d_f16x3_pack_veto_word) is NaN, or when the pieces do not hold (true maximum not finite or outlier-dominated --
 *       conv_guard.h's rule -- or more than 8 binary orders below the bound): in that last case a referee launch behind the sweep
 *       computes the volume again in fp32 (slow; the path of broken inputs).
 *   mvs_conv3d_c8p_f16x3_f32   conv0 (3x3x3, Cout 8, stride 1) on such pieces, eight-row tiles, copies straight into the plane ring:
 *       returns at once if redo != NULL and *redo != 0.  in_absmax = the block the pieces were scaled by (`hand`).  Same results bit
 *       for bit as mvs_conv3d_c8_f16x3_f32 on the fp32 volume with that block.  npair = 4.
 *   mvs_costreg_fwd4_f32   = fwd3 on the sweep's output: conv0 on the pieces, then mvs_conv3d_c8_f16x3_f32 on the fp32 volume under
 *       *redo as its run-only-if word -- exactly one of the two runs.
 *   mvs_c8_to_c8p_f32 / mvs_c8p_bytes   an fp32 MVS_LAYOUT_C8 volume into pieces (tests, other producers). */
size_t mvs_costvol_variance_handover_bytes(int B, int C, int D, int H, int W);
int mvs_costvol_variance_fwd_ws3_f32(const float *ref_fea, const float *src_feas, const float *rot_trans,
                                     const float *depth_values, int B, int V, int C, int D, int H, int W,
                                     int align_corners, int fea_layout, int flags, const void *fea_absmax,
                                     const void *reader_veto, void *out_volume, void *workspace, size_t workspace_bytes,
                                     void *var_absmax, void *hand, void *redo, void *stream);
size_t mvs_c8p_bytes(int B, int C, int D, int H, int W, int layout);
int mvs_c8_to_c8p_f32(const float *in_c8, const void *absmax, int B, int C, int D, int H, int W, int layout, void *out_pairs, void *stream);
int mvs_conv3d_c8p_f16x3_f32(const void *in_pairs, const void *in_absmax, const void *redo, const void *packed, const float *scale,
                             const float *shift, const float *residual, int relu, int B, int Cin, int D, int H, int W,
                             int layout, int npair, float *out, void *out_absmax, void *stream);
/* conv0 on whatever the sweep left: the kernel on the pieces, then mvs_conv3d_c8_f16x3_f32 on the fp32 volume under *redo as its
 * run-only-if word -- exactly one of the two writes out (and maxes into out_absmax) */
int mvs_conv3d_c8_handed_f16x3_f32(const void *volume, const void *hand, const void *redo, const void *var_absmax,
                                   const void *packed, const float *scale, const float *shift, const float *residual, int relu,
                                   int B, int Cin, int D, int H, int W, float *out, void *out_absmax, void *stream);
const void *mvs_conv3d_f16x3_pack_veto_word(const void *packed, int Cin);
int mvs_costreg_fwd4_f32(const void *in_volume, const void *hand, const void *redo, const void *var_absmax,
                         const mvs_conv_layer *layers, const void *const *packed_f16, const void *packed_tail, int B, int Cin,
                         int base, int D, int H, int W, int impl, void *workspace, size_t workspace_bytes, float *out_cost,
                         void *stream);

/* Weight gradient of one 3x3x3 layer (training, BASELINE config 5; the reference gets it from
 * autograd through nn.Conv3d / nn.ConvTranspose3d, module.py:26-33, mvsnet.py:66-79):
 *   grad_weight[co][ci][kz][ky][kx] += sum_o grad_out[o][co] * in[o*stride + k - 1][ci]
 * in [B,D,H,W,Cin] and grad_out [B,Do,Ho,Wo,Cout] channels-last, Do = (D-1)/stride + 1, ...;
 * grad_weight (Cout,Cin,3,3,3) is ACCUMULATED into (zero it for a plain gradient).
 * workspace: device scratch of mvs_conv3d_wgrad_workspace_bytes(...) bytes for the per-workgroup
 * partial sums (reduced by a second kernel); with NULL / too few bytes the partials meet in
 * grad_weight by atomic adds instead -- same result up to summation order, several times
 * slower for the small layers.  The last bits are not deterministic (as with the reference's cuDNN).
 * Transposed layer (stride 2, weight (Cin_t,Cout_t,3,3,3)): call with in := its grad_out
 * (the fine grid, Cout_t channels), grad_out := its input (Cin_t channels), stride 2; the
 * result is already in the transposed layer's weight layout.
 * Cin in {8,16,32,64}, Cout in {1,8,16,32,64}, stride in {1,2}. */
int mvs_conv3d_wgrad_f32(const float *in, const float *grad_out, int B, int Cin, int Cout, int D,
                         int H, int W, int stride, float *grad_weight, void *workspace,
                         size_t workspace_bytes, void *stream);
/* The same for the conv0-class layers (Cout = 8, stride 1, Cin in {8, 16, 32}) whose input is the 8-channel-blocked
 * volume [B,D,H,Cin/8,W,8] (MVS_LAYOUT_C8) that mvs_conv3d_c8_bf16x6_f32 reads: the training step runs conv0 on the
 * split-operand bf16 kernel and takes its weight gradient from the same tensor. */
int mvs_conv3d_wgrad_c8_f32(const float *in_c8, const float *grad_out, int B, int Cin, int D, int H, int W,
                            float *grad_weight, void *workspace, size_t workspace_bytes, void *stream);
/* conv0's weight gradient (Cin = 32, Cout = 8, 3x3x3, stride 1) on the 16-bit matrix pipe with two-piece fp16 operands
 * (mvs_amd/csrc/conv3d_wgrad_f16.hip; MVSNet/models/mvsnet.py:52 under autograd, train.py:222-226): the same sum as
 * mvs_conv3d_wgrad_c8_f32, products within 2^-22 relative of the fp32 products for operands within 2^-18 of their tensor's largest
 * magnitude.  in_absmax / grad_absmax: the MVS_ABSMAX_WORDS-word absmax blocks of in_c8 and grad_out (the sweep kernel's and
 * mvs_absmax_f32's outputs).  grad_weight (8,32,3,3,3) is ACCUMULATED into.  workspace: mvs_conv3d_wgrad_c8_f16_workspace_bytes.
 * mvs_conv3d_wgrad_c8_f16_supported: 1 if the shape has this kernel (Cin = 32, volume below 4 GB). */
int mvs_conv3d_wgrad_c8_f16_supported(int B, int Cin, int D, int H, int W);
size_t mvs_conv3d_wgrad_c8_f16_workspace_bytes(int B, int Cin, int D, int H, int W);
int mvs_conv3d_wgrad_c8_f16_f32(const float *in_c8, const unsigned *in_absmax, const float *grad_out, const unsigned *grad_absmax,
                                int B, int Cin, int D, int H, int W, float *grad_weight, void *workspace, size_t workspace_bytes,
                                void *stream);
size_t mvs_conv3d_wgrad_workspace_bytes(int B, int Cin, int Cout, int D, int H, int W, int stride);
int mvs_conv3d_wgrad_supported(int Cin, int Cout, int stride);

/* Training-mode BatchNorm fused with the ReLU and skip add that follow it in the reference's
 * blocks (module.py:26-33 ConvBnReLU3D = relu(bn(conv(x))); mvsnet.py:89-91 skip + relu(bn(deconv)));
 * channels-last rows x [N][C], N = B*D*H*W, C in {8,16,32,64}:
 *   y = relu((x - mean) * invstd * weight + bias) [+ skip]     (relu != 0; skip may be NULL)
 * mean / biased variance over the N rows (nn.BatchNorm3d in training mode); running_mean /
 * running_var (momentum, unbiased variance) and num_batches_tracked are updated in place when
 * not NULL; save_mean / save_invstd [C] are outputs for the backward.
 * The backward takes grad_y (gradient of y; the skip's gradient is grad_y itself) and returns
 * grad_x, grad_weight, grad_bias; the ReLU mask is recomputed from x.
 * workspace: mvs_bn_train_workspace_bytes(C) bytes of device scratch (fp64 partial sums). */
size_t mvs_bn_train_workspace_bytes(int C);
int mvs_bn_train_fwd_f32(const float *x, const float *weight, const float *bias, const float *skip,
                         int64_t N, int C, float eps, float momentum, int relu, float *running_mean,
                         float *running_var, long long *num_batches_tracked, float *save_mean,
                         float *save_invstd, float *y, void *workspace, size_t workspace_bytes,
                         void *stream);
int mvs_bn_train_bwd_f32(const float *grad_y, const float *x, const float *weight, const float *bias,
                         const float *save_mean, const float *save_invstd, int64_t N, int C, int relu,
                         float *grad_x, float *grad_weight, float *grad_bias, void *workspace,
                         size_t workspace_bytes, void *stream);
/* The same with GROUPS: x is G consecutive blocks of N rows, each normalised with its OWN batch statistics (save_mean /
 * save_invstd: [G][C]) -- the reference calls FeatureNet once per view (mvsnet.py:146 `[self.feature(img) for img in imgs]`),
 * so a batch of views [V*B, h, w, C] in one launch is V groups.  The running statistics take the groups in order, one
 * momentum update each, and num_batches_tracked grows by G: exactly what G separate calls leave behind.  grad_weight /
 * grad_bias are the sums over the groups.  1 <= G <= 16; the workspace is that of mvs_bn_train_workspace_bytes. */
int mvs_bn_train_fwd_groups_f32(const float *x, const float *weight, const float *bias, const float *skip, int G,
                                int64_t N, int C, float eps, float momentum, int relu, float *running_mean,
                                float *running_var, long long *num_batches_tracked, float *save_mean,
                                float *save_invstd, float *y, void *workspace, size_t workspace_bytes, void *stream);
int mvs_bn_train_bwd_groups_f32(const float *grad_y, const float *x, const float *weight, const float *bias,
                                const float *save_mean, const float *save_invstd, int G, int64_t N, int C, int relu,
                                float *grad_x, float *grad_weight, float *grad_bias, void *workspace,
                                size_t workspace_bytes, void *stream);

/* Per-pixel depth hypotheses of a cascade stage after the first (CasMVSNet/models/cas_mvsnet.py:129-152
 * + get_cur_depth_range_samples, module.py:485-502): prev_depth [B,hp,wp] (the previous stage's depth
 * map) is resized to [H,W] (bilinear, align_corners=False), each pixel gets D samples from
 * cur - half_range to cur + half_range (half_range = ndepth / 2 * interval), and that volume is resized
 * to [D,Hs,Ws] (trilinear, align_corners=False) -- evaluated directly at the output resolution.
 * out [B,D,Hs,Ws]. */
int mvs_cas_depth_hypotheses_f32(const float *prev_depth, int B, int hp, int wp, int H, int W, int Hs,
                                 int Ws, int D, float half_range, float *out, void *stream);

/* Sum over the image of |depth step| that moves a reference pixel's projection into the first source
 * view by `pixel_interval` along its epipolar line (CVP-MVSNet/models/modules.py:147-219, calDepthHypo
 * in test mode; the level's hypothesis interval is this sum / (H W)).  depth [H,W] float32;
 * mats: 59 doubles on the device = inverse(K_ref) (9), inverse(E_ref) (16), K_src (9), E_src (16),
 * K_ref R_ref inverse(K_src R_src) (9), row-major; sum_abs: one double on the device (set here). */
int mvs_cvp_interval_sum_f64(const float *depth, const double *mats, int H, int W, double pixel_interval,
                             double *sum_abs, void *stream);

/* Glue between the levels of the CVP-MVSNet pyramid, CVP-MVSNet/models/net.py:45,171
 * and modules.py:149-152,205-219:
 *   mvs_downsample_bilinear_half_f32  F.interpolate(x, scale_factor=0.5, mode='bilinear'): in [planes,H,W] ->
 *       out [planes,H/2,W/2], bit-identical to ATen's CPU kernel (the 2x2 block, weights 0.25, its summation order);
 *   mvs_upsample_bicubic2x_f32        F.interpolate(x, scale_factor=2, mode='bicubic'): in [planes,H,W] ->
 *       out [planes,2H,2W] (A = -0.75, border indices clamped; within 2 ulp of ATen's CPU kernel);
 *   mvs_cvp_hypothesis_mats_f64       the 59 doubles mvs_cvp_interval_sum_f64 takes, from the float32 camera
 *       matrices K_ref, K_src [3,3], E_ref, E_src [4,4] of one batch item (fp64 inverses and products on the device);
 *   mvs_cvp_hypotheses_f32            out [2d,H,W] = depth_up + (k - d) * float(sum_abs / (H W)), k = 0..2d-1. */
int mvs_downsample_bilinear_half_f32(const float *in, int64_t planes, int H, int W, float *out, void *stream);
int mvs_upsample_bicubic2x_f32(const float *in, int64_t planes, int H, int W, float *out, void *stream);
int mvs_cvp_hypothesis_mats_f64(const float *K_ref, const float *K_src, const float *E_ref, const float *E_src,
                                double *mats, void *stream);
int mvs_cvp_hypotheses_f32(const float *depth_up, const double *sum_abs, int H, int W, int d, float *out,
                           void *stream);

/* Geometric-consistency check of the depth filter that follows the path (SURVEY.md 8f rank 2;
 * MVSNet/eval.py:136-214 reproject_with_depth + check_geometric_consistency, sums of eval.py:239-262):
 * for one reference depth map [H,W] and S source depth maps [S,H,W] (all at the same resolution),
 * per source view: project the reference pixels with their depth into the source view, sample the
 * source depth there (cv2.remap INTER_LINEAR semantics), project back, and accept where the pixel
 * moved < 1 px and the depth changed < 1 %.
 * mats: [18 + 50 S] floats = inverse(K_ref), K_ref, then per view K_src, inverse(K_src),
 * E_src @ inverse(E_ref) (4x4), E_ref @ inverse(E_src) (4x4), row-major, computed by the caller as the
 * reference does (float32 numpy).  Outputs: mask [S,H,W] u8, depth_reprojected [S,H,W] (0 outside the
 * mask), xy_src [S,2,H,W] (all three optional: NULL), geo_mask_sum [H,W] int32, depth_averaged [H,W]
 * float64 = (sum of reprojected depths + reference depth) / (geo_mask_sum + 1). */
int mvs_geo_consistency_f32(const float *depth_ref, const float *depth_src, const float *mats, int S, int H,
                            int W, unsigned char *mask, float *depth_reprojected, float *xy_src,
                            int *geo_mask_sum, double *depth_averaged, void *stream);

/* Depth-map fusion (fusibile/fusibile.cu:138-277, launched per reference camera as
 * fusibile.cu:425-430 does).  normals_depths [N,H,W,4] = (nx, ny, nz, depth) per view; colors [N,H,W,4] or
 * NULL; cams [N,28] = per view P (3x4 row major), inverse(P[:, :3]) (3x3), P[:, 3], camera centre, f.
 * For reference view `ref`: out_points / out_normals (/ out_colors) [H,W,4]; a pixel without a fused point
 * gets zeros (the host keeps points whose three coordinates are non-zero, fusibile.cu:309).
 * Parity of this entry point is unpinned (CUDA texture filtering restated from its documentation): see
 * the header of csrc/fusibile.hip. */
int mvs_fusibile_fuse_f32(const float *normals_depths, const float *colors, const float *cams, int N, int H,
                          int W, int ref, float disp_thresh, float normal_thresh, int num_consistent,
                          float *out_points, float *out_normals, float *out_colors, void *stream);

/* ---- FeatureNet layers -- mvsnet.py:8-45 ("next" row 1) ---- */
/* One 2D convolution of FeatureNet on the fp32 matrix cores: k x k (3 stride 1, or 5
 * stride 2), pad k/2, no conv bias, then y = acc*scale[co] + shift[co] (BatchNorm(eval)
 * folded; for the last layer scale = NULL and shift = the conv bias) and the activation
 * `relu`: 0 none, 1 ReLU, 2 LeakyReLU(0.1) (CVP-MVSNet/models/modules.py:22-26).
 * layout_flags: bit 0 -- in is the reference's planar [B,3,H,W] image (3-channel layer only) instead
 * of channels-last [B,H,W,Cin]; bit 1 -- out is 4-channel blocked [B,Cout/4,Ho,Wo,4] (MVS_LAYOUT_C4,
 * what the persistent sweep kernel copies fastest; layers of the persistent kernel only) instead of
 * channels-last [B,Ho,Wo,Cout].  Supported (Cin,Cout,k,
 * stride): (3,8,3,1) (8,8,3,1) (8,16,5,2) (16,16,3,1) (16,32,5,2) (32,32,3,1); the
 * CasMVSNet FPN heads (32,32,1,1) (16,32,1,1) (8,32,1,1) (32,16,3,1) (32,8,3,1); the
 * CVP-MVSNet pyramid (3,64,3,1) (64,64,3,1) (64,32,3,1) (32,16,3,1).
 * coarse (or NULL): [B,Ho/2,Wo/2,Cout] channels-last, added to the result through a nearest-
 * neighbour x2 upsample -- the FPN top-down step `F.interpolate(intra_feat, scale_factor=2,
 * mode="nearest") + inner(conv)` of CasMVSNet/models/module.py:392,396 fused into the lateral
 * 1x1 convolution; stride-1 layers of the persistent kernel with even H, W only. */
int mvs_conv2d_f32(const float *in, const float *packed_weight, const float *scale,
                   const float *shift, const float *coarse, int relu, int B, int Cin, int Cout, int H,
                   int W, int ksize, int stride, int layout_flags, float *out, void *stream);
/* The same, and the largest magnitude of `out` max-ed INTO the absmax block out_absmax in the kernels' epilogues (the caller
 * clears it; NULL = none): for a layer whose output feeds a two-piece fp16 layer (mvs_conv_split_f16_f32). */
int mvs_conv2d_absmax_f32(const float *in, const float *packed_weight, const float *scale, const float *shift,
                          const float *coarse, int relu, int B, int Cin, int Cout, int H, int W, int ksize, int stride,
                          int layout_flags, float *out, void *out_absmax, void *stream);
/* FeatureNet's first two layers in one kernel: conv0 (3 -> 8, 3x3) + BN + ReLU + conv1 (8 -> 8, 3x3) + BN + ReLU
 * (MVSNet/models/mvsnet.py:11-12,33-34 `self.conv0 = ConvBnReLU(3, 8, 3, 1, 1)`, `self.conv1 = ConvBnReLU(8, 8, 3, 1, 1)`;
 * CasMVSNet/models/module.py:318-321 opens with the same pair).  conv0's 8-channel full-resolution output never
 * reaches HBM: conv0 runs on the vector ALU into LDS, conv1 on the BF16 matrix pipe with exactly split fp32 operands
 * (mvs_amd/csrc/feature_head.hip).  img: the reference's planar [N,3,H,W] batch; weight0: PyTorch layout (8,3,3,3);
 * packed1: mvs_feature_head_pack_weights_f32 of conv1's PyTorch-layout (8,8,3,3) weight
 * (mvs_feature_head_packed_bytes() bytes); scale / shift (or NULL): folded BatchNorm(eval) affines of the two
 * layers; out: [N,H,W,8] channels-last.  MVS_EUNSUPPORTED unless mvs_feature_head_supported(H, W) (W % 4 == 0). */
int mvs_feature_head_supported(int H, int W);
size_t mvs_feature_head_packed_bytes(void);
int mvs_feature_head_pack_weights_f32(const float *weight1, void *packed, void *stream);
int mvs_feature_head_f32(const float *img, const float *weight0, const float *scale0, const float *shift0,
                         const void *packed1, const float *scale1, const float *shift1, int N, int H, int W,
                         float *out, void *stream);
/* The same, and the largest magnitude of `out` max-ed INTO the absmax block out_absmax (the caller clears it; NULL = none):
 * the operand scale of the two-piece fp16 layer that reads `out` (mvs_conv_split_f16_f32). */
int mvs_feature_head_absmax_f32(const float *img, const float *w0, const float *scale0, const float *shift0,
                                const void *packed1, const float *scale1, const float *shift1, int N, int H, int W,
                                float *out, void *out_absmax, void *stream);
/* The full-resolution end of CasMVSNet's FPN in one kernel (CasMVSNet/models/module.py:396-398,
 * `intra_feat = F.interpolate(intra_feat, scale_factor=2, mode="nearest") + self.inner2(conv0)` followed by
 * `out = self.out3(intra_feat)`, final_chs = 32, base_channels = 8): the 32-channel full-resolution map stays in LDS
 * (mvs_amd/csrc/fpn_tail.hip).  fine: conv0's output [N,H,W,8]; coarse: the half-resolution map [N,H/2,W/2,32];
 * weight_inner / bias_inner: inner2 in PyTorch layout (32,8,1,1) / (32) (bias or NULL); packed_out:
 * mvs_fpn_tail_pack_weights_f32 of out3's PyTorch-layout (8,32,3,3) weight (mvs_fpn_tail_packed_bytes() bytes);
 * bias_out (8) or NULL; out [N,H,W,8]; all channels-last.  MVS_EUNSUPPORTED unless mvs_fpn_tail_supported(H, W)
 * (even H, W). */
int mvs_fpn_tail_supported(int H, int W);
size_t mvs_fpn_tail_packed_bytes(void);
int mvs_fpn_tail_pack_weights_f32(const float *weight_out, void *packed, void *stream);
int mvs_fpn_tail_f32(const float *fine, const float *coarse, const float *weight_inner, const float *bias_inner,
                     const void *packed_out, const float *bias_out, int N, int H, int W, float *out, void *stream);
int64_t mvs_conv2d_packed_weight_floats(int Cin, int Cout, int ksize, int stride);
/* weight: PyTorch layout (Cout,Cin,k,k) -> MFMA A-fragment order. */
int mvs_conv2d_pack_weights_f32(const float *weight, int Cin, int Cout, int ksize, int stride,
                                float *packed, void *stream);
int mvs_conv2d_supported(int Cin, int Cout, int ksize, int stride);

/* ---- K4+K5: softmax + expectation + confidence -- mvsnet.py:183-191 -- */
/* cost [B,D,H,W]; out_depth, out_conf [B,H,W]; out_prob [B,D,H,W] or NULL.
 * clamp_idx=1 is CasMVSNet's index clamp (cas_mvsnet.py:63). */
int mvs_softmax_regress_conf_f32(const float *cost, const float *depth_values, int depth_mode,
                                 int clamp_idx, int B, int D, int H, int W, float *out_depth,
                                 float *out_conf, float *out_prob, void *stream);
/* grad_cost [B,D,H,W] (overwritten) from grad_depth [B,H,W]; confidence
 * carries no gradient (mvsnet.py:187 no_grad). */
int mvs_softmax_regress_bwd_f32(const float *cost, const float *depth_values, int depth_mode,
                                const float *grad_depth, int B, int D, int H, int W,
                                float *grad_cost, void *stream);

/* ---- FeatureNet under autograd (training path, BASELINE configs[4]; MVSNet/models/mvsnet.py:8-45, train.py:222-226) ----
 * Weight gradient of a k x k, stride-s 2D convolution (3x3 stride 1 or 5x5 stride 2, up to 32 channels either side):
 * x [N,H,W,Cin] channels-last (planar: [N,Cin,H,W], the RGB layer, Cin <= 4), grad_out [N,Ho,Wo,Cout] -> grad_weight
 * (Cout,Cin,k,k).  The workspace holds one partial gradient per workgroup. */
size_t mvs_conv2d_wgrad_workspace_bytes(int N, int Cin, int Cout, int H, int W, int ksize, int stride);
int mvs_conv2d_wgrad_f32(const float *x, const float *grad_out, int N, int Cin, int Cout, int H, int W, int ksize,
                         int stride, int planar, float *grad_weight, void *workspace, size_t workspace_bytes,
                         void *stream);
/* classes [4,N,H,W,C] (class py*2+px = the pixels (2y+py, 2x+px)) -> out [N,2H,2W,C]: assembles the input gradient of a
 * stride-2 layer from its four output-parity classes, each a 3x3 stride-1 convolution of the output gradient. */
int mvs_interleave2x2_f32(const float *classes, int N, int H, int W, int C, float *out, void *stream);

/* rot_trans [V-1,B,12] (the rows of (src_proj @ inverse(ref_proj))[:3,:4], module.py:63-65) of every source view
 * from proj_matrices [B,V,4,4] on the device: float64 Gauss-Jordan + product, rounded once; no synchronisation, so it
 * can sit inside a captured HIP graph.  (The eval default evaluates these with the reference's own float32 LAPACK
 * call on the host, stream-ordered.) */
int mvs_rot_trans_f32(const float *proj_matrices, int B, int V, float *rot_trans, void *stream);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* MVS_HIP_H */
