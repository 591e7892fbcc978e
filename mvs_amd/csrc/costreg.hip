// Whole-network entry point of the 3D U-Net cost regularisation:
// MVSNet/models/mvsnet.py:48-93 (CostRegNet.forward) as ONE C call that sequences the layer
// kernels of mvs_conv3d_f32 over a caller-provided workspace.  Host code only.
#include "mvs_common.h"

namespace mvs {

// activation sizes (floats) of the four resolutions for batch B and `base` channels
constexpr int kFlagWords = 64;

struct CostRegPlan {
    int64_t n[4];          // voxels per batch item at levels 0..3
    size_t off[10];        // workspace offsets (floats): c0 t1 c2 t3 c4 t5 t6 d7 d9 d11
    size_t total;
};

static bool costreg_plan(int B, int base, int D, int H, int W, CostRegPlan &p) {
    if (B <= 0 || base <= 0 || D <= 0 || H <= 0 || W <= 0 || (D % 8) || (H % 8) || (W % 8)) return false;
    for (int l = 0; l < 4; ++l) p.n[l] = (int64_t)B * (D >> l) * (H >> l) * (W >> l);
    const int64_t sz[10] = {p.n[0] * base,     p.n[1] * 2 * base, p.n[1] * 2 * base, p.n[2] * 4 * base,
                            p.n[2] * 4 * base, p.n[3] * 8 * base, p.n[3] * 8 * base, p.n[2] * 4 * base,
                            p.n[1] * 2 * base, p.n[0] * base};
    size_t o = 0;
    for (int i = 0; i < 10; ++i) {
        p.off[i] = o;
        o += ((size_t)sz[i] + 63) & ~(size_t)63;   // 256-byte aligned slices
    }
    // + the absmax blocks of mvs_costreg_fwd2_f32: the input's (when the caller has none), nine activations'; + the flag words of
    // mvs_costreg_fwd3_f32 (the fused tail kernel's "I declined")
    p.total = o + 10 * kAbsmaxWords + kFlagWords;
    return true;
}

}  // namespace mvs

using namespace mvs;

extern "C" size_t mvs_costreg_workspace_bytes(int B, int base, int D, int H, int W) {
    CostRegPlan p;
    return costreg_plan(B, base, D, H, W, p) ? p.total * sizeof(float) : 0;
}

// conv11 + prob: the fused kernel (tail_fused.hip), then the two unfused layers enqueued behind it with the "run only if" word
// (mvs_common.h: conv_run_flag) = the flag the fused kernel's range guard sets when it declines -- they return at once otherwise.
// No host synchronisation.  *flag must be 0 on entry (a device word; the caller clears it).
extern "C" int mvs_costreg_tail_guarded_f16_f32(const float *in, const void *in_absmax, const float *skip, const void *skip_absmax,
                                                const void *packed_tail, const mvs_conv_layer *conv11, const void *conv11_f16,
                                                const mvs_conv_layer *prob, int B, int Di, int Hi, int Wi, float *d11_scratch,
                                                float *out_cost, void *flag, void *stream) {
    if (!conv11 || !conv11_f16 || !prob || !prob->weight || !d11_scratch || !flag) {
        set_error("mvs_costreg_tail_guarded_f16_f32: needs conv11's layer and two-piece pack, prob's layer, a [B, 2Di, 2Hi, 2Wi, 8] scratch "
                  "volume for the unfused path and a zeroed flag word");
        return MVS_EINVAL;
    }
    // the unfused `prob` behind the flag must be the Cout = 1 MFMA kernel (the only one that reads the word): it needs the packed
    // weights and nine output-resolution planes inside 32-bit offsets (capi.hip: window_ok) -- checked BEFORE anything is enqueued
    if (!prob->packed || (int64_t)9 * (2 * Hi) * (2 * Wi) * 8 * 4 >= 0xffffff00LL) {
        set_error("mvs_costreg_tail_guarded_f16_f32: the guarded fall-back needs prob's packed weights and an output plane of fewer than "
                  "%lld voxels (2Hi x 2Wi = %d x %d)", (long long)(0xffffff00LL / (9 * 8 * 4)), 2 * Hi, 2 * Wi);
        return MVS_EUNSUPPORTED;
    }
    int rc = mvs_costreg_tail_f16_f32(in, in_absmax, skip, skip_absmax, packed_tail, conv11->scale, conv11->shift, prob->weight,
                                      prob->scale, prob->shift, B, Di, Hi, Wi, out_cost, flag, stream);
    if (rc != MVS_OK) return rc;
    struct FlagScope {
        explicit FlagScope(const void *f) { conv_run_flag() = static_cast<const unsigned *>(f); }
        ~FlagScope() { conv_run_flag() = nullptr; }
    } scope(flag);
    rc = mvs_deconv_split_f16_f32(in, in_absmax, conv11_f16, conv11->scale, conv11->shift, skip, 1, B, 16, 8, Di, Hi, Wi, d11_scratch,
                                  nullptr, stream);
    if (rc != MVS_OK) return rc;
    return mvs_conv3d_absmax_f32(d11_scratch, prob->weight, prob->packed, prob->scale, prob->shift, nullptr, 0, 0, B, 8, 1, 2 * Di,
                                 2 * Hi, 2 * Wi, 1, MVS_LAYOUT_NHWC, 0, out_cost, nullptr, stream);
}

// layers: the eleven layers; f16 = their two-piece fp16 packs or NULL (the entry of before those existed)
// hand / redo: non-NULL with in_layout = MVS_LAYOUT_C8PH -- the volume of a hand-over sweep (mvs_costvol_variance_fwd_ws3_f32): two
// fp16 pieces per value scaled by the bound in `hand` if *redo == 0, an fp32 MVS_LAYOUT_C8 volume if *redo == 1; in_absmax = the
// block the sweep collected the volume's true maximum in
static int costreg_impl(const float *in, int in_layout, const mvs_conv_layer *layers, const void *const *f16, const void *tail_pack, int B,
                        int Cin, int base, int D, int H, int W, int impl, void *workspace,
                        size_t workspace_bytes, const void *in_absmax, float *out_cost, void *stream, const char *who,
                        const void *hand = nullptr, const void *redo = nullptr) {
    const bool handed = in_layout == MVS_LAYOUT_C8PH;
    if (!in || !layers || !out_cost || (in_layout != MVS_LAYOUT_NHWC && in_layout != MVS_LAYOUT_C8 && !handed) ||
        (handed && (!hand || !redo || !in_absmax || !f16 || !f16[0] || impl == 1 || base != 8 || mvs_conv3d_f16x3_packed_bytes(Cin) == 0))) {
        set_error("%s: invalid argument%s", who, handed ? " (a handed-over volume needs the hand-over block, the redo word, the volume's absmax "
                                                          "block, conv0's two-piece pack, base 8, Cin in {8, 16, 32}, impl 0 or 2)" : "");
        return MVS_EINVAL;
    }
    if (handed) in_layout = MVS_LAYOUT_C8;      // what the volume is when it is not pieces; the layer table below is written for that
    CostRegPlan p;
    if (!costreg_plan(B, base, D, H, W, p)) {
        set_error("%s: D, H, W = %d, %d, %d must be positive multiples of 8 (three "
                  "stride-2 levels whose transposed layers return exactly 2x)", who, D, H, W);
        return MVS_EINVAL;
    }
    if (!workspace || workspace_bytes < p.total * sizeof(float)) {
        set_error("%s: workspace of %zu bytes, need %zu", who, workspace_bytes, p.total * sizeof(float));
        return MVS_EWORKSPACE;
    }
    for (int i = 0; i < 11; ++i)
        if (!layers[i].weight) {
            set_error("%s: layer %d has no weight", who, i);
            return MVS_EINVAL;
        }
    float *ws = static_cast<float *>(workspace);
    float *c0 = ws + p.off[0], *t1 = ws + p.off[1], *c2 = ws + p.off[2], *t3 = ws + p.off[3],
          *c4 = ws + p.off[4], *t5 = ws + p.off[5], *t6 = ws + p.off[6], *d7 = ws + p.off[7],
          *d9 = ws + p.off[8], *d11 = ws + p.off[9];
    // absmax blocks (mvs_common.h): [0] the input's when the caller has none, [1 + i] activation i's -- what a layer on the
    // two-piece fp16 kernels scales its input by, collected in the epilogue of the layer that wrote it
    unsigned *const amax = reinterpret_cast<unsigned *>(ws + p.total - 10 * kAbsmaxWords - kFlagWords);
    unsigned *const flags = amax + 10 * kAbsmaxWords;
    auto block = [&](int i) { return static_cast<void *>(amax + (size_t)i * kAbsmaxWords); };
    const bool two_piece = f16 && impl != 1;
    if (two_piece) {
        const int rc = launch_zero_words(amax, 10 * kAbsmaxWords + kFlagWords, as_stream(stream));      // (mvs_common.h: not a memset node)
        if (rc != MVS_OK) return rc;
    }
    const int b = base;
    struct Step {
        int layer; const float *src; const float *skip; float *dst; int cin, cout, lvl, stride, transposed, relu, layout;
        int src_act, dst_act;   // activation index of src / dst (-1: the network's input / none)
    };
    // mvsnet.py:83-93: conv0 .. conv6, then x = conv4 + conv7(x); x = conv2 + conv9(x);
    // x = conv0 + conv11(x); prob (bias, no BN, no ReLU)
    const Step steps[11] = {
        {0, in, nullptr, c0, Cin, b, 0, 1, 0, 1, in_layout, -1, 0},
        {1, c0, nullptr, t1, b, 2 * b, 0, 2, 0, 1, MVS_LAYOUT_NHWC, 0, 1},
        {2, t1, nullptr, c2, 2 * b, 2 * b, 1, 1, 0, 1, MVS_LAYOUT_NHWC, 1, 2},
        {3, c2, nullptr, t3, 2 * b, 4 * b, 1, 2, 0, 1, MVS_LAYOUT_NHWC, 2, 3},
        {4, t3, nullptr, c4, 4 * b, 4 * b, 2, 1, 0, 1, MVS_LAYOUT_NHWC, 3, 4},
        {5, c4, nullptr, t5, 4 * b, 8 * b, 2, 2, 0, 1, MVS_LAYOUT_NHWC, 4, 5},
        {6, t5, nullptr, t6, 8 * b, 8 * b, 3, 1, 0, 1, MVS_LAYOUT_NHWC, 5, 6},
        {7, t6, c4, d7, 8 * b, 4 * b, 3, 2, 1, 1, MVS_LAYOUT_NHWC, 6, 7},
        {8, d7, c2, d9, 4 * b, 2 * b, 2, 2, 1, 1, MVS_LAYOUT_NHWC, 7, 8},
        {9, d9, c0, d11, 2 * b, b, 1, 2, 1, 1, MVS_LAYOUT_NHWC, 8, -1},
        {10, d11, nullptr, out_cost, b, 1, 0, 1, 0, 0, MVS_LAYOUT_NHWC, -1, -1},
    };
    // does the layer that reads activation i run on a two-piece kernel (its producer then collects the absmax block)?
    auto two_piece_layer = [&](const Step &s) {
        if (!two_piece || !f16[s.layer]) return false;
        if (s.layer == 0) return in_layout == MVS_LAYOUT_C8 && b == 8 && mvs_conv3d_f16x3_packed_bytes(Cin) != 0;
        if (s.transposed) return s.stride == 2 && mvs_deconv_split_supported(s.cin, s.cout) != 0;
        return s.layout == MVS_LAYOUT_NHWC && mvs_conv_split_supported(3, s.cin, s.cout, s.stride) != 0;
    };
    bool wanted[10] = {false};     // activation i's absmax block has a reader
    for (const Step &s : steps)
        if (s.src_act >= 0 && two_piece_layer(s)) wanted[s.src_act] = true;
    // conv11 -> prob as ONE kernel (tail_fused.hip) when both run on the two-piece path of a base-8 net: it needs the blocks of
    // conv11's input (activation 8) and of the skip volume (activation 0), both collected anyway.  The unfused layers stay enqueued
    // behind it with the "run only if" word: they return at once unless the fused kernel's range guard declined.
    const bool fuse_tail = tail_pack && b == 8 && two_piece_layer(steps[9]) && wanted[0] && wanted[8] &&
                           mvs_costreg_tail_supported(B, D >> 1, H >> 1, W >> 1) && layers[10].packed &&
                           (int64_t)9 * H * W * 8 * 4 < 0xffffff00LL;   // what mvs_costreg_tail_guarded_f16_f32's fall-back needs
    for (const Step &s : steps) {
        const mvs_conv_layer &L = layers[s.layer];
        const int d = D >> s.lvl, h = H >> s.lvl, w = W >> s.lvl;
        if (fuse_tail && s.layer == 9)      // conv11 + prob: the fused kernel, the two layers behind it under its flag
            return mvs_costreg_tail_guarded_f16_f32(s.src, block(1 + 8), s.skip, block(1 + 0), tail_pack, &L, f16[9], &layers[10],
                                                    B, d, h, w, s.dst, out_cost, flags, stream);
        void *const out_mx = (s.dst_act >= 0 && wanted[s.dst_act]) ? block(1 + s.dst_act) : nullptr;
        bool collected = false;     // did the kernel collect out_mx in its epilogue?
        int rc;
        if (two_piece_layer(s)) {
            const void *mx = s.src_act >= 0 ? block(1 + s.src_act) : in_absmax;
            if (!mx) {    // the network's input without a block from its producer: one more pass over it
                rc = mvs_absmax_f32(in, (int64_t)B * D * H * W * Cin, block(0), stream);
                if (rc != MVS_OK) return rc;
                mx = block(0);
            }
            if (s.layer == 0 && handed) {
                // conv0 on the pieces, then conv0 on the fp32 volume under *redo: exactly one of the two writes c0 and its absmax block
                rc = mvs_conv3d_c8_handed_f16x3_f32(s.src, hand, redo, mx, f16[0], L.scale, L.shift, nullptr, s.relu, B, Cin, D, H, W, s.dst,
                                                    out_mx, stream);
            } else if (s.layer == 0)     // conv0 on the fp16 matrix pipe with two-piece operands (conv_f16x3.hip)
                rc = mvs_conv3d_c8_f16x3_f32(s.src, mx, f16[0], L.scale, L.shift, nullptr, s.relu, B, Cin, D, H, W, s.dst, out_mx, stream);
            else if (s.transposed)       // conv7 / conv9 / conv11 (deconv_split.hip); s.lvl is the INPUT level
                rc = mvs_deconv_split_f16_f32(s.src, mx, f16[s.layer], L.scale, L.shift, s.skip, s.relu, B, s.cin, s.cout, d, h, w,
                                              s.dst, out_mx, stream);
            else                         // conv1 .. conv6 (conv_split.hip)
                rc = mvs_conv_split_f16_f32(s.src, mx, f16[s.layer], L.scale, L.shift, s.skip, s.relu, 3, s.stride, B, s.cin, s.cout,
                                            d, h, w, 0, s.dst, out_mx, stream);
            collected = true;
        } else if (s.layer == 0 && L.packed_split && in_layout == MVS_LAYOUT_C8 && b == 8 && impl != 1 &&
                   mvs_conv3d_bf16x6_packed_bytes(Cin) != 0) {
            // conv0 on the bf16 matrix pipe with exactly split fp32 operands (conv_bf16x6.hip)
            rc = mvs_conv3d_c8_bf16x6_f32(s.src, L.packed_split, L.scale, L.shift, nullptr, s.relu, B, Cin, D, H, W, s.dst, stream);
        } else if (L.packed_split && s.transposed && s.stride == 2 && impl != 1 && mvs_deconv_split_supported(s.cin, s.cout)) {
            rc = mvs_deconv_split_f32(s.src, L.packed_split, L.scale, L.shift, s.skip, s.relu, B, s.cin, s.cout, d, h, w, s.dst, stream);
        } else if (L.packed_split && !s.transposed && s.layout == MVS_LAYOUT_NHWC && impl != 1 &&
                   mvs_conv_split_supported(3, s.cin, s.cout, s.stride)) {
            rc = mvs_conv_split_f32(s.src, L.packed_split, L.scale, L.shift, s.skip, s.relu, 3, s.stride, B, s.cin, s.cout,
                                    d, h, w, 0, s.dst, stream);
        } else {      // (conv3, conv5 on the fp32 MFMA kernels: they collect out_mx in their epilogue too)
            rc = mvs_conv3d_absmax_f32(s.src, L.weight, L.packed, L.scale, L.shift, s.skip, s.relu,
                                       s.transposed, B, s.cin, s.cout, d, h, w, s.stride, s.layout, impl, s.dst, out_mx, stream);
            collected = true;
        }
        if (rc != MVS_OK) return rc;   // the layer call has set the error text
        if (out_mx && !collected) {    // a layer on the other kernels (conv3, conv5) in front of a two-piece one: one pass over its output
            const int so = s.transposed ? 1 : -1, q = s.stride == 2 ? 1 : 0;
            const int lo = s.lvl + (so < 0 ? q : -q);
            rc = mvs_absmax_f32(s.dst, (int64_t)B * (D >> lo) * (H >> lo) * (W >> lo) * s.cout, out_mx, stream);
            if (rc != MVS_OK) return rc;
        }
    }
    return MVS_OK;
}

extern "C" int mvs_costreg_fwd_f32(const float *in, int in_layout, const mvs_conv_layer *layers, int B,
                                   int Cin, int base, int D, int H, int W, int impl, void *workspace,
                                   size_t workspace_bytes, float *out_cost, void *stream) {
    return costreg_impl(in, in_layout, layers, nullptr, nullptr, B, Cin, base, D, H, W, impl, workspace, workspace_bytes, nullptr, out_cost,
                        stream, "mvs_costreg_fwd_f32");
}

extern "C" int mvs_costreg_fwd2_f32(const float *in, int in_layout, const mvs_conv_layer *layers, const void *const *packed_f16,
                                    int B, int Cin, int base, int D, int H, int W, int impl, void *workspace,
                                    size_t workspace_bytes, const void *in_absmax, float *out_cost, void *stream) {
    if (!packed_f16) {
        set_error("mvs_costreg_fwd2_f32: packed_f16 = the eleven layers' two-piece packs (NULL entries allowed)");
        return MVS_EINVAL;
    }
    return costreg_impl(in, in_layout, layers, packed_f16, nullptr, B, Cin, base, D, H, W, impl, workspace, workspace_bytes, in_absmax,
                        out_cost, stream, "mvs_costreg_fwd2_f32");
}

// fwd2 + packed_tail (mvs_costreg_tail_pack_weights_f32 of conv11's weight, or NULL = fwd2): conv11 and prob as one kernel
extern "C" int mvs_costreg_fwd3_f32(const float *in, int in_layout, const mvs_conv_layer *layers, const void *const *packed_f16,
                                    const void *packed_tail, int B, int Cin, int base, int D, int H, int W, int impl, void *workspace,
                                    size_t workspace_bytes, const void *in_absmax, float *out_cost, void *stream) {
    if (!packed_f16) {
        set_error("mvs_costreg_fwd3_f32: packed_f16 = the eleven layers' two-piece packs (NULL entries allowed)");
        return MVS_EINVAL;
    }
    return costreg_impl(in, in_layout, layers, packed_f16, packed_tail, B, Cin, base, D, H, W, impl, workspace, workspace_bytes,
                        in_absmax, out_cost, stream, "mvs_costreg_fwd3_f32");
}

// fwd3 on the volume of a hand-over sweep (mvs_costvol_variance_fwd_ws3_f32): in = its out_volume, hand / redo / var_absmax = its blocks
extern "C" int mvs_costreg_fwd4_f32(const void *in_volume, const void *hand, const void *redo, const void *var_absmax,
                                    const mvs_conv_layer *layers, const void *const *packed_f16, const void *packed_tail, int B, int Cin,
                                    int base, int D, int H, int W, int impl, void *workspace, size_t workspace_bytes, float *out_cost,
                                    void *stream) {
    return costreg_impl(static_cast<const float *>(in_volume), MVS_LAYOUT_C8PH, layers, packed_f16, packed_tail, B, Cin, base, D, H, W, impl,
                        workspace, workspace_bytes, var_absmax, out_cost, stream, "mvs_costreg_fwd4_f32", hand, redo);
}
