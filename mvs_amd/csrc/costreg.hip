// Whole-network entry point of the 3D U-Net cost regularisation:
// MVSNet/models/mvsnet.py:48-93 (CostRegNet.forward) as ONE C call that sequences the layer
// kernels of mvs_conv3d_f32 over a caller-provided workspace.  Host code only.
#include "mvs_common.h"

namespace mvs {

// activation sizes (floats) of the four resolutions for batch B and `base` channels
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
    p.total = o + 64;   // + the word mvs_costreg_fwd2_f32 collects the input's largest magnitude in when the caller has none
    return true;
}

}  // namespace mvs

using namespace mvs;

extern "C" size_t mvs_costreg_workspace_bytes(int B, int base, int D, int H, int W) {
    CostRegPlan p;
    return costreg_plan(B, base, D, H, W, p) ? p.total * sizeof(float) : 0;
}

extern "C" int mvs_costreg_fwd_f32(const float *in, int in_layout, const mvs_conv_layer *layers, int B,
                                   int Cin, int base, int D, int H, int W, int impl, void *workspace,
                                   size_t workspace_bytes, float *out_cost, void *stream) {
    return mvs_costreg_fwd2_f32(in, in_layout, layers, B, Cin, base, D, H, W, impl, workspace, workspace_bytes, nullptr,
                                nullptr, out_cost, stream);
}

extern "C" int mvs_costreg_fwd2_f32(const float *in, int in_layout, const mvs_conv_layer *layers, int B,
                                    int Cin, int base, int D, int H, int W, int impl, void *workspace,
                                    size_t workspace_bytes, const void *conv0_f16x3, const void *in_absmax,
                                    float *out_cost, void *stream) {
    if (!in || !layers || !out_cost || (in_layout != MVS_LAYOUT_NHWC && in_layout != MVS_LAYOUT_C8)) {
        set_error("mvs_costreg_fwd_f32: invalid argument");
        return MVS_EINVAL;
    }
    CostRegPlan p;
    if (!costreg_plan(B, base, D, H, W, p)) {
        set_error("mvs_costreg_fwd_f32: D, H, W = %d, %d, %d must be positive multiples of 8 (three "
                  "stride-2 levels whose transposed layers return exactly 2x)", D, H, W);
        return MVS_EINVAL;
    }
    if (!workspace || workspace_bytes < p.total * sizeof(float)) {
        set_error("mvs_costreg_fwd_f32: workspace of %zu bytes, need %zu", workspace_bytes,
                  p.total * sizeof(float));
        return MVS_EWORKSPACE;
    }
    for (int i = 0; i < 11; ++i)
        if (!layers[i].weight) {
            set_error("mvs_costreg_fwd_f32: layer %d has no weight", i);
            return MVS_EINVAL;
        }
    float *ws = static_cast<float *>(workspace);
    float *c0 = ws + p.off[0], *t1 = ws + p.off[1], *c2 = ws + p.off[2], *t3 = ws + p.off[3],
          *c4 = ws + p.off[4], *t5 = ws + p.off[5], *t6 = ws + p.off[6], *d7 = ws + p.off[7],
          *d9 = ws + p.off[8], *d11 = ws + p.off[9];
    const int b = base;
    struct Step {
        int layer; const float *src; const float *skip; float *dst; int cin, cout, lvl, stride, transposed, relu, layout;
    };
    // mvsnet.py:83-93: conv0 .. conv6, then x = conv4 + conv7(x); x = conv2 + conv9(x);
    // x = conv0 + conv11(x); prob (bias, no BN, no ReLU)
    const Step steps[11] = {
        {0, in, nullptr, c0, Cin, b, 0, 1, 0, 1, in_layout},
        {1, c0, nullptr, t1, b, 2 * b, 0, 2, 0, 1, MVS_LAYOUT_NHWC},
        {2, t1, nullptr, c2, 2 * b, 2 * b, 1, 1, 0, 1, MVS_LAYOUT_NHWC},
        {3, c2, nullptr, t3, 2 * b, 4 * b, 1, 2, 0, 1, MVS_LAYOUT_NHWC},
        {4, t3, nullptr, c4, 4 * b, 4 * b, 2, 1, 0, 1, MVS_LAYOUT_NHWC},
        {5, c4, nullptr, t5, 4 * b, 8 * b, 2, 2, 0, 1, MVS_LAYOUT_NHWC},
        {6, t5, nullptr, t6, 8 * b, 8 * b, 3, 1, 0, 1, MVS_LAYOUT_NHWC},
        {7, t6, c4, d7, 8 * b, 4 * b, 3, 2, 1, 1, MVS_LAYOUT_NHWC},
        {8, d7, c2, d9, 4 * b, 2 * b, 2, 2, 1, 1, MVS_LAYOUT_NHWC},
        {9, d9, c0, d11, 2 * b, b, 1, 2, 1, 1, MVS_LAYOUT_NHWC},
        {10, d11, nullptr, out_cost, b, 1, 0, 1, 0, 0, MVS_LAYOUT_NHWC},
    };
    for (const Step &s : steps) {
        const mvs_conv_layer &L = layers[s.layer];
        if (s.layer == 0 && conv0_f16x3 && in_layout == MVS_LAYOUT_C8 && b == 8 && impl != 1 &&
            mvs_conv3d_f16x3_packed_bytes(Cin) != 0) {
            // conv0 on the fp16 matrix pipe with two-piece operands (conv_f16x3.hip); the operand scale comes from the
            // producer of the volume, or from one more pass over it
            const void *mx = in_absmax;
            if (!mx) {
                void *w = ws + p.total - 64;
                const int rc = mvs_absmax_f32(in, (int64_t)B * D * H * W * Cin, w, stream);
                if (rc != MVS_OK) return rc;
                mx = w;
            }
            const int rc = mvs_conv3d_c8_f16x3_f32(s.src, mx, conv0_f16x3, L.scale, L.shift, nullptr, s.relu, B, Cin,
                                                   D, H, W, s.dst, stream);
            if (rc != MVS_OK) return rc;
            continue;
        }
        if (s.layer == 0 && L.packed_split && in_layout == MVS_LAYOUT_C8 && b == 8 && impl != 1 &&
            mvs_conv3d_bf16x6_packed_bytes(Cin) != 0) {
            // conv0 on the bf16 matrix pipe with exactly split fp32 operands (conv_bf16x6.hip)
            const int rc = mvs_conv3d_c8_bf16x6_f32(s.src, L.packed_split, L.scale, L.shift, nullptr, s.relu, B, Cin,
                                                    D, H, W, s.dst, stream);
            if (rc != MVS_OK) return rc;
            continue;
        }
        if (L.packed_split && s.transposed && s.stride == 2 && impl != 1 && mvs_deconv_split_supported(s.cin, s.cout)) {
            // conv7 / conv9 / conv11 (deconv_split.hip); s.lvl is the INPUT level
            const int rc = mvs_deconv_split_f32(s.src, L.packed_split, L.scale, L.shift, s.skip, s.relu, B, s.cin, s.cout,
                                                D >> s.lvl, H >> s.lvl, W >> s.lvl, s.dst, stream);
            if (rc != MVS_OK) return rc;
            continue;
        }
        if (L.packed_split && !s.transposed && s.layout == MVS_LAYOUT_NHWC && impl != 1 &&
            mvs_conv_split_supported(3, s.cin, s.cout, s.stride)) {
            // conv1 .. conv6: the split-operand kernel (conv_split.hip)
            const int rc = mvs_conv_split_f32(s.src, L.packed_split, L.scale, L.shift, s.skip, s.relu, 3, s.stride, B, s.cin, s.cout,
                                              D >> s.lvl, H >> s.lvl, W >> s.lvl, 0, s.dst, stream);
            if (rc != MVS_OK) return rc;
            continue;
        }
        const int rc = mvs_conv3d_f32(s.src, L.weight, L.packed, L.scale, L.shift, s.skip, s.relu,
                                      s.transposed, B, s.cin, s.cout, D >> s.lvl, H >> s.lvl, W >> s.lvl,
                                      s.stride, s.layout, impl, s.dst, stream);
        if (rc != MVS_OK) return rc;   // the layer call has set the error text
    }
    return MVS_OK;
}
