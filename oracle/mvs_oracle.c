/*
 * oracle/mvs_oracle.c -- CPU restatement of the MVSNet cost-volume path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker.  The product path
 * (mvs_amd/ -> libmvs_hip.so) never links, imports or falls back to it.
 *
 * Parity status: PINNED.  Every function here is checked against golden
 * vectors produced by importing the reference's own Python
 * (/root/reference/MVSNet/models, CasMVSNet/models, CVP-MVSNet/models) in the
 * build container -- tests/golden/make_golden.py, tests/test_oracle_golden.py.
 *
 * Each function cites the reference lines it restates.  The arithmetic is
 * written out op by op in fp32 (compile with -ffp-contract=off) in the order
 * the reference's ATen composition evaluates it; OpenMP only parallelises
 * independent output elements, so results do not depend on thread count.
 *
 * All tensors are contiguous float32, NCHW / NCDHW.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_OK 0
#define ORC_EINVAL (-1)

/* ------------------------------------------------------------------ */
/* Plane-sweep coordinate: MVSNet/models/module.py:62-81 + the ATen
 * grid_sampler un-normalisation the call at module.py:83-84 runs with
 * (align_corners=False is the torch>=1.3 default; SURVEY.md 7 "Hard parts").
 * rt = rows of (src_proj @ inverse(ref_proj))[:3,:4]                   */
static inline void orc_sample_coord(const float *rt, float x, float y, float d,
                                    int H, int W, int align_corners,
                                    float *ix, float *iy)
{
    /* module.py:73  rot_xyz = rot @ (x, y, 1): ATen's sgemm accumulates
     * k = 0,1,2 with FMA -- r0*x, fma(r1,y,.), fma(r2,1,.) -- found bit-exact
     * against the golden vectors (tests/test_oracle_golden.py). */
    float rx = fmaf(rt[1], y, rt[0] * x) + rt[2];
    float ry = fmaf(rt[5], y, rt[4] * x) + rt[6];
    float rz = fmaf(rt[9], y, rt[8] * x) + rt[10];
    /* module.py:74-76  * depth, + trans */
    float X = rx * d + rt[3];
    float Y = ry * d + rt[7];
    float Z = rz * d + rt[11];
    /* module.py:77 */
    float px = X / Z;
    float py = Y / Z;
    /* module.py:78-79 */
    float gx = px / (float)((W - 1) / 2.0) - 1.0f;
    float gy = py / (float)((H - 1) / 2.0) - 1.0f;
    /* ATen grid_sampler unnormalize */
    if (align_corners) {
        *ix = (gx + 1.0f) * (float)((W - 1) / 2.0);
        *iy = (gy + 1.0f) * (float)((H - 1) / 2.0);
    } else {
        /* (g + 1) * (size / 2) - 0.5, contracted to one FMA by ATen's
         * vectorised CPU kernel (bit-exact against the golden vectors) */
        *ix = fmaf(gx + 1.0f, (float)(W / 2.0), -0.5f);
        *iy = fmaf(gy + 1.0f, (float)(H / 2.0), -0.5f);
    }
}

/* Bilinear sample with zeros padding of one channel plane (ATen
 * grid_sampler_2d, bilinear/zeros, as invoked at module.py:83-84). */
static inline float orc_bilinear(const float *plane, int H, int W, float ix, float iy)
{
    float x0f = floorf(ix), y0f = floorf(iy);
    float w = ix - x0f, e = 1.0f - w;
    float n = iy - y0f, s = 1.0f - n;
    float nw = s * e, ne = s * w, sw = n * e, se = n * w;
    int x0ok = (x0f >= 0.0f) && (x0f <= (float)(W - 1));
    int x1ok = (x0f >= -1.0f) && (x0f <= (float)(W - 2));
    int y0ok = (y0f >= 0.0f) && (y0f <= (float)(H - 1));
    int y1ok = (y0f >= -1.0f) && (y0f <= (float)(H - 2));
    int x0 = x0ok ? (int)x0f : 0, x1 = x1ok ? (int)x0f + 1 : 0;
    int y0 = y0ok ? (int)y0f : 0, y1 = y1ok ? (int)y0f + 1 : 0;
    float v00 = (x0ok && y0ok) ? plane[(size_t)y0 * W + x0] : 0.0f;
    float v01 = (x1ok && y0ok) ? plane[(size_t)y0 * W + x1] : 0.0f;
    float v10 = (x0ok && y1ok) ? plane[(size_t)y1 * W + x0] : 0.0f;
    float v11 = (x1ok && y1ok) ? plane[(size_t)y1 * W + x1] : 0.0f;
    /* nw*v00, then FMA in ne, sw, se order (ATen vectorised kernel) */
    return fmaf(v11, se, fmaf(v10, sw, fmaf(v01, ne, v00 * nw)));
}

static inline float orc_depth_at(const float *depth, int depth_mode, int b, int d,
                                 int y, int x, int D, int H, int W)
{
    /* depth_mode 0: [B,D] (MVSNet module.py:74); 1: [B,D,H,W] per-pixel
     * hypotheses (CasMVSNet/models/module.py:249,267; CVP modules.py:253) */
    if (depth_mode == 0) return depth[(size_t)b * D + d];
    return depth[(((size_t)b * D + d) * H + y) * W + x];
}

/* homo_warping: MVSNet/models/module.py:46-87.
 * src [B,C,H,W], rt [B,12], out [B,C,D,H,W] */
int orc_warp_f32(const float *src, const float *rt, const float *depth, int depth_mode,
                 int B, int C, int D, int H, int W, int align_corners, float *out)
{
    if (!src || !rt || !depth || !out || B <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0)
        return ORC_EINVAL;
    for (int b = 0; b < B; ++b) {
#pragma omp parallel for collapse(2) schedule(static)
        for (int d = 0; d < D; ++d)
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    float dv = orc_depth_at(depth, depth_mode, b, d, y, x, D, H, W);
                    float ix, iy;
                    orc_sample_coord(rt + (size_t)b * 12, (float)x, (float)y, dv, H, W,
                                     align_corners, &ix, &iy);
                    for (int c = 0; c < C; ++c) {
                        const float *pl = src + ((size_t)b * C + c) * H * W;
                        out[((((size_t)b * C + c) * D + d) * H + y) * W + x] =
                            orc_bilinear(pl, H, W, ix, iy);
                    }
                }
    }
    return ORC_OK;
}

/* Variance cost volume: MVSNet/models/mvsnet.py:152-170 (eval and train
 * branches compute the same values); CasMVSNet/models/cas_mvsnet.py:24-46.
 * alias_quirk=1 reproduces CVP-MVSNet/models/modules.py:228-229 and
 * net.py:129-130, where pow_ aliases volume_sum so S0 = Q0 = ref^2.
 * ref [B,C,H,W], srcs [V-1,B,C,H,W], rt [V-1,B,12], out [B,C,D,H,W] */
int orc_costvol_variance_f32(const float *ref, const float *srcs, const float *rt,
                             const float *depth, int depth_mode,
                             int B, int V, int C, int D, int H, int W,
                             int align_corners, int alias_quirk, float *out)
{
    if (!ref || !depth || !out || B <= 0 || V < 1 || C <= 0 || D <= 0 || H <= 0 || W <= 0)
        return ORC_EINVAL;
    if (V > 1 && (!srcs || !rt)) return ORC_EINVAL;
    const size_t plane = (size_t)H * W;
    const float fV = (float)V;
    for (int b = 0; b < B; ++b) {
#pragma omp parallel for collapse(2) schedule(static)
        for (int d = 0; d < D; ++d)
            for (int y = 0; y < H; ++y) {
                float *ixs = (float *)malloc(sizeof(float) * 2 * (size_t)(V > 1 ? V - 1 : 1));
                for (int x = 0; x < W; ++x) {
                    float dv = orc_depth_at(depth, depth_mode, b, d, y, x, D, H, W);
                    for (int v = 0; v < V - 1; ++v)
                        orc_sample_coord(rt + ((size_t)v * B + b) * 12, (float)x, (float)y, dv,
                                         H, W, align_corners, &ixs[2 * v], &ixs[2 * v + 1]);
                    for (int c = 0; c < C; ++c) {
                        float r = ref[((size_t)b * C + c) * plane + (size_t)y * W + x];
                        float q = r * r;               /* mvsnet.py:154 */
                        float s = alias_quirk ? q : r; /* mvsnet.py:153 / CVP alias */
                        for (int v = 0; v < V - 1; ++v) {
                            const float *pl = srcs + (((size_t)v * B + b) * C + c) * plane;
                            float wv = orc_bilinear(pl, H, W, ixs[2 * v], ixs[2 * v + 1]);
                            s = s + wv;      /* mvsnet.py:164 */
                            q = q + wv * wv; /* mvsnet.py:165 */
                        }
                        float sm = s / fV; /* mvsnet.py:170 */
                        out[((((size_t)b * C + c) * D + d) * H + y) * W + x] = q / fV - sm * sm;
                    }
                }
                free(ixs);
            }
    }
    return ORC_OK;
}

/* Backward of the variance volume w.r.t. the feature maps.  The sampling
 * grid is built under no_grad (module.py:62), so gradient flows only through
 * the bilinear taps (scatter-add) and the broadcast of ref over D
 * (mvsnet.py:152).  d var / d w_v = 2 w_v / V - 2 S / V^2 (same for ref with
 * its own value).  grad_ref [B,C,H,W], grad_srcs [V-1,B,C,H,W] are
 * overwritten.  Accumulation is serial (deterministic). alias_quirk is
 * inference-only in the reference and not differentiated here. */
int orc_costvol_variance_bwd_f32(const float *grad_var, const float *ref, const float *srcs,
                                 const float *rt, const float *depth, int depth_mode,
                                 int B, int V, int C, int D, int H, int W, int align_corners,
                                 float *grad_ref, float *grad_srcs)
{
    if (!grad_var || !ref || !depth || !grad_ref || B <= 0 || V < 1) return ORC_EINVAL;
    const size_t plane = (size_t)H * W;
    memset(grad_ref, 0, sizeof(float) * (size_t)B * C * plane);
    if (V > 1) memset(grad_srcs, 0, sizeof(float) * (size_t)(V - 1) * B * C * plane);
    const float fV = (float)V;
    float *wv = (float *)malloc(sizeof(float) * (size_t)(V > 1 ? V - 1 : 1));
    float *co = (float *)malloc(sizeof(float) * 2 * (size_t)(V > 1 ? V - 1 : 1));
    for (int b = 0; b < B; ++b)
        for (int d = 0; d < D; ++d)
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    float dv = orc_depth_at(depth, depth_mode, b, d, y, x, D, H, W);
                    for (int v = 0; v < V - 1; ++v)
                        orc_sample_coord(rt + ((size_t)v * B + b) * 12, (float)x, (float)y, dv,
                                         H, W, align_corners, &co[2 * v], &co[2 * v + 1]);
                    for (int c = 0; c < C; ++c) {
                        float g = grad_var[((((size_t)b * C + c) * D + d) * H + y) * W + x];
                        float r = ref[((size_t)b * C + c) * plane + (size_t)y * W + x];
                        float s = r;
                        for (int v = 0; v < V - 1; ++v) {
                            const float *pl = srcs + (((size_t)v * B + b) * C + c) * plane;
                            wv[v] = orc_bilinear(pl, H, W, co[2 * v], co[2 * v + 1]);
                            s += wv[v];
                        }
                        float k = 2.0f * s / (fV * fV);
                        grad_ref[((size_t)b * C + c) * plane + (size_t)y * W + x] +=
                            g * (2.0f * r / fV - k);
                        for (int v = 0; v < V - 1; ++v) {
                            float gw = g * (2.0f * wv[v] / fV - k);
                            float *gp = grad_srcs + (((size_t)v * B + b) * C + c) * plane;
                            float ix = co[2 * v], iy = co[2 * v + 1];
                            float x0f = floorf(ix), y0f = floorf(iy);
                            float w_ = ix - x0f, e = 1.0f - w_, n = iy - y0f, s_ = 1.0f - n;
                            int x0ok = (x0f >= 0.0f) && (x0f <= (float)(W - 1));
                            int x1ok = (x0f >= -1.0f) && (x0f <= (float)(W - 2));
                            int y0ok = (y0f >= 0.0f) && (y0f <= (float)(H - 1));
                            int y1ok = (y0f >= -1.0f) && (y0f <= (float)(H - 2));
                            int x0 = x0ok ? (int)x0f : 0, x1 = x1ok ? (int)x0f + 1 : 0;
                            int y0 = y0ok ? (int)y0f : 0, y1 = y1ok ? (int)y0f + 1 : 0;
                            if (x0ok && y0ok) gp[(size_t)y0 * W + x0] += gw * (s_ * e);
                            if (x1ok && y0ok) gp[(size_t)y0 * W + x1] += gw * (s_ * w_);
                            if (x0ok && y1ok) gp[(size_t)y1 * W + x0] += gw * (n * e);
                            if (x1ok && y1ok) gp[(size_t)y1 * W + x1] += gw * (n * w_);
                        }
                    }
                }
    free(wv);
    free(co);
    return ORC_OK;
}

/* ------------------------------------------------------------------ */
/* Conv3d k=3 pad=1 stride s, bias-free, followed by the per-channel affine
 * that eval-mode BatchNorm3d reduces to, optional ReLU, optional residual
 * add AFTER the ReLU (the skip adds of mvsnet.py:89-91).
 * MVSNet/models/module.py:26-33 (ConvBnReLU3D), mvsnet.py:48-93.
 * in [B,Ci,D,H,W], w [Co,Ci,3,3,3], scale/shift [Co] (NULL = identity),
 * out [B,Co,Do,Ho,Wo] with Xo = floor((X+2-3)/s)+1.                    */
int orc_conv3d_f32(const float *in, const float *w, const float *scale, const float *shift,
                   const float *residual, int relu,
                   int B, int Ci, int Co, int D, int H, int W, int stride, float *out)
{
    if (!in || !w || !out || stride < 1) return ORC_EINVAL;
    const int Do = (D - 1) / stride + 1, Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    for (int b = 0; b < B; ++b) {
#pragma omp parallel for collapse(2) schedule(static)
        for (int co = 0; co < Co; ++co)
            for (int z = 0; z < Do; ++z)
                for (int y = 0; y < Ho; ++y)
                    for (int x = 0; x < Wo; ++x) {
                        float acc = 0.0f;
                        for (int ci = 0; ci < Ci; ++ci)
                            for (int kz = 0; kz < 3; ++kz) {
                                int iz = z * stride + kz - 1;
                                if (iz < 0 || iz >= D) continue;
                                for (int ky = 0; ky < 3; ++ky) {
                                    int iy = y * stride + ky - 1;
                                    if (iy < 0 || iy >= H) continue;
                                    for (int kx = 0; kx < 3; ++kx) {
                                        int ixx = x * stride + kx - 1;
                                        if (ixx < 0 || ixx >= W) continue;
                                        acc += in[((((size_t)b * Ci + ci) * D + iz) * H + iy) * W + ixx] *
                                               w[((((size_t)co * Ci + ci) * 3 + kz) * 3 + ky) * 3 + kx];
                                    }
                                }
                            }
                        size_t o = ((((size_t)b * Co + co) * Do + z) * Ho + y) * Wo + x;
                        float v = acc;
                        if (scale) v = v * scale[co];
                        if (shift) v = v + shift[co];
                        if (relu && v < 0.0f) v = 0.0f;
                        if (residual) v = residual[o] + v;
                        out[o] = v;
                    }
    }
    return ORC_OK;
}

/* ConvTranspose3d k=3 stride=2 pad=1 output_padding=1 (mvsnet.py:66-79),
 * weight layout (Ci,Co,3,3,3); out[co,o] += in[ci,i]*w[ci,co,k], o=2i-1+k;
 * output size exactly 2x.  Same affine / ReLU / residual epilogue.
 * stride=1 (output_padding 0) covers CVP-MVSNet/models/net.py:66-69.  */
int orc_deconv3d_f32(const float *in, const float *w, const float *scale, const float *shift,
                     const float *residual, int relu,
                     int B, int Ci, int Co, int D, int H, int W, int stride, float *out)
{
    if (!in || !w || !out || (stride != 1 && stride != 2)) return ORC_EINVAL;
    const int Do = D * stride, Ho = H * stride, Wo = W * stride;
    for (int b = 0; b < B; ++b) {
#pragma omp parallel for collapse(2) schedule(static)
        for (int co = 0; co < Co; ++co)
            for (int z = 0; z < Do; ++z)
                for (int y = 0; y < Ho; ++y)
                    for (int x = 0; x < Wo; ++x) {
                        float acc = 0.0f;
                        for (int ci = 0; ci < Ci; ++ci)
                            for (int kz = 0; kz < 3; ++kz) {
                                int tz = z + 1 - kz;
                                if (tz < 0 || tz % stride) continue;
                                int iz = tz / stride;
                                if (iz >= D) continue;
                                for (int ky = 0; ky < 3; ++ky) {
                                    int ty = y + 1 - ky;
                                    if (ty < 0 || ty % stride) continue;
                                    int iy = ty / stride;
                                    if (iy >= H) continue;
                                    for (int kx = 0; kx < 3; ++kx) {
                                        int tx = x + 1 - kx;
                                        if (tx < 0 || tx % stride) continue;
                                        int ixx = tx / stride;
                                        if (ixx >= W) continue;
                                        acc += in[((((size_t)b * Ci + ci) * D + iz) * H + iy) * W + ixx] *
                                               w[((((size_t)ci * Co + co) * 3 + kz) * 3 + ky) * 3 + kx];
                                    }
                                }
                            }
                        size_t o = ((((size_t)b * Co + co) * Do + z) * Ho + y) * Wo + x;
                        float v = acc;
                        if (scale) v = v * scale[co];
                        if (shift) v = v + shift[co];
                        if (relu && v < 0.0f) v = 0.0f;
                        if (residual) v = residual[o] + v;
                        out[o] = v;
                    }
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------ */
/* softmax over D + expectation + photometric confidence:
 * mvsnet.py:183-191, module.py:91-103; Cas index clamp cas_mvsnet.py:63.
 * cost [B,D,H,W]; depth [B,D] (mode 0) or [B,D,H,W] (mode 1);
 * out_depth/out_conf [B,H,W]; out_prob [B,D,H,W] optional (NULL to skip). */
int orc_softmax_regress_conf_f32(const float *cost, const float *depth, int depth_mode,
                                 int clamp_idx, int B, int D, int H, int W,
                                 float *out_depth, float *out_conf, float *out_prob)
{
    if (!cost || !depth || !out_depth || !out_conf) return ORC_EINVAL;
    const size_t plane = (size_t)H * W;
    for (int b = 0; b < B; ++b) {
#pragma omp parallel for schedule(static)
        for (int p = 0; p < (int)plane; ++p) {
            float *pr = (float *)malloc(sizeof(float) * (size_t)D);
            const float *c = cost + (size_t)b * D * plane + p;
            float m = c[0];
            for (int d = 1; d < D; ++d) m = fmaxf(m, c[d * plane]);
            float sum = 0.0f;
            for (int d = 0; d < D; ++d) { pr[d] = expf(c[d * plane] - m); sum += pr[d]; }
            float dep = 0.0f, fidx = 0.0f;
            for (int d = 0; d < D; ++d) {
                pr[d] = pr[d] / sum;
                float dv = (depth_mode == 0) ? depth[(size_t)b * D + d]
                                             : depth[((size_t)b * D + d) * plane + p];
                dep += pr[d] * dv;            /* module.py:102 */
                fidx += pr[d] * (float)d;     /* mvsnet.py:189 */
                if (out_prob) out_prob[((size_t)b * D + d) * plane + p] = pr[d];
            }
            long idx = (long)fidx; /* .long(): truncation */
            if (clamp_idx) { if (idx < 0) idx = 0; if (idx > D - 1) idx = D - 1; }
            /* mvsnet.py:188: 4*avg_pool over padded (1,2) window = p[i-1..i+2] */
            float s4 = 0.0f;
            for (long k = idx - 1; k <= idx + 2; ++k)
                if (k >= 0 && k < D) s4 += pr[k];
            out_depth[(size_t)b * plane + p] = dep;
            out_conf[(size_t)b * plane + p] = s4;
            free(pr);
        }
    }
    return ORC_OK;
}

/* smooth-L1 (beta=1) mean over mask>0.5: mvsnet.py:201-203 */
int orc_masked_smooth_l1_f32(const float *est, const float *gt, const float *mask, size_t n,
                             float *out_loss)
{
    double acc = 0.0; size_t cnt = 0;
    for (size_t i = 0; i < n; ++i)
        if (mask[i] > 0.5f) {
            float e = fabsf(est[i] - gt[i]);
            acc += (e < 1.0f) ? 0.5f * e * e : e - 0.5f;
            ++cnt;
        }
    *out_loss = cnt ? (float)(acc / (double)cnt) : NAN;
    return ORC_OK;
}
