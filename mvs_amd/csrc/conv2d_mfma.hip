// FeatureNet (MVSNet/models/mvsnet.py:8-45) convolutions on the fp32 matrix cores:
// SURVEY.md section 8(f), "next" row 1.  Same machinery as conv3d_mfma.hip in two
// dimensions: implicit GEMM on v_mfma_f32_16x16x4_f32 with A = weights (16 output
// channels x 4 input channels), B = inputs (4 input channels x 16 pixels along x),
// channels-last activations, an LDS halo tile per block staged as 4 planes
// [kq][pixel][CK/4], BatchNorm(eval) affine + ReLU (+ bias) in the epilogue.
// Layers: 3x3 stride 1 and 5x5 stride 2 (x de-interleaved in LDS for stride 2), input
// channels 3 (read straight from the planar [B,3,H,W] image, padded to 4), 8, 16, 32.
// The two full-resolution 8-channel layers are HBM-bound (303 MB each way for 5 views);
// the kernel's job there is to stream whole lines and keep the MFMA work off the
// critical path, which MIOpen's generic fp32 igemm does not (1.2 ms per layer).
#include <cstdlib>
#include "mvs_common.h"
#include "conv_persistent.h"

namespace mvs {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int round_up2(int v, int m) { return (v + m - 1) / m * m; }

template <int CIN_, int COUT_, int KH_, int STRIDE_, int CK_, int TY_>
struct Conv2Cfg {
    static constexpr int CIN = CIN_;     // padded input channels (multiple of 4)
    static constexpr int COUT = COUT_, KH = KH_, STRIDE = STRIDE_, CK = CK_, TY = TY_;
    static constexpr int KS = CK / 4;    // floats per lane read: 1, 2 or 4
    static constexpr int MT = (COUT + 15) / 16;
    static constexpr int NTAPS = KH * KH;
    static constexpr int XT = 15 * STRIDE + KH;
    static constexpr int YT = (TY - 1) * STRIDE + KH;
    static constexpr int XH = (XT + 1) / 2;
    static constexpr int XTP = STRIDE == 2 ? 2 * XH : XT;
    static constexpr int NPIX = YT * XTP;
    // plane stride (pixel slots): conflict-free operand reads (see conv3d_mfma.hip)
    static constexpr int PLANE = KS == 4 ? round_up2(NPIX, 16) : round_up2(NPIX, 32) + 16;
    static constexpr int NCHUNK = CIN / CK;
    static constexpr int RPW = TY / 4;
    static constexpr int NITEMS = round_up2(NPIX, 16) * 4;
    static constexpr int NIT = (NITEMS + 255) / 256;
    static constexpr int LDS_FLOATS = 4 * PLANE * KS;
    static_assert(CIN % CK == 0 && (CK == 4 || CK == 8 || CK == 16), "bad chunk");
    static_assert(TY % 4 == 0 && NIT <= 32, "bad tile");
};

struct Conv2Args {
    const float *in, *wpk, *scale, *shift;
    float *out;
    int B, H, W;        // input size
    int Ho, Wo;         // output size
    int cin_real;       // channels present in memory (3 for the RGB layer)
    int in_planar;      // input is [B,cin_real,H,W] instead of [B,H,W,CIN]
    int tiles_x, tiles_y;
    int relu;
    unsigned *out_absmax;   // NULL, or the absmax block (mvs_common.h) the largest magnitude stored is max-ed into
};

template <class Cfg>
__global__ __launch_bounds__(256) void conv2d_mfma_kernel(Conv2Args a) {
    constexpr int CIN = Cfg::CIN, COUT = Cfg::COUT, KH = Cfg::KH, S = Cfg::STRIDE, CK = Cfg::CK;
    constexpr int KS = Cfg::KS, MT = Cfg::MT, RPW = Cfg::RPW, TY = Cfg::TY, NTAPS = Cfg::NTAPS;
    constexpr int XT = Cfg::XT, XH = Cfg::XH, XTP = Cfg::XTP;
    constexpr int NPIX = Cfg::NPIX, PLANE = Cfg::PLANE, NIT = Cfg::NIT, NITEMS = Cfg::NITEMS;
    constexpr int PAD = KH / 2;
    __shared__ __attribute__((aligned(16))) float lds[Cfg::LDS_FLOATS];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = lane & 15, kq = lane >> 4;
    int bid = blockIdx.x;
    const int tx = bid % a.tiles_x; bid /= a.tiles_x;
    const int ty = bid % a.tiles_y;
    const int b = bid / a.tiles_y;
    const int ox0 = tx * 16, oy0 = ty * TY;
    const int ix0 = ox0 * S - PAD, iy0 = oy0 * S - PAD;

    // staging geometry, once per block (piece-major lane order inside 64-item groups)
    int g_off[NIT];
    unsigned okmask = 0;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int e = min(tid + it * 256, NITEMS - 1);
        const int ekq = (e >> 4) & 3, v = ((e >> 6) << 4) | (e & 15);
        const int vc = min(v, NPIX - 1);
        const int lxp = vc % XTP, ly = vc / XTP;
        const int lx = (S == 2) ? (lxp < XH ? 2 * lxp : 2 * (lxp - XH) + 1) : lxp;
        const int gx = ix0 + lx, gy = iy0 + ly;
        bool ok = v < NPIX && lx < XT && gx >= 0 && gx < a.W && gy >= 0 && gy < a.H;
        const int cx = min(max(gx, 0), a.W - 1), cy = min(max(gy, 0), a.H - 1);
        if (a.in_planar) {   // [B,cin_real,H,W], KS == 1: piece kq is channel kq
            ok = ok && ekq < a.cin_real;
            g_off[it] = (min(ekq, a.cin_real - 1) * a.H + cy) * a.W + cx;
        } else {
            g_off[it] = (cy * a.W + cx) * CIN + ekq * KS;
        }
        okmask |= ok ? (1u << it) : 0u;
    }
    const float *in_b = a.in + (int64_t)b * a.H * a.W * (a.in_planar ? a.cin_real : CIN);

    f32x4 acc[RPW][MT];
#pragma unroll
    for (int r = 0; r < RPW; ++r)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[r][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int rd_base = (kq * PLANE + n) * KS;

#pragma unroll 1
    for (int ch = 0; ch < Cfg::NCHUNK; ++ch) {
        if (ch) __syncthreads();
        {
            float stg[NIT][KS];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const float *src = in_b + g_off[it] + ch * CK;
                if constexpr (KS == 4) {
                    const float4 t = *reinterpret_cast<const float4 *>(src);
                    stg[it][0] = t.x; stg[it][1] = t.y; stg[it][2] = t.z; stg[it][3] = t.w;
                } else if constexpr (KS == 2) {
                    const float2 t = *reinterpret_cast<const float2 *>(src);
                    stg[it][0] = t.x; stg[it][1] = t.y;
                } else {
                    stg[it][0] = *src;
                }
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int e = tid + it * 256;
                const int ekq = (e >> 4) & 3, v = ((e >> 6) << 4) | (e & 15);
                if (e >= NITEMS || v >= NPIX) continue;
                const bool ok = (okmask >> it) & 1u;
                float *dst = lds + (ekq * PLANE + v) * KS;
                if constexpr (KS == 4)
                    *reinterpret_cast<float4 *>(dst) =
                        make_float4(ok ? stg[it][0] : 0.f, ok ? stg[it][1] : 0.f,
                                    ok ? stg[it][2] : 0.f, ok ? stg[it][3] : 0.f);
                else if constexpr (KS == 2)
                    *reinterpret_cast<float2 *>(dst) =
                        make_float2(ok ? stg[it][0] : 0.f, ok ? stg[it][1] : 0.f);
                else
                    *dst = ok ? stg[it][0] : 0.f;
            }
        }
        __syncthreads();

        const float *wch = a.wpk + (int64_t)ch * NTAPS * MT * 64 * KS + lane * KS;
#pragma unroll
        for (int tap = 0; tap < NTAPS; ++tap) {
            const int ky = tap / KH, kx = tap % KH;
            const int xoff = (S == 2) ? ((kx & 1) * XH + (kx >> 1)) : kx;
            float af[MT][KS];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const float *wp = wch + (tap * MT + m) * 64 * KS;
                if constexpr (KS == 4) {
                    const float4 t = *reinterpret_cast<const float4 *>(wp);
                    af[m][0] = t.x; af[m][1] = t.y; af[m][2] = t.z; af[m][3] = t.w;
                } else if constexpr (KS == 2) {
                    const float2 t = *reinterpret_cast<const float2 *>(wp);
                    af[m][0] = t.x; af[m][1] = t.y;
                } else {
                    af[m][0] = *wp;
                }
            }
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int yr = wv * RPW + r;   // wave-uniform row of the tile
                const float *rp = lds + rd_base + ((yr * S + ky) * XTP + xoff) * KS;
                float bf[KS];
                if constexpr (KS == 4) {
                    const float4 t = *reinterpret_cast<const float4 *>(rp);
                    bf[0] = t.x; bf[1] = t.y; bf[2] = t.z; bf[3] = t.w;
                } else if constexpr (KS == 2) {
                    const float2 t = *reinterpret_cast<const float2 *>(rp);
                    bf[0] = t.x; bf[1] = t.y;
                } else {
                    bf[0] = *rp;
                }
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int s2 = 0; s2 < KS; ++s2)
                        acc[r][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m][s2], bf[s2],
                                                                         acc[r][m], 0, 0, 0);
            }
        }
    }

    float vmax = 0.0f;
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int oy = oy0 + wv * RPW + r, ox = ox0 + n;
        if (oy >= a.Ho || ox >= a.Wo) continue;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int c0 = m * 16 + kq * 4;
            if (c0 >= COUT) continue;
            f32x4 v = acc[r][m];
            if (a.scale) {
                const float4 sc = *reinterpret_cast<const float4 *>(a.scale + c0);
                v[0] *= sc.x; v[1] *= sc.y; v[2] *= sc.z; v[3] *= sc.w;
            }
            if (a.shift) {
                const float4 sh = *reinterpret_cast<const float4 *>(a.shift + c0);
                v[0] += sh.x; v[1] += sh.y; v[2] += sh.z; v[3] += sh.w;
            }
            if (a.relu == 1) {
                v[0] = relu_nan(v[0]); v[1] = relu_nan(v[1]);
                v[2] = relu_nan(v[2]); v[3] = relu_nan(v[3]);
            } else if (a.relu == 2) {   // LeakyReLU(0.1)
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : v[j] * 0.1f;
            }
            *reinterpret_cast<float4 *>(a.out + (((int64_t)b * a.Ho + oy) * a.Wo + ox) * COUT + c0) =
                make_float4(v[0], v[1], v[2], v[3]);
            vmax = amax4_nan(vmax, v[0], v[1], v[2], v[3]);
        }
    }
    publish_absmax(a.out_absmax, vmax);
}

// PyTorch (Cout,Cin,KH,KW) -> packed[ch][tap][mt][lane][s]; input channel =
// ch*CK + kq*KS + s (channels >= cin_real are zero: the padded RGB layer).
struct Pack2Args {
    const float *w;
    float *packed;
    int cin_real, Cout, KH, ck, mt;
};

__global__ __launch_bounds__(256) void conv2d_pack_kernel(Pack2Args p, int64_t total) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int KS = p.ck / 4, NT = p.KH * p.KH;
    int64_t t = i;
    const int s = (int)(t % KS); t /= KS;
    const int lane = (int)(t % 64); t /= 64;
    const int mt = (int)(t % p.mt); t /= p.mt;
    const int tap = (int)(t % NT); t /= NT;
    const int ch = (int)t;
    const int m = lane & 15, kq = lane >> 4;
    const int cin = ch * p.ck + kq * KS + s, co = mt * 16 + m;
    float val = 0.0f;
    if (cin < p.cin_real && co < p.Cout) val = p.w[((int64_t)co * p.cin_real + cin) * NT + tap];
    p.packed[i] = val;
}

// Layout of the persistent kernel (conv_persistent.h), 8-channel chunks:
// packed[ch][ky][kx'][mt][lane][s], input channel = ch*8 + 2*kq + s.  shifted (Cout = 8): 4
// x-taps, row m = (shift m>>3, channel m&7) holds w[kx' - shift] (zero outside 0..2).
struct PackP2Args {
    const float *w;
    float *packed;
    int Cin, Cout, KH, nkx, mt, shifted;
};

__global__ __launch_bounds__(256) void conv2d_pack_persistent_kernel(PackP2Args p, int64_t total) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int64_t t = i;
    const int s = (int)(t % 2); t /= 2;
    const int lane = (int)(t % 64); t /= 64;
    const int mt = (int)(t % p.mt); t /= p.mt;
    const int kx = (int)(t % p.nkx); t /= p.nkx;
    const int ky = (int)(t % p.KH); t /= p.KH;
    const int ch = (int)t;
    const int m = lane & 15, kq = lane >> 4;
    const int cin = ch * 8 + kq * 2 + s;
    const int co = p.shifted ? (m & 7) : mt * 16 + m;
    const int kxr = p.shifted ? kx - (m >> 3) : kx;
    float val = 0.0f;
    if (cin < p.Cin && co < p.Cout && kxr >= 0 && kxr < p.KH)
        val = p.w[(((int64_t)co * p.Cin + cin) * p.KH + ky) * p.KH + kxr];
    p.packed[i] = val;
}

// FeatureNet layers that run on the persistent DMA-fed kernel: the batch of images is a
// volume whose planes are the images, the kernel is one plane deep.
struct Persist2Info {
    int mode, ty, xout, nkx, mt, nchunk;
    void (*kernel)(ConvArgs, int);
};

static bool lookup_persist2(int Cin, int Cout, int ksize, int stride, Persist2Info &pi) {
#define MVS_P2(cin, cout, kh, st, mode, ty)                                                      \
    if (Cin == cin && Cout == cout && ksize == kh && stride == st) {                             \
        using P = PersistCfg<cin, cout, mode, 1, ty, 1, kh>;                                     \
        pi = Persist2Info{mode, ty, P::XOUT, P::NKX, P::MT, P::NCHUNK,                            \
                          conv3d_c8_persistent_kernel<P>};                                       \
        return true;                                                                             \
    }
    MVS_P2(8, 8, 3, 1, 2, 32)     // feature.conv1
    MVS_P2(8, 16, 5, 2, 1, 16)    // feature.conv2
    MVS_P2(16, 16, 3, 1, 0, 32)   // feature.conv3, conv4
    MVS_P2(16, 32, 5, 2, 1, 16)   // feature.conv5
    MVS_P2(32, 32, 3, 1, 0, 32)   // feature.conv6, feature.feature
    // CasMVSNet FPN heads (CasMVSNet/models/module.py:330-341)
    MVS_P2(32, 32, 1, 1, 0, 32)   // out1
    MVS_P2(16, 32, 1, 1, 0, 32)   // inner1
    MVS_P2(8, 32, 1, 1, 0, 32)    // inner2
    MVS_P2(32, 16, 3, 1, 0, 32)   // out2
    MVS_P2(32, 8, 3, 1, 2, 32)    // out3
    MVS_P2(16, 8, 3, 1, 2, 32)    // input gradient of feature.conv2 (5x5 stride 2, 8 -> 16): its four parity classes
    // CVP-MVSNet feature pyramid
    MVS_P2(64, 32, 3, 1, 0, 32)   // conv0bc
    // ((32,16) conv0bf, (32,32) conv0bd/be and (16,16) conv0bg/bh: entries above)
#undef MVS_P2
    return false;
}

static bool persist2_enabled() {
    const char *e = getenv("MVS_CONV2D_PERSISTENT");   // tuning / A-B: 0 = per-tile kernels
    return !e || atoi(e) != 0;
}

struct Cfg2Info {
    int cin_pad, ck, mt, ty, ntaps;
    void (*kernel)(Conv2Args);
};

template <class Cfg>
static Cfg2Info info2() {
    return Cfg2Info{Cfg::CIN, Cfg::CK, Cfg::MT, Cfg::TY, Cfg::NTAPS, conv2d_mfma_kernel<Cfg>};
}

static bool lookup2(int Cin, int Cout, int ksize, int stride, Cfg2Info &ci) {
#define MVS_C2(cin_real, cin_pad, cout, kh, st, ck, ty)                        \
    if (Cin == cin_real && Cout == cout && ksize == kh && stride == st) {      \
        ci = info2<Conv2Cfg<cin_pad, cout, kh, st, ck, ty>>();                 \
        return true;                                                           \
    }
    MVS_C2(3, 4, 8, 3, 1, 4, 32)      // feature.conv0 (RGB, planar input)
    MVS_C2(8, 8, 8, 3, 1, 8, 32)      // feature.conv1
    MVS_C2(8, 8, 16, 5, 2, 8, 16)     // feature.conv2
    MVS_C2(16, 16, 16, 3, 1, 16, 32)  // feature.conv3, conv4
    MVS_C2(16, 16, 32, 5, 2, 16, 8)   // feature.conv5
    MVS_C2(32, 32, 32, 3, 1, 16, 32)  // feature.conv6, feature.feature
    // CVP-MVSNet feature pyramid (CVP-MVSNet/models/net.py:28-37)
    MVS_C2(3, 4, 64, 3, 1, 4, 16)     // conv0aa (RGB, planar input)
    MVS_C2(64, 64, 64, 3, 1, 16, 16)  // conv0ba, conv0bb
#undef MVS_C2
    return false;
}

int conv2d_supported(int Cin, int Cout, int ksize, int stride) {
    Persist2Info pi;
    if (persist2_enabled() && lookup_persist2(Cin, Cout, ksize, stride, pi)) return 1;
    Cfg2Info ci;
    return lookup2(Cin, Cout, ksize, stride, ci) ? 1 : 0;
}

int64_t conv2d_packed_floats(int Cin, int Cout, int ksize, int stride) {
    Persist2Info pi;
    if (persist2_enabled() && lookup_persist2(Cin, Cout, ksize, stride, pi))
        return (int64_t)pi.nchunk * ksize * pi.nkx * pi.mt * 128;
    Cfg2Info ci;
    if (!lookup2(Cin, Cout, ksize, stride, ci)) return 0;
    return (int64_t)(ci.cin_pad / ci.ck) * ci.ntaps * ci.mt * 64 * (ci.ck / 4);
}

int conv2d_pack_launch(const float *weight, int Cin, int Cout, int ksize, int stride,
                       float *packed, hipStream_t st) {
    Persist2Info pi;
    if (persist2_enabled() && lookup_persist2(Cin, Cout, ksize, stride, pi)) {
        PackP2Args pp{weight, packed, Cin, Cout, ksize, pi.nkx, pi.mt, pi.mode == 2};
        const int64_t total = conv2d_packed_floats(Cin, Cout, ksize, stride);
        hipLaunchKernelGGL(conv2d_pack_persistent_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256),
                           0, st, pp, total);
        return check_launch("mvs_conv2d_pack_weights_f32");
    }
    Cfg2Info ci;
    if (!lookup2(Cin, Cout, ksize, stride, ci)) {
        set_error("mvs_conv2d_pack_weights_f32: no configuration for Cin=%d Cout=%d k=%d stride=%d",
                  Cin, Cout, ksize, stride);
        return MVS_EUNSUPPORTED;
    }
    Pack2Args p{weight, packed, Cin, Cout, ksize, ci.ck, ci.mt};
    const int64_t total = conv2d_packed_floats(Cin, Cout, ksize, stride);
    hipLaunchKernelGGL(conv2d_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       p, total);
    return check_launch("mvs_conv2d_pack_weights_f32");
}

int conv2d_launch(const float *in, const float *packed, const float *scale, const float *shift,
                  const float *coarse, int relu, int B, int Cin, int Cout, int H, int W, int ksize, int stride,
                  int layout_flags, float *out, hipStream_t st, unsigned *out_absmax) {
    const int in_planar = layout_flags & 1, out_c4 = (layout_flags >> 1) & 1;
    Persist2Info pi;
    const bool persistent = !in_planar && lookup_persist2(Cin, Cout, ksize, stride, pi);
    if (out_c4 && (!persistent || !persist2_enabled() || coarse || (Cout & 3))) {
        set_error("mvs_conv2d_f32: 4-channel blocked output is written by the persistent kernel's layers only "
                  "(Cin=%d Cout=%d k=%d stride=%d)", Cin, Cout, ksize, stride);
        return MVS_EUNSUPPORTED;
    }
    if (coarse && (!persistent || stride != 1 || (H & 1) || (W & 1))) {
        set_error("mvs_conv2d_f32: the upsampled residual needs a stride-1 layer of the persistent kernel "
                  "and even H, W (Cin=%d Cout=%d k=%d)", Cin, Cout, ksize);
        return MVS_EUNSUPPORTED;
    }
    if (persistent && (coarse || persist2_enabled())) {
        if ((int64_t)H * W * Cin * 4 >= 0xffffff00LL) return bare_error(MVS_EINVAL, __func__, __LINE__);   // 32-bit offsets inside one image
        ConvArgs a;
        a.in = in; a.wpk = packed; a.scale = scale; a.shift = shift; a.residual = coarse; a.out = out;
        a.res_up2 = coarse ? 1 : 0;
        a.B = 1; a.D = B; a.H = H; a.W = W;
        const int pad = ksize / 2;
        a.Do = B;
        a.Ho = (H + 2 * pad - ksize) / stride + 1;
        a.Wo = (W + 2 * pad - ksize) / stride + 1;
        a.tiles_x = (a.Wo + pi.xout - 1) / pi.xout;
        a.tiles_y = (a.Ho + pi.ty - 1) / pi.ty;
        a.tiles_z = B;
        a.relu = relu; a.in_c8 = 0; a.ystrip = 4; a.out_c4 = out_c4; a.out_absmax = out_absmax;
        const int64_t nt = (int64_t)a.tiles_x * a.tiles_y * a.tiles_z;
        if (nt <= 0 || nt > 0x7fffffffLL) return bare_error(MVS_EINVAL, __func__, __LINE__);
        const int n_cu = device_cu_count();
        hipLaunchKernelGGL(pi.kernel, dim3((unsigned)(nt < n_cu ? nt : n_cu)), dim3(512), 0, st, a, (int)nt);
        return check_launch("mvs_conv2d_f32(persistent)");
    }
    Cfg2Info ci;
    if (!lookup2(Cin, Cout, ksize, stride, ci)) {
        set_error("mvs_conv2d_f32: no configuration for Cin=%d Cout=%d k=%d stride=%d", Cin, Cout,
                  ksize, stride);
        return MVS_EUNSUPPORTED;
    }
    if (in_planar && ci.ck != 4) {
        set_error("mvs_conv2d_f32: planar input is only taken by the 3-channel layer");
        return MVS_EUNSUPPORTED;
    }
    if (!in_planar && Cin != ci.cin_pad) {
        set_error("mvs_conv2d_f32: channels-last input needs Cin %% 4 == 0");
        return MVS_EUNSUPPORTED;
    }
    if ((int64_t)H * W * ci.cin_pad >= (1ll << 31)) {
        set_error("mvs_conv2d_f32: image too large for 32-bit offsets");
        return MVS_EINVAL;
    }
    Conv2Args a;
    a.in = in; a.wpk = packed; a.scale = scale; a.shift = shift; a.out = out;
    a.B = B; a.H = H; a.W = W;
    const int pad = ksize / 2;
    a.Ho = (H + 2 * pad - ksize) / stride + 1;
    a.Wo = (W + 2 * pad - ksize) / stride + 1;
    a.cin_real = Cin; a.in_planar = in_planar;
    a.tiles_x = (a.Wo + 15) / 16;
    a.tiles_y = (a.Ho + ci.ty - 1) / ci.ty;
    a.relu = relu; a.out_absmax = out_absmax;
    const int64_t nblk = (int64_t)B * a.tiles_x * a.tiles_y;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return bare_error(MVS_EINVAL, __func__, __LINE__);
    hipLaunchKernelGGL(ci.kernel, dim3((unsigned)nblk), dim3(256), 0, st, a);
    return check_launch("mvs_conv2d_f32");
}

}  // namespace mvs

using namespace mvs;

extern "C" int mvs_conv2d_supported(int Cin, int Cout, int ksize, int stride) {
    return conv2d_supported(Cin, Cout, ksize, stride);
}

extern "C" int64_t mvs_conv2d_packed_weight_floats(int Cin, int Cout, int ksize, int stride) {
    return conv2d_packed_floats(Cin, Cout, ksize, stride);
}

extern "C" int mvs_conv2d_pack_weights_f32(const float *weight, int Cin, int Cout, int ksize,
                                           int stride, float *packed, void *stream) {
    if (!weight || !packed) {
        set_error("mvs_conv2d_pack_weights_f32: null pointer");
        return MVS_EINVAL;
    }
    return conv2d_pack_launch(weight, Cin, Cout, ksize, stride, packed, as_stream(stream));
}

extern "C" int mvs_conv2d_f32(const float *in, const float *packed_weight, const float *scale,
                              const float *shift, const float *coarse, int relu, int B, int Cin, int Cout,
                              int H, int W, int ksize, int stride, int layout_flags, float *out, void *stream) {
    return mvs_conv2d_absmax_f32(in, packed_weight, scale, shift, coarse, relu, B, Cin, Cout, H, W, ksize, stride, layout_flags,
                                 out, nullptr, stream);
}

// out_absmax: NULL, or the absmax block the largest magnitude of `out` is max-ed INTO in the kernels' epilogues (the caller clears it)
extern "C" int mvs_conv2d_absmax_f32(const float *in, const float *packed_weight, const float *scale,
                                     const float *shift, const float *coarse, int relu, int B, int Cin, int Cout,
                                     int H, int W, int ksize, int stride, int layout_flags, float *out, void *out_absmax,
                                     void *stream) {
    if (!in || !packed_weight || !out || B <= 0 || H <= 0 || W <= 0) {
        set_error("mvs_conv2d_f32: invalid argument");
        return MVS_EINVAL;
    }
    return conv2d_launch(in, packed_weight, scale, shift, coarse, relu, B, Cin, Cout, H, W, ksize, stride,
                         layout_flags, out, as_stream(stream), static_cast<unsigned *>(out_absmax));
}
