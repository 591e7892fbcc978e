// K3 fast path: 3x3x3 convolution as an implicit GEMM on the fp32 matrix cores
// (v_mfma_f32_16x16x4_f32), channels-last activations [B,D,H,W,C].
// Replaces nn.Conv3d + BatchNorm3d(eval) + ReLU (+ skip add) of
// MVSNet/models/module.py:26-33 and mvsnet.py:48-93.
//
// fp32 in / fp32 accumulate everywhere: the 1e-3 mm parity gate on depth leaves
// no room for bf16/fp16 operands, so the roofline for this
// kernel is the 157 TFLOP/s fp32-matrix peak.
//
// GEMM orientation (one MFMA = D[16x16] += A[16x4] * B[4x16]):
//   A = weights,  rows  = 16 output channels, k = 4 input channels
//   B = inputs,   k     = 4 input channels,   cols = 16 voxels along x
//   D lane (n = lane&15, q = lane>>4) holds output channels 4q..4q+3 of voxel n
//   -> the epilogue is one 16-byte store per lane and the wave writes a fully
//   contiguous run of voxels.
// The block stages a (TZ,TY,16)-voxel output tile's input halo for CK input
// channels at a time into LDS as 4 planes [kq][voxel][CK/4]: lane (n,kq) reads
// its CK/4 channels of one voxel with a single ds_read_b128 / b64 that feeds
// CK/4 MFMAs, conflict-free by construction (plane stride = 0 mod 16 slots for
// b128; = 16 mod 32 slots for b64).
//
// MODE 0: stride 1.  MODE 1: stride 2 (x de-interleaved in LDS so the stride-2
// reads of a wave stay on consecutive slots).  MODE 2: stride 1 with Cout = 8:
// the 16 MFMA rows carry (8 channels) x (2 x-shifts) against a 4-tap x window,
// i.e. 75 % useful MFMA work instead of the 50 % a zero-padded M tile gives --
// this is conv0, 68 % of CostRegNet's FLOPs.
#include "mvs_common.h"

namespace mvs {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int round_up_c(int v, int m) { return (v + m - 1) / m * m; }

template <int CIN_, int COUT_, int MODE_, int CK_, int TZ_, int TY_>
struct ConvCfg {
    static constexpr int CIN = CIN_, COUT = COUT_, MODE = MODE_, CK = CK_, TZ = TZ_, TY = TY_;
    static constexpr int SX = (MODE == 0) ? 1 : 2;    // x step of a wave's B reads
    static constexpr int SZY = (MODE == 1) ? 2 : 1;   // conv stride in y, z
    static constexpr int NKX = (MODE == 2) ? 4 : 3;   // x taps
    static constexpr int NTAPS = 9 * NKX;
    static constexpr int KS = CK / 4;                 // MFMA k-steps (floats per lane read)
    static constexpr int MT = (MODE == 2) ? 1 : (COUT + 15) / 16;
    static constexpr int XT = 15 * SX + NKX;          // staged x extent
    static constexpr int YT = (TY - 1) * SZY + 3;
    static constexpr int ZT = (TZ - 1) * SZY + 3;
    static constexpr int XH = (XT + 1) / 2;
    static constexpr int XTP = (SX == 2) ? 2 * XH : XT;
    static constexpr int NVOX = ZT * YT * XTP;
    static constexpr int PLANE = (CK == 16) ? round_up_c(NVOX, 16) : round_up_c(NVOX, 32) + 16;
    static constexpr int NCHUNK = CIN / CK;
    static constexpr int ROWS = TZ * TY;
    static constexpr int RPW = ROWS / 4;              // N-tiles (rows) per wave
    static constexpr int XOUT = (MODE == 2) ? 32 : 16;
    static constexpr int LDS_FLOATS = 4 * PLANE * KS;
    static_assert(CIN % CK == 0, "CIN must be a multiple of the chunk");
    static_assert(CK == 8 || CK == 16, "chunk is 8 or 16 channels");
    static_assert(ROWS % 4 == 0, "rows split over 4 waves");
    static_assert(MODE != 2 || COUT == 8, "MODE 2 is the Cout=8 shifted form");
};

struct ConvArgs {
    const float *in, *wpk, *scale, *shift, *residual;
    float *out;
    int B, D, H, W;        // input dims
    int Do, Ho, Wo;        // output dims
    int tiles_x, tiles_y, tiles_z;
    int relu;
};

// XCD-aware bijective remap: consecutive tiles land on the same XCD (same L2)
// so halo re-reads of neighbouring tiles hit that L2 (guide T1).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

template <class Cfg>
__global__ __launch_bounds__(256) void conv3d_mfma_kernel(ConvArgs a) {
    constexpr int CIN = Cfg::CIN, COUT = Cfg::COUT, MODE = Cfg::MODE, CK = Cfg::CK;
    constexpr int KS = Cfg::KS, MT = Cfg::MT, RPW = Cfg::RPW, TY = Cfg::TY, TZ = Cfg::TZ;
    constexpr int SX = Cfg::SX, SZY = Cfg::SZY, NKX = Cfg::NKX, NTAPS = Cfg::NTAPS;
    constexpr int XT = Cfg::XT, YT = Cfg::YT, XH = Cfg::XH, XTP = Cfg::XTP;
    constexpr int NVOX = Cfg::NVOX, PLANE = Cfg::PLANE;
    __shared__ __attribute__((aligned(16))) float lds[Cfg::LDS_FLOATS];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = lane & 15, kq = lane >> 4;

    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tx = bid % a.tiles_x; bid /= a.tiles_x;
    const int ty = bid % a.tiles_y; bid /= a.tiles_y;
    const int tz = bid % a.tiles_z;
    const int b = bid / a.tiles_z;
    const int ox0 = tx * Cfg::XOUT, oy0 = ty * TY, oz0 = tz * TZ;
    const int ix0 = (MODE == 1 ? 2 * ox0 : ox0) - 1;
    const int iy0 = oy0 * SZY - 1, iz0 = oz0 * SZY - 1;

    f32x4 acc[RPW][MT];
#pragma unroll
    for (int r = 0; r < RPW; ++r)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[r][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const float *in_b = a.in + (int64_t)b * a.D * a.H * a.W * CIN;
    // per-lane LDS read base (floats): plane kq, slot n
    const int rd_base = (kq * PLANE + n) * KS;

#pragma unroll 1
    for (int ch = 0; ch < Cfg::NCHUNK; ++ch) {
        if (ch) __syncthreads();
        // ---- stage the halo tile of CK channels: global (channels-last) -> LDS planes
        for (int e = tid; e < 4 * NVOX; e += 256) {
            const int ekq = e / NVOX, v = e - ekq * NVOX;
            const int lxp = v % XTP, t2 = v / XTP;
            const int ly = t2 % YT, lz = t2 / YT;
            const int lx = (SX == 2) ? (lxp < XH ? 2 * lxp : 2 * (lxp - XH) + 1) : lxp;
            const int gx = ix0 + lx, gy = iy0 + ly, gz = iz0 + lz;
            const bool ok = lx < XT && gx >= 0 && gx < a.W && gy >= 0 && gy < a.H && gz >= 0 &&
                            gz < a.D;
            const float *src =
                in_b + (((int64_t)gz * a.H + gy) * a.W + gx) * CIN + ch * CK + ekq * KS;
            float *dst = lds + (ekq * PLANE + v) * KS;
            if constexpr (KS == 4) {
                float4 val = ok ? *reinterpret_cast<const float4 *>(src)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4 *>(dst) = val;
            } else {
                float2 val = ok ? *reinterpret_cast<const float2 *>(src) : make_float2(0.f, 0.f);
                *reinterpret_cast<float2 *>(dst) = val;
            }
        }
        __syncthreads();

        const float *wch = a.wpk + (int64_t)ch * NTAPS * MT * 64 * KS + lane * KS;
#pragma unroll 1
        for (int kz = 0; kz < 3; ++kz) {
            const float *rdz = lds + rd_base + kz * (YT * XTP) * KS;
            const float *wkz = wch + kz * (3 * NKX) * MT * 64 * KS;
#pragma unroll
            for (int kyx = 0; kyx < 3 * NKX; ++kyx) {
                const int ky = kyx / NKX, kx = kyx % NKX;
                const int xoff = (SX == 2) ? ((kx & 1) * XH + (kx >> 1)) : kx;
                float af[MT][KS];
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const float *wp = wkz + (kyx * MT + m) * 64 * KS;
                    if constexpr (KS == 4) {
                        float4 t = *reinterpret_cast<const float4 *>(wp);
                        af[m][0] = t.x; af[m][1] = t.y; af[m][2] = t.z; af[m][3] = t.w;
                    } else {
                        float2 t = *reinterpret_cast<const float2 *>(wp);
                        af[m][0] = t.x; af[m][1] = t.y;
                    }
                }
#pragma unroll
                for (int r = 0; r < RPW; ++r) {
                    const int row = wv * RPW + r;  // wave-uniform
                    const int zr = row / TY, yr = row % TY;
                    const float *rp = rdz + (((zr * SZY) * YT + (yr * SZY + ky)) * XTP + xoff) * KS;
                    float bf[KS];
                    if constexpr (KS == 4) {
                        float4 t = *reinterpret_cast<const float4 *>(rp);
                        bf[0] = t.x; bf[1] = t.y; bf[2] = t.z; bf[3] = t.w;
                    } else {
                        float2 t = *reinterpret_cast<const float2 *>(rp);
                        bf[0] = t.x; bf[1] = t.y;
                    }
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int s = 0; s < KS; ++s)
                            acc[r][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m][s], bf[s],
                                                                             acc[r][m], 0, 0, 0);
                }
            }
        }
    }

    // ---- epilogue: BN affine, ReLU, skip add, one 16-byte store per lane
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int row = wv * RPW + r;
        const int oz = oz0 + row / TY, oy = oy0 + row % TY;
        if (oz >= a.Do || oy >= a.Ho) continue;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            int ox, c0;
            if (MODE == 2) {
                ox = ox0 + 2 * n + (kq >> 1);
                c0 = (kq & 1) * 4;
            } else {
                ox = ox0 + n;
                c0 = m * 16 + kq * 4;
            }
            if (ox >= a.Wo || c0 >= COUT) continue;
            f32x4 v = acc[r][m];
            if (a.scale) {
                const float4 sc = *reinterpret_cast<const float4 *>(a.scale + c0);
                v[0] *= sc.x; v[1] *= sc.y; v[2] *= sc.z; v[3] *= sc.w;
            }
            if (a.shift) {
                const float4 sh = *reinterpret_cast<const float4 *>(a.shift + c0);
                v[0] += sh.x; v[1] += sh.y; v[2] += sh.z; v[3] += sh.w;
            }
            if (a.relu) {
                v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f);
                v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
            }
            const int64_t o = ((((int64_t)b * a.Do + oz) * a.Ho + oy) * a.Wo + ox) * COUT + c0;
            if (a.residual) {
                const float4 rs = *reinterpret_cast<const float4 *>(a.residual + o);
                v[0] += rs.x; v[1] += rs.y; v[2] += rs.z; v[3] += rs.w;
            }
            *reinterpret_cast<float4 *>(a.out + o) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

// ---------------------------------------------------------------------
// Weight packing: PyTorch (Cout,Cin,3,3,3) -> A-fragment order
// packed[ch][tap][mt][lane][s], lane = (m = lane&15, kq = lane>>4),
// input channel = ch*CK + kq*KS + s.
struct PackArgs {
    const float *w;
    float *packed;
    int Cin, Cout, mode, ck, mt;
};

__global__ __launch_bounds__(256) void conv3d_pack_kernel(PackArgs p, int64_t total) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int KS = p.ck / 4;
    const int NKX = p.mode == 2 ? 4 : 3;
    const int NTAPS = 9 * NKX;
    int64_t t = i;
    const int s = (int)(t % KS); t /= KS;
    const int lane = (int)(t % 64); t /= 64;
    const int mt = (int)(t % p.mt); t /= p.mt;
    const int tap = (int)(t % NTAPS); t /= NTAPS;
    const int ch = (int)t;
    const int m = lane & 15, kq = lane >> 4;
    const int cin = ch * p.ck + kq * KS + s;
    const int kz = tap / (3 * NKX), ky = (tap / NKX) % 3, kxp = tap % NKX;
    float val = 0.0f;
    if (p.mode == 2) {
        const int sft = m >> 3, co = m & 7, kx = kxp - sft;
        if (kx >= 0 && kx <= 2)
            val = p.w[((int64_t)co * p.Cin + cin) * 27 + kz * 9 + ky * 3 + kx];
    } else {
        const int co = mt * 16 + m;
        if (co < p.Cout) val = p.w[((int64_t)co * p.Cin + cin) * 27 + kz * 9 + ky * 3 + kxp];
    }
    p.packed[i] = val;
}

// ---------------------------------------------------------------------
// Layer-shape -> kernel configuration table.
struct CfgInfo {
    int mode, ck, mt, tz, ty, xout, ntaps;
    void (*kernel)(ConvArgs);
};

template <class Cfg>
static CfgInfo info_of() {
    return CfgInfo{Cfg::MODE, Cfg::CK, Cfg::MT, Cfg::TZ, Cfg::TY, Cfg::XOUT, Cfg::NTAPS,
                   conv3d_mfma_kernel<Cfg>};
}

static bool lookup(int transposed, int Cin, int Cout, int stride, CfgInfo &ci) {
    if (transposed) return false;
#define MVS_CFG(cin, cout, mode, ck, tz, ty)                       \
    if (Cin == cin && Cout == cout) {                              \
        ci = info_of<ConvCfg<cin, cout, mode, ck, tz, ty>>();      \
        return true;                                               \
    }
    if (stride == 1) {
        // Cout = 8: shifted form (conv0 of MVSNet / CasMVSNet stages / CVP-free)
        MVS_CFG(32, 8, 2, 8, 4, 8)
        MVS_CFG(16, 8, 2, 8, 4, 8)
        MVS_CFG(8, 8, 2, 8, 4, 8)
        MVS_CFG(8, 16, 0, 8, 4, 8)
        MVS_CFG(16, 16, 0, 16, 4, 8)
        MVS_CFG(32, 32, 0, 16, 4, 8)
        MVS_CFG(64, 64, 0, 16, 4, 8)
        MVS_CFG(16, 32, 0, 16, 4, 8)
        MVS_CFG(32, 64, 0, 16, 4, 8)
    } else if (stride == 2) {
        MVS_CFG(8, 16, 1, 8, 2, 4)
        MVS_CFG(16, 32, 1, 8, 2, 4)
        MVS_CFG(32, 64, 1, 8, 2, 4)
    }
#undef MVS_CFG
    return false;
}

int conv3d_mfma_supported(int transposed, int Cin, int Cout, int stride) {
    CfgInfo ci;
    return lookup(transposed, Cin, Cout, stride, ci) ? 1 : 0;
}

int64_t conv3d_packed_floats(int transposed, int Cin, int Cout, int stride) {
    CfgInfo ci;
    if (!lookup(transposed, Cin, Cout, stride, ci)) return 0;
    return (int64_t)(Cin / ci.ck) * ci.ntaps * ci.mt * 64 * (ci.ck / 4);
}

int conv3d_pack_launch(const float *weight, int transposed, int Cin, int Cout, int stride,
                       float *packed, hipStream_t st) {
    CfgInfo ci;
    if (!lookup(transposed, Cin, Cout, stride, ci)) {
        set_error("mvs_conv3d_pack_weights_f32: no MFMA configuration for %s Cin=%d Cout=%d stride=%d",
                  transposed ? "deconv" : "conv", Cin, Cout, stride);
        return MVS_EUNSUPPORTED;
    }
    PackArgs p{weight, packed, Cin, Cout, ci.mode, ci.ck, ci.mt};
    const int64_t total = conv3d_packed_floats(transposed, Cin, Cout, stride);
    hipLaunchKernelGGL(conv3d_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       p, total);
    return check_launch("mvs_conv3d_pack_weights_f32");
}

int conv3d_mfma_launch(const float *in, const float *packed, const float *scale,
                       const float *shift, const float *residual, int relu, int transposed, int B,
                       int Cin, int Cout, int D, int H, int W, int stride, float *out,
                       hipStream_t st) {
    CfgInfo ci;
    if (!lookup(transposed, Cin, Cout, stride, ci)) {
        set_error("mvs_conv3d_f32(mfma): no configuration for %s Cin=%d Cout=%d stride=%d",
                  transposed ? "deconv" : "conv", Cin, Cout, stride);
        return MVS_EUNSUPPORTED;
    }
    ConvArgs a;
    a.in = in; a.wpk = packed; a.scale = scale; a.shift = shift; a.residual = residual;
    a.out = out;
    a.B = B; a.D = D; a.H = H; a.W = W;
    a.Do = (D - 1) / stride + 1; a.Ho = (H - 1) / stride + 1; a.Wo = (W - 1) / stride + 1;
    a.tiles_x = (a.Wo + ci.xout - 1) / ci.xout;
    a.tiles_y = (a.Ho + ci.ty - 1) / ci.ty;
    a.tiles_z = (a.Do + ci.tz - 1) / ci.tz;
    a.relu = relu;
    const int64_t nblk = (int64_t)B * a.tiles_x * a.tiles_y * a.tiles_z;
    if (nblk <= 0 || nblk > 0x7fffffffLL) {
        set_error("mvs_conv3d_f32(mfma): bad grid");
        return MVS_EINVAL;
    }
    hipLaunchKernelGGL(ci.kernel, dim3((unsigned)nblk), dim3(256), 0, st, a);
    return check_launch("mvs_conv3d_f32(mfma)");
}

}  // namespace mvs
