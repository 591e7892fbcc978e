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
#include <type_traits>
#include "mvs_common.h"
#include "conv_persistent.h"

#include <cstdlib>

namespace mvs {



template <int CIN_, int COUT_, int MODE_, int CK_, int TZ_, int TY_, int SB_ = 8, int PREA_ = -1>
struct ConvCfg {
    static constexpr int SB = SB_;   // staging loads in flight per thread
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
    // Preload a whole chunk's weight fragments into registers ahead of the staging
    // phase when they fit (conv0: 36 taps x 2 floats = 72 VGPRs): the MFMA loop then
    // never waits on a global load.
    static constexpr bool PREA = PREA_ < 0 ? (NTAPS * MT * KS <= 80) : (PREA_ != 0);
    static_assert(CIN % CK == 0, "CIN must be a multiple of the chunk");
    static_assert(CK == 8 || CK == 16, "chunk is 8 or 16 channels");
    static_assert(ROWS % 4 == 0, "rows split over 4 waves");
    static_assert(MODE != 2 || COUT == 8, "MODE 2 is the Cout=8 shifted form");
};



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

    const TileIdx tile = decode_tile(a, blockIdx.x, gridDim.x);
    const int tx = tile.tx, ty = tile.ty, tz = tile.tz, b = tile.b;
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

    // ---- staging geometry of this thread's halo items: identical for every chunk, so
    // the div/mod-by-constant address arithmetic (quarter-rate integer multiplies) is
    // paid once per block instead of once per chunk.  Within each group of 64 items the
    // lanes are ordered (piece, voxel): a wave's loads still cover one contiguous run,
    // and its ds_writes land on 16 consecutive slots of one plane (conflict-free).
    constexpr int NITEMS = round_up_c(NVOX, 16) * 4;   // whole 64-item groups (16 voxels x 4 pieces)
    constexpr int NIT = (NITEMS + 255) / 256;
    // Loads are buffer-addressed: descriptor base = the tile's first halo plane, a per-lane
    // byte offset fixed for all chunks (halo voxels outside the volume get one past
    // num_records and load zeros: no clamps, no selects), the chunk offset in an SGPR.
    unsigned voff[NIT];
    const int64_t plane_in = (int64_t)a.H * a.W * CIN;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(in_b + (int64_t)iz0 * plane_in), 0,
        (int)(unsigned)min((int64_t)Cfg::ZT * plane_in * 4, (int64_t)0xffffff00u), 0x00020000);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int e = tid + it * 256;
        const int ec = min(e, NITEMS - 1);
        const int ekq = (ec >> 4) & 3, v = ((ec >> 6) << 4) | (ec & 15);
        const int vc = min(v, NVOX - 1);
        const int lxp = vc % XTP, t2 = vc / XTP;
        const int ly = t2 % YT, lz = t2 / YT;
        const int lx = (SX == 2) ? (lxp < XH ? 2 * lxp : 2 * (lxp - XH) + 1) : lxp;
        const int gx = ix0 + lx, gy = iy0 + ly, gz = iz0 + lz;
        const bool ok = v < NVOX && lx < XT && gx >= 0 && gx < a.W && gy >= 0 && gy < a.H &&
                        gz >= 0 && gz < a.D;
        const int64_t off = a.in_c8 ? ((((int64_t)lz * a.H + gy) * (CIN / 8)) * a.W + gx) * 8 + ekq * KS
                                    : (((int64_t)lz * a.H + gy) * a.W + gx) * CIN + ekq * KS;
        voff[it] = ok ? (unsigned)(off * 4) : 0xffffff00u;
    }
    const int ch_step = a.in_c8 ? a.W * 8 * (CK / 8) : CK;

#pragma unroll 1
    for (int ch = 0; ch < Cfg::NCHUNK; ++ch) {
        const float *wch = a.wpk + (int64_t)ch * NTAPS * MT * 64 * KS + lane * KS;
        float apre[Cfg::PREA ? NTAPS : 1][MT][KS];
        if constexpr (Cfg::PREA) {
#pragma unroll
            for (int t = 0; t < NTAPS; ++t)
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const float *wp = wch + (t * MT + m) * 64 * KS;
                    if constexpr (KS == 4) {
                        float4 w4 = *reinterpret_cast<const float4 *>(wp);
                        apre[t][m][0] = w4.x; apre[t][m][1] = w4.y; apre[t][m][2] = w4.z; apre[t][m][3] = w4.w;
                    } else {
                        float2 w2 = *reinterpret_cast<const float2 *>(wp);
                        apre[t][m][0] = w2.x; apre[t][m][1] = w2.y;
                    }
                }
        }
        if (ch) __syncthreads();
        // ---- stage the halo tile of CK channels: global -> LDS planes, SB loads in
        // flight per thread per batch (the volume streams from HBM)
        {
            constexpr int SB = Cfg::SB;
#pragma unroll
            for (int it0 = 0; it0 < NIT; it0 += SB) {
                float stg[SB][KS];
#pragma unroll
                for (int j = 0; j < SB; ++j) {
                    if (it0 + j >= NIT) continue;
                    if constexpr (KS == 4) {
                        const auto val = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[it0 + j], ch * ch_step * 4, 0);
                        stg[j][0] = __uint_as_float(val[0]); stg[j][1] = __uint_as_float(val[1]);
                        stg[j][2] = __uint_as_float(val[2]); stg[j][3] = __uint_as_float(val[3]);
                    } else {
                        const auto val = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff[it0 + j], ch * ch_step * 4, 0);
                        stg[j][0] = __uint_as_float(val[0]); stg[j][1] = __uint_as_float(val[1]);
                    }
                }
#pragma unroll
                for (int j = 0; j < SB; ++j) {
                    if (it0 + j >= NIT) continue;
                    const int e = tid + (it0 + j) * 256;
                    const int ekq = (e >> 4) & 3, v = ((e >> 6) << 4) | (e & 15);
                    if (e >= NITEMS || v >= NVOX) continue;
                    float *dst = lds + (ekq * PLANE + v) * KS;
                    if constexpr (KS == 4)
                        *reinterpret_cast<float4 *>(dst) = make_float4(stg[j][0], stg[j][1], stg[j][2], stg[j][3]);
                    else
                        *reinterpret_cast<float2 *>(dst) = make_float2(stg[j][0], stg[j][1]);
                }
            }
        }
        __syncthreads();

        constexpr int KZ_UNROLL = Cfg::PREA ? 3 : 1;
#pragma unroll KZ_UNROLL
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
                    if constexpr (Cfg::PREA) {
#pragma unroll
                        for (int k = 0; k < KS; ++k) af[m][k] = apre[kz * 3 * NKX + kyx][m][k];
                    } else if constexpr (KS == 4) {
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
                        for (int s = 0; s < KS; ++s) {
                            acc[r][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m][s], bf[s], acc[r][m], 0, 0, 0);
                        }
                }
            }
        }
    }
    // ---- epilogue: BN affine, ReLU, skip add, one 16-byte store per lane
    float vmax = 0.0f;      // largest magnitude this lane stores (-> a.out_absmax)
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
                v[0] = relu_nan(v[0]); v[1] = relu_nan(v[1]);
                v[2] = relu_nan(v[2]); v[3] = relu_nan(v[3]);
            }
            const int64_t o = ((((int64_t)b * a.Do + oz) * a.Ho + oy) * a.Wo + ox) * COUT + c0;
            if (a.residual) {
                const float4 rs = *reinterpret_cast<const float4 *>(a.residual + o);
                v[0] += rs.x; v[1] += rs.y; v[2] += rs.z; v[3] += rs.w;
            }
            *reinterpret_cast<float4 *>(a.out + o) = make_float4(v[0], v[1], v[2], v[3]);
            vmax = amax4_nan(vmax, v[0], v[1], v[2], v[3]);
        }
    }
    publish_absmax(a.out_absmax, vmax);
}

// ---------------------------------------------------------------------
// Transposed convolution k=3, stride 2, pad 1, output_padding 1 (mvsnet.py:66-79)
// on the same MFMA machinery.  out[o] += in[i]*w[k] with o = 2i-1+k, so per
// dimension an even output (o=2j) sees one tap (k=1, i=j) and an odd output
// (o=2j+1) two (k=2, i=j) and (k=0, i=j+1).  The 8 output parity classes are 8
// small convolutions on the INPUT grid with 1..8 taps (27 in total): a wave's
// N-tile is 16 consecutive input positions jx, and each class keeps its own
// accumulators, so no MFMA work is wasted on structural zeros.
// PXM (Cout = 8): the two x parities share one MFMA -- rows = (px, 8 channels),
// k = (dx in {0,1}, ci) -- 18 instead of 27 half-empty class-taps, and the
// epilogue becomes one fully contiguous 1-KiB store per wave.
template <int CIN_, int COUT_, int CK_, int TZ_, int TY_, bool PXM_>
struct DeconvCfg {
    static constexpr int CIN = CIN_, COUT = COUT_, CK = CK_, TZ = TZ_, TY = TY_;
    static constexpr bool PXM = PXM_;
    static constexpr int KS = CK / 4;
    static constexpr int MT = PXM ? 1 : (COUT + 15) / 16;
    static constexpr int XT = 17, YT = TY + 1, ZT = TZ + 1;
    static constexpr int NVOX = ZT * YT * XT;
    static constexpr int PLANE = (CK == 16) ? round_up_c(NVOX, 16) : round_up_c(NVOX, 32) + 16;
    static constexpr int NCHUNK = CIN / CK;
    static constexpr int ROWS = TZ * TY;
    static constexpr int RPW = ROWS / 4;
    static constexpr int NCLS = PXM ? 4 : 8;
    static constexpr int NCT = PXM ? 18 : 27;
    static constexpr int LDS_FLOATS = 4 * PLANE * KS;
    static_assert(CIN % CK == 0 && ROWS % 4 == 0, "bad deconv tile");
    static_assert(!PXM || COUT == 8, "x-parity merge is the Cout=8 form");
};

// per-dimension tap t of parity p -> (kernel index k, input offset d)
__host__ __device__ constexpr int dc_k(int p, int t) { return p == 0 ? 1 : (t == 0 ? 2 : 0); }
__host__ __device__ constexpr int dc_d(int p, int t) { return (p == 1 && t == 1) ? 1 : 0; }

template <class Cfg>
__global__ __launch_bounds__(256) void deconv3d_mfma_kernel(ConvArgs a) {
    constexpr int CIN = Cfg::CIN, COUT = Cfg::COUT, CK = Cfg::CK, KS = Cfg::KS, MT = Cfg::MT;
    constexpr int RPW = Cfg::RPW, TY = Cfg::TY, TZ = Cfg::TZ, XT = Cfg::XT, YT = Cfg::YT;
    constexpr int NVOX = Cfg::NVOX, PLANE = Cfg::PLANE, NCT = Cfg::NCT, NCLS = Cfg::NCLS;
    constexpr bool PXM = Cfg::PXM;
    __shared__ __attribute__((aligned(16))) float lds[Cfg::LDS_FLOATS];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = lane & 15, kq = lane >> 4;
    const TileIdx tile = decode_tile(a, blockIdx.x, gridDim.x);
    const int tx = tile.tx, ty = tile.ty, tz = tile.tz, b = tile.b;
    const int jx0 = tx * 16, jy0 = ty * TY, jz0 = tz * TZ;

    f32x4 acc[NCLS][RPW][MT];
#pragma unroll
    for (int c = 0; c < NCLS; ++c)
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[c][r][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const float *in_b = a.in + (int64_t)b * a.D * a.H * a.W * CIN;
    const int rd_base = (kq * PLANE + n) * KS;

    // skip-connection values (mvsnet.py:89-91) for every output this lane will write:
    // issued now, consumed in the epilogue, so their HBM latency hides under the staging
    // and MFMA phases instead of stalling the store tail
    float4 res[NCLS][RPW][MT];
    if (a.residual) {
#pragma unroll
        for (int pz = 0; pz < 2; ++pz)
#pragma unroll
            for (int py = 0; py < 2; ++py)
#pragma unroll
                for (int px = 0; px < (PXM ? 1 : 2); ++px)
#pragma unroll
                    for (int r = 0; r < RPW; ++r)
#pragma unroll
                        for (int m = 0; m < MT; ++m) {
                            const int cls = (pz * 2 + py) * (PXM ? 1 : 2) + px;
                            const int row = wv * RPW + r;
                            const int jz = min(jz0 + row / TY, a.D - 1), jy = min(jy0 + row % TY, a.H - 1);
                            const int jx = min(jx0 + n, a.W - 1);
                            const int oz = 2 * jz + pz, oy = 2 * jy + py;
                            const int ox = 2 * jx + (PXM ? (kq >> 1) : px);
                            const int c0 = min(PXM ? (kq & 1) * 4 : m * 16 + kq * 4, COUT - 4);
                            res[cls][r][m] = *reinterpret_cast<const float4 *>(
                                a.residual + ((((int64_t)b * a.Do + oz) * a.Ho + oy) * a.Wo + ox) * COUT + c0);
                        }
    }

#pragma unroll 1
    for (int ch = 0; ch < Cfg::NCHUNK; ++ch) {
        if (ch) __syncthreads();
        {
            constexpr int NIT = (4 * NVOX + 255) / 256;
            constexpr int SB = 8;
#pragma unroll 1
            for (int it0 = 0; it0 < NIT; it0 += SB) {
                float stg[SB][KS];
                int dsto[SB];
#pragma unroll
                for (int j = 0; j < SB; ++j) {
                    const int e = tid + (it0 + j) * 256;
                    const int ec = min(e, 4 * NVOX - 1);
                    const int ekq = ec & 3, v = ec >> 2;
                    const int lx = v % XT, t2 = v / XT;
                    const int ly = t2 % YT, lz = t2 / YT;
                    const int gx = jx0 + lx, gy = jy0 + ly, gz = jz0 + lz;
                    const bool ok = gx < a.W && gy < a.H && gz < a.D;
                    const int cx = min(gx, a.W - 1), cy = min(gy, a.H - 1), cz = min(gz, a.D - 1);
                    const float *src =
                        in_b + (((int64_t)cz * a.H + cy) * a.W + cx) * CIN + ch * CK + ekq * KS;
                    dsto[j] = (e < 4 * NVOX && it0 + j < NIT) ? (ekq * PLANE + v) * KS : -1;
                    if constexpr (KS == 4) {
                        float4 val = *reinterpret_cast<const float4 *>(src);
                        stg[j][0] = ok ? val.x : 0.f; stg[j][1] = ok ? val.y : 0.f;
                        stg[j][2] = ok ? val.z : 0.f; stg[j][3] = ok ? val.w : 0.f;
                    } else {
                        float2 val = *reinterpret_cast<const float2 *>(src);
                        stg[j][0] = ok ? val.x : 0.f; stg[j][1] = ok ? val.y : 0.f;
                    }
                }
#pragma unroll
                for (int j = 0; j < SB; ++j) {
                    if (dsto[j] < 0) continue;
                    float *dst = lds + dsto[j];
                    if constexpr (KS == 4)
                        *reinterpret_cast<float4 *>(dst) =
                            make_float4(stg[j][0], stg[j][1], stg[j][2], stg[j][3]);
                    else
                        *reinterpret_cast<float2 *>(dst) = make_float2(stg[j][0], stg[j][1]);
                }
            }
        }
        __syncthreads();

        const float *wch = a.wpk + (int64_t)ch * NCT * MT * 64 * KS + lane * KS;
        int ct = 0;  // compile-time after unrolling
#pragma unroll
        for (int pz = 0; pz < 2; ++pz)
#pragma unroll
            for (int py = 0; py < 2; ++py)
#pragma unroll
                for (int px = 0; px < (PXM ? 1 : 2); ++px) {
                    const int cls = (pz * 2 + py) * (PXM ? 1 : 2) + px;
#pragma unroll
                    for (int tzz = 0; tzz <= pz; ++tzz)
#pragma unroll
                        for (int tyy = 0; tyy <= py; ++tyy)
#pragma unroll
                            for (int txx = 0; txx <= (PXM ? 1 : px); ++txx) {
                                const int dz = dc_d(pz, tzz), dy = dc_d(py, tyy);
                                const int dx = PXM ? txx : dc_d(px, txx);
                                float af[MT][KS];
#pragma unroll
                                for (int m = 0; m < MT; ++m) {
                                    const float *wp = wch + (ct * MT + m) * 64 * KS;
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
                                    const int row = wv * RPW + r;
                                    const int zr = row / TY, yr = row % TY;
                                    const float *rp =
                                        lds + rd_base + (((zr + dz) * YT + (yr + dy)) * XT + dx) * KS;
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
                                            acc[cls][r][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                                af[m][s], bf[s], acc[cls][r][m], 0, 0, 0);
                                }
                                ++ct;
                            }
                }
    }

#pragma unroll
    for (int pz = 0; pz < 2; ++pz)
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
            for (int px = 0; px < (PXM ? 1 : 2); ++px) {
                const int cls = (pz * 2 + py) * (PXM ? 1 : 2) + px;
#pragma unroll
                for (int r = 0; r < RPW; ++r) {
                    const int row = wv * RPW + r;
                    const int jz = jz0 + row / TY, jy = jy0 + row % TY, jx = jx0 + n;
                    if (jz >= a.D || jy >= a.H || jx >= a.W) continue;
                    const int oz = 2 * jz + pz, oy = 2 * jy + py;
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        const int ox = 2 * jx + (PXM ? (kq >> 1) : px);
                        const int c0 = PXM ? (kq & 1) * 4 : m * 16 + kq * 4;
                        if (c0 >= COUT) continue;
                        f32x4 v = acc[cls][r][m];
                        if (a.scale) {
                            const float4 sc = *reinterpret_cast<const float4 *>(a.scale + c0);
                            v[0] *= sc.x; v[1] *= sc.y; v[2] *= sc.z; v[3] *= sc.w;
                        }
                        if (a.shift) {
                            const float4 sh = *reinterpret_cast<const float4 *>(a.shift + c0);
                            v[0] += sh.x; v[1] += sh.y; v[2] += sh.z; v[3] += sh.w;
                        }
                        if (a.relu) {
                            v[0] = relu_nan(v[0]); v[1] = relu_nan(v[1]);
                            v[2] = relu_nan(v[2]); v[3] = relu_nan(v[3]);
                        }
                        const int64_t o =
                            ((((int64_t)b * a.Do + oz) * a.Ho + oy) * a.Wo + ox) * COUT + c0;
                        if (a.residual) {
                            const float4 rs = res[cls][r][m];
                            v[0] += rs.x; v[1] += rs.y; v[2] += rs.z; v[3] += rs.w;
                        }
                        *reinterpret_cast<float4 *>(a.out + o) = make_float4(v[0], v[1], v[2], v[3]);
                    }
                }
            }
}

// ---------------------------------------------------------------------
// Cout = 1 convolution (the `prob` layer, mvsnet.py:81,92: 8 -> 1 with bias).
// One output channel cannot fill an MFMA tile, so the layer runs on the VALU from an LDS
// halo tile: thread (x,y) of a 32x8 tile produces outputs along z and reads every staged
// voxel once for all of them; weights are broadcast LDS reads.  (Cin = 8 -- MVSNet's `prob` --
// runs on the marching kernel below; this one serves Cin = 16 and MVS_PROB_MARCH=0.)
template <int CIN, int NWV>
__global__ __launch_bounds__(NWV * 64) void conv3d_cout1_kernel(ConvArgs a, const float *__restrict__ w) {
    constexpr int CQ = CIN / 4, TX = 32, TY = 8, TZ = 4;   // (TZ = 2 measured slower: 0.57 vs 0.46 ms)
    // NWV = 8: waves 0-3 produce the tile's outputs z 0, 1, waves 4-7 z 2, 3 -- a wave issues a vector instruction
    // every ~6 cycles at best, and every wave is blocked ~250 cycles per copy it issues: twice the waves halve both
    constexpr int ZS = NWV / 4, TZL = TZ / ZS;             // z slabs of the tile, outputs per thread
    constexpr int XT = TX + 2, YT = TY + 2, ZT = TZ + 2, NVOX = ZT * YT * XT;
    constexpr int PLANE = round_up_c(NVOX, 64);            // whole 64-lane DMA instructions
    __shared__ __attribute__((aligned(16))) float lds[CQ * PLANE * 4];
    __shared__ __attribute__((aligned(16))) float wl[27 * CIN];
    if (a.run_flag && *a.run_flag == 0u) return;
    const int tid = threadIdx.x, lx = tid & 31, ly = (tid >> 5) & 7, zh = (tid >> 8) * TZL;
    const TileIdx tile = decode_tile(a, blockIdx.x, gridDim.x);
    const int tx = tile.tx, ty = tile.ty, tz = tile.tz, b = tile.b;
    const int x0 = tx * TX, y0 = ty * TY, z0 = tz * TZ;
    const float *in_b = a.in + (int64_t)b * a.D * a.H * a.W * CIN;
    {
        // halo tile HBM -> LDS by buffer-addressed DMA, planes [q][voxel][4 channels]:
        // instruction i of wave w covers voxels (i*NWV + w)*64 + lane of every channel quad q;
        // the descriptor's base is the tile's first halo plane, out-of-volume voxels (and the
        // plane's tail) get an offset past num_records and fetch zeros
        const int lane = tid & 63;
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        const unsigned lds_base = (unsigned)(uintptr_t)lds;
        const int64_t plane_in = (int64_t)a.H * a.W * CIN;
        const mvs_srd_t srd = make_srd(in_b + (int64_t)(z0 - 1) * plane_in,
                                       (unsigned)min((int64_t)ZT * plane_in * 4, (int64_t)0xffffff00u));
        constexpr int NI = (PLANE / 64 + NWV - 1) / NWV;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int vb = i * NWV + wv;
            if (PLANE / 64 % NWV && vb >= PLANE / 64) continue;   // wave-uniform
            const int v = vb * 64 + lane;
            const int vc = min(v, NVOX - 1);
            const int vx = vc % XT, t2 = vc / XT, vy = t2 % YT, vz = t2 / YT;
            const int gx = x0 + vx - 1, gy = y0 + vy - 1, gz = z0 + vz - 1;
            const bool ok = v < NVOX && (unsigned)gx < (unsigned)a.W && (unsigned)gy < (unsigned)a.H &&
                            (unsigned)gz < (unsigned)a.D;
            const unsigned off = ok ? (unsigned)(((int64_t)vz * plane_in + ((int64_t)gy * a.W + gx) * CIN) * 4)
                                    : 0xffffff00u;
#pragma unroll
            for (int q = 0; q < CQ; ++q)
                glds16_buf(off, srd, (unsigned)(q * 16), lds_base + (unsigned)((q * PLANE + vb * 64) * 16));
        }
        for (int i = tid; i < 27 * CIN; i += NWV * 64) {
            const int kyx = i / (3 * CIN), r = i - kyx * (3 * CIN), kz = r / CIN, ci = r - kz * CIN;
            wl[i] = w[ci * 27 + kz * 9 + kyx];   // PyTorch layout (1,CIN,3,3,3)
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    float acc[TZL];
#pragma unroll
    for (int z = 0; z < TZL; ++z) acc[z] = 0.f;
    // runtime loop over the 9 (ky,kx) taps; its 3*CIN weights come from LDS as broadcast
    // reads ([kyx][kz][ci], staged once per block).  Scalar loads per tap stalled the loop:
    // SMEM returns out of order, so every LDS wait behind them became lgkmcnt(0).
#pragma unroll 1
    for (int kyx = 0; kyx < 9; ++kyx) {
        const int ky = kyx / 3, kx = kyx - ky * 3;
        float wk[3][CIN];
#pragma unroll
        for (int kz = 0; kz < 3; ++kz)
#pragma unroll
            for (int c4 = 0; c4 < CIN / 4; ++c4) {
                const float4 t = *reinterpret_cast<const float4 *>(wl + (kyx * 3 + kz) * CIN + c4 * 4);
                wk[kz][c4 * 4 + 0] = t.x; wk[kz][c4 * 4 + 1] = t.y;
                wk[kz][c4 * 4 + 2] = t.z; wk[kz][c4 * 4 + 3] = t.w;
            }
        const float *lp = lds + ((zh * YT + ly + ky) * XT + lx + kx) * 4;
#pragma unroll
        for (int dz = 0; dz < TZL + 2; ++dz)
#pragma unroll
            for (int q = 0; q < CQ; ++q) {
                const float4 t = *reinterpret_cast<const float4 *>(lp + (q * PLANE + dz * YT * XT) * 4);
#pragma unroll
                for (int z = 0; z < TZL; ++z) {
                    const int kz = dz - z;
                    if (kz < 0 || kz > 2) continue;
                    acc[z] = fmaf(t.x, wk[kz][q * 4 + 0], acc[z]);
                    acc[z] = fmaf(t.y, wk[kz][q * 4 + 1], acc[z]);
                    acc[z] = fmaf(t.z, wk[kz][q * 4 + 2], acc[z]);
                    acc[z] = fmaf(t.w, wk[kz][q * 4 + 3], acc[z]);
                }
            }
    }
    const int ox = x0 + lx, oy = y0 + ly;
    if (ox >= a.Wo || oy >= a.Ho) return;
#pragma unroll
    for (int z = 0; z < TZL; ++z) {
        const int oz = z0 + zh + z;
        if (oz >= a.Do) continue;
        float v = acc[z];
        if (a.scale) v *= a.scale[0];
        if (a.shift) v += a.shift[0];
        if (a.relu) v = relu_nan(v);
        const int64_t o = (((int64_t)b * a.Do + oz) * a.Ho + oy) * a.Wo + ox;
        if (a.residual) v += a.residual[o];
        a.out[o] = v;
    }
}

// The same layer (Cin = 8) as a persistent kernel that MARCHES along z.  The per-tile kernel above pays per
// workgroup: launch, weights from global memory and copy geometry (0.11 ms of its 0.33 at config 2, measured with
// the copies and the arithmetic removed), and it fetches every z plane 1.5 times (a 6-plane halo per 4 output
// planes: 1.5 GB through LDS at the ~10-14 B / cycle a CU sustains is 0.23 ms by itself).  Here one workgroup per
// CU -- 8 computing waves, 4 copy waves -- owns a (y, x) tile and a depth segment: a ring of 12 z planes in LDS,
// per step the copy waves bring the next FOUR planes (each plane is fetched once per column) while the computing
// waves produce 4 output planes from the six that are there; one barrier per step; the weights are staged once.
// A copy wave owns three (channel quad, 64-voxel group) pieces of every plane, so its per-lane offsets are three
// registers per column and a copy costs one add and one select.
// Measured at config 2 (MVS_PROB_ABL: 1 no copies, 2 no arithmetic): 0.26-0.28 ms whole, 0.25 arithmetic only,
// 0.21 copies only, 0.06 neither (stores, barriers).  The arithmetic is what is left: a wave issues one instruction
// per ~7 cycles whatever its mix (scripts/micro/coissue.hip), 8 waves x (432 FMA + 81 ds_read_b128 + waits) per step
// are ~4600 cycles, and the same 81 reads x 8 waves x 8 cycles of the 128 B / cycle LDS are 5200 -- more waves trade
// one bound for the other (16 waves on half the planes each read 1.5x as much), weights held in registers
// (54 reads) left it at 0.24.  Sixteen accumulator chains per lane instead of four were worth 0.28 -> 0.25.
namespace march {
constexpr int TX = 32, TY = 8, TZ = 4, XT = TX + 2, YT = TY + 2;
constexpr int PVOX = 384;                 // voxels of a plane slot: 340 in whole 64-lane copies
constexpr int RING = 12;
constexpr int SLOT_FLOATS = 2 * PVOX * 4; // two channel quads
constexpr int LDS_FLOATS = RING * SLOT_FLOATS + 27 * 8;
static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS budget");
}  // namespace march

struct MarchArgs {
    ConvArgs c;
    int ncols, nseg, sps, nunits;     // (y, x) tiles per batch item x B; depth segments; steps per segment
    int abl;                          // tuning: 1 no copies, 2 no arithmetic (garbage results)
};

__global__ __launch_bounds__(768) void conv3d_cout1_march_kernel(MarchArgs m, const float *__restrict__ w) {
    using namespace march;
    constexpr int CIN = 8;
    const ConvArgs &a = m.c;
    if (a.run_flag && *a.run_flag == 0u) return;
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    float *wl = lds + RING * SLOT_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds_base = (unsigned)(uintptr_t)lds;
    const int nsteps = (a.D + TZ - 1) / TZ;
    const int64_t plane_in = (int64_t)a.H * a.W * CIN;
    int u = blockIdx.x;
    if (u >= m.nunits) return;
    // unit -> (segment, batch item, tile); segment-major: the units in flight are neighbouring columns
    int b, ty, tx, s, s_end;
    auto open_unit = [&](int uu) {
        const int seg = uu / m.ncols;
        int col = uu - seg * m.ncols;
        tx = col % a.tiles_x; col /= a.tiles_x;
        ty = col % a.tiles_y; b = col / a.tiles_y;
        s = seg * m.sps;
        s_end = min(s + m.sps, nsteps);
    };
    open_unit(u);
    int r0 = 0;     // ring slot of plane z = 4 s - 1 of the current step

    if (wv >= 8) {
        // ---------------------------------------------------------------- copy waves
        const int cw = wv - 8;
        unsigned offxy[3], ldst[3];
        auto column = [&]() {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int c = cw + 4 * i, q = c / 6, k = c % 6;       // wave-uniform
                const int v = k * 64 + lane;
                const int vy = v / XT, vx = v - vy * XT;
                const int gx = tx * TX + vx - 1, gy = ty * TY + vy - 1;
                const bool ok = v < XT * YT && (unsigned)gx < (unsigned)a.W && (unsigned)gy < (unsigned)a.H;
                offxy[i] = ok ? (unsigned)((((gy * a.W + gx) * CIN) + q * 4) * 4) : 0xffffff00u;
                ldst[i] = (unsigned)((q * PVOX + k * 64) * 16);
            }
        };
        column();
        asm volatile("s_barrier" ::: "memory");   // the computing waves' weight staging
        mvs_srd_t srd = make_srd(a.in + (int64_t)b * a.D * plane_in,
                                 (unsigned)min((int64_t)a.D * plane_in * 4, (int64_t)0xffffff00u));
        const unsigned plane_bytes = (unsigned)(plane_in * 4);
        auto issue = [&](int gz0, int nplanes, int slot0) {
            if (m.abl & 1) return;
            for (int p = 0; p < nplanes; ++p) {
                const int gz = gz0 + p;
                int slot = slot0 + p;
                if (slot >= RING) slot -= RING;
                const bool zok = (unsigned)gz < (unsigned)a.D;
                const unsigned zoff = zok ? (unsigned)gz * plane_bytes : 0u;
                const unsigned base = lds_base + (unsigned)(slot * SLOT_FLOATS * 4);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const unsigned off = (zok && offxy[i] != 0xffffff00u) ? zoff + offxy[i] : 0xffffff00u;
                    glds16_buf(off, srd, 0u, base + ldst[i]);
                }
            }
        };
        issue(s * TZ - 1, TZ + 2, r0);
        for (;;) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (s + 1 < s_end) {
                issue((s + 1) * TZ + 1, TZ, (r0 + TZ + 2) % RING);
                r0 = (r0 + TZ) % RING; ++s;
            } else {
                u += gridDim.x;
                if (u >= m.nunits) break;
                const int bprev = b;
                open_unit(u);
                column();
                if (b != bprev)
                    srd = make_srd(a.in + (int64_t)b * a.D * plane_in,
                                   (unsigned)min((int64_t)a.D * plane_in * 4, (int64_t)0xffffff00u));
                r0 = (r0 + TZ + 2) % RING;
                issue(s * TZ - 1, TZ + 2, r0);
            }
        }
        return;
    }

    // -------------------------------------------------------------------- computing waves
    // The arithmetic is bound by LDS reads (a 16-byte read per lane is 8 cycles of the CU's 128 B / cycle whether it
    // is data or a broadcast weight), so a wave = one row of the tile, its lower 32 lanes take channels 0-3 and its
    // upper 32 channels 4-7 of the same 32 pixels, each for all four output planes of the step: 54 data + 27 weight
    // reads per lane and step instead of 108 + 54; the two halves meet through one cross-lane add per output.
    const int lx = tid & 31, cq = (tid >> 5) & 1, ly = tid >> 6;
    const float sc = a.scale ? a.scale[0] : 1.0f, sh = a.shift ? a.shift[0] : 0.0f;
    for (int i = tid; i < 27 * CIN; i += 512) {
        const int kyx = i / (3 * CIN), r = i - kyx * (3 * CIN), kz = r / CIN, ci = r - kz * CIN;
        wl[i] = w[ci * 27 + kz * 9 + kyx];   // PyTorch layout (1,CIN,3,3,3)
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // weights staged (the copy waves take part)
    for (;;) {
        __syncthreads();
        int pl[TZ + 2];     // float offsets of the six input planes (this lane's channel quad)
#pragma unroll
        for (int k = 0; k < TZ + 2; ++k) {
            int slot = r0 + k;
            if (slot >= RING) slot -= RING;
            pl[k] = slot * SLOT_FLOATS + cq * PVOX * 4;
        }
        // sixteen accumulator chains (output plane x channel): a dependent v_fma_f32 chain advances every ~50 cycles,
        // with four chains the wave would spend 13 cycles per FMA
        float acc[TZ], acc4[TZ][4];
#pragma unroll
        for (int z = 0; z < TZ; ++z)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc4[z][c] = 0.f;
        if (!(m.abl & 2)) {
            // nine batches = taps (ky, kx): six plane reads + three weight reads, one batch ahead of the arithmetic;
            // fenced per batch (left alone the scheduler gathers the reads of a whole step)
            float4 t[2][TZ + 2], wk[2][3];
            auto fetch = [&](int slot, int kyx) {
                const int ky = kyx / 3, kx = kyx - ky * 3;
                const int vo = ((ly + ky) * XT + lx + kx) * 4;
#pragma unroll
                for (int kz = 0; kz < 3; ++kz) wk[slot][kz] = *reinterpret_cast<const float4 *>(wl + (kyx * 3 + kz) * CIN + cq * 4);
#pragma unroll
                for (int i = 0; i < TZ + 2; ++i) t[slot][i] = *reinterpret_cast<const float4 *>(lds + pl[i] + vo);
            };
            fetch(0, 0);
#pragma unroll
            for (int kyx = 0; kyx < 9; ++kyx) {
                if (kyx + 1 < 9) fetch((kyx + 1) & 1, kyx + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < TZ + 2; ++i) {
                    const float4 v = t[kyx & 1][i];
#pragma unroll
                    for (int z = 0; z < TZ; ++z) {
                        const int kz = i - z;
                        if (kz < 0 || kz > 2) continue;
                        acc4[z][0] = fmaf(v.x, wk[kyx & 1][kz].x, acc4[z][0]);
                        acc4[z][1] = fmaf(v.y, wk[kyx & 1][kz].y, acc4[z][1]);
                        acc4[z][2] = fmaf(v.z, wk[kyx & 1][kz].z, acc4[z][2]);
                        acc4[z][3] = fmaf(v.w, wk[kyx & 1][kz].w, acc4[z][3]);
                    }
                }
#pragma unroll
                for (int z = 0; z < TZ; ++z)
#pragma unroll
                    for (int c = 0; c < 4; ++c) asm volatile("" : "+v"(acc4[z][c]));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int z = 0; z < TZ; ++z) acc[z] = (acc4[z][0] + acc4[z][1]) + (acc4[z][2] + acc4[z][3]);
#pragma unroll
        for (int z = 0; z < TZ; ++z) acc[z] += __shfl_xor(acc[z], 32);
        const int ox = tx * TX + lx, oy = ty * TY + ly;
        if (ox < a.Wo && oy < a.Ho) {
#pragma unroll
            for (int zz = 0; zz < 2; ++zz) {     // lower half stores planes 0, 1 of the step, upper half 2, 3
                const int oz = s * TZ + cq * 2 + zz;
                if (oz >= a.Do) continue;
                float v = (cq ? acc[2 + zz] : acc[zz]) * sc + sh;
                if (a.relu) v = relu_nan(v);
                const int64_t o = (((int64_t)b * a.Do + oz) * a.Ho + oy) * a.Wo + ox;
                if (a.residual) v += a.residual[o];
                a.out[o] = v;
            }
        }
        if (s + 1 < s_end) {
            r0 = (r0 + TZ) % RING; ++s;
        } else {
            u += gridDim.x;
            if (u >= m.nunits) break;
            open_unit(u);
            r0 = (r0 + TZ + 2) % RING;
        }
    }
}

// ---------------------------------------------------------------------
// Weight packing: PyTorch layout -> A-fragment order
// packed[ch][tap][mt][lane][s], lane = (m = lane&15, kq = lane>>4),
// input channel = ch*CK + kq*KS + s.
struct PackArgs {
    const float *w;
    float *packed;
    int Cin, Cout, mode, ck, mt, ntaps;   // mode 0/1/2 conv, 3 deconv, 4 deconv x-parity-merged
};

__global__ __launch_bounds__(256) void conv3d_pack_kernel(PackArgs p, int64_t total) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int KS = p.ck / 4;
    int64_t t = i;
    const int s = (int)(t % KS); t /= KS;
    const int lane = (int)(t % 64); t /= 64;
    const int mt = (int)(t % p.mt); t /= p.mt;
    const int tap = (int)(t % p.ntaps); t /= p.ntaps;
    const int ch = (int)t;
    const int m = lane & 15, kq = lane >> 4;
    const int cin = ch * p.ck + kq * KS + s;
    float val = 0.0f;
    if (p.mode <= 2) {
        const int NKX = p.mode == 2 ? 4 : 3;
        const int kz = tap / (3 * NKX), ky = (tap / NKX) % 3, kxp = tap % NKX;
        if (p.mode == 2) {
            const int sft = m >> 3, co = m & 7, kx = kxp - sft;
            if (kx >= 0 && kx <= 2)
                val = p.w[((int64_t)co * p.Cin + cin) * 27 + kz * 9 + ky * 3 + kx];
        } else {
            const int co = mt * 16 + m;
            if (co < p.Cout) val = p.w[((int64_t)co * p.Cin + cin) * 27 + kz * 9 + ky * 3 + kxp];
        }
    } else {
        const bool pxm = p.mode == 4;
        int ct = 0, kz = -1, ky = -1, kx = -1;
        bool found = false;
        for (int pz = 0; pz < 2 && !found; ++pz)
            for (int py = 0; py < 2 && !found; ++py)
                for (int px = 0; px < (pxm ? 1 : 2) && !found; ++px)
                    for (int tz = 0; tz <= pz && !found; ++tz)
                        for (int ty = 0; ty <= py && !found; ++ty)
                            for (int tx = 0; tx <= (pxm ? 1 : px) && !found; ++tx) {
                                if (ct == tap) {
                                    kz = dc_k(pz, tz);
                                    ky = dc_k(py, ty);
                                    if (pxm) {
                                        const int prow = m >> 3;   // x parity of this MFMA row
                                        kx = prow == 0 ? (tx == 0 ? 1 : -1) : (tx == 0 ? 2 : 0);
                                    } else {
                                        kx = dc_k(px, tx);
                                    }
                                    found = true;
                                }
                                ++ct;
                            }
        const int co = pxm ? (m & 7) : mt * 16 + m;
        if (found && kx >= 0 && co < p.Cout)   // weight (Cin,Cout,3,3,3)
            val = p.w[((int64_t)cin * p.Cout + co) * 27 + kz * 9 + ky * 3 + kx];
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

template <class Cfg>
static CfgInfo dinfo_of() {
    return CfgInfo{Cfg::PXM ? 4 : 3, Cfg::CK, Cfg::MT, Cfg::TZ, Cfg::TY, 16, Cfg::NCT,
                   deconv3d_mfma_kernel<Cfg>};
}

static bool lookup(int transposed, int Cin, int Cout, int stride, CfgInfo &ci) {
#define MVS_CFG(cin, cout, mode, ck, tz, ty)                       \
    if (Cin == cin && Cout == cout) {                              \
        ci = info_of<ConvCfg<cin, cout, mode, ck, tz, ty>>();      \
        return true;                                               \
    }
#define MVS_DCFG(cin, cout, ck, tz, ty, pxm)                       \
    if (Cin == cin && Cout == cout) {                              \
        ci = dinfo_of<DeconvCfg<cin, cout, ck, tz, ty, pxm>>();    \
        return true;                                               \
    }
    if (transposed) {
        if (stride != 2) return false;
        // tiles sized so accumulators (classes x rows x m-tiles x 4) stay at 64 VGPRs:
        // two waves per SIMD hide the epilogue's residual loads and the staging latency
        MVS_DCFG(64, 32, 16, 2, 2, false)
        MVS_DCFG(32, 16, 16, 2, 4, false)
        MVS_DCFG(16, 8, 16, 2, 4, true)
        return false;
    }
    if (stride == 1) {
        // Cout = 8: shifted form (conv0 of MVSNet / of the CasMVSNet stages)
        if (Cin == 32 && Cout == 8) {   // conv0: 16 staging loads in flight, no A preload (measured best)
            ci = info_of<ConvCfg<32, 8, 2, 8, 4, 8, 16, 0>>();
            return true;
        }
        MVS_CFG(16, 8, 2, 8, 4, 8)
        MVS_CFG(8, 8, 2, 8, 4, 8)
        MVS_CFG(8, 16, 0, 8, 4, 8)
        MVS_CFG(8, 32, 0, 8, 4, 8)      // input gradient of conv0
        MVS_CFG(16, 16, 0, 16, 4, 8)
        MVS_CFG(32, 32, 0, 16, 4, 8)
        MVS_CFG(64, 64, 0, 16, 4, 8)
        MVS_CFG(16, 32, 0, 16, 4, 8)
        MVS_CFG(32, 64, 0, 16, 4, 8)
        MVS_CFG(64, 32, 0, 16, 4, 8)    // CVP-MVSNet's stride-1 transposed layer as a convolution
    } else if (stride == 2) {
        MVS_CFG(8, 16, 1, 8, 2, 4)
        MVS_CFG(16, 32, 1, 8, 2, 4)
        MVS_CFG(32, 64, 1, 8, 2, 4)
    }
#undef MVS_CFG
#undef MVS_DCFG
    return false;
}

static bool is_cout1(int transposed, int Cin, int Cout, int stride) {
    return !transposed && stride == 1 && Cout == 1 && (Cin == 8 || Cin == 16);
}

int conv3d_mfma_supported(int transposed, int Cin, int Cout, int stride) {
    CfgInfo ci;
    return (lookup(transposed, Cin, Cout, stride, ci) || is_cout1(transposed, Cin, Cout, stride)) ? 1 : 0;
}

int64_t conv3d_packed_floats(int transposed, int Cin, int Cout, int stride) {
    CfgInfo ci;
    if (is_cout1(transposed, Cin, Cout, stride)) return (int64_t)Cin * 27;  // used as is
    if (!lookup(transposed, Cin, Cout, stride, ci)) return 0;
    return (int64_t)(Cin / ci.ck) * ci.ntaps * ci.mt * 64 * (ci.ck / 4);
}

int conv3d_pack_launch(const float *weight, int transposed, int Cin, int Cout, int stride,
                       float *packed, hipStream_t st) {
    CfgInfo ci;
    if (is_cout1(transposed, Cin, Cout, stride)) {
        if (hipMemcpyAsync(packed, weight, sizeof(float) * (size_t)Cin * 27,
                           hipMemcpyDeviceToDevice, st) != hipSuccess)
            return check_launch("mvs_conv3d_pack_weights_f32(copy)");
        return MVS_OK;
    }
    if (!lookup(transposed, Cin, Cout, stride, ci)) {
        set_error("mvs_conv3d_pack_weights_f32: no fast-path configuration for %s Cin=%d Cout=%d stride=%d",
                  transposed ? "deconv" : "conv", Cin, Cout, stride);
        return MVS_EUNSUPPORTED;
    }
    PackArgs p{weight, packed, Cin, Cout, ci.mode, ci.ck, ci.mt, ci.ntaps};
    const int64_t total = conv3d_packed_floats(transposed, Cin, Cout, stride);
    hipLaunchKernelGGL(conv3d_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       p, total);
    return check_launch("mvs_conv3d_pack_weights_f32");
}

// out_absmax: NULL, or the absmax block the output's largest magnitude is max-ed into by the kernels that collect it in their
// epilogue (the convolutions with Cout > 1); *collected tells the caller whether the launched kernel was one of them
int conv3d_mfma_launch(const float *in, const float *packed, const float *scale,
                       const float *shift, const float *residual, int relu, int transposed, int B,
                       int Cin, int Cout, int D, int H, int W, int stride, int in_c8, float *out,
                       hipStream_t st, unsigned *out_absmax, bool *collected) {
    ConvArgs a;
    const bool collects = !transposed && !is_cout1(transposed, Cin, Cout, stride);
    a.out_absmax = collects ? out_absmax : nullptr;
    if (collected) *collected = collects && out_absmax;
    a.in = in; a.wpk = packed; a.scale = scale; a.shift = shift; a.residual = residual;
    a.out = out;
    a.B = B; a.D = D; a.H = H; a.W = W;
    a.relu = relu;
    a.in_c8 = in_c8;
    a.res_up2 = 0;
    {
        const char *ys = getenv("MVS_CONV_YSTRIP");   // tuning; default: strips of 4 tile rows
        a.ystrip = ys ? atoi(ys) : 4;
    }

    if (in_c8 && (transposed || is_cout1(transposed, Cin, Cout, stride) || Cin % 8)) {
        set_error("mvs_conv3d_f32: the 8-channel-blocked input layout is only taken by the conv kernels");
        return MVS_EUNSUPPORTED;
    }
    if (is_cout1(transposed, Cin, Cout, stride)) {
        a.run_flag = conv_run_flag();
        a.Do = D; a.Ho = H; a.Wo = W;
        a.tiles_x = (W + 31) / 32; a.tiles_y = (H + 7) / 8; a.tiles_z = (D + 3) / 4;
        const int64_t nblk = (int64_t)B * a.tiles_x * a.tiles_y * a.tiles_z;
        if (nblk <= 0 || nblk > 0x7fffffffLL) return bare_error(MVS_EINVAL, __func__, __LINE__);
        static const bool use_march = !(getenv("MVS_PROB_MARCH") && atoi(getenv("MVS_PROB_MARCH")) == 0);
        if (Cin == 8 && use_march && (int64_t)D * H * W * Cin * 4 < 0xffffff00LL) {
            // depth segments: enough units for ~4 per CU, at least 8 steps each
            MarchArgs m;
            m.c = a;
            m.ncols = a.tiles_x * a.tiles_y * B;
            const int n_cu = device_cu_count();
            int nseg = (4 * n_cu + m.ncols - 1) / m.ncols;
            nseg = nseg < 1 ? 1 : nseg;
            if (nseg > (a.tiles_z + 7) / 8) nseg = (a.tiles_z + 7) / 8;
            m.sps = (a.tiles_z + nseg - 1) / nseg;
            m.nseg = (a.tiles_z + m.sps - 1) / m.sps;
            m.nunits = m.ncols * m.nseg;
#ifdef MVS_TUNING
            static const int abl = getenv("MVS_PROB_ABL") ? atoi(getenv("MVS_PROB_ABL")) : 0;
#else
            constexpr int abl = 0;
#endif
            m.abl = abl;
            hipLaunchKernelGGL(conv3d_cout1_march_kernel, dim3((unsigned)(m.nunits < n_cu ? m.nunits : n_cu)), dim3(768), 0, st,
                               m, packed);
            return check_launch("mvs_conv3d_f32(cout1, marching)");
        }
        if (Cin == 8)
            hipLaunchKernelGGL((conv3d_cout1_kernel<8, 8>), dim3((unsigned)nblk), dim3(512), 0, st, a, packed);
        else
            hipLaunchKernelGGL((conv3d_cout1_kernel<16, 8>), dim3((unsigned)nblk), dim3(512), 0, st, a, packed);
        return check_launch("mvs_conv3d_f32(cout1)");
    }
    CfgInfo ci;
    if (!lookup(transposed, Cin, Cout, stride, ci)) {
        set_error("mvs_conv3d_f32(mfma): no configuration for %s Cin=%d Cout=%d stride=%d",
                  transposed ? "deconv" : "conv", Cin, Cout, stride);
        return MVS_EUNSUPPORTED;
    }
    if (transposed) {
        a.Do = D * 2; a.Ho = H * 2; a.Wo = W * 2;
        a.tiles_x = (W + 15) / 16;          // tiles on the INPUT grid
        a.tiles_y = (H + ci.ty - 1) / ci.ty;
        a.tiles_z = (D + ci.tz - 1) / ci.tz;
    } else {
        a.Do = (D - 1) / stride + 1; a.Ho = (H - 1) / stride + 1; a.Wo = (W - 1) / stride + 1;
        a.tiles_x = (a.Wo + ci.xout - 1) / ci.xout;
        a.tiles_y = (a.Ho + ci.ty - 1) / ci.ty;
        a.tiles_z = (a.Do + ci.tz - 1) / ci.tz;
        // small volumes (the bottom of the U-Net: 24x37x50 at config 2 is 120 tiles for 256
        // CUs): a quarter-size tile of the same layer -- same packed weights -- fills the chip;
        // up to a few tiles per CU and wave slot it also wins on the tail (conv2: 0.43 -> 0.40 ms)
        if (stride == 1 && ci.mode == 0 && ci.tz == 4 && ci.ty == 8 &&
            (int64_t)B * a.tiles_x * a.tiles_y * a.tiles_z < 8192) {
            CfgInfo small;
            bool have = true;
            if (Cin == 64 && Cout == 64) small = info_of<ConvCfg<64, 64, 0, 16, 2, 4>>();
            else if (Cin == 32 && Cout == 32) small = info_of<ConvCfg<32, 32, 0, 16, 2, 4>>();
            else if (Cin == 16 && Cout == 16) small = info_of<ConvCfg<16, 16, 0, 16, 2, 4>>();
            else have = false;
            if (have) {
                ci = small;
                a.tiles_y = (a.Ho + ci.ty - 1) / ci.ty;
                a.tiles_z = (a.Do + ci.tz - 1) / ci.tz;
            }
        }
    }
    const int64_t nblk = (int64_t)B * a.tiles_x * a.tiles_y * a.tiles_z;
    if (nblk <= 0 || nblk > 0x7fffffffLL) {
        set_error("mvs_conv3d_f32(mfma): bad grid");
        return MVS_EINVAL;
    }
    // Persistent DMA-fed kernel: the Cout = 8 stride-1 layers on the 8-channel-blocked
    // volume (conv0 and the cascade's first layers) and the stride-2 layers whose weights
    // fit in LDS beside two halo buffers (conv1 8->16, conv3 16->32).
    // (MVS_CONV0_VARIANT=10 selects the per-tile kernel instead: tuning / A-B testing)
    {
        const bool c8_class = !transposed && Cout == 8 && stride == 1 && in_c8 &&
                              (Cin == 32 || Cin == 16 || Cin == 8);
        const bool s2_class = !transposed && stride == 2 && !in_c8 &&
                              ((Cin == 8 && Cout == 16) || (Cin == 16 && Cout == 32));
        const char *var = getenv("MVS_CONV0_VARIANT");
        const int v = var ? atoi(var) : 0;
        // (its copies address a tile's halo planes with 32-bit byte offsets)
        const bool window_ok = (int64_t)9 * H * W * Cin * 4 < 0xffffff00LL;
        if ((c8_class || s2_class) && window_ok && (v == 0 || v >= 20)) {
            const int n_cu = device_cu_count();
            // Cin = 8 has ONE 8-channel chunk per tile: a 4 x 8-row tile doubles the MFMA work behind each
            // tile's fixed costs (0.68 -> 0.63 ms on the cascade's finest stage); for Cin = 16 it no longer fits LDS
            const bool tall = c8_class && Cin == 8;
            if (c8_class) {
                a.tiles_x = (a.Wo + 31) / 32; a.tiles_y = (a.Ho + (tall ? 7 : 3)) / (tall ? 8 : 4); a.tiles_z = (a.Do + 3) / 4;
            } else {
                a.tiles_x = (a.Wo + 15) / 16; a.tiles_y = (a.Ho + 3) / 4; a.tiles_z = (a.Do + 1) / 2;
            }
            const int64_t nt = (int64_t)B * a.tiles_x * a.tiles_y * a.tiles_z;
            if (nt <= 0 || nt > 0x7fffffffLL) return bare_error(MVS_EINVAL, __func__, __LINE__);
            if (!getenv("MVS_CONV_YSTRIP")) a.ystrip = 8;
            const dim3 grid((unsigned)(nt < n_cu ? nt : n_cu)), blk(512);
            const int ntl = (int)nt;
            if (s2_class && Cin == 8)
                hipLaunchKernelGGL((conv3d_c8_persistent_kernel<PersistCfg<8, 16, 1, 2, 4>>), grid, blk, 0, st, a, ntl);
            else if (s2_class)
                hipLaunchKernelGGL((conv3d_c8_persistent_kernel<PersistCfg<16, 32, 1, 2, 4>>), grid, blk, 0, st, a, ntl);
            else if (Cin == 32 && v == 28)
                hipLaunchKernelGGL((conv3d_c8_persistent_kernel<PersistCfg<32>, 8>), grid, blk, 0, st, a, ntl);
            else if (Cin == 32 && v == 36)
                hipLaunchKernelGGL((conv3d_c8_persistent_kernel<PersistCfg<32>, 16>), grid, blk, 0, st, a, ntl);
            else if (Cin == 32)
                hipLaunchKernelGGL((conv3d_c8_persistent_kernel<PersistCfg<32>>), grid, blk, 0, st, a, ntl);
            else if (Cin == 16)
                hipLaunchKernelGGL((conv3d_c8_persistent_kernel<PersistCfg<16>>), grid, blk, 0, st, a, ntl);
            else if (tall)
                hipLaunchKernelGGL((conv3d_c8_persistent_kernel<PersistCfg<8, 8, 2, 4, 8>>), grid, blk, 0, st, a, ntl);
            else
                hipLaunchKernelGGL((conv3d_c8_persistent_kernel<PersistCfg<8>>), grid, blk, 0, st, a, ntl);
            return check_launch("mvs_conv3d_f32(mfma, persistent)");
        }
    }
    hipLaunchKernelGGL(ci.kernel, dim3((unsigned)nblk), dim3(256), 0, st, a);
    return check_launch("mvs_conv3d_f32(mfma)");
}

}  // namespace mvs
