// Weight gradient of the 3x3x3 convolutions of CostRegNet (training path, BASELINE config 5)
// on the fp32 matrix cores.
//
//   dW[co][ci][kz,ky,kx] = sum over output voxels o of  g[o][co] * x[o*s + k - 1][ci]
//
// (x = layer input, g = gradient of the layer output, both channels-last, s = stride; the
// transposed layers' weight gradient is the same sum with the roles of x and g swapped --
// see mvs_conv3d_wgrad_f32 in mvs_hip.h).  It replaces 27 strided-view GEMMs per layer in
// torch (each a copy of the input + a split-K bmm): 44 of the 68 ms of a training step.
//
// GEMM orientation: the reduction dimension is the VOXELS.  One MFMA = D[16 co x 16 ci] +=
// A[16 co x 4 voxels] * B[4 voxels x 16 ci]; A is a fragment of g, B a fragment of x at the
// tap's offset, 4 consecutive output voxels along x per MFMA.  Accumulators are the weight
// gradient itself, so they stay in registers for the whole kernel: a workgroup is PERSISTENT,
// owns one 16-channel slice of ci and a stream of output tiles, its 4 waves split the 27
// taps, and the result goes out once.  With a workspace each workgroup stores its partial dW
// in accumulator-register order (fully coalesced) and wgrad_reduce_kernel sums the
// workgroups; without one the partials meet in grad_weight through atomic adds, which for
// the small layers is the whole cost (512 workgroups x 27.6 k adds on the same 55 k
// addresses: 0.72 of conv5's 0.75 ms).
// Tiles (g: TZ x TY x 16 voxels, x: its halo) are staged through LDS with buffer loads
// (out-of-volume voxels load zeros = the convolution's padding and the tile overhang).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mvs_common.h"

namespace mvs {

typedef float wg_f32x4 __attribute__((ext_vector_type(4)));

// num_records of a buffer resource: the bytes of the window, capped below 4 GB (the out-of-range offset the loaders use)
__device__ __forceinline__ int rsrc_bytes(int64_t n) { return (int)(unsigned)(n < 0xffffff00LL ? n : 0xffffff00LL); }

struct WgradArgs {
    const float *x, *g;
    float *gw;
    float *partial;       // [workgroup][wave][tap of the wave][m][j][lane], or null
    int B, Cin, Cout;
    int D, H, W;          // input (x) grid
    int Do, Ho, Wo;       // output (g) grid
    int tiles_x, tiles_y, tiles_z;
    int x_c8 = 0;         // x is [B,D,H,Cin/8,W,8] (8-channel blocked: the variance volume as conv0's bf16 kernel reads it)
};

template <int COUT_T, int CK, int S>
struct WgradCfg {
    // (8-channel slices at stride 2 -- conv1 and the last decoder layer, full-resolution inputs -- take 4 rows: their x halo is
    // 32 bytes a voxel, and 32-voxel tiles were mostly per-tile overhead)
    static constexpr int TZ = (S == 1) ? 2 : 1, TY = (S == 1 || CK == 8) ? 4 : 2;
    static constexpr int ROWS = TZ * TY, NOUT = ROWS * 16;
    static constexpr int XT = 15 * S + 3, YT = (TY - 1) * S + 3, ZT = (TZ - 1) * S + 3;
    static constexpr int NVOX = ZT * YT * XT;
    static constexpr int MT = COUT_T / 16;
    // voxel strides (floats) chosen so the two 16-lane runs of a ds_read_b32 half-wave land
    // on different banks: consecutive voxels of a fragment are GP apart in g, S*XP in x
    static constexpr int GP = (COUT_T % 32 == 0) ? COUT_T + 16 : COUT_T;
    // CK = 8: 8 floats per voxel, lanes 8..15 of a B fragment read the next voxel's channels -- columns of the product nothing
    // stores (a 4-voxel fragment is then 64 consecutive floats at stride 2)
    static constexpr int QX = CK / 4;                     // 16-byte pieces per x voxel
    static constexpr int XP = CK == 8 ? 8 : ((S == 1) ? 16 : 24);
    static constexpr int LDS_FLOATS = NOUT * GP + NVOX * XP + 8;   // (+8: the last voxel's lanes 8..15 at CK = 8)
    static constexpr int TAPS_PER_WAVE = 7;
};

template <int COUT_T, int CK, int S>
__global__ __launch_bounds__(256, 2) void conv3d_wgrad_kernel(WgradArgs a, int ntiles, int ncc) {
    using C = WgradCfg<COUT_T, CK, S>;
    constexpr int MT = C::MT, GP = C::GP, XP = C::XP, XT = C::XT, YT = C::YT;
    constexpr int TPW = C::TAPS_PER_WAVE;
    __shared__ __attribute__((aligned(16))) float lds[C::LDS_FLOATS];
    float *gl = lds, *xl = lds + C::NOUT * GP;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, kv = lane >> 4;
    const int cc = blockIdx.x % ncc;                      // this workgroup's ci slice
    const int t0 = blockIdx.x / ncc, tstep = gridDim.x / ncc;
    const int tap0 = wv * TPW, ntap = min(TPW, 27 - tap0);

    wg_f32x4 acc[TPW][MT];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[t][m] = (wg_f32x4){0.f, 0.f, 0.f, 0.f};

    if (t0 < ntiles && blockIdx.x < tstep * ncc) {
        const int64_t gplane = (int64_t)a.Ho * a.Wo * a.Cout, xplane = (int64_t)a.H * a.W * a.Cin;
        // A tile's global loads are issued BEFORE the previous tile's K loop and land in registers (gr / xr) while the matrix
        // pipe works; they go to LDS between the two barriers at the top of the next round.  (Loading, waiting and computing
        // in turn left a workgroup idle for one memory round trip per tile: 30 tiles x ~5 us of the stride-2 layers' 190 us.)
        constexpr int QG = COUT_T / 4;      // float4 pieces per g voxel
        constexpr int QX = C::QX;
        constexpr int GI = (C::NOUT * QG + 255) / 256, XI = (C::NVOX * QX + 255) / 256;
        float4 gr[GI], xr[XI];
        auto fetch = [&](int t) {
            int bid = t;
            const int tx = bid % a.tiles_x; bid /= a.tiles_x;
            const int ty = bid % a.tiles_y; bid /= a.tiles_y;
            const int tz = bid % a.tiles_z;
            const int b = bid / a.tiles_z;
            const int ox0 = tx * 16, oy0 = ty * C::TY, oz0 = tz * C::TZ;
            const int ix0 = ox0 * S - 1, iy0 = oy0 * S - 1, iz0 = oz0 * S - 1;
            {   // g tile: [row][x][Cout]; zero beyond Cout and the volume
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float *>(a.g + ((int64_t)b * a.Do + oz0) * gplane), 0,
                    rsrc_bytes((int64_t)C::TZ * gplane * 4), 0x00020000);
#pragma unroll
                for (int i = 0; i < GI; ++i) {
                    const int e = tid + i * 256;
                    const int q = e % QG, v = e / QG;
                    const int x = v & 15, row = v >> 4;
                    const int oz = row / C::TY, oy = oy0 + row % C::TY, ox = ox0 + x;
                    const bool ok = e < C::NOUT * QG && oz0 + oz < a.Do && oy < a.Ho && ox < a.Wo && q * 4 < a.Cout;
                    const unsigned off = ok ? (unsigned)(((int64_t)oz * gplane + ((int64_t)oy * a.Wo + ox) * a.Cout + q * 4) * 4)
                                            : 0xffffff00u;
                    if (a.Cout == 1) {   // the `prob` layer: one channel per voxel, no 16-byte pieces
                        gr[i] = make_float4(__uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0)), 0.f, 0.f, 0.f);
                    } else {
                        const auto val = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
                        gr[i] = make_float4(__uint_as_float(val[0]), __uint_as_float(val[1]), __uint_as_float(val[2]),
                                            __uint_as_float(val[3]));
                    }
                }
            }
            {   // x halo: channels cc*CK .. +CK of every halo voxel
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float *>(a.x + ((int64_t)b * a.D + iz0) * xplane), 0,
                    rsrc_bytes((int64_t)C::ZT * xplane * 4), 0x00020000);
#pragma unroll
                for (int i = 0; i < XI; ++i) {
                    const int e = tid + i * 256;
                    const int q = e % QX, v = e / QX;
                    const int lx = v % XT, t2 = v / XT, ly = t2 % YT, lz = t2 / YT;
                    const int gx = ix0 + lx, gy = iy0 + ly, gz = iz0 + lz;
                    const bool ok = e < C::NVOX * QX && (unsigned)gx < (unsigned)a.W && (unsigned)gy < (unsigned)a.H &&
                                    (unsigned)gz < (unsigned)a.D;
                    const unsigned off = ok ? (unsigned)(((int64_t)lz * xplane + ((int64_t)gy * a.W + gx) * a.Cin +
                                                          cc * CK + q * 4) * 4)
                                            : 0xffffff00u;
                    const auto val = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
                    xr[i] = make_float4(__uint_as_float(val[0]), __uint_as_float(val[1]), __uint_as_float(val[2]),
                                        __uint_as_float(val[3]));
                }
            }
        };
        auto stash = [&]() {   // registers -> gl[voxel * GP + co], xl[voxel * XP + ci]
#pragma unroll
            for (int i = 0; i < GI; ++i) {
                const int e = tid + i * 256;
                if (e < C::NOUT * QG) *reinterpret_cast<float4 *>(gl + (e / QG) * GP + (e % QG) * 4) = gr[i];
            }
#pragma unroll
            for (int i = 0; i < XI; ++i) {
                const int e = tid + i * 256;
                if (e < C::NVOX * QX) *reinterpret_cast<float4 *>(xl + (e / QX) * XP + (e % QX) * 4) = xr[i];
            }
        };
        fetch(t0);
        for (int t = t0; t < ntiles; t += tstep) {
            __syncthreads();   // every wave is done with the previous tile
            stash();
            __syncthreads();
            if (t + tstep < ntiles) fetch(t + tstep);
            // ---- K loop: 4 output voxels along x per MFMA
#pragma unroll 1
            for (int row = 0; row < C::ROWS; ++row) {
                const int rz = row / C::TY, ry = row % C::TY;
#pragma unroll
                for (int xs = 0; xs < 4; ++xs) {
                    float af[MT];
#pragma unroll
                    for (int m = 0; m < MT; ++m) af[m] = gl[(row * 16 + xs * 4 + kv) * GP + m * 16 + c];
                    const float *xb = xl + (((rz * S) * YT + ry * S) * XT + (xs * 4 + kv) * S) * XP + c;
#pragma unroll
                    for (int t2 = 0; t2 < TPW; ++t2) {
                        const int tap = min(tap0 + t2, 26);          // wave-uniform; the 28th slot repeats tap 26 (never stored)
                        const int kz = tap / 9, ky = (tap / 3) % 3, kx = tap % 3;
                        const float bf = xb[((kz * YT + ky) * XT + kx) * XP];
#pragma unroll
                        for (int m = 0; m < MT; ++m)
                            acc[t2][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m], bf, acc[t2][m], 0, 0, 0);
                    }
                }
            }
        }
    }
    // ---- flush: D lane (n = ci, q) holds co = m*16 + 4q + j
    if (a.partial) {
        float *dst = a.partial + ((size_t)blockIdx.x * 4 + wv) * (TPW * MT * 256) + lane;
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int j = 0; j < 4; ++j) dst[((t * MT + m) * 4 + j) * 64] = acc[t][m][j];
        return;
    }
    const int ci = cc * CK + c;
    if (c < CK && ci < a.Cin) {
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            if (t >= ntap) break;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int co = m * 16 + kv * 4 + j;
                    if (co < a.Cout) unsafeAtomicAdd(a.gw + ((int64_t)co * a.Cin + ci) * 27 + tap0 + t, acc[t][m][j]);
                }
        }
    }
}

// Second stage: grad_weight += sum over the workgroups' partials.  One thread per slot of a
// workgroup's region (x the ci slices), so every load is coalesced; SPLIT thread groups share
// the workgroups of a slot and meet in grad_weight with one atomic each.
template <int MT, int CK>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ partial, int streams, int ncc,
                                                           int Cin, int Cout, float *__restrict__ gw) {
    constexpr int TPW = 7, PB = 4 * TPW * MT * 256;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= ncc * PB) return;
    const int cc = e / PB, r = e % PB;
    const int lane = r & 63, j = (r >> 6) & 3, m = (r >> 8) % MT, t = (r / (256 * MT)) % TPW, wv = r / (256 * MT * TPW);
    const int c = lane & 15, co = m * 16 + (lane >> 4) * 4 + j, ci = cc * CK + c, tap = wv * TPW + t;
    if (c >= CK || ci >= Cin || co >= Cout || tap >= 27) return;
    const int s0 = blockIdx.y, sstep = gridDim.y;
    const float *p = partial + (size_t)cc * PB + r;
    const size_t ps = (size_t)ncc * PB;
    float u0 = 0.f, u1 = 0.f, u2 = 0.f, u3 = 0.f;      // four loads in flight (one chain of `streams` loads was 40 us for 60 streams)
    int s = s0;
    for (; s + 3 * sstep < streams; s += 4 * sstep) {
        u0 += p[(size_t)s * ps]; u1 += p[(size_t)(s + sstep) * ps];
        u2 += p[(size_t)(s + 2 * sstep) * ps]; u3 += p[(size_t)(s + 3 * sstep) * ps];
    }
    for (; s < streams; s += sstep) u0 += p[(size_t)s * ps];
    const float sum = (u0 + u1) + (u2 + u3);
    float *dst = gw + ((int64_t)co * Cin + ci) * 27 + tap;
    if (sstep == 1) *dst += sum; else unsafeAtomicAdd(dst, sum);
}

// A stream = one workgroup's run of tiles; it ends in a partial dW of 28 MT KB that the reduce kernel reads back.  The small
// layers have fewer tiles than the chip has workgroup slots: one tile per workgroup made the partials (55 MB for 32 -> 64
// stride 2) cost more than the tiles, so a stream takes at least 4 stride-2 / 2 stride-1 tiles.
static int wgrad_streams(int ntiles, int ncc, int stride) {
    int streams = (2 * device_cu_count()) / ncc;   // tile streams per ci slice
    const int per = stride == 2 ? 4 : 2;
    const int few = (ntiles + per - 1) / per;
    if (streams > few) streams = few;
    return streams < 1 ? 1 : streams;
}

template <int COUT_T, int CK, int S>
static int launch_wgrad(WgradArgs a, int ntiles, void *workspace, size_t workspace_bytes, hipStream_t st) {
    constexpr int MT = COUT_T / 16, PB = 4 * 7 * MT * 256;
    const int ncc = (a.Cin + CK - 1) / CK;
    const int streams = wgrad_streams(ntiles, ncc, S);
    const bool two_stage = workspace && workspace_bytes >= (size_t)streams * ncc * PB * sizeof(float);
    a.partial = two_stage ? static_cast<float *>(workspace) : nullptr;
    hipLaunchKernelGGL((conv3d_wgrad_kernel<COUT_T, CK, S>), dim3((unsigned)(streams * ncc)), dim3(256), 0, st,
                       a, ntiles, ncc);
    if (two_stage) {
        // thread groups per slot: 16 fill the chip (28 MT ncc blocks each) and keep a thread's chain of loads short
        const int split = streams >= 256 ? 16 : (streams >= 32 ? 8 : (streams >= 8 ? 4 : 1));
        hipLaunchKernelGGL((wgrad_reduce_kernel<MT, CK>), dim3((unsigned)((ncc * PB + 255) / 256), (unsigned)split),
                           dim3(256), 0, st, a.partial, streams, ncc, a.Cin, a.Cout, a.gw);
    }
    return check_launch("mvs_conv3d_wgrad_f32");
}

// ---------------------------------------------------------------------
// Stride-1 layers with few output channels (conv0: 32 -> 8, `prob`: 8 -> 1): anchored on the
// INPUT voxel,
//   dW[co][ci][tap] = sum over input voxels v of  g[v - (tap - 1)][co] * x[v][ci],
// every row of the MFMA's A operand may look at a different tap: A[(h, co)][k] =
// g[v_k - off(tap_h)][co], B[k][ci] = x[v_k][ci], so one MFMA carries 16 / Cout taps instead
// of one padded with zeros (27 -> 14 MFMAs per 4 voxels for Cout = 8, 27 -> 2 for Cout = 1).
// The halo moves to g (8 or 1 channels, cheap) and the wide operand x is staged without a halo
// and read ONCE: a workgroup covers all of Cin.  Units = (tap slot) x (16-channel ci tile); a
// wave owns NU of them (its accumulators) and, when there are fewer than 4 NU units, a share
// of the tile's rows.  x and g tiles arrive by LDS-DMA (x granules XOR-swizzled on the source
// side so the B reads of 4 voxels x 16 channels hit 64 banks).  Needs the workspace.
template <int CIN, int COUT>
struct WgradXCfg {
    static constexpr int TZ = 2, TY = 4, ROWS = TZ * TY, NOUT = ROWS * 16;
    static constexpr int TPM = 16 / COUT;                  // taps per MFMA
    static constexpr int NP = (27 + TPM - 1) / TPM;        // tap slots
    static constexpr int NT = CIN >= 16 ? CIN / 16 : 1;    // ci tiles
    static constexpr int NU = TPM == 2 ? 7 : 2;            // units per wave
    static constexpr int UNITS = NP * NT, NG = UNITS / NU, NRS = 4 / NG;
    static_assert(NG * NU == UNITS && (NG == 1 || NG == 2 || NG == 4), "unit split");
    static constexpr int Q = CIN / 4;                      // 16-byte granules per x voxel
    static constexpr int GZT = TZ + 2, GYT = TY + 2, GXT = 18, GVOX = GZT * GYT * GXT;
    static constexpr int X_FLOATS = NOUT * CIN;
    static constexpr int G_INSTR = (GVOX * COUT / 4 + 255) / 256;          // DMA instructions per wave (Cout = 8)
    static constexpr int G_FLOATS = COUT == 1 ? ((GVOX + 3) & ~3) : G_INSTR * 1024;
    static constexpr int PB = 4 * NU * 256;                // floats of one workgroup's partials
};

template <int CIN, int COUT>
__global__ __launch_bounds__(256, 2) void conv3d_wgrad_xanchor_kernel(WgradArgs a, int ntiles) {
    using C = WgradXCfg<CIN, COUT>;
    constexpr int NU = C::NU, Q = C::Q, GYT = C::GYT, GXT = C::GXT;
    // two tile buffers: the LDS-DMA of tile t + 1 runs under the K loop of tile t (one barrier per tile; loading, waiting
    // and computing in turn left the matrix pipe idle for a memory round trip per 128-voxel tile)
    constexpr int BUF = C::X_FLOATS + C::G_FLOATS;
    __shared__ __attribute__((aligned(16))) float lds[2 * BUF];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kv = lane >> 4;
    const int grp = wv % C::NG, rs = wv / C::NG;
    const int nt = (grp * NU) / C::NP;

    // A operand: row m = lane & 15 -> (tap slot half h, co); offset of g[v - off(tap)][co] from g[v]'s halo slot
    int aoff[NU];
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        const int p = (grp * NU + i) % C::NP;
        const int tap = min(p * C::TPM + n / COUT, 26);   // slot past the 27th tap: computed, never read back
        const int kz = tap / 9, ky = (tap / 3) % 3, kx = tap % 3;
        aoff[i] = ((((2 - kz) * GYT + (2 - ky)) * GXT + (2 - kx)) + kv) * COUT + n % COUT;
    }
    // B operand: x[v][ch]; granule (v, ch / 4) sits in slot v * Q + ((ch / 4) ^ swz(v))
    const int ch = nt * 16 + (CIN >= 16 ? n : (n & (CIN - 1)));
    const int boff = (((ch >> 2) ^ (CIN == 32 ? (kv >> 1) * 4 : 0)) + kv * Q) * 4 + (ch & 3);

    wg_f32x4 acc[NU];
#pragma unroll
    for (int i = 0; i < NU; ++i) acc[i] = (wg_f32x4){0.f, 0.f, 0.f, 0.f};

    const int64_t plane = (int64_t)a.H * a.W;
    // Tile t + 1 is loaded into registers (xr / gr / g1) before the K loop of tile t and written to the other LDS buffer
    // after it: one barrier per tile.  (As LDS-DMA the 8 copies per wave and tile cost ~250 cycles of issue each, a quarter
    // of the tile's 7168 MFMA cycles; loading, waiting and computing in turn cost a memory round trip per tile on top.)
    constexpr int XI = C::NOUT * Q / 256;       // 16-byte granules of the x tile per thread
    constexpr int G1 = (C::GVOX + 255) / 256;   // Cout = 1: one float per halo voxel
    constexpr int GI = COUT == 1 ? 1 : C::G_INSTR;
    float4 xr[XI], gr[GI];
    float g1[G1];
    auto fetch = [&](int t) {
        int bid = t;
        const int tx = bid % a.tiles_x; bid /= a.tiles_x;
        const int ty = bid % a.tiles_y; bid /= a.tiles_y;
        const int tz = bid % a.tiles_z;
        const int b = bid / a.tiles_z;
        const int x0 = tx * 16, y0 = ty * C::TY, z0 = tz * C::TZ;
        {   // x tile, no halo: slot s <- granule (v = s / Q, q = (s % Q) ^ swz(v))
            const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float *>(a.x + ((int64_t)b * a.D + z0) * plane * CIN), 0,
                rsrc_bytes((int64_t)C::TZ * plane * CIN * 4), 0x00020000);
#pragma unroll
            for (int it = 0; it < XI; ++it) {
                const int s = (it * 4 + wv) * 64 + lane;
                const int v = s / Q, q = (s % Q) ^ (CIN == 32 ? ((v >> 1) & 1) * 4 : 0);
                const int x = v & 15, row = v >> 4, rz = row / C::TY, ry = row % C::TY;
                const bool ok = z0 + rz < a.D && y0 + ry < a.H && x0 + x < a.W;
                const unsigned off = !ok ? 0xffffff00u
                    : a.x_c8 ? (unsigned)(((((int64_t)rz * a.H + (y0 + ry)) * (CIN / 8) + (q >> 1)) * a.W + x0 + x) * 32 + (q & 1) * 16)
                             : (unsigned)(((rz * plane + (int64_t)(y0 + ry) * a.W + x0 + x) * CIN + q * 4) * 4);
                const auto val = __builtin_amdgcn_raw_buffer_load_b128(rsx, off, 0, 0);
                xr[it] = make_float4(__uint_as_float(val[0]), __uint_as_float(val[1]), __uint_as_float(val[2]),
                                     __uint_as_float(val[3]));
            }
        }
        if (COUT == 1) {   // g halo, one float per voxel
            const __amdgpu_buffer_rsrc_t rsg = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float *>(a.g + ((int64_t)b * a.D + z0 - 1) * plane), 0,
                rsrc_bytes((int64_t)C::GZT * plane * 4), 0x00020000);
#pragma unroll
            for (int i = 0; i < G1; ++i) {
                const int e = tid + i * 256;
                const int lx = e % GXT, t2 = e / GXT, ly = t2 % GYT, lz = t2 / GYT;
                const int gx = x0 - 1 + lx, gy = y0 - 1 + ly, gz = z0 - 1 + lz;
                const bool ok = e < C::GVOX && (unsigned)gx < (unsigned)a.W && (unsigned)gy < (unsigned)a.H &&
                                (unsigned)gz < (unsigned)a.D;
                const unsigned off = ok ? (unsigned)((lz * plane + (int64_t)gy * a.W + gx) * 4) : 0xffffff00u;
                g1[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsg, off, 0, 0));
            }
        } else {           // g halo: [halo voxel][Cout], 16-byte granules
            constexpr int GQ = COUT / 4;
            const __amdgpu_buffer_rsrc_t rsg = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float *>(a.g + ((int64_t)b * a.D + z0 - 1) * plane * COUT), 0,
                rsrc_bytes((int64_t)C::GZT * plane * COUT * 4), 0x00020000);
#pragma unroll
            for (int it = 0; it < GI; ++it) {
                const int s = (it * 4 + wv) * 64 + lane;
                const int hv = s / GQ, q = s % GQ;
                const int lx = hv % GXT, t2 = hv / GXT, ly = t2 % GYT, lz = t2 / GYT;
                const int gx = x0 - 1 + lx, gy = y0 - 1 + ly, gz = z0 - 1 + lz;
                const bool ok = hv < C::GVOX && (unsigned)gx < (unsigned)a.W && (unsigned)gy < (unsigned)a.H &&
                                (unsigned)gz < (unsigned)a.D;
                const unsigned off = ok ? (unsigned)(((lz * plane + (int64_t)gy * a.W + gx) * COUT + q * 4) * 4) : 0xffffff00u;
                const auto val = __builtin_amdgcn_raw_buffer_load_b128(rsg, off, 0, 0);
                gr[it] = make_float4(__uint_as_float(val[0]), __uint_as_float(val[1]), __uint_as_float(val[2]),
                                     __uint_as_float(val[3]));
            }
        }
    };
    auto stash = [&](int buf) {   // registers -> LDS, slot order (the layouts the LDS-DMA form wrote)
        float *xb = lds + buf * BUF, *gb = xb + C::X_FLOATS;
#pragma unroll
        for (int it = 0; it < XI; ++it) *reinterpret_cast<float4 *>(xb + ((it * 4 + wv) * 64 + lane) * 4) = xr[it];
        if (COUT == 1) {
#pragma unroll
            for (int i = 0; i < G1; ++i)
                if (tid + i * 256 < C::GVOX) gb[tid + i * 256] = g1[i];
        } else {
#pragma unroll
            for (int it = 0; it < GI; ++it) *reinterpret_cast<float4 *>(gb + ((it * 4 + wv) * 64 + lane) * 4) = gr[it];
        }
    };
    if ((int)blockIdx.x < ntiles) {
        fetch(blockIdx.x);
        stash(0);
    }
    __syncthreads();
    int buf = 0;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x, buf ^= 1) {
        const float *xl = lds + buf * BUF, *gl = xl + C::X_FLOATS;
        const bool more = t + (int)gridDim.x < ntiles;
        if (more) fetch(t + gridDim.x);
        // ---- K loop over this wave's rows: 4 input voxels along x per MFMA
        constexpr int RPW = C::ROWS / C::NRS;
#pragma unroll 1
        for (int r = 0; r < RPW; ++r) {
            const int row = rs * RPW + r, rz = row / C::TY, ry = row % C::TY;
#pragma unroll
            for (int xs = 0; xs < 4; ++xs) {
                const float bf = xl[(row * 16 + xs * 4) * CIN + boff];
                const float *gb = gl + ((rz * GYT + ry) * GXT + xs * 4) * COUT;
#pragma unroll
                for (int i = 0; i < NU; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(gb[aoff[i]], bf, acc[i], 0, 0, 0);
            }
        }
        if (more) stash(buf ^ 1);   // the other buffer was last read in round t - 1: every wave passed that round's barrier
        __syncthreads();
    }
    float *dst = a.partial + ((size_t)blockIdx.x * 4 + wv) * (NU * 256) + lane;
#pragma unroll
    for (int i = 0; i < NU; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[(i * 4 + j) * 64] = acc[i][j];
}

template <int CIN, int COUT>
__global__ __launch_bounds__(256) void wgrad_xanchor_reduce_kernel(const float *__restrict__ partial, int nblocks,
                                                                   float *__restrict__ gw) {
    using C = WgradXCfg<CIN, COUT>;
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= C::PB) return;
    const int lane = r & 63, j = (r >> 6) & 3, i = (r >> 8) % C::NU, wv = r / (256 * C::NU);
    const int u = (wv % C::NG) * C::NU + i, nt = u / C::NP, p = u % C::NP;
    const int m = (lane >> 4) * 4 + j, tap = p * C::TPM + m / COUT, co = m % COUT, ci = nt * 16 + (lane & 15);
    if (tap >= 27 || ci >= CIN) return;
    float sum = 0.f;
    float u1 = 0.f, u2 = 0.f, u3 = 0.f;
    int s = blockIdx.y;
    const int ss = gridDim.y;
    for (; s + 3 * ss < nblocks; s += 4 * ss) {
        sum += partial[(size_t)s * C::PB + r]; u1 += partial[(size_t)(s + ss) * C::PB + r];
        u2 += partial[(size_t)(s + 2 * ss) * C::PB + r]; u3 += partial[(size_t)(s + 3 * ss) * C::PB + r];
    }
    for (; s < nblocks; s += ss) sum += partial[(size_t)s * C::PB + r];
    sum = (sum + u1) + (u2 + u3);
    unsafeAtomicAdd(gw + ((int64_t)co * CIN + ci) * 27 + tap, sum);
}

static bool wgrad_xanchor_shape(int Cin, int Cout, int stride) {
    return stride == 1 && (Cout == 8 || Cout == 1) && (Cin == 8 || Cin == 16 || Cin == 32);
}
static int wgrad_xanchor_blocks(int ntiles) { return min(ntiles, 2 * device_cu_count()); }
static size_t wgrad_xanchor_bytes(int Cout, int ntiles) {
    return (size_t)wgrad_xanchor_blocks(ntiles) * (4 * (Cout == 8 ? 7 : 2) * 256) * sizeof(float);
}

template <int CIN, int COUT>
static int launch_wgrad_xanchor(WgradArgs a, int ntiles, void *workspace, hipStream_t st) {
    using C = WgradXCfg<CIN, COUT>;
    const int nblocks = wgrad_xanchor_blocks(ntiles);
    a.partial = static_cast<float *>(workspace);
    hipLaunchKernelGGL((conv3d_wgrad_xanchor_kernel<CIN, COUT>), dim3((unsigned)nblocks), dim3(256), 0, st, a, ntiles);
    hipLaunchKernelGGL((wgrad_xanchor_reduce_kernel<CIN, COUT>), dim3((unsigned)((C::PB + 255) / 256), 32u), dim3(256),
                       0, st, a.partial, nblocks, a.gw);
    return check_launch("mvs_conv3d_wgrad_f32(x-anchored)");
}

}  // namespace mvs

using namespace mvs;

extern "C" int mvs_conv3d_wgrad_supported(int Cin, int Cout, int stride) {
    const bool cin_ok = Cin == 8 || Cin == 16 || Cin == 32 || Cin == 64;
    const bool cout_ok = Cout == 1 || Cout == 8 || Cout == 16 || Cout == 32 || Cout == 64;
    return (cin_ok && cout_ok && (stride == 1 || stride == 2)) ? 1 : 0;
}

static bool wgrad_geometry(int B, int Cin, int Cout, int D, int H, int W, int stride, WgradArgs &a, int64_t &nt) {
    a.B = B; a.Cin = Cin; a.Cout = Cout; a.D = D; a.H = H; a.W = W;
    a.Do = (D - 1) / stride + 1; a.Ho = (H - 1) / stride + 1; a.Wo = (W - 1) / stride + 1;
    const int tz = stride == 1 ? 2 : 1, ty = (stride == 1 || Cin < 16) ? 4 : 2;     // WgradCfg::TZ, TY (CK = 8 for Cin < 16)
    a.tiles_x = (a.Wo + 15) / 16; a.tiles_y = (a.Ho + ty - 1) / ty; a.tiles_z = (a.Do + tz - 1) / tz;
    nt = (int64_t)B * a.tiles_x * a.tiles_y * a.tiles_z;
    return nt > 0 && nt <= 0x7fffffffLL && (int64_t)9 * H * W * Cin * 4 < 0xffffff00LL;
}

extern "C" size_t mvs_conv3d_wgrad_workspace_bytes(int B, int Cin, int Cout, int D, int H, int W, int stride) {
    WgradArgs a;
    int64_t nt;
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || !mvs_conv3d_wgrad_supported(Cin, Cout, stride) ||
        !wgrad_geometry(B, Cin, Cout, D, H, W, stride, a, nt))
        return 0;
    if (wgrad_xanchor_shape(Cin, Cout, stride)) return wgrad_xanchor_bytes(Cout, (int)nt);
    const int ck = Cin >= 16 ? 16 : 8, ncc = (Cin + ck - 1) / ck;
    const int mt = Cout <= 16 ? 1 : (Cout <= 32 ? 2 : 4);
    return (size_t)wgrad_streams((int)nt, ncc, stride) * ncc * (4 * 7 * mt * 256) * sizeof(float);
}

static int conv3d_wgrad_impl(const float *in, const float *grad_out, int B, int Cin, int Cout, int D, int H, int W, int stride,
                             int in_c8, float *grad_weight, void *workspace, size_t workspace_bytes, void *stream);

extern "C" int mvs_conv3d_wgrad_f32(const float *in, const float *grad_out, int B, int Cin, int Cout,
                                    int D, int H, int W, int stride, float *grad_weight, void *workspace,
                                    size_t workspace_bytes, void *stream) {
    return conv3d_wgrad_impl(in, grad_out, B, Cin, Cout, D, H, W, stride, 0, grad_weight, workspace, workspace_bytes, stream);
}

// the conv0-class layers (Cout 8, stride 1) with the 8-channel-blocked input [B,D,H,Cin/8,W,8] their forward kernel reads
extern "C" int mvs_conv3d_wgrad_c8_f32(const float *in_c8, const float *grad_out, int B, int Cin, int D, int H, int W,
                                       float *grad_weight, void *workspace, size_t workspace_bytes, void *stream) {
    if (!(Cin == 8 || Cin == 16 || Cin == 32) || !workspace) {
        set_error("mvs_conv3d_wgrad_c8_f32: Cin in {8, 16, 32} (Cout = 8, stride 1) and a workspace of mvs_conv3d_wgrad_workspace_bytes");
        return MVS_EUNSUPPORTED;
    }
    return conv3d_wgrad_impl(in_c8, grad_out, B, Cin, 8, D, H, W, 1, 1, grad_weight, workspace, workspace_bytes, stream);
}

static int conv3d_wgrad_impl(const float *in, const float *grad_out, int B, int Cin, int Cout, int D, int H, int W, int stride,
                             int in_c8, float *grad_weight, void *workspace, size_t workspace_bytes, void *stream) {
    if (!in || !grad_out || !grad_weight || B <= 0 || D <= 0 || H <= 0 || W <= 0) {
        set_error("mvs_conv3d_wgrad_f32: bad argument");
        return MVS_EINVAL;
    }
    if (!mvs_conv3d_wgrad_supported(Cin, Cout, stride)) {
        set_error("mvs_conv3d_wgrad_f32: no kernel for Cin=%d Cout=%d stride=%d", Cin, Cout, stride);
        return MVS_EUNSUPPORTED;
    }
    WgradArgs a;
    int64_t nt;
    if (!wgrad_geometry(B, Cin, Cout, D, H, W, stride, a, nt)) return bare_error(MVS_EINVAL, __func__, __LINE__);
    a.x = in; a.g = grad_out; a.gw = grad_weight; a.partial = nullptr; a.x_c8 = in_c8;
    hipStream_t st = as_stream(stream);
    if (in_c8 && !(wgrad_xanchor_shape(Cin, Cout, stride) && workspace && workspace_bytes >= wgrad_xanchor_bytes(Cout, (int)nt))) {
        set_error("mvs_conv3d_wgrad_c8_f32: workspace of %zu bytes too small", workspace_bytes);
        return MVS_EWORKSPACE;
    }
    if (wgrad_xanchor_shape(Cin, Cout, stride) && workspace && workspace_bytes >= wgrad_xanchor_bytes(Cout, (int)nt)) {
#define MVS_WX(ci, co) if (Cin == ci && Cout == co) return launch_wgrad_xanchor<ci, co>(a, (int)nt, workspace, st);
        MVS_WX(32, 8) MVS_WX(16, 8) MVS_WX(8, 8) MVS_WX(32, 1) MVS_WX(16, 1) MVS_WX(8, 1)
#undef MVS_WX
    }
    const int ck = Cin >= 16 ? 16 : 8;
#define MVS_WG(co)                                                                               \
    if (Cout <= co) {                                                                            \
        if (stride == 1)                                                                         \
            return ck == 16 ? launch_wgrad<co, 16, 1>(a, (int)nt, workspace, workspace_bytes, st) \
                            : launch_wgrad<co, 8, 1>(a, (int)nt, workspace, workspace_bytes, st); \
        return ck == 16 ? launch_wgrad<co, 16, 2>(a, (int)nt, workspace, workspace_bytes, st)    \
                        : launch_wgrad<co, 8, 2>(a, (int)nt, workspace, workspace_bytes, st);    \
    }
    MVS_WG(16) MVS_WG(32) MVS_WG(64)
#undef MVS_WG
    return bare_error(MVS_EUNSUPPORTED, __func__, __LINE__);
}
