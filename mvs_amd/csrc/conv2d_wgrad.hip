// Training path of FeatureNet (MVSNet/models/mvsnet.py:8-45 under autograd, train.py:222-226) without MIOpen:
//
//   mvs_conv2d_wgrad_f32     weight gradient of a k x k, stride-s 2D convolution on the fp32 matrix cores
//                            dW[co][ci][ky][kx] = sum over images and output pixels o of g[o][co] * x[o*s + k - pad][ci]
//   mvs_interleave2x2_f32    the four output-parity classes of a stride-2 transposed convolution -> one map
//                            (the input gradient of the 5x5 stride-2 layers = four 3x3 stride-1 convolutions of the
//                            output gradient, each on the existing 2D kernels, then this interleave)
//
// Weight gradient: the reduction dimension of the MFMA is the PIXELS, as in conv3d_wgrad.hip -- one
// v_mfma_f32_16x16x4_f32 = D[16 co x 16 ci] += A[16 co x 4 pixels] * B[4 pixels x 16 ci], A a fragment of the output
// gradient, B the input at the tap's offset, four consecutive output pixels along x per MFMA -- so the accumulators ARE
// the weight gradient and stay in registers for the whole kernel: persistent workgroups walk (8 x 16)-pixel output
// tiles staged through LDS (zero padding = zeros in LDS; channels padded to 16), the four waves split the k*k taps,
// each workgroup stores its partial dW once and conv2d_wgrad_reduce_kernel sums the workgroups.  FeatureNet's eight
// layers are 3 GFLOP per view at 640x512: the kernel is sized for simplicity, not for the matrix pipe's peak.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mvs_common.h"

namespace mvs {

typedef float w2_f32x4 __attribute__((ext_vector_type(4)));

struct Wgrad2dArgs {
    const float *x, *g;
    float *partial;        // [workgroup][tap][m][n][lane][4]
    int N, Cin, Cout, H, W, Ho, Wo, planar;
    int tiles_x, tiles_y;
};

constexpr int kW2TY = 8;    // output rows per tile (16 columns)

// K = kernel size, S = stride, MT / NT = 16-channel tiles of Cout / Cin (channels beyond the real count are zeros)
template <int K, int S, int MT, int NT>
__global__ __launch_bounds__(256, (K == 5 && NT == 2) ? 1 : 2) void conv2d_wgrad_kernel(Wgrad2dArgs a, int ntiles) {
    constexpr int TY = kW2TY, PAD = K / 2;
    constexpr int YT = (TY - 1) * S + K, XT = 15 * S + K;
    constexpr int GP = MT * 16 + (MT == 2 ? 16 : 0);    // floats per output pixel in LDS (bank spread, see conv3d_wgrad.hip)
    constexpr int XP = NT * 16 + (S == 2 ? 8 : 0);      // floats per input pixel
    constexpr int NTAP = K * K, TPW = (NTAP + 3) / 4;   // taps per wave
    __shared__ __attribute__((aligned(16))) float gl[TY * 16 * GP];
    __shared__ __attribute__((aligned(16))) float xl[YT * XT * XP];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, kv = lane >> 4;

    w2_f32x4 acc[TPW][MT][NT];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[t][m][n] = (w2_f32x4){0.f, 0.f, 0.f, 0.f};

    const int64_t gimg = (int64_t)a.Ho * a.Wo * a.Cout, ximg = (int64_t)a.H * a.W * a.Cin;
    // The next tile's global loads are issued before this tile's K loop and wait in registers (gr / xr / xs1) while the matrix
    // pipe works; they go to LDS between the two barriers of the next round (as conv3d_wgrad.hip).
    constexpr int GI = (TY * 16 * MT * 4 + 255) / 256, XI = (YT * XT * NT * 4 + 255) / 256;
    constexpr int PI = (YT * XT * 4 + 255) / 256;   // planar input (the image: Cin <= 4): one float per item, channels 0..3
    float4 gr[GI], xr[XI];
    float xs1[PI];
    auto fetch = [&](int t) {
        int bid = t;
        const int tx = bid % a.tiles_x; bid /= a.tiles_x;
        const int ty = bid % a.tiles_y;
        const int n = bid / a.tiles_y;
        const int ox0 = tx * 16, oy0 = ty * TY, ix0 = ox0 * S - PAD, iy0 = oy0 * S - PAD;
        // output-gradient tile, zeros beyond Cout and the image (16-byte pieces: Cout % 4 == 0)
#pragma unroll
        for (int i = 0; i < GI; ++i) {
            const int e = tid + i * 256;
            const int q = e % (MT * 4), v = e / (MT * 4);
            const int ox = ox0 + (v & 15), oy = oy0 + (v >> 4);
            gr[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < TY * 16 * MT * 4 && q * 4 < a.Cout && ox < a.Wo && oy < a.Ho)
                gr[i] = *reinterpret_cast<const float4 *>(a.g + (int64_t)n * gimg + ((int64_t)oy * a.Wo + ox) * a.Cout + q * 4);
        }
        // input halo, zeros beyond Cin and the image (= the convolution's padding)
        if (a.planar) {
#pragma unroll
            for (int i = 0; i < PI; ++i) {
                const int e = tid + i * 256;
                const int v = e % (YT * XT), ci = e / (YT * XT);      // pixel fastest: planar rows are contiguous in x
                const int gx = ix0 + v % XT, gy = iy0 + v / XT;
                xs1[i] = 0.f;
                if (ci < 4 && ci < a.Cin && (unsigned)gx < (unsigned)a.W && (unsigned)gy < (unsigned)a.H)
                    xs1[i] = a.x[((int64_t)n * a.Cin + ci) * a.H * a.W + (int64_t)gy * a.W + gx];
            }
        } else {
#pragma unroll
            for (int i = 0; i < XI; ++i) {
                const int e = tid + i * 256;
                const int q = e % (NT * 4), v = e / (NT * 4);
                const int gx = ix0 + v % XT, gy = iy0 + v / XT;
                xr[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e < YT * XT * NT * 4 && q * 4 < a.Cin && (unsigned)gx < (unsigned)a.W && (unsigned)gy < (unsigned)a.H)
                    xr[i] = *reinterpret_cast<const float4 *>(a.x + (int64_t)n * ximg + ((int64_t)gy * a.W + gx) * a.Cin + q * 4);
            }
        }
    };
    auto stash = [&]() {   // registers -> gl[pixel][co], xl[pixel][ci]
#pragma unroll
        for (int i = 0; i < GI; ++i) {
            const int e = tid + i * 256;
            if (e < TY * 16 * MT * 4) *reinterpret_cast<float4 *>(gl + (e / (MT * 4)) * GP + (e % (MT * 4)) * 4) = gr[i];
        }
        if (a.planar) {
#pragma unroll
            for (int i = 0; i < PI; ++i) {
                const int e = tid + i * 256;
                if (e < YT * XT * 4) xl[(e % (YT * XT)) * XP + e / (YT * XT)] = xs1[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < XI; ++i) {
                const int e = tid + i * 256;
                if (e < YT * XT * NT * 4) *reinterpret_cast<float4 *>(xl + (e / (NT * 4)) * XP + (e % (NT * 4)) * 4) = xr[i];
            }
        }
    };
    if (a.planar)      // columns 4 .. of a planar input are zeros for the whole kernel
        for (int e = tid; e < YT * XT * XP; e += 256) xl[e] = 0.f;
    if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        __syncthreads();   // every wave is done with the previous tile
        stash();
        __syncthreads();
        if (t + (int)gridDim.x < ntiles) fetch(t + gridDim.x);
#pragma unroll 1
        for (int row = 0; row < TY; ++row) {
#pragma unroll
            for (int xs = 0; xs < 4; ++xs) {
                float af[MT];
#pragma unroll
                for (int m = 0; m < MT; ++m) af[m] = gl[(row * 16 + xs * 4 + kv) * GP + m * 16 + c];
#pragma unroll
                for (int tt = 0; tt < TPW; ++tt) {
                    // wave-uniform; a slot past the last tap repeats it (computed, never stored): a branch around the
                    // MFMA made every LDS read wait for its own round trip (96 x ~130 cycles per tile instead of 96 x 32)
                    const int tap = min(wv + tt * 4, NTAP - 1);
                    const int ky = tap / K, kx = tap % K;
                    const float *xb = xl + ((row * S + ky) * XT + (xs * 4 + kv) * S + kx) * XP + c;
#pragma unroll
                    for (int nn = 0; nn < NT; ++nn) {
                        const float bf = xb[nn * 16];
#pragma unroll
                        for (int m = 0; m < MT; ++m)
                            acc[tt][m][nn] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m], bf, acc[tt][m][nn], 0, 0, 0);
                    }
                }
            }
        }
    }
    // partial dW of this workgroup, accumulator-register order: [tap][m][n][lane][4] (zeros if it had no tile)
    float *out = a.partial + (int64_t)blockIdx.x * (NTAP * MT * NT * 256);
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt) {
        const int tap = wv + tt * 4;
        if (tap >= NTAP) continue;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int nn = 0; nn < NT; ++nn)
                *reinterpret_cast<float4 *>(out + ((tap * MT + m) * NT + nn) * 256 + lane * 4) =
                    make_float4(acc[tt][m][nn][0], acc[tt][m][nn][1], acc[tt][m][nn][2], acc[tt][m][nn][3]);
    }
}

// sum the workgroups' partials.  A block owns 16 consecutive accumulator elements; thread (e = tid & 15, s = tid >> 4)
// adds the partials of workgroups s, s + 16, ... in four independent chains (one thread per element walking all
// workgroups was a 512-long chain of dependent loads: 88 us per layer), the 16 slices meet in LDS.  element -> (co, ci,
// tap) by the MFMA's D layout (lane (j = lane & 15, r = lane >> 4), register e: row 4 r + e = co, column j = ci)
__global__ __launch_bounds__(256) void conv2d_wgrad_reduce_kernel(const float *__restrict__ partial, int nwg, int ntap, int MT,
                                                                   int NT, int Cout, int Cin, float *__restrict__ gw) {
    __shared__ float red[16][17];
    const int total = ntap * MT * NT * 256;
    const int el = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + el;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < total) {
        int w = sl;
        for (; w + 48 < nwg; w += 64) {
            s0 += partial[(int64_t)w * total + i];
            s1 += partial[(int64_t)(w + 16) * total + i];
            s2 += partial[(int64_t)(w + 32) * total + i];
            s3 += partial[(int64_t)(w + 48) * total + i];
        }
        for (; w < nwg; w += 16) s0 += partial[(int64_t)w * total + i];
    }
    red[sl][el] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sl == 0 && i < total) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += red[k][el];
        const int e = i & 3, lane = (i >> 2) & 63, blk = i >> 8;
        const int nn = blk % NT, m = (blk / NT) % MT, tap = blk / (NT * MT);
        const int co = m * 16 + 4 * (lane >> 4) + e, ci = nn * 16 + (lane & 15);
        if (co < Cout && ci < Cin) gw[((int64_t)co * Cin + ci) * ntap + tap] = s;
    }
}

// out[n][2y + py][2x + px][c] = cls[py * 2 + px][n][y][x][c]
__global__ __launch_bounds__(256) void interleave2x2_kernel(const float4 *__restrict__ cls, float4 *__restrict__ out, int N,
                                                            int H, int W, int C4) {
    const int64_t per = (int64_t)N * H * W * C4, total = per * 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int64_t r = i;
        const int c = (int)(r % C4); r /= C4;
        const int X = (int)(r % (2 * W)); r /= 2 * W;
        const int Y = (int)(r % (2 * H));
        const int64_t n = r / (2 * H);
        const int k = (Y & 1) * 2 + (X & 1);
        out[i] = cls[k * per + ((n * H + (Y >> 1)) * W + (X >> 1)) * C4 + c];
    }
}

static int wgrad2d_grid(int ntiles) {
    const int cap = 2 * device_cu_count();
    return ntiles < cap ? ntiles : cap;
}

}  // namespace mvs

using namespace mvs;

static bool wgrad2d_shape(int Cin, int Cout, int k, int stride, int *mt, int *nt) {
    if (!((k == 3 && stride == 1) || (k == 5 && stride == 2))) return false;
    if (Cin < 1 || Cin > 32 || Cout < 1 || Cout > 32) return false;
    *mt = (Cout + 15) / 16; *nt = (Cin + 15) / 16;
    return true;
}

extern "C" size_t mvs_conv2d_wgrad_workspace_bytes(int N, int Cin, int Cout, int H, int W, int ksize, int stride) {
    int mt, nt;
    if (N <= 0 || H <= 0 || W <= 0 || !wgrad2d_shape(Cin, Cout, ksize, stride, &mt, &nt)) return 0;
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const int64_t ntiles = (int64_t)N * ((Ho + kW2TY - 1) / kW2TY) * ((Wo + 15) / 16);
    if (ntiles > 0x7fffffffLL) return 0;
    return (size_t)wgrad2d_grid((int)ntiles) * ksize * ksize * mt * nt * 256 * sizeof(float);
}

extern "C" int mvs_conv2d_wgrad_f32(const float *x, const float *grad_out, int N, int Cin, int Cout, int H, int W,
                                    int ksize, int stride, int planar, float *grad_weight, void *workspace,
                                    size_t workspace_bytes, void *stream) {
    int mt, nt;
    if (!x || !grad_out || !grad_weight || N <= 0 || H <= 0 || W <= 0) {
        set_error("mvs_conv2d_wgrad_f32: invalid argument");
        return MVS_EINVAL;
    }
    if (planar && Cin > 4) {
        set_error("mvs_conv2d_wgrad_f32: a planar input is the image (Cin <= 4), got Cin=%d", Cin);
        return MVS_EUNSUPPORTED;
    }
    if (!wgrad2d_shape(Cin, Cout, ksize, stride, &mt, &nt)) {
        set_error("mvs_conv2d_wgrad_f32: 3x3 stride 1 or 5x5 stride 2 with up to 32 channels, got k=%d stride=%d Cin=%d Cout=%d",
                  ksize, stride, Cin, Cout);
        return MVS_EUNSUPPORTED;
    }
    const size_t need = mvs_conv2d_wgrad_workspace_bytes(N, Cin, Cout, H, W, ksize, stride);
    if (need == 0) return bare_error(MVS_EINVAL, __func__, __LINE__);
    if (!workspace || workspace_bytes < need) {
        set_error("mvs_conv2d_wgrad_f32: workspace of %zu bytes, need %zu", workspace_bytes, need);
        return MVS_EWORKSPACE;
    }
    Wgrad2dArgs a;
    a.x = x; a.g = grad_out; a.partial = static_cast<float *>(workspace);
    a.N = N; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W; a.planar = planar;
    a.Ho = (H - 1) / stride + 1; a.Wo = (W - 1) / stride + 1;
    a.tiles_x = (a.Wo + 15) / 16; a.tiles_y = (a.Ho + kW2TY - 1) / kW2TY;
    const int ntiles = N * a.tiles_x * a.tiles_y, grid = wgrad2d_grid(ntiles);
    hipStream_t st = as_stream(stream);
#define MVS_W2(K_, S_, M_, N_)                                                                                   \
    if (ksize == K_ && mt == M_ && nt == N_) {                                                                   \
        hipLaunchKernelGGL((conv2d_wgrad_kernel<K_, S_, M_, N_>), dim3(grid), dim3(256), 0, st, a, ntiles);      \
    } else
    MVS_W2(3, 1, 1, 1) MVS_W2(3, 1, 2, 2) MVS_W2(3, 1, 1, 2) MVS_W2(3, 1, 2, 1)
    MVS_W2(5, 2, 1, 1) MVS_W2(5, 2, 2, 1) MVS_W2(5, 2, 1, 2) MVS_W2(5, 2, 2, 2)
    return bare_error(MVS_EUNSUPPORTED, __func__, __LINE__);
#undef MVS_W2
    const int total = ksize * ksize * mt * nt * 256;
    hipLaunchKernelGGL(conv2d_wgrad_reduce_kernel, dim3((total + 15) / 16), dim3(256), 0, st, a.partial, grid, ksize * ksize,
                       mt, nt, Cout, Cin, grad_weight);
    return check_launch("mvs_conv2d_wgrad_f32");
}

extern "C" int mvs_interleave2x2_f32(const float *classes, int N, int H, int W, int C, float *out, void *stream) {
    if (!classes || !out || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) {
        set_error("mvs_interleave2x2_f32: invalid argument (C must be a multiple of 4)");
        return MVS_EINVAL;
    }
    const int64_t total = (int64_t)N * H * W * C;    // float4 items of the output
    const unsigned grid = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(interleave2x2_kernel, dim3(grid), dim3(256), 0, as_stream(stream), reinterpret_cast<const float4 *>(classes),
                       reinterpret_cast<float4 *>(out), N, H, W, C / 4);
    return check_launch("mvs_interleave2x2_f32");
}
