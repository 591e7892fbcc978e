// Weight gradient of conv0 (32 -> 8, 3x3x3, stride 1; MVSNet/models/mvsnet.py:52 under autograd, train.py:222-226) on the
// 16-bit matrix pipe with the two-piece fp16 operands of the forward kernels (conv_f16x3.hip has the arithmetic and its
// error bound):
//
//   dW[co][ci][kz,ky,kx] = sum over input voxels v of  g[v - (k - 1)][co] * x[v][ci]
//
// x = the variance volume, 8-channel blocked [B,D,H,Cin/8,W,8] as conv0's forward reads it, g = the gradient of conv0's raw
// output [B,D,H,W,8].  On the fp32 matrix pipe (conv3d_wgrad_xanchor_kernel) this layer is 27.5 M v_mfma_f32_16x16x4_f32 per
// step = 0.36 ms at the pipe's peak and 0.52-0.56 ms measured, the largest weight gradient of a training step by 3x.
// v_mfma_f32_16x16x32_f16 carries 8x the voxels per instruction at half the cycles; with three products per fp32 product
// (hi*hi + hi*lo + lo*hi of x * 2^(14-ex) = hi + lo, g likewise) that is 5.3x the rate.
//
// The reduction dimension of the MFMA is 32 consecutive voxels along x, and BOTH operands want 8 consecutive voxels of one
// channel per lane -- the transpose of how either tensor lies in memory.  No LDS: a lane loads its 8 voxels as 8 dwords, 32
// bytes apart, straight from HBM/L2 (the 8 loads of a lane walk the same cache lines, the 16 lanes of a row cover the 32-byte
// channel groups), splits them in registers and feeds the pipe:
//   A (M = 16 input channels, two tiles, three x shifts): x[z][y][x0 + 8 kq + i + dx][ci],   i = 0..7, dx = kx - 1
//   B (N = 2 tap rows x 8 output channels):               g[z - dz_h][y - dy_h][x0 + 8 kq + i][co]
// The x shift of a tap rides on the x operand (sum_v g[v - dx] x[v] = sum_u g[u] x[u + dx]), which is loaded once per chunk
// while g is loaded for each of the five row pairs the nine (dz, dy) tap rows make: ten loads per lane and tile (elements
// -1..8) give all three shifts -- dx = 0 packs the pairs (0,1)(2,3)(4,5)(6,7), dx = -1 the pairs (-1,0)(1,2)(3,4)(5,6), dx = +1
// (1,2)(3,4)(5,6)(7,8) -- so a chunk of 32 voxels costs 60 dword loads and 38 pair splits for its 90 MFMAs.  A wave owns all
// 2 x 15 accumulator tiles (120 registers = the whole weight gradient) and a stream of (row, 32-voxel segment) chunks; it
// stores them once, the reduce kernel sums the waves and undoes the two scales.
//
// Measured (640x512 training step, 3.9 M voxels): 0.23 ms + 0.02 ms reduce against 0.54 + 0.02 on the fp32 pipe.  Ablation
// of the first version (0.28 ms in isolation, shifts on g: 66 loads, 53 splits; scripts/exp_wgrad_f16_abl.py times the op):
// without the MFMAs -0.10 ms (19 cycles each: the pipe's own time), without the g splits -0.085, without the loads -0.125 --
// the three add up: a wave takes them in turn and two waves per SIMD (230 registers) overlap little.  Tried on top of it:
// loads two pairs / a whole chunk ahead (no change: not latency); 16-byte loads with a 4 x 4 DPP transpose per lane quad (24
// load instructions instead of 66, +224 selects and moves: 0.30 ms); the four waves of a workgroup on adjacent rows sharing
// one g halo tile in LDS (19.6 KB instead of 4 x 10.9 KB through the texture path: 0.27 ms); the next pair's splits placed
// between this pair's three MFMA groups (kept, no measurable change).  Moving the shifts to x (this version) took 0.02 ms.
// The split asm must stay `volatile`: without it the compiler mis-handles the half-register writes of v_fma_mixlo / mixhi
// (wrong gradients).
//
// Arithmetic: products are within 2^-22 relative of the fp32 products for operands within 2^-18 of their tensor's largest
// magnitude and within 2^-40 of (max |x| max |g|) absolute below that (an element far below its tensor's maximum loses
// relative precision, as in the forward layers); accumulation is fp32 as before.  Non-finite inputs give non-finite
// gradients, as the fp32 kernel does.  There is no range guard on this path: it is taken by the training node only
// (ops._VarianceConv0), MVS_WGRAD_F16=0 keeps the fp32 kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mvs_common.h"
#include "conv_split_common.h"

namespace mvs {

struct WgradF16Args {
    const float *x, *g;
    const unsigned *x_absmax, *g_absmax;
    float *partial;          // [wave][mt][tile][j][lane]
    float *gw;
    int B, D, H, W;
    int nseg;                // 32-voxel segments per row
    int nchunks;             // B * D * H * nseg
    int nwaves;
};

constexpr int kWFTiles = 15;                       // 5 row pairs x 3 x-shifts
constexpr int kWFSlots = 2 * kWFTiles * 256;       // floats of one wave's partial

// (a * s, b * s) -> hi = fp16 pair (round to nearest even), lo = fp16 pair of the residuals
__device__ __forceinline__ void split_pair(float a, float b, float s, unsigned &h, unsigned &l) {
    asm volatile(
        "v_fma_mixlo_f16 %2, %0, %4, 0\n\t"
        "v_fma_mixhi_f16 %2, %1, %4, 0\n\t"
        "v_fma_mix_f32 %0, %0, %4, -%2 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mix_f32 %1, %1, %4, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_cvt_pk_f16_f32 %3, %0, %1"
        : "+v"(a), "+v"(b), "=&v"(h), "=&v"(l)
        : "s"(s));
}

__device__ __forceinline__ f16x8 frag(unsigned a, unsigned b, unsigned c, unsigned d) {
    return __builtin_bit_cast(f16x8, (u32x4){a, b, c, d});
}

__device__ __forceinline__ int wf_rsrc_bytes(int64_t n) { return (int)(unsigned)(n < 0xffffff00LL ? n : 0xffffff00LL); }

// FULLW: W is a multiple of 32 (no element of a segment lies beyond the row)
template <bool FULLW>
__global__ __launch_bounds__(256, 2) void conv3d_wgrad_c8_f16_kernel(WgradF16Args a) {
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 15, kq = lane >> 4;
    const int h = n >> 3, co = n & 7;
    const float sx = pow2f(14 - absmax_exponent(load_absmax(a.x_absmax)));
    const float sg = pow2f(14 - absmax_exponent(load_absmax(a.g_absmax)));

    f32x4 acc[2][kWFTiles];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int t = 0; t < kWFTiles; ++t) acc[mt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int64_t vox = (int64_t)a.B * a.D * a.H * a.W;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.x), 0, wf_rsrc_bytes(vox * 128), 0x00020000);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.g), 0, wf_rsrc_bytes(vox * 32), 0x00020000);
    constexpr unsigned OOB = 0xffffff00u;
    const int gwave = blockIdx.x * 4 + wv;

    // chunk c = ((b D + z) H + y) nseg + seg, c = gwave, gwave + nwaves, ...: the four coordinates advance by the digits of
    // nwaves with carries (four integer divisions per chunk were a third of the loop's vector instructions)
    struct Pos { int seg, y, z, b; };
    Pos cur, nxt;
    {
        int r = gwave;
        nxt.seg = r % a.nseg; r /= a.nseg;
        nxt.y = r % a.H; r /= a.H;
        nxt.z = r % a.D;
        nxt.b = r / a.D;
    }
    int sseg, sy, sz, sb;
    {
        int r = a.nwaves;
        sseg = r % a.nseg; r /= a.nseg;
        sy = r % a.H; r /= a.H;
        sz = r % a.D;
        sb = r / a.D;
    }
    const int niter = gwave < a.nchunks ? (a.nchunks - gwave + a.nwaves - 1) / a.nwaves : 0;
    auto advance = [&](Pos q) {
        q.seg += sseg; int cy = 0;
        if (q.seg >= a.nseg) { q.seg -= a.nseg; cy = 1; }
        q.y += sy + cy; cy = 0;
        if (q.y >= a.H) { q.y -= a.H; cy = 1; }
        q.z += sz + cy; cy = 0;
        if (q.z >= a.D) { q.z -= a.D; cy = 1; }
        q.b += sb + cy;
        return q;
    };
    // ---- x: 2 tiles x elements -1 .. 8 of channel mt * 16 + n of the chunk at q; blocked layout [row][Cin/8][W][8].
    // The x shift of a tap rides on THIS operand (sum_v g[v - dx] x[v] = sum_u g[u] x[u + dx]): it is loaded once per chunk,
    // g five times, so the odd-pair splits and the two halo loads are paid per chunk instead of per row pair.
    float xv[2][10];
    auto load_x = [&](const Pos &q) {
        const int xb = q.seg * 32 + kq * 8, lim = a.W - xb;    // this lane's first voxel; elements e < lim are inside the row
        const int64_t row = ((int64_t)q.b * a.D + q.z) * a.H + q.y;
        const unsigned base = (unsigned)((((row * 4 + (n >> 3)) * a.W + xb) * 8 + (n & 7)) * 4);
        const unsigned off = lim > 0 ? base : OOB;
        const unsigned offl = (lim > 0 && xb > 0) ? base - 32u : OOB, offr = 8 < lim ? base + 256u : OOB;
        const int tile = a.W * 64;                             // bytes from channel group g to g + 2
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            xv[mt][0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, offl, mt * tile, 0));
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                xv[mt][1 + i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, off + i * 32, mt * tile, 0));
                if (!FULLW && i >= lim) xv[mt][1 + i] = 0.f;
            }
            xv[mt][9] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, offr, mt * tile, 0));
        }
    };
    // ---- g: per row pair, elements 0 .. 7 of output channel co in tap row 2 p + h of the chunk at q
    auto load_pair = [&](const Pos &q, int p, float (&gv)[8]) {
        const int xb = q.seg * 32 + kq * 8, lim = a.W - xb;
        const int rr = 2 * p + h;              // tap row (kz, ky); the tenth does not exist
        const int kz = rr / 3, ky = rr - 3 * kz;
        const int gz = q.z - (kz - 1), gy = q.y - (ky - 1);
        const bool rowok = rr < 9 && (unsigned)gz < (unsigned)a.D && (unsigned)gy < (unsigned)a.H && lim > 0;
        const int64_t row = ((int64_t)q.b * a.D + gz) * a.H + gy;
        const unsigned off = rowok ? (unsigned)(((row * a.W + xb) * 8 + co) * 4) : OOB;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            gv[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rg, off + i * 32, 0, 0));
            if (!FULLW && i >= lim) gv[i] = 0.f;
        }
    };
    // Software pipeline over the row pairs (and across chunks): while the 18 MFMAs of pair p run, the four splits of pair
    // p + 1 sit between their three groups and the loads of pair p + 2 are in flight -- issued in source order, a wave's
    // splits, loads and MFMAs took turns (ablation: MFMAs 0.10 ms, splits 0.085, loads 0.125 of 0.28, adding up).  x is
    // loaded a whole chunk ahead (HBM: the volume is read once); g mostly hits L2 (a g row serves nine tap rows).
    float g1[8], g2[8];                 // raw g of the next pair (landed) and of the one after (in flight)
    f16x8 Bh, Bl;
    if (niter > 0) {
        load_x(nxt);
        float g0[8];
        load_pair(nxt, 0, g0);
        load_pair(nxt, 1, g1);
        unsigned bh[4], bl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) split_pair(g0[2 * j], g0[2 * j + 1], sg, bh[j], bl[j]);
        Bh = frag(bh[0], bh[1], bh[2], bh[3]); Bl = frag(bl[0], bl[1], bl[2], bl[3]);
    }
#pragma unroll 1
    for (int it = 0; it < niter; ++it) {
        // A fragments: x-shift index 0: kx = 0, x at u - 1 (elements -1..6 = odd pairs 0..3); 1: kx = 1 (even pairs);
        // 2: kx = 2, x at u + 1 (elements 1..8 = odd pairs 1..4)
        f16x8 Ah[2][3], Al[2][3];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            unsigned eh[4], el[4], oh[5], ol[5];
#pragma unroll
            for (int j = 0; j < 4; ++j) split_pair(xv[mt][1 + 2 * j], xv[mt][2 + 2 * j], sx, eh[j], el[j]);
#pragma unroll
            for (int j = 0; j < 5; ++j) split_pair(xv[mt][2 * j], xv[mt][2 * j + 1], sx, oh[j], ol[j]);
            Ah[mt][0] = frag(oh[0], oh[1], oh[2], oh[3]); Al[mt][0] = frag(ol[0], ol[1], ol[2], ol[3]);
            Ah[mt][1] = frag(eh[0], eh[1], eh[2], eh[3]); Al[mt][1] = frag(el[0], el[1], el[2], el[3]);
            Ah[mt][2] = frag(oh[1], oh[2], oh[3], oh[4]); Al[mt][2] = frag(ol[1], ol[2], ol[3], ol[4]);
        }
        const bool more = it + 1 < niter;
        cur = nxt;
        nxt = advance(cur);
        if (more) load_x(nxt);       // (xv is free: its pieces are in Ah / Al)
#pragma unroll
        for (int p = 0; p < 5; ++p) {
            // pair p + 2 of this chunk, or pair p - 3 of the next one
            if (p < 3) load_pair(cur, p + 2, g2);
            else if (more) load_pair(nxt, p - 3, g2);
            unsigned nh[4], nl[4];
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    acc[mt][p * 3 + s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[mt][s], Bl, acc[mt][p * 3 + s], 0, 0, 0);
            split_pair(g1[0], g1[1], sg, nh[0], nl[0]);
            split_pair(g1[2], g1[3], sg, nh[1], nl[1]);
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    acc[mt][p * 3 + s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al[mt][s], Bh, acc[mt][p * 3 + s], 0, 0, 0);
            split_pair(g1[4], g1[5], sg, nh[2], nl[2]);
            split_pair(g1[6], g1[7], sg, nh[3], nl[3]);
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    acc[mt][p * 3 + s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[mt][s], Bh, acc[mt][p * 3 + s], 0, 0, 0);
            Bh = frag(nh[0], nh[1], nh[2], nh[3]); Bl = frag(nl[0], nl[1], nl[2], nl[3]);
#pragma unroll
            for (int i = 0; i < 8; ++i) g1[i] = g2[i];
        }
    }
    // ---- this wave's partial, accumulator order (coalesced): [wave][mt][tile][j][lane]
    float *dst = a.partial + (size_t)gwave * kWFSlots + lane;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int t = 0; t < kWFTiles; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) dst[((mt * kWFTiles + t) * 4 + j) * 64] = acc[mt][t][j];
}

// grad_weight += 2^(ex - 14) 2^(eg - 14) * sum over the waves.  blockIdx.y slices the waves; a slot's slices meet in
// grad_weight through one atomic each.  D layout: lane (n = (h, co), q), register j -> ci = 16 mt + 4 q + j.
__global__ __launch_bounds__(256) void conv3d_wgrad_c8_f16_reduce_kernel(WgradF16Args a) {
    const int slot = blockIdx.x * 256 + threadIdx.x;   // < kWFSlots (a multiple of 256)
    const unsigned bx = load_absmax(a.x_absmax), bg = load_absmax(a.g_absmax);
    const int lane = slot & 63, j = (slot >> 6) & 3, t = (slot >> 8) % kWFTiles, mt = slot / (256 * kWFTiles);
    const int n = lane & 15, q = lane >> 4, h = n >> 3, co = n & 7;
    const int rr = 2 * (t / 3) + h, kx = t % 3;
    if (rr >= 9) return;
    const int ci = mt * 16 + 4 * q + j;
    float u0 = 0.f, u1 = 0.f, u2 = 0.f, u3 = 0.f;
    int w = blockIdx.y;
    const int ws = gridDim.y;
    for (; w + 3 * ws < a.nwaves; w += 4 * ws) {
        u0 += a.partial[(size_t)w * kWFSlots + slot]; u1 += a.partial[(size_t)(w + ws) * kWFSlots + slot];
        u2 += a.partial[(size_t)(w + 2 * ws) * kWFSlots + slot]; u3 += a.partial[(size_t)(w + 3 * ws) * kWFSlots + slot];
    }
    for (; w < a.nwaves; w += ws) u0 += a.partial[(size_t)w * kWFSlots + slot];
    const float unscale = pow2f(absmax_exponent(bx) - 14) * pow2f(absmax_exponent(bg) - 14);
    unsafeAtomicAdd(a.gw + ((int64_t)co * 32 + ci) * 27 + rr * 3 + kx, ((u0 + u1) + (u2 + u3)) * unscale);
}

static int wgrad_f16_waves() { return 2 * device_cu_count() * 4; }

}  // namespace mvs

using namespace mvs;

extern "C" int mvs_conv3d_wgrad_c8_f16_supported(int B, int Cin, int D, int H, int W) {
    if (Cin != 32 || B <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
    const int64_t vox = (int64_t)B * D * H * W, chunks = (int64_t)B * D * H * ((W + 31) / 32);
    return vox * 128 < 0xffffff00LL && chunks < 0x7fffffffLL ? 1 : 0;      // one buffer resource per tensor; chunk index in an int
}

extern "C" size_t mvs_conv3d_wgrad_c8_f16_workspace_bytes(int B, int Cin, int D, int H, int W) {
    if (!mvs_conv3d_wgrad_c8_f16_supported(B, Cin, D, H, W)) return 0;
    return (size_t)wgrad_f16_waves() * kWFSlots * sizeof(float);
}

extern "C" int mvs_conv3d_wgrad_c8_f16_f32(const float *in_c8, const unsigned *in_absmax, const float *grad_out,
                                           const unsigned *grad_absmax, int B, int Cin, int D, int H, int W, float *grad_weight,
                                           void *workspace, size_t workspace_bytes, void *stream) {
    if (!in_c8 || !in_absmax || !grad_out || !grad_absmax || !grad_weight || !workspace) {
        set_error("mvs_conv3d_wgrad_c8_f16_f32: null argument (both absmax blocks and the workspace are required)");
        return MVS_EINVAL;
    }
    if (!mvs_conv3d_wgrad_c8_f16_supported(B, Cin, D, H, W)) {
        set_error("mvs_conv3d_wgrad_c8_f16_f32: Cin = 32 (Cout = 8, 3x3x3, stride 1) and a volume below 4 GB, got Cin=%d %dx%dx%dx%d",
                  Cin, B, D, H, W);
        return MVS_EUNSUPPORTED;
    }
    const size_t need = mvs_conv3d_wgrad_c8_f16_workspace_bytes(B, Cin, D, H, W);
    if (workspace_bytes < need) {
        set_error("mvs_conv3d_wgrad_c8_f16_f32: workspace of %zu bytes, need %zu", workspace_bytes, need);
        return MVS_EWORKSPACE;
    }
    WgradF16Args a;
    a.x = in_c8; a.g = grad_out; a.x_absmax = in_absmax; a.g_absmax = grad_absmax;
    a.partial = static_cast<float *>(workspace); a.gw = grad_weight;
    a.B = B; a.D = D; a.H = H; a.W = W;
    a.nseg = (W + 31) / 32;
    a.nchunks = (int)((int64_t)B * D * H * a.nseg);
    a.nwaves = wgrad_f16_waves();
    hipStream_t st = as_stream(stream);
    if (W % 32 == 0)
        hipLaunchKernelGGL((conv3d_wgrad_c8_f16_kernel<true>), dim3((unsigned)(a.nwaves / 4)), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((conv3d_wgrad_c8_f16_kernel<false>), dim3((unsigned)(a.nwaves / 4)), dim3(256), 0, st, a);
    hipLaunchKernelGGL(conv3d_wgrad_c8_f16_reduce_kernel, dim3(kWFSlots / 256, 32), dim3(256), 0, st, a);
    return check_launch("mvs_conv3d_wgrad_c8_f16_f32");
}
