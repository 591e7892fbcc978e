// 3x3(x3) convolutions with Cout = 16 or 32 per launch (Cin = 16, 32, 64; stride 2: Cin = 8, 16, 32, and FeatureNet's 5x5
// layers 8 -> 16, 16 -> 32 of mvsnet.py:13,16; channels-last) on the BF16
// matrix pipe at FP32 accuracy: CostRegNet's conv2 / conv4 / conv6 (mvsnet.py:68-72), FeatureNet's 16 -> 16 and
// 32 -> 32 layers (mvsnet.py:21-27), the same-shaped layers of the cascade (CasMVSNet/models/module.py) and of
// CVP-MVSNet (net.py:22-97).  Operand splitting and error bound: conv_split_common.h / conv_bf16x6.hip.
//
// One step = one 8-channel chunk of one output tile.  A workgroup = 8 multiplying waves + 4 copy waves:
//   copy waves   request the fp32 halo of the NEXT step by LDS-DMA (`buffer_load ... lds`; a copy blocks the wave
//                that issues it for ~250 cycles, so the multiplying waves issue none), and at a chunk change the
//                next weight chunk; wait for them; take part in the split pass
//   barrier      the halo has landed, every wave is done with the previous step's bf16 parts
//   split pass   all 12 waves: every halo voxel once, fp32 -> hi, mid, lo -> three bf16 planes in LDS
//   barrier
//   MFMA phase   one v_mfma_f32_16x16x32_bf16 = 4 taps x 8 channels (K = 32) x 16 output channels x 16 voxels
//                along x; per (row block, tap group) the B fragment is three ds_read_b128 (one per part), per
//                (tap group, 16 output channels) the A fragment three more; six MFMAs per pair
// Tiles of a group share the chunk's weights in LDS and keep their accumulators in registers over the chunk loop.
#include <algorithm>
#include <vector>

#include "conv_split_common.h"
#include "conv_guard.h"

#include <cstdlib>

namespace mvs {

struct SplitArgs {
    const float *in;          // [B, D, H, W, Cin]
    const unsigned char *wpk; // [chunk][group][m-tile][part][lane][8 bf16]
    const float *scale, *shift, *residual;   // per output channel (of this launch) / [B, D, H, W, ldc] like out
    float *out;               // [B, D, H, W, ldc], this launch writes channels [0, COUT)
    int B, D, H, W, ldc;      // D, H, W: input dims
    int Do, Ho, Wo;           // output dims (= input dims for stride 1)
    int tiles_x, tiles_y, tiles_z, ystrip;
    int relu;                 // 0 none, 1 ReLU, 2 LeakyReLU(0.1)
    int nco;                  // output channels of this launch that exist (8 for an 8-channel layer on the 16-row tile)
    int co0, out_c4;          // out_c4: the WHOLE output tensor is [B*D, ldc/4, H, W, 4] (4-channel blocks), co0 = first channel of the launch
    // two-piece fp16 form (NP = 2): the input's absmax block (mvs_common.h), what undoes the weights' scale
    // (device float, the trailer of the packed weights)
    const unsigned *in_absmax;
    const float *w_iscale;
    unsigned *out_absmax;     // NULL, or the absmax block that collects the largest magnitude this launch stores (any form)
    // NP = 2, the range guard (conv_guard.h): the layer's original fp32 weights (PyTorch layout, behind the trailer of the
    // packed weights) and the device counter of launches that fell back to them
    const float *w_f32;
    unsigned long long *guard_cnt;
    const unsigned *run_flag;   // NULL, or a device word: the launch returns at once when it is 0 (mvs_common.h: conv_run_flag)
};

// NP: operand pieces -- 3 = bf16 hi/mid/lo, six products (exact split); 2 = scaled fp16 hi/lo, three products (conv_f16x3.hip
// has the arithmetic and its error bound)
template <int CIN_, int COUT_, int KD_, int S_ = 1, int KH_ = 3, int NP_ = 3>
struct SplitCfg {
    static constexpr int NP = NP_;
    // S: stride (2: the 3D down-sampling layers, and FeatureNet's 5x5 layers = KD 1, KH 5)
    static constexpr int CIN = CIN_, COUT = COUT_, KD = KD_, S = S_, KH = KH_;
    // 8-channel chunks per step: two for the 2D layers (a K = 32 step is then 2 taps x 16 channels: 9 taps fill 18 of
    // 20 slots instead of 9 of 12, and a tile has half as many steps -- barriers, split passes -- for the same MFMAs)
    static constexpr int CPS = (KD == 1 && S == 1 && CIN_ >= 16) ? 2 : 1;
    static constexpr int NCHUNK = CIN / (8 * CPS), MT = COUT / 16;
    static constexpr int NTAP = KD * KH * KH, NSLOT = NTAP * CPS, G = (NSLOT + 3) / 4;
    // output tile (TZ, TY, 16 XB) and its halo
    // (Cout 32 in 3D: two weight chunks of 42 KiB leave room for the smaller halo only; stride 2: the halo of a (2, 4, 16)
    // output tile is 5 x 9 x 33 input voxels)
    // (2D stride 2, 5x5: a 16 x 16 output tile needs 35 x 35 inputs; with 32 output channels the weights leave room for 8 rows)
    static constexpr int TZ = KD == 3 ? (S == 2 ? 2 : 4) : 1;
    static constexpr int TY = KD == 3 ? (S == 2 || COUT_ == 32 ? 4 : 8) : (S == 2 && COUT_ == 32 ? 8 : 16);
    static constexpr int XB = (KD == 3 || S == 2) ? 1 : 2, TX = 16 * XB;
    static constexpr int ZT = (TZ - 1) * S + KD, YT = (TY - 1) * S + KH, XP = (TX - 1) * S + KH, NVOX = ZT * YT * XP;
    static constexpr int RB = TZ * TY * XB, RPW = RB / 8;                 // 16-voxel row blocks, per multiplying wave
    // voxels per chunk plane: a multiple of 16 when a step holds two chunks -- the two 8-lane halves of a ds_read_b128
    // service group (same tap, chunk 0 / chunk 1) then read voxels n .. and 16 k + n ..: complementary 16-byte slots
    static constexpr int NVP = CPS == 1 ? NVOX : (NVOX + 15) / 16 * 16;
    static constexpr int NPIECE = 2 * NVP * CPS, NCOPY = (NPIECE + 63) / 64;   // 16-byte pieces (chunk, voxel, channel half)
    static constexpr int FBYTES = NCOPY * 1024, SPART = NVP * 16 * CPS, SBYTES = NP * SPART;
    static constexpr int WBYTES = G * MT * NP * 1024;                     // A fragments of one chunk
    // tiles per group (they share a chunk's weights and hold their accumulators, RPW x MT x 4 registers each, over the
    // chunk loop): bounded by the 168 registers of a 12-wave workgroup
    static constexpr int T = 48 / (RPW * MT * 4) >= 2 ? 2 : 1;
    // copy waves: a stride-2 step carries 47 copies for 42 MFMAs per wave -- eight copy waves (16 waves, 128 registers)
    static constexpr int NCW = S == 2 ? 8 : 4, NTHREADS = 512 + 64 * NCW;
    // DB2 (two-piece form, where LDS has room for a second staging buffer): the copies run two STEPS ahead and have a whole step
    // to land.  Behind the second barrier of a step with a
    // single buffer they had only the MFMA phase, which the two-piece form halved: the stride-2 layer conv1 (47 KiB of halo
    // per step) waited for them every step (0.28 ms for 1.0 GB)
    static constexpr int NWBUF = NCHUNK == 1 ? 1 : 2;      // (one chunk per tile: the weights never change)
    // (measured, profiles/r03_split_f16.json: conv1 0.268 -> 0.230 ms, conv4 0.081 -> 0.076, FeatureNet's 5x5 8 -> 16 layer 0.135 -> 0.120;
    // the multi-chunk 2D layers lose 5 %: they keep the single buffer)
    static constexpr bool DB2 = NP_ == 2 && (NCHUNK == 1 || KD_ == 3) && NWBUF * WBYTES + 2 * FBYTES + SBYTES + 2 * COUT * 4 <= 160 * 1024;
    static constexpr int F_OFF = (DB2 ? NWBUF : 2) * WBYTES, S_OFF = F_OFF + (DB2 ? 2 : 1) * FBYTES, AFF_OFF = S_OFF + SBYTES;   // + scale, shift of the launch
    static constexpr int LDS_BYTES = AFF_OFF + 2 * COUT * 4;
    // (3D stride 2: 16 output channels per launch in the bf16 form -- its three part planes leave no room for a second M tile;
    // 32 in the two-piece form, round 4: conv3 / conv5 then read and split their input once per 32 output channels)
    static_assert(RB % 8 == 0 && LDS_BYTES <= 160 * 1024 && SPART * 2 + 16 * 1024 < 65536 && (S == 1 || KD == 1 || MT == 1 || (MT == 2 && NP_ == 2)) &&
                  (KH == 3 || (KH == 5 && KD == 1 && S == 2)), "tile / LDS budget");
};


__device__ __forceinline__ TileIdx split_decode(const SplitArgs &a, int bid) {
    TileIdx t;
    const int per_b = a.tiles_x * a.tiles_y * a.tiles_z;
    t.b = bid / per_b; bid -= t.b * per_b;
    const int full = a.ystrip * a.tiles_z * a.tiles_x;
    const int s = bid / full; bid -= s * full;
    const int y0 = s * a.ystrip;
    const int hs = min(a.ystrip, a.tiles_y - y0);   // the last strip may be short
    t.ty = y0 + bid % hs; bid /= hs;
    t.tz = bid % a.tiles_z;
    t.tx = bid / a.tiles_z;
    return t;
}

// LAPS (tuning build, MVS_CONV_SPLIT_LAPS=1): cycles of each multiplying wave per phase, summed over the workgroup's steps,
// into the buffer passed as `residual` (int64 [workgroup][wave][8]): barrier, split pass, barrier, MFMA phase, epilogue
template <class C, bool LAPS = false>
__global__ __launch_bounds__(C::NTHREADS) void conv_split_kernel(SplitArgs a, int ntiles) {
    constexpr int CIN = C::CIN, NCHUNK = C::NCHUNK, MT = C::MT, G = C::G, T = C::T, RPW = C::RPW;
    constexpr int YT = C::YT, XP = C::XP, NVOX = C::NVOX, NPIECE = C::NPIECE, NCOPY = C::NCOPY;
    constexpr int SPART = C::SPART, WBYTES = C::WBYTES, F_OFF = C::F_OFF, S_OFF = C::S_OFF;
    constexpr int NC = C::NCW, IPW = (NCOPY + NC - 1) / NC, NT = C::NTHREADS;
    constexpr int NWC = (WBYTES / 1024 + NC - 1) / NC;            // weight copies per copy wave
    __shared__ __attribute__((aligned(16))) unsigned char lds[C::LDS_BYTES];

    if (a.run_flag && *a.run_flag == 0u) return;      // (uniform: a fused kernel in front of this launch did the work)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4;
    const unsigned lds_base = (unsigned)(uintptr_t)lds;
    const bool copier = wv >= 8;
    const int cw = wv - 8;
    constexpr int NP = C::NP;
    // NP = 2: operand scale of the input; what undoes it and the weights' scale goes into the epilogue's per-channel scale
    float sx = 1.0f, unscale = 1.0f, unscale2 = 1.0f;
    if constexpr (NP == 2) {
        const AbsmaxVerdict verdict = absmax_verdict(a.in_absmax);
        const int xe = absmax_exponent(verdict.bits);
        const float isw = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, *a.w_iscale)));
        sx = pow2f(14 - xe);
        // what undoes both operand scales is 2^E: the part within +-60 rides in the per-channel scale of the epilogue, the rest
        // (inputs or weights beyond 1e+-18: folded into one float it would underflow) is a second multiply there
        const int E = xe - 14 + (int)((__builtin_bit_cast(unsigned, isw) >> 23) & 255u) - 127;
        const int e1 = E < -60 ? -60 : (E > 60 ? 60 : E), e2 = E - e1 < -126 ? -126 : (E - e1 > 127 ? 127 : E - e1);
        unscale = pow2f(e1);
        unscale2 = pow2f(e2);
        // the range guard (conv_guard.h): a non-finite or outlier-dominated input, or non-finite weights -> plain fp32
        if (verdict.code != 0 || isw != isw) {
            GuardConv g;
            g.in = a.in; g.w = a.w_f32; g.scale = a.scale; g.shift = a.shift; g.residual = a.residual; g.out = a.out;
            g.out_absmax = a.out_absmax; g.counter = a.guard_cnt;
            g.B = a.B; g.D = a.D; g.H = a.H; g.W = a.W; g.Cin = CIN; g.Do = a.Do; g.Ho = a.Ho; g.Wo = a.Wo;
            g.ldc = a.ldc; g.co0 = a.co0; g.nco = a.nco; g.kd = C::KD; g.kh = C::KH; g.stride = C::S; g.transposed = 0;
            g.relu = a.relu; g.in_c8 = 0; g.out_c4 = a.out_c4;
            guard_direct_conv(g);
            return;
        }
    }
    // the per-channel affine of the epilogue waits in LDS (a global load there would put its latency into every tile)
    if (tid < 2 * C::COUT) {
        const int c = tid % C::COUT;
        const float v = c >= a.nco ? 0.0f : tid < C::COUT ? (a.scale ? a.scale[c] : 1.0f) * unscale : (a.shift ? a.shift[c] : 0.0f);
        *reinterpret_cast<float *>(lds + C::AFF_OFF + tid * 4) = v;
    }

    // this workgroup's tiles: t0 + k * t_step, k < ntw (XCD x owns a contiguous range of the ordered tile list)
    int t0, t_step, ntw;
    {
        const int nb = gridDim.x;
        if ((nb & 7) == 0) {
            const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, per = nb >> 3;
            const int lo = (int)((int64_t)ntiles * xcd / 8), hi = (int)((int64_t)ntiles * (xcd + 1) / 8);
            t0 = lo + j; t_step = per; ntw = (hi - t0 + per - 1) / per;
        } else {
            t0 = blockIdx.x; t_step = nb; ntw = (ntiles - t0 + nb - 1) / nb;
        }
        if (ntw < 0) ntw = 0;
    }

    // split pass (all waves): piece P = ps*NT + tid = (voxel P >> 1, channel half P & 1): 16 bytes of the fp32 buffer
    // -> 8 bytes of each bf16 part at the same position
    constexpr int NPS = (NPIECE + NT - 1) / NT;
    auto split_pass = [&](int par = 0) {     // par: the staging buffer of this step (DB2)
        f32x4 x[NPS];
        const unsigned fp = lds_base + (unsigned)(F_OFF + par * C::FBYTES + tid * 16), sp = lds_base + (unsigned)(S_OFF + tid * 8);
        const unsigned sp1 = sp + (unsigned)SPART, sp2 = sp + (unsigned)(2 * SPART);     // (offset field: 16 bits)
        static_for<0, NPS>([&](auto pc) {
            constexpr int ps = decltype(pc)::value;
            x[ps] = lds_read_b128<ps * NT * 16>(fp);
        });
        lds_wait_n<0>();
        static_for<0, (NPS + 1) / 2>([&](auto pc) {
            constexpr int p0 = 2 * decltype(pc)::value, p1 = (p0 + 1 < NPS) ? p0 + 1 : p0;
            f32x4 &x0 = x[p0], &x1 = x[p1];
            asm volatile("" : "+v"(x0), "+v"(x1));
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            u32x4 hu, mu, lu;
            if constexpr (NP == 2) {
                split2_block(x0, x1, sx, hu, mu);
            } else {
                bf16x8 h, m, l;
                split3_block(x0, x1, h, m, l);
                hu = __builtin_bit_cast(u32x4, h); mu = __builtin_bit_cast(u32x4, m); lu = __builtin_bit_cast(u32x4, l);
            }
            if (p0 * NT + tid < NPIECE) {
                lds_write_b64<p0 * NT * 8>(sp, hu[0], hu[1]);
                lds_write_b64<p0 * NT * 8>(sp1, mu[0], mu[1]);
                if constexpr (NP == 3) lds_write_b64<p0 * NT * 8>(sp2, lu[0], lu[1]);
            }
            if (p1 != p0 && p1 * NT + tid < NPIECE) {
                lds_write_b64<p1 * NT * 8>(sp, hu[2], hu[3]);
                lds_write_b64<p1 * NT * 8>(sp1, mu[2], mu[3]);
                if constexpr (NP == 3) lds_write_b64<p1 * NT * 8>(sp2, lu[2], lu[3]);
            }
        });
        lds_wait_n<0>();
    };

    if (copier) {
        // ================================================================ copy waves
        // copy g = i*NC + cw brings pieces g*64 + lane: (voxel, half) of the halo, voxel = (lz, ly, lx) x-fastest
        int loc[IPW];
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            const int P = (i * NC + cw) * 64 + lane;
            const bool ok = P < NPIECE;
            const int vraw = (P % (2 * C::NVP)) >> 1, c = ok ? P / (2 * C::NVP) : 0;
            const bool okv = ok && vraw < NVOX;          // (the padding voxels of a chunk plane arrive as zeros)
            const int v = okv ? vraw : 0;
            loc[i] = (v % XP) | (((v / XP) % YT) << 8) | ((v / (XP * YT)) << 16) | ((P & 1) << 24) | (c << 25) |
                     (okv ? 0 : (int)0x80000000);
        }
        const int64_t plane_in = (int64_t)a.H * a.W * CIN;
        const int row_in = a.W * CIN;
        const unsigned window_bytes = (unsigned)min((int64_t)C::ZT * plane_in * 4, (int64_t)0xffffff00u);
        unsigned voff[T][IPW];
        mvs_srd_t srd[T];
        auto geometry = [&](auto jc, int t) {
            constexpr int j = decltype(jc)::value;
            const TileIdx tile = split_decode(a, t);
            const int ix0 = tile.tx * C::TX * C::S - C::KH / 2, iy0 = tile.ty * C::TY * C::S - C::KH / 2;
            const int iz0 = C::KD == 3 ? tile.tz * C::TZ * C::S - 1 : tile.tz * C::TZ;      // (images are not strided)
            srd[j] = make_srd(a.in + ((int64_t)tile.b * a.D + iz0) * plane_in, window_bytes);
#pragma unroll
            for (int i = 0; i < IPW; ++i) {
                const int gx = ix0 + (loc[i] & 255), gy = iy0 + ((loc[i] >> 8) & 255);
                const int lz = (loc[i] >> 16) & 255, h = (loc[i] >> 24) & 1, c = (loc[i] >> 25) & 3;
                const bool ok = loc[i] >= 0 && (unsigned)gx < (unsigned)a.W && (unsigned)gy < (unsigned)a.H &&
                                (unsigned)(iz0 + lz) < (unsigned)a.D;
                voff[j][i] = ok ? (unsigned)(((int64_t)lz * plane_in + (int64_t)gy * row_in + gx * CIN + c * 8 + h * 4) * 4) : 0xffffff00u;
            }
        };
        auto issue_halo = [&](auto jc, int ch) {
            constexpr int j = decltype(jc)::value;
            const unsigned soff = (unsigned)(ch * 32 * C::CPS);
#pragma unroll
            for (int i = 0; i < IPW; ++i) {
                if (i * NC + cw >= NCOPY) continue;   // wave-uniform
                glds16_buf(voff[j][i], srd[j], soff, lds_base + (unsigned)(F_OFF + (i * NC + cw) * 1024));
            }
        };
        auto issue_weights = [&](int ch, int sel) {
#pragma unroll
            for (int i = 0; i < NWC; ++i) {
                const int g = i * NC + cw;
                if (g < WBYTES / 1024) glds16(a.wpk + (size_t)ch * WBYTES + (size_t)g * 1024 + lane * 16,
                                              lds_base + (unsigned)(sel * WBYTES + g * 1024));
            }
        };
        if constexpr (C::DB2) {
            // step s of this workgroup -> staging buffer s & 1; behind the second barrier of step s: wait for the copies of step
            // s + 1 (issued a step ago), request step s + 2.  The issue iterator walks (group, chunk, tile) like the consumers.
            int it_k0 = 0, it_ch = 0, it_j = 0, it_par = 0;
            bool it_done = ntw <= 0;
            auto it_issue = [&]() {
                if (it_done) return;
                // (a tile's offsets are computed with its first chunk and kept for the others: slot = tile of the group)
                static_for<0, T>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    if (it_j != j) return;   // wave-uniform
                    if (it_ch == 0) geometry(jc, t0 + (it_k0 + j) * t_step);
                    const unsigned soff = (unsigned)(it_ch * 32 * C::CPS);
#pragma unroll
                    for (int q = 0; q < IPW; ++q) {
                        if (q * NC + cw >= NCOPY) continue;   // wave-uniform
                        glds16_buf(voff[j][q], srd[j], soff, lds_base + (unsigned)(F_OFF + it_par * C::FBYTES + (q * NC + cw) * 1024));
                    }
                });
                it_par ^= 1;
                if (++it_j >= min(T, ntw - it_k0)) {
                    it_j = 0;
                    if (++it_ch >= NCHUNK) {
                        it_ch = 0;
                        it_k0 += T;
                        if (it_k0 >= ntw) it_done = true;
                    }
                }
            };
            int wsel = 0, par = 0;
            if (ntw > 0) {
                issue_weights(0, 0);
                it_issue();
                it_issue();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            for (int k0 = 0; k0 < ntw; k0 += T) {
                const int nvalid = min(T, ntw - k0);
#pragma unroll 1
                for (int ch = 0; ch < NCHUNK; ++ch) {
                    // the weights of the next chunk (of this group, or chunk 0 of the next) go out behind this chunk's first step, when
                    // the buffer of the previous chunk is free; a single-tile group needs them by its very next step
                    const bool more = NCHUNK > 1 && (ch + 1 < NCHUNK || k0 + T < ntw);
                    const int nch = ch + 1 < NCHUNK ? ch + 1 : 0;
#pragma unroll 1
                    for (int j = 0; j < nvalid; ++j) {
                        __syncthreads();
                        split_pass(par);
                        __syncthreads();
                        par ^= 1;
                        if (more && j == 0 && nvalid == 1) issue_weights(nch, wsel ^ 1);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        if (more && j == 0 && nvalid > 1) issue_weights(nch, wsel ^ 1);
                        it_issue();
                    }
                    if (NCHUNK > 1) wsel ^= 1;
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            return;
        }
        int wsel = 0;
        if (ntw > 0) {
            geometry(std::integral_constant<int, 0>{}, t0);
            issue_halo(std::integral_constant<int, 0>{}, 0);
            issue_weights(0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        for (int k0 = 0; k0 < ntw; k0 += T) {
            const int nvalid = min(T, ntw - k0);
#pragma unroll 1
            for (int ch = 0; ch < NCHUNK; ++ch) {
                static_for<0, T>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    if (j >= nvalid) return;   // wave-uniform
                    __syncthreads();           // the halo of this step is in the fp32 buffer (this wave has waited for it)
                    split_pass();
                    __syncthreads();           // ... and has been split: the fp32 buffer is free
                    if (j + 1 < nvalid) {
                        if (ch == 0) geometry(std::integral_constant<int, (j + 1) % T>{}, t0 + (k0 + j + 1) * t_step);
                        issue_halo(std::integral_constant<int, (j + 1) % T>{}, ch);
                    } else if (ch + 1 < NCHUNK) {
                        issue_halo(std::integral_constant<int, 0>{}, ch + 1);
                        issue_weights(ch + 1, wsel ^ 1);
                    } else if (k0 + T < ntw) {
                        geometry(std::integral_constant<int, 0>{}, t0 + (k0 + T) * t_step);
                        issue_halo(std::integral_constant<int, 0>{}, 0);
                        if (NCHUNK > 1) issue_weights(0, wsel ^ 1);      // (one chunk per tile: the weights never change)
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                });
                if (NCHUNK > 1) wsel ^= 1;
            }
        }
        return;
    }

    // ================================================================ multiplying waves
    // row block rb = wv*RPW + r -> (z, y, x block); this lane's B voxel of tap (dz, dy, dx): (z+dz, y+dy, xb*16 + n + dx)
    unsigned rbo[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int rb = wv * RPW + r;
        const int z = rb / (C::TY * C::XB), y = (rb / C::XB) % C::TY, xb = rb % C::XB;
        rbo[r] = (unsigned)(((z * C::S * YT + y * C::S) * XP + (xb * 16 + n) * C::S) * 16);
    }
    // slot 4 g + kq of this lane's K group = (tap, chunk of the step); slots past the kernel read voxel 0 against
    // zero weights
    unsigned tapo[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int sl = 4 * g + kq, t = sl / C::CPS, c = sl % C::CPS;
        constexpr int KK = C::KH * C::KH;
        const int dz = C::KD == 3 ? t / KK : 0, dy = (t % KK) / C::KH, dx = t % C::KH;
        tapo[g] = sl < C::NSLOT ? (unsigned)((((dz * YT + dy) * XP + dx) + c * C::NVP) * 16) : 0u;
    }

    f32x4 acc[T][RPW][MT];
#pragma unroll
    for (int j = 0; j < T; ++j)
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[j][r][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // split_decode without integer division: quotients through float reciprocals plus one correction step, exact for
    // tile indices below 2^23 (the launcher guarantees it)
    const int per_b = a.tiles_x * a.tiles_y * a.tiles_z, full = a.ystrip * a.tiles_z * a.tiles_x;
    const float r_per_b = 1.0f / (float)per_b, r_full = 1.0f / (float)full, r_tz = 1.0f / (float)a.tiles_z;
    const int hs_last = a.tiles_y % a.ystrip ? a.tiles_y % a.ystrip : a.ystrip;
    const float r_hs = 1.0f / (float)a.ystrip, r_hs_last = 1.0f / (float)hs_last;
    auto divmod = [](int nn, int d, float rd, int &q, int &r) {
        q = (int)((float)nn * rd);
        r = nn - q * d;
        if (r < 0) { q -= 1; r += d; } else if (r >= d) { q += 1; r -= d; }
    };
    auto fast_decode = [&](int t) {
        TileIdx ti;
        int rem, strip, rem2, tyl, rem3;
        divmod(t, per_b, r_per_b, ti.b, rem);
        divmod(rem, full, r_full, strip, rem2);
        const int ys = strip * a.ystrip;
        const bool last = ys + a.ystrip > a.tiles_y;
        divmod(rem2, last ? hs_last : a.ystrip, last ? r_hs_last : r_hs, rem3, tyl);
        divmod(rem3, a.tiles_z, r_tz, ti.tx, ti.tz);
        ti.ty = ys + tyl;
        return ti;
    };
    // in-tile position (z << 16 | y << 8 | x) and element offset of this lane's store for each of its row blocks; out_c4: a
    // lane's four channels are one block of [image, C/4, H, W, 4] (the sweep kernel's input), else channels-last
    const int64_t hw4 = (int64_t)a.Ho * a.Wo * 4, mstep = a.out_c4 ? 4 * hw4 : 16;
    int rpos[RPW], loff[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int rb = wv * RPW + r;
        const int zr = rb / (C::TY * C::XB), yr = (rb / C::XB) % C::TY, xr = (rb % C::XB) * 16 + n;
        rpos[r] = (zr << 16) | (yr << 8) | xr;
        loff[r] = a.out_c4 ? (int)(zr * (a.ldc >> 2) * hw4 + (yr * a.Wo + xr) * 4 + kq * hw4)
                           : ((zr * a.Ho + yr) * a.Wo + xr) * a.ldc + kq * 4;
    }
    float vmax = 0.0f;          // largest magnitude this lane has stored (-> a.out_absmax)
    long long tsum[5] = {0, 0, 0, 0, 0};
    long long tprev = 0;
    if constexpr (LAPS) tprev = clock64();
#define MVS_LAP(k) do { if constexpr (LAPS) { const long long tn = clock64(); tsum[k] += tn - tprev; tprev = tn; } } while (0)
    int wsel = 0, spar = 0;     // spar: staging buffer of the step (DB2)
    for (int k0 = 0; k0 < ntw; k0 += T) {
        const int nvalid = min(T, ntw - k0);
#pragma unroll 1
        for (int ch = 0; ch < NCHUNK; ++ch) {
            static_for<0, T>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                if (j >= nvalid) return;   // wave-uniform
                MVS_LAP(4);
                __syncthreads();
                MVS_LAP(0);
                split_pass(C::DB2 ? spar : 0);
                spar ^= 1;
                MVS_LAP(1);
                __syncthreads();
                MVS_LAP(2);
                // ---- MFMA phase: items (g, r) = (tap group, row block); the reads of the next item go out before the
                // MFMAs of the current one
                const unsigned aA = lds_base + (unsigned)(wsel * WBYTES + lane * 16);
                const unsigned aS = lds_base + (unsigned)S_OFF;
                bf16x8 A[2][MT][NP], Bf[2][NP];     // (NP = 2: the same registers hold fp16 pairs)
                auto read_a = [&](auto gc) {
                    constexpr int g = decltype(gc)::value;
                    static_for<0, MT * NP>([&](auto ic) {
                        constexpr int i = decltype(ic)::value;
                        A[g & 1][i / NP][i % NP] = __builtin_bit_cast(bf16x8, lds_read_b128<(g * MT * NP + i) * 1024>(aA));
                    });
                };
                auto read_b = [&](auto ic) {
                    constexpr int it = decltype(ic)::value, g = it / RPW, r = it % RPW;
                    const unsigned ad = aS + rbo[r] + tapo[g];
                    static_for<0, NP>([&](auto pc) {
                        constexpr int sp = decltype(pc)::value;
                        Bf[it & 1][sp] = __builtin_bit_cast(bf16x8, lds_read_b128<sp * SPART>(ad));
                    });
                };
                read_a(std::integral_constant<int, 0>{});
                read_b(std::integral_constant<int, 0>{});
                static_for<0, G * RPW>([&](auto ic) {
                    constexpr int it = decltype(ic)::value, g = it / RPW, r = it % RPW;
                    lds_wait_n<0>();
                    static_for<0, NP>([&](auto pc) {
                        bf16x8 &b0 = Bf[it & 1][decltype(pc)::value];
                        asm volatile("" : "+v"(b0));
                    });
                    if constexpr (r == 0) {
                        static_for<0, MT * NP>([&](auto qc) {
                            bf16x8 &aa = A[g & 1][decltype(qc)::value / NP][decltype(qc)::value % NP];
                            asm volatile("" : "+v"(aa));
                        });
                    }
                    if constexpr (it + 1 < G * RPW) {
                        if constexpr (r == RPW - 1) read_a(std::integral_constant<int, g + 1>{});
                        read_b(std::integral_constant<int, it + 1>{});
                    }
                    __builtin_amdgcn_sched_barrier(0);   // the reads go out BEFORE this item's MFMAs
                    if constexpr (NP == 2) {
                        // three partial products per 16 output channels, small terms first: al bh, ah bl, ah bh
                        static_for<0, 3>([&](auto tc) {
                            constexpr int t = decltype(tc)::value;
                            static_for<0, MT>([&](auto mc) {
                                constexpr int m = decltype(mc)::value;
                                const f16x8 bb = __builtin_bit_cast(f16x8, Bf[it & 1][t == 1 ? 1 : 0]);
                                const f16x8 aa = __builtin_bit_cast(f16x8, A[g & 1][m][t == 0 ? 1 : 0]);
                                f32x4 &cc = acc[j][r][m];
                                cc = __builtin_amdgcn_mfma_f32_16x16x32_f16(aa, bb, cc, 0, 0, 0);
                            });
                        });
                    } else {
                    const bf16x8 bh = Bf[it & 1][0], bm = Bf[it & 1][1], bl = Bf[it & 1][NP - 1];
                    // six partial products per 16 output channels, small terms first; M tiles interleaved
                    static_for<0, 6>([&](auto tc) {
                        constexpr int t = decltype(tc)::value;
                        constexpr int as = (t == 0 || t == 3) ? 1 : (t == 1 ? 2 : 0);       // am al ah am ah ah
                        static_for<0, MT>([&](auto mc) {
                            constexpr int m = decltype(mc)::value;
                            const bf16x8 &bb = (t == 0 || t == 4) ? bm : (t == 2 ? bl : bh);   // bm bh bl bh bm bh
                            const bf16x8 &aa = A[g & 1][m][as < NP ? as : 0];
                            f32x4 &cc = acc[j][r][m];
                            cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aa, bb, cc, 0, 0, 0);
                        });
                    });
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
                if constexpr (LAPS) {
                    f32x4 &c0 = acc[j][0][0];
                    asm volatile("" : "+v"(c0));
                    asm volatile("s_nop 0" ::: "memory");
                }
                MVS_LAP(3);
            });
            if (NCHUNK > 1) wsel ^= 1;
        }
        // ---- epilogue of the group: per-channel affine, activation, skip add, one 16-byte store per lane, row block, M tile.
        // (A wave issues an instruction every ~7 cycles and the matrix pipe idles meanwhile: the tile's base address is
        // scalar arithmetic, a lane adds its precomputed in-tile offset -- 120 instructions per store became ~30.)
        static_for<0, T>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if (j >= nvalid) return;
            const TileIdx cur = fast_decode(t0 + (k0 + j) * t_step);
            const int tb = __builtin_amdgcn_readfirstlane(cur.b);
            const int oz0 = __builtin_amdgcn_readfirstlane(cur.tz) * C::TZ, oy0 = __builtin_amdgcn_readfirstlane(cur.ty) * C::TY;
            const int ox0 = __builtin_amdgcn_readfirstlane(cur.tx) * C::TX;
            const int64_t base = a.out_c4
                ? (((int64_t)tb * a.Do + oz0) * (a.ldc >> 2) + (a.co0 >> 2)) * hw4 + ((int64_t)oy0 * a.Wo + ox0) * 4
                : ((((int64_t)tb * a.Do + oz0) * a.Ho + oy0) * a.Wo + ox0) * a.ldc;
            float *const ob = a.out + base;
            const float *const rp = (a.residual && !LAPS) ? a.residual + base : nullptr;
            float4 sc[MT], sh[MT];      // (read here, once per tile: held over the MFMA phase they cost 8 registers per M tile)
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int c0 = m * 16 + kq * 4;
                sc[m] = *reinterpret_cast<const float4 *>(lds + C::AFF_OFF + c0 * 4);
                sh[m] = *reinterpret_cast<const float4 *>(lds + C::AFF_OFF + (C::COUT + c0) * 4);
            }
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const bool inside = oz0 + (rpos[r] >> 16) < a.Do && oy0 + ((rpos[r] >> 8) & 255) < a.Ho && ox0 + (rpos[r] & 255) < a.Wo;
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    f32x4 v = acc[j][r][m];
                    acc[j][r][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (!inside || m * 16 + kq * 4 >= a.nco) continue;
                    if (unscale2 != 1.0f) { v[0] *= unscale2; v[1] *= unscale2; v[2] *= unscale2; v[3] *= unscale2; }   // (wave-uniform; extreme magnitudes only)
                    v[0] = v[0] * sc[m].x + sh[m].x; v[1] = v[1] * sc[m].y + sh[m].y;
                    v[2] = v[2] * sc[m].z + sh[m].z; v[3] = v[3] * sc[m].w + sh[m].w;
                    if (a.relu == 1) {
                        v[0] = relu_nan(v[0]); v[1] = relu_nan(v[1]);
                        v[2] = relu_nan(v[2]); v[3] = relu_nan(v[3]);
                    } else if (a.relu == 2) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] = v[q] > 0.f ? v[q] : v[q] * 0.1f;
                    }
                    const int64_t o = (int64_t)loff[r] + m * mstep;
                    if (rp) {
                        const float4 rs = *reinterpret_cast<const float4 *>(rp + o);
                        v[0] += rs.x; v[1] += rs.y; v[2] += rs.z; v[3] += rs.w;
                    }
                    *reinterpret_cast<float4 *>(ob + o) = make_float4(v[0], v[1], v[2], v[3]);
                    vmax = amax4_nan(vmax, v[0], v[1], v[2], v[3]);
                }
            }
        });
    }
    publish_absmax(a.out_absmax, vmax);
    if constexpr (LAPS) {
        MVS_LAP(4);
        if (lane == 0) {
            long long *dbg = reinterpret_cast<long long *>(const_cast<float *>(a.residual)) + ((int64_t)blockIdx.x * 8 + wv) * 8;
            for (int k = 0; k < 5; ++k) dbg[k] = tsum[k];
        }
    }
#undef MVS_LAP
}

// PyTorch-layout weight (Cout_total, Cin, [kd,] 3, 3), output channels [co0, co0 + COUT) ->
// [step][group][m-tile][part][lane][8 bf16]; lane (mrow, kq): output channel co0 + m*16 + mrow, slot 4 g + kq = (tap,
// chunk c of the step) (zero past the kernel), channel (step*cps + c)*8 + j
// one element of a packed bf16 hi / mid / lo fragment buffer: i = ((step * G + g) * MT + m) * 512 + lane * 8 + j
__device__ __forceinline__ void pack_split_element(const float *__restrict__ w, int Cin, int Cout, int ntap, int cps, int G, int MT, int co0,
                                                   unsigned short *__restrict__ out, int i) {
    const int j = i & 7, lane = (i >> 3) & 63;
    int rest = i >> 9;
    const int m = rest % MT; rest /= MT;
    const int g = rest % G, ch = rest / G;
    const int mrow = lane & 15, kq = lane >> 4, sl = 4 * g + kq, t = sl / cps, c = sl % cps;   // ch = step of the tile
    float x = 0.0f;
    if (t < ntap && co0 + m * 16 + mrow < Cout) x = w[((int64_t)(co0 + m * 16 + mrow) * Cin + (ch * cps + c) * 8 + j) * ntap + t];
    const __bf16 h = (__bf16)x;
    const float r1 = x - (float)h;
    const __bf16 mm = (__bf16)r1;
    const float r2 = r1 - (float)mm;
    const __bf16 l = (__bf16)r2;
    unsigned short *o = out + ((size_t)((ch * G + g) * MT + m) * 3) * 512 + lane * 8 + j;
    o[0] = __builtin_bit_cast(unsigned short, h);
    o[512] = __builtin_bit_cast(unsigned short, mm);
    o[1024] = __builtin_bit_cast(unsigned short, l);
}

__global__ __launch_bounds__(256) void pack_split_kernel(const float *__restrict__ w, int Cin, int Cout, int ntap, int cps, int G, int MT,
                                                         int co0, unsigned short *__restrict__ out, int total) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    pack_split_element(w, Cin, Cout, ntap, cps, G, MT, co0, out, i);
}

// Several packs in one launch (mvs_pack_batch_begin / _end): a training step packs every layer's weights again, 26 launches
// of a few microseconds of work each behind ~4 us of launch; recorded between begin and end they are ONE launch whose blocks
// look their job up in the kernel arguments.
struct PackSplitJob {
    const float *w;
    unsigned short *out;
    int Cin, Cout, ntap, cps, G, MT, co0, total, block0;
};
constexpr int kPackBatch = 24;
struct PackSplitBatch {
    PackSplitJob job[kPackBatch];
    int n;
};
__global__ __launch_bounds__(256) void pack_split_batch_kernel(PackSplitBatch b) {
    PackSplitJob q = b.job[0];
#pragma unroll
    for (int k = 1; k < kPackBatch; ++k)
        if (k < b.n && (int)blockIdx.x >= b.job[k].block0) q = b.job[k];     // (wave-uniform selects: no indexed copy of the table)
    const int i = ((int)blockIdx.x - q.block0) * 256 + threadIdx.x;
    if (i >= q.total) return;
    pack_split_element(q.w, q.Cin, q.Cout, q.ntap, q.cps, q.G, q.MT, q.co0, q.out, i);
}

struct PackRecorder {
    bool active = false;
    std::vector<PackSplitJob> jobs;
};
static PackRecorder &pack_recorder() {
    static thread_local PackRecorder r;
    return r;
}
static int flush_pack_jobs(PackRecorder &r, hipStream_t st) {
    for (size_t first = 0; first < r.jobs.size(); first += kPackBatch) {
        PackSplitBatch b;
        b.n = (int)std::min<size_t>(kPackBatch, r.jobs.size() - first);
        int blocks = 0;
        for (int k = 0; k < kPackBatch; ++k) {
            b.job[k] = r.jobs[first + (k < b.n ? k : 0)];
            if (k < b.n) { b.job[k].block0 = blocks; blocks += (b.job[k].total + 255) / 256; }
        }
        hipLaunchKernelGGL(pack_split_batch_kernel, dim3((unsigned)blocks), dim3(256), 0, st, b);
    }
    r.jobs.clear();
    return check_launch("mvs_pack_batch_end");
}

// the two-piece fp16 form of the same fragments: [step][group][m-tile][hi,lo][lane][8 fp16] of w * 2^(14 - exponent(max |w|))
// (the layer's largest weight, *wmax); block 0 also writes what undoes the scale into *iscale
__global__ __launch_bounds__(256) void pack_split_f16_kernel(const float *__restrict__ w, int Cin, int Cout, int ntap, int cps, int G, int MT,
                                                             int co0, unsigned short *__restrict__ out, int total,
                                                             const unsigned *__restrict__ wmax, float *__restrict__ iscale) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int e = absmax_exponent(*wmax);
    // what undoes the scale; NaN = "the weights are not finite", which sends every launch to the guard's fp32 path
    if (i == 0) *iscale = *wmax >= 0x7f800000u ? __builtin_nanf("") : pow2f(e - 14);
    if (i >= total) return;
    const int j = i & 7, lane = (i >> 3) & 63;
    int rest = i >> 9;
    const int m = rest % MT; rest /= MT;
    const int g = rest % G, ch = rest / G;
    const int mrow = lane & 15, kq = lane >> 4, sl = 4 * g + kq, t = sl / cps, c = sl % cps;   // ch = step of the tile
    float x = 0.0f;
    if (t < ntap && co0 + m * 16 + mrow < Cout)
        x = w[((int64_t)(co0 + m * 16 + mrow) * Cin + (ch * cps + c) * 8 + j) * ntap + t] * pow2f(14 - e);
    const _Float16 h = (_Float16)x;
    const _Float16 l = (_Float16)(x - (float)h);
    unsigned short *o = out + ((size_t)((ch * G + g) * MT + m) * 2) * 512 + lane * 8 + j;
    o[0] = __builtin_bit_cast(unsigned short, h);
    o[512] = __builtin_bit_cast(unsigned short, l);
}

template <class C>
static int launch_split(const SplitArgs &a0, hipStream_t st) {
    SplitArgs a = a0;
    a.Do = (a.D - 1) / (C::KD == 3 ? C::S : 1) + 1; a.Ho = (a.H - 1) / C::S + 1; a.Wo = (a.W - 1) / C::S + 1;
    a.tiles_x = (a.Wo + C::TX - 1) / C::TX; a.tiles_y = (a.Ho + C::TY - 1) / C::TY; a.tiles_z = (a.Do + C::TZ - 1) / C::TZ;
    a.ystrip = C::KD == 3 ? 4 : 2;
    const int64_t nt = (int64_t)a.B * a.tiles_x * a.tiles_y * a.tiles_z;
    if (nt <= 0 || nt >= (1 << 23)) return bare_error(MVS_EUNSUPPORTED, __func__, __LINE__);   // the kernel's float tile decode: callers fall back to mvs_conv3d_f32 / mvs_conv2d_f32
    const int n_cu = device_cu_count();
#ifdef MVS_TUNING   // phase-stamp build: int64 cycle counters written through `residual`
    static const bool laps = [] { const char *e = getenv("MVS_CONV_SPLIT_LAPS"); return e && e[0] == '1'; }();
#else
    constexpr bool laps = false;
#endif
    if (laps && C::NP == 3 && a.residual && (C::CIN == 64 || C::CIN == 16) && C::COUT == 16 * (C::CIN == 64 ? 2 : 1))   // tuning builds: 64 -> 32 and 16 -> 16
        hipLaunchKernelGGL((conv_split_kernel<C, true>), dim3((unsigned)(nt < n_cu ? nt : n_cu)), dim3(C::NTHREADS), 0, st, a, (int)nt);
    else
        hipLaunchKernelGGL((conv_split_kernel<C>), dim3((unsigned)(nt < n_cu ? nt : n_cu)), dim3(C::NTHREADS), 0, st, a, (int)nt);
    return check_launch("mvs_conv_split_f32");
}

}  // namespace mvs

using namespace mvs;

// output channels of ONE launch: 32 where the layer has a multiple of 32, else 16; stride 2: always 16 (its halo fills LDS)
// (kd = 1 with stride 2 is FeatureNet's 5x5 form: the whole layer, 16 or 32 output channels, in one launch)
static int split_cout_step(int kd, int Cout, int stride, int np) {
    return (stride == 1 || kd == 1 || np == 2) && Cout % 32 == 0 ? 32 : 16;
}
static int split_cps(int kd, int Cin, int stride) { return kd == 1 && stride == 1 && Cin >= 16 ? 2 : 1; }
static int split_ntap(int kd, int stride) { return kd == 1 && stride == 2 ? 25 : kd * 9; }

extern "C" int mvs_conv_split_supported(int kd, int Cin, int Cout, int stride) {
    // (FeatureNet's 8 -> 8 layer was tried on half-filled 16-row tiles: 0.24 ms against 0.20 for the fp32 shifted form)
    const bool cout_ok = Cout == 16 || Cout == 32 || Cout == 64;
    if (stride == 2 && kd == 1) return (Cin == 8 && Cout == 16) || (Cin == 16 && Cout == 32);   // FeatureNet's 5x5 layers
    if (stride == 2) return kd == 3 && cout_ok && (Cin == 8 || Cin == 16 || Cin == 32);     // conv1, conv3, conv5
    if (stride == 1 && kd == 3 && Cin == 8) return Cout == 32;      // the input gradient of a 32 -> 8 layer (conv0) in training
    return stride == 1 && (kd == 1 || kd == 3) && (Cin == 16 || Cin == 32 || Cin == 64) && cout_ok;
}

static size_t split_packed_bytes(int kd, int Cin, int Cout, int stride, int np) {
    if (!mvs_conv_split_supported(kd, Cin, Cout, stride)) return 0;
    const int cps = split_cps(kd, Cin, stride), G = (split_ntap(kd, stride) * cps + 3) / 4;
    return (size_t)(Cin / (8 * cps)) * G * ((Cout + 15) / 16) * np * 1024;
}

extern "C" size_t mvs_conv_split_packed_bytes(int kd, int Cin, int Cout, int stride) {
    return split_packed_bytes(kd, Cin, Cout, stride, 3);
}

// two-piece fp16 form: the fragments + a 16-byte trailer (what undoes the weights' scale; the weights' largest magnitude)
// + the original fp32 weights (the range guard's, conv_guard.h)
extern "C" size_t mvs_conv_split_f16_packed_bytes(int kd, int Cin, int Cout, int stride) {
    const size_t n = split_packed_bytes(kd, Cin, Cout, stride, 2);
    return n ? n + 16 + (size_t)Cout * Cin * split_ntap(kd, stride) * 4 : 0;
}

namespace mvs { int launch_absmax_word(const float *x, int64_t n, unsigned *word, hipStream_t st); }   // conv_f16x3.hip

extern "C" int mvs_conv_split_pack_weights_f16_f32(const float *weight, int kd, int Cin, int Cout, int stride, void *packed,
                                                   void *stream) {
    if (!weight || !packed || !mvs_conv_split_supported(kd, Cin, Cout, stride)) {
        set_error("mvs_conv_split_pack_weights_f16_f32: unsupported layer shape (see mvs_conv_split_supported)");
        return MVS_EINVAL;
    }
    const int ntap = split_ntap(kd, stride);
    const int step = split_cout_step(kd, Cout, stride, 2), cps = split_cps(kd, Cin, stride), G = (ntap * cps + 3) / 4, MT = step / 16;
    const size_t per_launch = (size_t)(Cin / (8 * cps)) * G * MT * 2 * 1024, body = split_packed_bytes(kd, Cin, Cout, stride, 2);
    unsigned char *pk = static_cast<unsigned char *>(packed);
    unsigned *wmax = reinterpret_cast<unsigned *>(pk + body + 4);
    const int rc = launch_absmax_word(weight, (int64_t)Cout * Cin * ntap, wmax, as_stream(stream));
    if (rc != MVS_OK) return rc;
    for (int co0 = 0; co0 < Cout; co0 += step) {
        const int total = (Cin / (8 * cps)) * G * MT * 512;
        hipLaunchKernelGGL(pack_split_f16_kernel, dim3((total + 255) / 256), dim3(256), 0, as_stream(stream), weight, Cin, Cout, ntap, cps, G, MT,
                           co0, reinterpret_cast<unsigned short *>(pk + (co0 / step) * per_launch), total, wmax,
                           reinterpret_cast<float *>(pk + body));
    }
    if (launch_guard_weights(weight, 0, Cin, Cout, ntap, reinterpret_cast<float *>(pk + body + 16), as_stream(stream)) != MVS_OK)
        return bare_error(MVS_ELAUNCH, __func__, __LINE__);
    return check_launch("mvs_conv_split_pack_weights_f16_f32");
}

extern "C" int mvs_conv_split_pack_weights_f32(const float *weight, int kd, int Cin, int Cout, int stride, void *packed,
                                               void *stream) {
    if (!weight || !packed || !mvs_conv_split_supported(kd, Cin, Cout, stride)) {
        set_error("mvs_conv_split_pack_weights_f32: needs a (Cout, Cin, [kd,] 3, 3) weight with kd in {1, 3}, Cin and Cout in {16, 32, 64} (or a supported stride-2 shape)");
        return MVS_EINVAL;
    }
    // one block of the packed buffer per launch of the layer (Cout / step launches, each `step` output channels)
    const int ntap = split_ntap(kd, stride);
    const int step = split_cout_step(kd, Cout, stride, 3), cps = split_cps(kd, Cin, stride), G = (ntap * cps + 3) / 4, MT = step / 16;
    const size_t per_launch = (size_t)(Cin / (8 * cps)) * G * MT * 3 * 1024;
    for (int co0 = 0; co0 < Cout; co0 += step) {
        const int total = (Cin / (8 * cps)) * G * MT * 512;
        unsigned short *dst = reinterpret_cast<unsigned short *>(static_cast<unsigned char *>(packed) + (co0 / step) * per_launch);
        PackRecorder &rec = pack_recorder();
        if (rec.active) {     // between mvs_pack_batch_begin and _end: one launch for all of them at _end
            PackSplitJob q = {weight, dst, Cin, Cout, ntap, cps, G, MT, co0, total, 0};
            rec.jobs.push_back(q);
            continue;
        }
        hipLaunchKernelGGL(pack_split_kernel, dim3((total + 255) / 256), dim3(256), 0, as_stream(stream), weight, Cin, Cout, ntap, cps, G, MT,
                           co0, dst, total);
    }
    return check_launch("mvs_conv_split_pack_weights_f32");
}

extern "C" int mvs_pack_batch_begin(void) {
    PackRecorder &rec = pack_recorder();
    if (rec.active) {
        set_error("mvs_pack_batch_begin: a batch is already open on this thread");
        return MVS_EINVAL;
    }
    rec.active = true;
    rec.jobs.clear();
    return MVS_OK;
}

extern "C" int mvs_pack_batch_end(void *stream) {
    PackRecorder &rec = pack_recorder();
    if (!rec.active) {
        set_error("mvs_pack_batch_end: no batch is open on this thread");
        return MVS_EINVAL;
    }
    rec.active = false;
    return flush_pack_jobs(rec, as_stream(stream));
}

// np = 3: the bf16 form; np = 2: the fp16 form (in_absmax required).  out_absmax: NULL, or the absmax block the largest
// magnitude of `out` is atomically max-ed INTO (the caller clears it: one memset serves the blocks of a whole network)
namespace mvs {
int launch_conv_s2_march(const float *in, const void *in_absmax, const void *packed, const float *w_iscale, const float *scale,
                         const float *shift, int relu, int B, int D, int H, int W, float *out, void *out_absmax,
                         unsigned long long *guard_cnt, const unsigned *run_flag, hipStream_t st);     // conv_s2_march.hip
}

static int conv_split_impl(const float *in, const void *in_absmax, const void *packed, const float *scale, const float *shift,
                           const float *residual, int relu, int kd, int stride, int B, int Cin, int Cout, int D, int H,
                           int W, int out_c4, float *out, void *out_absmax, int np, void *stream) {
    if (!in || (np == 2 && !in_absmax) || !packed || !out || B <= 0 || D <= 0 || H <= 0 || W <= 0 || relu < 0 || relu > 2 ||
        !mvs_conv_split_supported(kd, Cin, Cout, stride) || (out_c4 && residual)) {
        set_error("mvs_conv_split_f32: invalid argument (kd in {1, 3}; Cin, Cout in {16, 32, 64}, stride 1; or kd 3, stride 2, Cin in {8, 16, 32}; or kd 1, stride 2 = the 5x5 layers 8 -> 16, 16 -> 32; channels-last)");
        return MVS_EINVAL;
    }
    if ((int64_t)(kd + 3) * H * W * Cin * 4 >= 0xffffff00LL) return bare_error(MVS_EUNSUPPORTED, __func__, __LINE__);   // 32-bit halo offsets: callers fall back to the fp32 kernels
    // conv1 of the 3D U-Net (8 -> 16, stride 2) in the two-piece form: the z-marching kernel of conv_s2_march.hip on the SAME
    // packed fragments (1.10 instead of 1.45 input voxels copied per voxel used); MVS_CONV1_MARCH=0: the per-tile kernel below
    static const bool s2_march = !(getenv("MVS_CONV1_MARCH") && atoi(getenv("MVS_CONV1_MARCH")) == 0);
    if (s2_march && np == 2 && kd == 3 && stride == 2 && Cin == 8 && Cout == 16 && !residual && !out_c4 && split_cout_step(kd, Cout, stride, np) == 16) {
        const float *w_iscale = reinterpret_cast<const float *>(static_cast<const unsigned char *>(packed) + split_packed_bytes(kd, Cin, Cout, stride, 2));
        const int rc = launch_conv_s2_march(in, in_absmax, packed, w_iscale, scale, shift, relu, B, D, H, W, out, out_absmax, guard_counter(),
                                            conv_run_flag(), as_stream(stream));
        if (rc != MVS_EUNSUPPORTED) return rc;
    }
    const int step = split_cout_step(kd, Cout, stride, np), cps = split_cps(kd, Cin, stride), G = (split_ntap(kd, stride) * cps + 3) / 4;
    const size_t per_launch = (size_t)(Cin / (8 * cps)) * G * (step / 16) * np * 1024;
    hipStream_t st = as_stream(stream);
    for (int co0 = 0; co0 < Cout; co0 += step) {
        SplitArgs a;
        a.in_absmax = static_cast<const unsigned *>(in_absmax);
        a.w_iscale = reinterpret_cast<const float *>(static_cast<const unsigned char *>(packed) + split_packed_bytes(kd, Cin, Cout, stride, 2));
        a.out_absmax = static_cast<unsigned *>(out_absmax);
        a.w_f32 = a.w_iscale + 4;
        a.guard_cnt = np == 2 ? guard_counter() : nullptr;
        a.run_flag = conv_run_flag();
        a.in = in; a.wpk = static_cast<const unsigned char *>(packed) + (co0 / step) * per_launch;
        a.scale = scale ? scale + co0 : nullptr; a.shift = shift ? shift + co0 : nullptr;
        a.residual = residual ? residual + co0 : nullptr; a.out = out + co0;
        a.B = B; a.D = D; a.H = H; a.W = W; a.ldc = Cout; a.relu = relu;
        a.nco = Cout - co0 < step ? Cout - co0 : step; a.co0 = co0; a.out_c4 = out_c4;
        if (out_c4) a.out = out;      // (the block index carries the channel offset)
        int rc = MVS_EUNSUPPORTED;
#define MVS_SPLIT_CASE(CI, CO, KD) if (stride == 1 && Cin == CI && step == CO && kd == KD) \
        rc = np == 2 ? launch_split<SplitCfg<CI, CO, KD, 1, 3, 2>>(a, st) : launch_split<SplitCfg<CI, CO, KD>>(a, st);
        MVS_SPLIT_CASE(8, 32, 3)
        MVS_SPLIT_CASE(16, 16, 3) MVS_SPLIT_CASE(32, 16, 3) MVS_SPLIT_CASE(64, 16, 3)
        MVS_SPLIT_CASE(16, 32, 3) MVS_SPLIT_CASE(32, 32, 3) MVS_SPLIT_CASE(64, 32, 3)
        MVS_SPLIT_CASE(16, 16, 1) MVS_SPLIT_CASE(32, 16, 1) MVS_SPLIT_CASE(64, 16, 1)
        MVS_SPLIT_CASE(16, 32, 1) MVS_SPLIT_CASE(32, 32, 1) MVS_SPLIT_CASE(64, 32, 1)
#undef MVS_SPLIT_CASE
#define MVS_SPLIT_S2(CI, CO, KD, KH) if (stride == 2 && kd == KD && Cin == CI && step == CO) \
        rc = np == 2 ? launch_split<SplitCfg<CI, CO, KD, 2, KH, 2>>(a, st) : launch_split<SplitCfg<CI, CO, KD, 2, KH>>(a, st);
        MVS_SPLIT_S2(8, 16, 3, 3) MVS_SPLIT_S2(16, 16, 3, 3) MVS_SPLIT_S2(32, 16, 3, 3)
        MVS_SPLIT_S2(8, 16, 1, 5) MVS_SPLIT_S2(16, 32, 1, 5)
#undef MVS_SPLIT_S2
        // two-piece form, 32 output channels per launch (conv3: 16 -> 32, conv5: 32 -> 64 as two launches; 8 -> 32 for completeness)
        if (np == 2 && stride == 2 && kd == 3 && step == 32) {
            if (Cin == 8) rc = launch_split<SplitCfg<8, 32, 3, 2, 3, 2>>(a, st);
            else if (Cin == 16) rc = launch_split<SplitCfg<16, 32, 3, 2, 3, 2>>(a, st);
            else if (Cin == 32) rc = launch_split<SplitCfg<32, 32, 3, 2, 3, 2>>(a, st);
        }
        if (rc != MVS_OK) return rc;
    }
    return MVS_OK;
}

extern "C" int mvs_conv_split_f32(const float *in, const void *packed, const float *scale, const float *shift,
                                  const float *residual, int relu, int kd, int stride, int B, int Cin, int Cout, int D, int H,
                                  int W, int out_c4, float *out, void *stream) {
    return conv_split_impl(in, nullptr, packed, scale, shift, residual, relu, kd, stride, B, Cin, Cout, D, H, W, out_c4, out,
                           nullptr, 3, stream);
}

extern "C" int mvs_conv_split_f16_f32(const float *in, const void *in_absmax, const void *packed, const float *scale,
                                      const float *shift, const float *residual, int relu, int kd, int stride, int B, int Cin,
                                      int Cout, int D, int H, int W, int out_c4, float *out, void *out_absmax, void *stream) {
    return conv_split_impl(in, in_absmax, packed, scale, shift, residual, relu, kd, stride, B, Cin, Cout, D, H, W, out_c4, out,
                           out_absmax, 2, stream);
}
