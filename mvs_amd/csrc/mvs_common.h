// Shared host/device helpers for libmvs_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/mvs_hip.h"

namespace mvs {

void set_error(const char *fmt, ...);

// A size / argument guard that has no message of its own: still leaves a fresh error text behind
// (mvs_last_error_string() must never describe an earlier call).
inline int bare_error(int code, const char *func, int line) {
    set_error("%s: %s (guard at line %d)", func,
              code == MVS_EUNSUPPORTED ? "shape or layout not supported by this build" : "invalid argument or size", line);
    return code;
}

inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return MVS_ELAUNCH;
    }
    return MVS_OK;
}

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// "Run only if this device word is non-zero": while set (per host thread), the launchers of deconv_split.hip and of the Cout = 1
// kernels of conv3d_mfma.hip hand the word to their kernels, which return at once when it is 0.  mvs_costreg_fwd3_f32 enqueues
// the unfused conv11 / prob layers that way behind the fused tail kernel (tail_fused.hip), whose range guard sets the word.
const unsigned *&conv_run_flag();
// Clears n 32-bit words with a KERNEL.  The library's small clears (absmax blocks, flag words, queue headers, counters) do not use
// hipMemsetAsync: captured into a HIP graph, the memset node of mvs_costreg_fwd2_f32's 10 KiB of blocks took effect on the first
// replay only (round 6, ROCm 7.2 -- every two-piece layer of a replayed forward then judged stale blocks and fell back to fp32;
// tests/test_gpu_handover.py captures the eval forward and replays it on new pixels).
int launch_zero_words(void *p, int n, hipStream_t st);

// Compute units of the current device (grid size of the persistent kernels); 256 if the
// runtime cannot tell.  Queried per call: cheap, and correct when a process drives several GPUs.
inline int device_cu_count() {
    int dev = 0, cu = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu <= 0)
        cu = 256;
    return cu;
}

constexpr int kWave = 64;  // CDNA wavefront

// ---------------------------------------------------------------------
// Plane-sweep sampling coordinate, op-for-op as the reference evaluates it
// (MVSNet/models/module.py:73-79 followed by ATen's grid_sampler
// un-normalisation).  The FMA placement below reproduces ATen's CPU kernels
// bit for bit (oracle/mvs_oracle.c, pinned by tests/test_oracle_golden.py);
// this file is compiled with -ffp-contract=off so nothing else is fused.
struct SweepCam {
    float r[12];  // rows of (src_proj @ inverse(ref_proj))[:3,:4]
};

__device__ __forceinline__ void sweep_ray(const float *__restrict__ r, float x, float y,
                                          float &rx, float &ry, float &rz) {
    rx = __fmaf_rn(r[1], y, r[0] * x) + r[2];
    ry = __fmaf_rn(r[5], y, r[4] * x) + r[6];
    rz = __fmaf_rn(r[9], y, r[8] * x) + r[10];
}

// half_w = (W-1)/2, half_h = (H-1)/2 (module.py:78-79); unn_* is W/2,H/2 for
// align_corners=0 or (W-1)/2,(H-1)/2 for align_corners=1.
__device__ __forceinline__ void sweep_coord(const float *__restrict__ r, float rx, float ry,
                                            float rz, float d, float half_w, float half_h,
                                            float unn_w, float unn_h, int align_corners,
                                            float &ix, float &iy) {
    float X = rx * d + r[3];
    float Y = ry * d + r[7];
    float Z = rz * d + r[11];
    float px = X / Z;
    float py = Y / Z;
    float gx = px / half_w - 1.0f;
    float gy = py / half_h - 1.0f;
    if (align_corners) {
        ix = (gx + 1.0f) * unn_w;
        iy = (gy + 1.0f) * unn_h;
    } else {
        ix = __fmaf_rn(gx + 1.0f, unn_w, -0.5f);
        iy = __fmaf_rn(gy + 1.0f, unn_h, -0.5f);
    }
}

// ---- the same coordinates, bit for bit, with cheaper divisions (round 6: the persistent sweep's FAST mode) -----------------
// IEEE-correct a / b from a reciprocal of b refined by one Newton step: the quotient and two residual corrections -- the
// compiler's own fp32 division sequence without its range scaling (v_div_scale / v_div_fixup), so exact for operands whose
// quotient and residuals neither overflow nor underflow; sweep_coord_safe() below has what happens otherwise.
__device__ __forceinline__ float refined_rcp(float b) {
    const float y = __builtin_amdgcn_rcpf(b);
    return __fmaf_rn(__fmaf_rn(-b, y, 1.0f), y, y);
}
__device__ __forceinline__ float div_with_rcp(float a, float b, float y) {
    float q = a * y;
    q = __fmaf_rn(__fmaf_rn(-b, q, a), y, q);
    return __fmaf_rn(__fmaf_rn(-b, q, a), y, q);
}
// When is that sequence's result the compiler's?  Whenever no intermediate over- or underflows -- and when one does, the FINAL
// sampling coordinate still agrees in everything the kernels use: a quotient that overflows is Inf for the compiler and Inf or NaN
// here, both "not finite" for the tap logic (NaN weights, as the reference's 0 * NaN); one in the subnormals vanishes in the
// `- 1.0f` that follows it.  The one case that would differ is a SUBNORMAL or zero divisor Z (a point on the camera plane to within
// 1e-38): v_rcp flushes it to Inf where the true quotient may be a huge finite number (taps outside the image: zero, not NaN).
// So: safe = Z is a normal number (one v_cmp_class); a wave with any other Z takes the compiler's divisions.
__device__ __forceinline__ bool sweep_coord_safe(float Z) {
    return __builtin_amdgcn_classf(Z, 0x108);     // negative normal | positive normal (the f32 test: `class` promotes to f64)
}
// sweep_coord with X/Z and Y/Z sharing one refined reciprocal and the divisions by (W-1)/2, (H-1)/2 through their refined
// reciprocals rhw / rhh (wave-uniform): 23 vector instructions where four compiler divisions take 44.  The caller has checked
// sweep_coord_safe(Z) for the whole wave (else: sweep_coord).
__device__ __forceinline__ void sweep_coord_shared(float X, float Y, float Z, float half_w, float half_h, float rhw, float rhh,
                                                   float unn_w, float unn_h, int align_corners, float &ix, float &iy) {
    const float y = refined_rcp(Z);
    const float px = div_with_rcp(X, Z, y), py = div_with_rcp(Y, Z, y);
    const float gx = div_with_rcp(px, half_w, rhw) - 1.0f;
    const float gy = div_with_rcp(py, half_h, rhh) - 1.0f;
    // one FMA for both conventions: fma(a, b, +0) is the rounded product a b (align_corners: the reference's plain multiplication;
    // a product of -0 comes out +0, which floor and the tap weights do not distinguish), fma(a, b, -0.5) the other
    const float off = align_corners ? 0.0f : -0.5f;
    ix = __fmaf_rn(gx + 1.0f, unn_w, off);
    iy = __fmaf_rn(gy + 1.0f, unn_h, off);
}

// Bilinear taps with zeros padding (ATen grid_sampler_2d bilinear/zeros).
struct Taps {
    float nw, ne, sw, se;  // weights of (y0,x0) (y0,x1) (y1,x0) (y1,x1)
    int x0, x1, y0, y1;    // clamped to valid indices
    bool x0ok, x1ok, y0ok, y1ok;
};

__device__ __forceinline__ Taps make_taps(float ix, float iy, int H, int W) {
    Taps t;
    float x0f = floorf(ix), y0f = floorf(iy);
    float w = ix - x0f, e = 1.0f - w;
    float n = iy - y0f, s = 1.0f - n;
    t.nw = s * e;
    t.ne = s * w;
    t.sw = n * e;
    t.se = n * w;
    t.x0ok = (x0f >= 0.0f) && (x0f <= (float)(W - 1));
    t.x1ok = (x0f >= -1.0f) && (x0f <= (float)(W - 2));
    t.y0ok = (y0f >= 0.0f) && (y0f <= (float)(H - 1));
    t.y1ok = (y0f >= -1.0f) && (y0f <= (float)(H - 2));
    t.x0 = t.x0ok ? (int)x0f : 0;
    t.x1 = t.x1ok ? (int)x0f + 1 : 0;
    t.y0 = t.y0ok ? (int)y0f : 0;
    t.y1 = t.y1ok ? (int)y0f + 1 : 0;
    return t;
}

__device__ __forceinline__ float blend(const Taps &t, float v00, float v01, float v10,
                                       float v11) {
    return __fmaf_rn(v11, t.se, __fmaf_rn(v10, t.sw, __fmaf_rn(v01, t.ne, v00 * t.nw)));
}

#ifdef __HIPCC__
// LDS-DMA: each lane's 16 bytes at `gsrc` -> LDS byte address `lds_byte_addr` + 16*lane
// (global_load_lds_dwordx4; the destination base is wave-uniform and travels in M0, which
// the compiler reserves -- saved and restored inside the one statement).  Honours EXEC.
// The compiler does not count these: pair with an explicit s_waitcnt vmcnt.
__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_byte_addr)
                 : "memory");
}
// The same copy addressed through a buffer resource: 16 bytes at (srd.base + soffset +
// voffset) per lane.  voffset is a per-lane VGPR that does not change between chunks and
// soffset a wave-uniform SGPR, so re-issuing a tile's copy for the next channel chunk costs
// no vector ALU work; lanes whose voffset is beyond srd.num_records fetch zeros.
typedef int mvs_srd_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ mvs_srd_t make_srd(const void *base, unsigned bytes) {
    const unsigned long long b = (unsigned long long)base;
    mvs_srd_t r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));   // stride 0, no swizzle
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;                                                  // raw dword buffer
    return r;
}
__device__ __forceinline__ void glds16_buf(unsigned voffset, mvs_srd_t srd, unsigned soffset,
                                           unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
                 "buffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voffset), "s"(srd), "s"(soffset), "s"(lds_byte_addr)
                 : "memory");
}

// "absmax block": kAbsmaxWords 32-bit words that together hold the largest magnitude of an array as the bit pattern of |x|
// (non-negative floats order like their bit patterns) -- word i collects the workgroups with blockIdx.x % 256 == i, the
// reader takes the maximum of all words.  One word for everybody made every wave's atomic queue up behind 2000-4000 others at
// the end of a kernel (+6 us per launch on the small layers, +50 us on the sweep).  The consumer is the operand scale of the
// two-piece fp16 convolutions (conv_f16x3.hip).
constexpr int kAbsmaxWords = 256;
// NaN-propagating maximum (IEEE 754-2019 `maximum`: v_maximum3_f32 on gfx950, one instruction like v_max_f32).  The reference's
// ReLU is torch's: relu(NaN) = NaN, relu(-Inf) = 0 -- fmaxf(NaN, 0) = 0 would hide a poisoned voxel; and an absmax block must SEE
// a NaN (its bit pattern is above every number's), because the two-piece layers decide on it whether their arithmetic holds
// (conv_guard.h).
__device__ __forceinline__ float max_nan(float a, float b) { return __builtin_elementwise_maximum(a, b); }
__device__ __forceinline__ float relu_nan(float v) { return __builtin_elementwise_maximum(v, 0.0f); }
// running largest magnitude of a lane's stored values: vmax = max(vmax, |a|, |b|, |c|, |d|), a NaN sticks
__device__ __forceinline__ float amax4_nan(float vmax, float a, float b, float c, float d) {
    return max_nan(max_nan(max_nan(vmax, __builtin_fabsf(a)), __builtin_fabsf(b)), max_nan(__builtin_fabsf(c), __builtin_fabsf(d)));
}
__device__ __forceinline__ void publish_absmax(unsigned *absmax, float vmax) {
    if (!absmax) return;
#pragma unroll
    for (int o = 32; o; o >>= 1) vmax = max_nan(vmax, __shfl_xor(vmax, o));
    if ((threadIdx.x & 63) == 0 && !(vmax <= 0.0f))
        atomicMax(absmax + (blockIdx.x & (kAbsmaxWords - 1)), __float_as_uint(vmax) & 0x7fffffffu);
}
// wave-uniform maximum of an absmax block
__device__ __forceinline__ unsigned load_absmax(const unsigned *absmax) {
    const uint4 v = reinterpret_cast<const uint4 *>(absmax)[threadIdx.x & 63];
    unsigned m = max(max(v.x, v.y), max(v.z, v.w));
#pragma unroll
    for (int o = 32; o; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    return (unsigned)__builtin_amdgcn_readfirstlane((int)m);
}
#endif

}  // namespace mvs
