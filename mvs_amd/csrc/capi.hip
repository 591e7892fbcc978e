// C-ABI glue of libmvs_hip.so: error text, version, layout helpers and the
// mvs_conv3d_f32 dispatcher (include/mvs_hip.h).
#include "mvs_common.h"

#include <atomic>
#include <cstring>

namespace mvs {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

namespace {
__global__ __launch_bounds__(256) void zero_words_kernel(unsigned *__restrict__ p, int n) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) p[i] = 0u;
}
}  // namespace
int launch_zero_words(void *p, int n, hipStream_t st) {
    if (!p || n <= 0) return MVS_OK;
    const int blocks = (n + 255) / 256;
    hipLaunchKernelGGL(zero_words_kernel, dim3(blocks < 64 ? blocks : 64), dim3(256), 0, st, static_cast<unsigned *>(p), n);
    return check_launch("launch_zero_words");
}

const unsigned *&conv_run_flag() {
    static thread_local const unsigned *flag = nullptr;
    return flag;
}

// launches that took the range guard's fallback (conv_guard.h), per device
__device__ unsigned long long g_guard_fallbacks;
unsigned long long *guard_counter() {
    // one slot per device, published with release / acquire: concurrent host threads (nn.DataParallel replicas) may race to
    // resolve the same slot, and both then store the same address
    static std::atomic<unsigned long long *> ptr[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    unsigned long long *q = ptr[dev].load(std::memory_order_acquire);
    if (!q) {
        void *p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_guard_fallbacks)) != hipSuccess) return nullptr;
        q = static_cast<unsigned long long *>(p);
        ptr[dev].store(q, std::memory_order_release);
    }
    return q;
}

int conv3d_direct_launch(const float *, const float *, const float *, const float *, const float *,
                         int, int, int, int, int, int, int, int, int, int, float *, hipStream_t);
int conv3d_mfma_launch(const float *, const float *, const float *, const float *, const float *,
                       int, int, int, int, int, int, int, int, int, int, float *, hipStream_t, unsigned *, bool *);
int conv3d_pack_launch(const float *, int, int, int, int, float *, hipStream_t);
int64_t conv3d_packed_floats(int, int, int, int);
int conv3d_mfma_supported(int, int, int, int);

// [B,C,S] <-> [B,S,C] through a 32x33 LDS tile (both sides coalesced).
__global__ __launch_bounds__(256) void transpose_cs_kernel(const float *__restrict__ in,
                                                           float *__restrict__ out, int R,
                                                           int64_t S) {
    // in: [B][R][S] -> out: [B][S][R]
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int64_t s0 = (int64_t)blockIdx.x * 32;
    const int r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const float *ip = in + (int64_t)b * R * S;
    float *op = out + (int64_t)b * R * S;
#pragma unroll
    for (int k = 0; k < 32; k += 8) {
        int r = r0 + ty + k;
        int64_t s = s0 + tx;
        tile[ty + k][tx] = (r < R && s < S) ? ip[(int64_t)r * S + s] : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 32; k += 8) {
        int64_t s = s0 + ty + k;
        int r = r0 + tx;
        if (r < R && s < S) op[s * R + r] = tile[tx][ty + k];
    }
}

// [B][R][S] -> [B][S][R] for tiny R (RGB images): one thread per position, R
// coalesced plane reads, one R*4-byte contiguous write.
template <int R>
__global__ __launch_bounds__(256) void transpose_small_r_kernel(const float *__restrict__ in,
                                                                float *__restrict__ out,
                                                                int64_t S) {
    const int b = blockIdx.y;
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= S) return;
    const float *ip = in + (int64_t)b * R * S + s;
    float v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = ip[(int64_t)r * S];
    float *op = out + ((int64_t)b * S + s) * R;
#pragma unroll
    for (int r = 0; r < R; ++r) op[r] = v[r];
}

static int launch_transpose(const float *in, float *out, int B, int R, int64_t S, hipStream_t st,
                            const char *what) {
    if (!in || !out || B <= 0 || R <= 0 || S <= 0) {
        set_error("%s: invalid argument", what);
        return MVS_EINVAL;
    }
    if (R <= 4 && B <= 65535 && (S + 255) / 256 <= 0x7fffffffLL) {
        const dim3 g((unsigned)((S + 255) / 256), (unsigned)B);
        switch (R) {
            case 1: hipLaunchKernelGGL(transpose_small_r_kernel<1>, g, dim3(256), 0, st, in, out, S); break;
            case 2: hipLaunchKernelGGL(transpose_small_r_kernel<2>, g, dim3(256), 0, st, in, out, S); break;
            case 3: hipLaunchKernelGGL(transpose_small_r_kernel<3>, g, dim3(256), 0, st, in, out, S); break;
            default: hipLaunchKernelGGL(transpose_small_r_kernel<4>, g, dim3(256), 0, st, in, out, S); break;
        }
        return check_launch(what);
    }
    // generic: in [B][R][S] -> out [B][S][R]
    int64_t gx = (S + 31) / 32;
    int gy = (R + 31) / 32;
    if (gx > 0x7fffffffLL || gy > 65535 || B > 65535) {
        set_error("%s: problem too large", what);
        return MVS_EINVAL;
    }
    hipLaunchKernelGGL(transpose_cs_kernel, dim3((unsigned)gx, (unsigned)gy, (unsigned)B),
                       dim3(256), 0, st, in, out, R, S);
    return check_launch(what);
}

}  // namespace mvs

using namespace mvs;

// the changelog is in include/mvs_hip.h above mvs_version(): 0.1.1 re-query the *_f16*_packed_bytes sizes, 0.1.2 re-query
// mvs_costreg_workspace_bytes, 0.1.3 the hand-over entry points
extern "C" int mvs_version(void) { return 103; /* 0.1.3 */ }
extern "C" const char *mvs_last_error_string(void) { return g_err; }
extern "C" const char *mvs_arch(void) { return "gfx950"; }

// Launches of the current device that took the range guard's fallback (conv_guard.h) since the library was loaded.
// Synchronises with the device (a diagnostic: call it after the work, not between the layers).
extern "C" int mvs_guard_fallback_count(unsigned long long *count) {
    unsigned long long *p = guard_counter();
    if (!count || !p) {
        set_error("mvs_guard_fallback_count: no counter on this device");
        return MVS_EINVAL;
    }
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(count, p, sizeof(*count), hipMemcpyDeviceToHost) != hipSuccess)
        return bare_error(MVS_ELAUNCH, __func__, __LINE__);
    return MVS_OK;
}

// Resolve the guard counter of EVERY visible device now (hipGetSymbolAddress on first use may load a code object: not something
// a launch should meet for the first time inside a HIP-graph capture on a device that was not current at load time).
extern "C" int mvs_guard_resolve_all_devices(void) {
    int n = 0, cur = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || hipGetDevice(&cur) != hipSuccess) return bare_error(MVS_ELAUNCH, __func__, __LINE__);
    int rc = MVS_OK;
    for (int d = 0; d < n && d < 64; ++d) {
        if (hipSetDevice(d) != hipSuccess || !guard_counter()) rc = MVS_ELAUNCH;
    }
    (void)hipSetDevice(cur);
    if (rc != MVS_OK) set_error("mvs_guard_resolve_all_devices: a device has no counter");
    return rc;
}

// The id of the HIP-graph capture `stream` is in (0 = not capturing): lets a host-side cache tell one capture from the next.
extern "C" int mvs_stream_capture_id(void *stream, unsigned long long *id) {
    if (!id) {
        set_error("mvs_stream_capture_id: id is NULL");
        return MVS_EINVAL;
    }
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    unsigned long long cid = 0;
    if (hipStreamGetCaptureInfo(as_stream(stream), &st, &cid) != hipSuccess) return bare_error(MVS_ELAUNCH, __func__, __LINE__);
    *id = (st == hipStreamCaptureStatusActive) ? (cid ? cid : 1ull) : 0ull;
    return MVS_OK;
}

extern "C" int mvs_nchw_to_nhwc_f32(const float *in, float *out, int B, int C, int64_t S,
                                    void *stream) {
    return launch_transpose(in, out, B, C, S, as_stream(stream), "mvs_nchw_to_nhwc_f32");
}

extern "C" int mvs_nhwc_to_nchw_f32(const float *in, float *out, int B, int C, int64_t S,
                                    void *stream) {
    // in [B][S][C] -> out [B][C][S]: same kernel with the roles of R and S swapped
    if (S > 0x7fffffffLL) {
        set_error("mvs_nhwc_to_nchw_f32: S too large");
        return MVS_EINVAL;
    }
    return launch_transpose(in, out, B, (int)S, (int64_t)C, as_stream(stream),
                            "mvs_nhwc_to_nchw_f32");
}

extern "C" int mvs_conv3d_mfma_supported(int transposed, int Cin, int Cout, int stride) {
    return conv3d_mfma_supported(transposed, Cin, Cout, stride);
}

extern "C" int64_t mvs_conv3d_packed_weight_floats(int transposed, int Cin, int Cout, int stride) {
    return conv3d_packed_floats(transposed, Cin, Cout, stride);
}

extern "C" int mvs_conv3d_pack_weights_f32(const float *weight, int transposed, int Cin, int Cout,
                                           int stride, float *packed, void *stream) {
    if (!weight || !packed) {
        set_error("mvs_conv3d_pack_weights_f32: null pointer");
        return MVS_EINVAL;
    }
    return conv3d_pack_launch(weight, transposed, Cin, Cout, stride, packed, as_stream(stream));
}

extern "C" int mvs_absmax_f32(const float *x, int64_t n, void *absmax, void *stream);   // conv_f16x3.hip

extern "C" int mvs_conv3d_f32(const float *in, const float *weight, const float *packed_weight,
                              const float *scale, const float *shift, const float *residual,
                              int relu, int transposed, int B, int Cin, int Cout, int D, int H,
                              int W, int stride, int layout, int impl, float *out, void *stream) {
    return mvs_conv3d_absmax_f32(in, weight, packed_weight, scale, shift, residual, relu, transposed, B, Cin, Cout, D, H, W,
                                 stride, layout, impl, out, nullptr, stream);
}

extern "C" int mvs_conv3d_absmax_f32(const float *in, const float *weight, const float *packed_weight,
                                     const float *scale, const float *shift, const float *residual,
                                     int relu, int transposed, int B, int Cin, int Cout, int D, int H,
                                     int W, int stride, int layout, int impl, float *out, void *out_absmax, void *stream) {
    if (!in || !out || B <= 0 || Cin <= 0 || Cout <= 0 || D <= 0 || H <= 0 || W <= 0 ||
        (stride != 1 && stride != 2) ||
        (layout != MVS_LAYOUT_NCHW && layout != MVS_LAYOUT_NHWC && layout != MVS_LAYOUT_C8) ||
        impl < 0 || impl > 2) {
        set_error("mvs_conv3d_f32: invalid argument");
        return MVS_EINVAL;
    }
    hipStream_t st = as_stream(stream);
    // the MFMA kernels address a tile's halo planes with 32-bit byte offsets: 9 input planes
    // (the deepest halo, stride 2) must stay below 4 GiB -- 12 M pixels at 8 channels
    const bool window_ok = (int64_t)9 * H * W * Cin * 4 < 0xffffff00LL;
    const bool mfma_ok = layout != MVS_LAYOUT_NCHW && packed_weight && window_ok &&
                         conv3d_mfma_supported(transposed, Cin, Cout, stride);
    if (impl == 2 && !mfma_ok) {
        set_error("mvs_conv3d_f32: MFMA path needs channels-last, packed weights and a supported "
                  "shape (%s Cin=%d Cout=%d stride=%d)",
                  transposed ? "deconv" : "conv", Cin, Cout, stride);
        return MVS_EUNSUPPORTED;
    }
    // out_absmax: the MFMA convolutions collect it in their epilogue; behind every other kernel, one more pass over `out`
    const int64_t nout = transposed ? (int64_t)B * Cout * (2 * D) * (2 * H) * (2 * W)
                                    : (int64_t)B * Cout * ((D - 1) / stride + 1) * ((H - 1) / stride + 1) * ((W - 1) / stride + 1);
    // a "run only if" word is in scope (mvs_common.h: conv_run_flag): only the Cout = 1 MFMA kernels read it -- any other kernel would
    // run unconditionally on a scratch volume nobody wrote, so refuse instead
    if (conv_run_flag() && !((impl == 2 || (impl == 0 && mfma_ok)) && !transposed && Cout == 1 && stride == 1)) {
        set_error("mvs_conv3d_f32: a run-only-if flag is in scope and this shape (%s Cin=%d Cout=%d stride=%d, packed %s, plane window %s) "
                  "has no kernel that honours it", transposed ? "deconv" : "conv", Cin, Cout, stride, packed_weight ? "yes" : "no",
                  window_ok ? "ok" : "too large");
        return MVS_EUNSUPPORTED;
    }
    if (impl == 2 || (impl == 0 && mfma_ok)) {
        bool collected = false;
        const int rc = conv3d_mfma_launch(in, packed_weight, scale, shift, residual, relu, transposed, B,
                                          Cin, Cout, D, H, W, stride, layout == MVS_LAYOUT_C8, out, st,
                                          static_cast<unsigned *>(out_absmax), &collected);
        if (rc != MVS_OK || !out_absmax || collected) return rc;
        return mvs_absmax_f32(out, nout, out_absmax, stream);
    }
    if (layout == MVS_LAYOUT_C8) {
        set_error("mvs_conv3d_f32: the 8-channel-blocked layout is input-only for the MFMA path");
        return MVS_EUNSUPPORTED;
    }
    if (!weight) {
        set_error("mvs_conv3d_f32: direct path needs the PyTorch-layout weight");
        return MVS_EINVAL;
    }
    const int rc = conv3d_direct_launch(in, weight, scale, shift, residual, relu, transposed, B, Cin, Cout,
                                        D, H, W, stride, layout, out, st);
    if (rc != MVS_OK || !out_absmax) return rc;
    return mvs_absmax_f32(out, nout, out_absmax, stream);
}
