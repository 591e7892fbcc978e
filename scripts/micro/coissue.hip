// Do vector-ALU instructions of one wave issue beside the MFMA stream of the other wave on its SIMD?
// One 512-thread workgroup per CU: waves 0-3 run an MFMA stream, waves 4-7 (the same four SIMDs) a v_fma_f32 stream;
// each role is also timed alone.  MODE of the MFMA stream: 0 back-to-back on four accumulators, 1 one dependent
// accumulator chain, 2 back-to-back with `s_nop` padding of about the MFMA's pipe time behind every MFMA,
// 3 as 0 at lowered priority (s_setprio 0 vs 3 for the VALU waves).
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/coissue.hip -o scripts/micro/coissue && scripts/micro/coissue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE, bool BF16>
__global__ __launch_bounds__(512) void k(long long *out, int nm, int nv, int roles) {
    const int wv = threadIdx.x >> 6;
    const bool mf = wv < 4;
    if ((mf && !(roles & 1)) || (!mf && !(roles & 2))) return;
    __syncthreads();
    long long t0 = clock64();
    if (mf && MODE != 6) {
        f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        float a = threadIdx.x * 1e-3f, b = 2e-3f;
        bf16x8 ha, hb;
        for (int i = 0; i < 8; ++i) { ha[i] = (__bf16)a; hb[i] = (__bf16)b; }
        if (MODE == 3) __builtin_amdgcn_s_setprio(0);
        for (int it = 0; it < nm / 4; ++it) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 &c = acc[MODE == 1 ? 0 : j];
                if (BF16) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, c, 0, 0, 0);
                else c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
                if (MODE == 2) {
                    if (BF16) asm volatile("s_nop 7\n\ts_nop 3");
                    else asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 3");
                }
                if (MODE == 4) asm volatile("s_nop 3");
                if (MODE == 5) asm volatile("s_nop 7");
            }
        }
        out[(blockIdx.x * 8 + wv) * 2 + 1] = (long long)(acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3]);
    } else {
        if (MODE == 3) __builtin_amdgcn_s_setprio(3);
        float x[16];
        for (int j = 0; j < 16; ++j) x[j] = 1.f + j;
        const float s = 1.0001f + threadIdx.x * 1e-7f;
        for (int it = 0; it < nv / 16; ++it) {
#pragma unroll
            for (int j = 0; j < 16; ++j) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[j]) : "v"(s));
        }
        float t = 0;
        for (int j = 0; j < 16; ++j) t += x[j];
        out[(blockIdx.x * 8 + wv) * 2 + 1] = (long long)t;
    }
    out[(blockIdx.x * 8 + wv) * 2] = clock64() - t0;
}

template <int MODE, bool BF16>
void run(const char *name, int nm, int nv) {
    long long *out, h[256 * 16];
    hipMalloc(&out, sizeof(h));
    for (int roles = 1; roles <= 3; ++roles) {
        hipMemset(out, 0, sizeof(h));
        hipLaunchKernelGGL((k<MODE, BF16>), dim3(256), dim3(512), 0, 0, out, nm, nv, roles);
        hipDeviceSynchronize();
        hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
        double m = 0, v = 0;
        for (int b = 0; b < 256; ++b)
            for (int w = 0; w < 8; ++w) (w < 4 ? m : v) += (double)h[(b * 8 + w) * 2] / 1024.0;
        printf("%-34s %-10s mfma waves %8.0f cycles (%5.1f / MFMA)   valu waves %8.0f cycles (%5.2f / FMA)\n", name,
               roles == 1 ? "mfma only" : roles == 2 ? "valu only" : "both", m, m / nm, v, v / nv);
    }
    hipFree(out);
}

int main() {
    run<0, false>("f32 16x16x4, 4 accumulators", 2000, 16000);
    run<1, false>("f32 16x16x4, dependent chain", 2000, 16000);
    run<2, false>("f32 16x16x4, s_nop padded", 2000, 16000);
    run<3, false>("f32 16x16x4, low priority", 2000, 16000);
    run<0, true>("bf16 16x16x32, 4 accumulators", 4000, 16000);
    run<1, true>("bf16 16x16x32, dependent chain", 4000, 16000);
    run<2, true>("bf16 16x16x32, s_nop padded", 4000, 16000);
    run<3, true>("bf16 16x16x32, low priority", 4000, 16000);
    run<4, true>("bf16 16x16x32, s_nop 3", 4000, 16000);
    run<5, true>("bf16 16x16x32, s_nop 7", 4000, 16000);
    run<4, false>("f32 16x16x4, s_nop 3", 2000, 16000);
    run<5, false>("f32 16x16x4, s_nop 7", 2000, 16000);
    run<6, false>("both waves of a SIMD: v_fma_f32", 16000, 16000);
    return 0;
}
