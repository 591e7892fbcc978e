// v_fma_f32 vs v_pk_fma_f32 issue rate on gfx950 (one question: does a packed FMA retire two
// lanes' worth of work per issue slot, or is it two passes?).
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/pk_fma.hip -o scripts/micro/pk_fma && scripts/micro/pk_fma
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float s) {
    float a[16];
    f32x2 p[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x * 0.001f + i; p[i] = f32x2{a[i], a[i] + 0.5f}; }
    const f32x2 s2 = {s, s};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(s));
            else if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(s2));
            else asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(s2));
        }
    }
    float acc = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += a[i] + p[i][0] + p[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int MODE>
static void run(const char *name, float *d, int iters, int waves_per_simd) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * waves_per_simd;   // 256 CUs x 4 SIMDs, 256 threads = 4 waves = 1 per SIMD
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 16, 0.999f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 0.999f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double inst = (double)blocks * 4 * iters * 16;           // wave-instructions
    const double cyc_per_inst = ms * 1e-3 * 2.4e9 / (inst / (256.0 * 4));   // per SIMD, at 2.4 GHz nominal
    printf("%-14s waves/SIMD %d: %.3f ms, %.2f cycles per wave-instruction per SIMD (2.4 GHz nominal), %.1f G lane-results/s\n",
           name, waves_per_simd, ms, cyc_per_inst, inst * 64 * (MODE == 0 ? 1 : 2) / ms / 1e6);
}

int main() {
    float *d;
    hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_fma_f32", d, 20000, w);
        run<1>("v_pk_fma_f32", d, 20000, w);
        run<2>("v_pk_mul_f32", d, 20000, w);
    }
    return 0;
}
