// Microbenchmark: v_mfma_f32_16x16x32_f16 issued back to back by ONE wave per SIMD -- all on the same accumulator (each waits for
// the one before it), or round-robin over 2 / 3 / 4 / 6 accumulators.  Prints cycles per MFMA (s_memtime).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_dep scripts/micro/mfma_dep.hip && /tmp/mfma_dep
#include <hip/hip_runtime.h>
#include <cstdio>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k(float *out, long long *cyc, int iters) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 1e-3f + i); b[i] = (_Float16)(i * 0.5f - threadIdx.x * 1e-3f); }
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 12; ++u)      // (inline asm: the builtin form made the compiler shuffle accumulation registers in the loop)
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[u % NACC]) : "v"(a), "v"(b));
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NACC>
void run() {
    float *out; long long *cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
    const int iters = 20000;
    k<NACC><<<256, 256>>>(out, cyc, iters);     // one workgroup per CU: one wave per SIMD
    k<NACC><<<256, 256>>>(out, cyc, iters);
    hipDeviceSynchronize();
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("accumulators %d: %.2f clock64 ticks per MFMA\n", NACC, (double)c / (iters * 12.0));
    hipFree(out); hipFree(cyc);
}

int main() {
    run<1>(); run<2>(); run<3>(); run<4>(); run<6>();
    return 0;
}
