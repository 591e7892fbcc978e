// Microbenchmark: attainable v_mfma_f32_16x16x4_f32 rate on gfx950 (tuning aid for conv0).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak scripts/micro/mfma_peak.hip && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int DEP>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < DEP; ++d)
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        asm volatile("" : "+v"(a), "+v"(b));
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, int DEP>
void run(const char *name, int blocks_per_cu) {
    int nblk = 256 * blocks_per_cu;
    float *out;
    hipMalloc(&out, (size_t)nblk * 256 * 4);
    int iters = 20000 / (NACC * DEP);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, DEP>), dim3(nblk), dim3(256), 0, 0, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, DEP>), dim3(nblk), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)nblk * 4 * iters * NACC * DEP * 2048.0;
    printf("%-28s blocks/CU=%d  %.3f ms  %.1f TFLOP/s\n", name, blocks_per_cu, ms, flops / ms / 1e9);
    hipFree(out);
}

int main() {
    for (int bpc = 1; bpc <= 4; bpc *= 2) {
        run<8, 1>("8 acc independent", bpc);
        run<8, 2>("8 acc, pairs interleaved", bpc);
        run<1, 8>("1 acc dependent chain", bpc);
        run<2, 4>("2 acc", bpc);
        run<4, 2>("4 acc", bpc);
    }
    return 0;
}
