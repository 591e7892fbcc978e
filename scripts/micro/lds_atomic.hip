// Microbenchmark: LDS atomic-add issue rate on gfx950 (tuning aid for the variance backward).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_atomic scripts/micro/lds_atomic.hip && /tmp/lds_atomic
// Prints CU cycles per wave-instruction with 8 waves resident per CU.
#include <hip/hip_runtime.h>
#include <cstdio>

// MODE 0: ds_add_f32, 1: ds_add_u32, 2: ds_add_rtn_f32, 3: read+add+write (non-atomic), 4: ds_pk_add_f16,
// 5: ds_add_u64
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, int stride, int active, int spread) {
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // address pattern: lane -> (lane / spread) * stride  (spread lanes share an address)
    const unsigned addr = (unsigned)(uintptr_t)lds + (unsigned)(((lane / spread) * stride + wv * 2048) & 8191) * 4u;
    (void)hipSuccess;
    float v = lane * 1e-3f, acc = 0.f;
    if (lane < active) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (MODE == 0) asm volatile("ds_add_f32 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(u * 256) : "memory");
                if (MODE == 1) asm volatile("ds_add_u32 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(u * 256) : "memory");
                if (MODE == 2) { float r; asm volatile("ds_add_rtn_f32 %0, %1, %2 offset:%3" : "=v"(r) : "v"(addr), "v"(v), "n"(u * 256) : "memory"); acc += 0.f * 0; asm volatile("" ::"v"(r)); }
                if (MODE == 3) { float r; asm volatile("ds_read_b32 %0, %1 offset:%2\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr), "n"(u * 256) : "memory"); r += v; asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(addr), "v"(r), "n"(u * 256) : "memory"); }
                if (MODE == 5) { unsigned long long q = (unsigned long long)lane; asm volatile("ds_add_u64 %0, %1 offset:%2" ::"v"(addr), "v"(q), "n"(u * 256) : "memory"); }
                if (MODE == 4) asm volatile("ds_pk_add_f16 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(u * 256) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = lds[threadIdx.x] + acc;
}

template <int MODE>
void run(const char *name, int stride, int active, int spread) {
    const int nblk = 256 * 2, iters = 2000;
    float *out;
    hipMalloc(&out, (size_t)nblk * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE>), dim3(nblk), dim3(256), 0, 0, out, iters, stride, active, spread);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(nblk), dim3(256), 0, 0, out, iters, stride, active, spread);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_cu = 8.0 * iters * 8;   // 8 waves x iters x 8
    printf("%-18s stride=%2d active=%2d share=%2d : %7.1f cycles / wave-instr (2.4 GHz)\n", name, stride, active, spread,
           ms * 1e-3 * 2.4e9 / instr_per_cu);
    hipFree(out);
}

int main() {
    run<0>("ds_add_f32", 1, 64, 1);
    run<0>("ds_add_f32", 1, 32, 1);
    run<0>("ds_add_f32", 1, 16, 1);
    run<0>("ds_add_f32", 1, 8, 1);
    run<0>("ds_add_f32", 4, 64, 1);
    run<0>("ds_add_f32", 1, 64, 2);
    run<0>("ds_add_f32", 1, 64, 4);
    run<0>("ds_add_f32", 1, 64, 64);
    run<0>("ds_add_f32", 16, 64, 1);
    run<1>("ds_add_u32", 1, 64, 1);
    run<1>("ds_add_u32", 4, 64, 1);
    run<1>("ds_add_u32", 1, 64, 4);
    run<2>("ds_add_rtn_f32", 1, 64, 1);
    run<3>("read+add+write", 1, 64, 1);
    run<4>("ds_pk_add_f16", 1, 64, 1);
    run<5>("ds_add_u64", 2, 64, 1);
    run<5>("ds_add_u64", 2, 32, 1);
    run<5>("ds_add_u64", 8, 64, 1);
    run<5>("ds_add_u64", 32, 64, 1);
    run<5>("ds_add_u64", 2, 64, 2);
    run<5>("ds_add_u64", 2, 64, 4);
    run<1>("ds_add_u32", 16, 64, 1);
    run<1>("ds_add_u32", 1, 64, 2);
    return 0;
}
