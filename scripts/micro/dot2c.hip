// What v_dot2c_f32_bf16 computes on gfx950, and whether it can subtract one bf16 of a packed pair from an
// fp32 number exactly (the split-operand convolution, mvs_amd/csrc/conv_bf16x6.hip).
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/dot2c.hip -o scripts/micro/dot2c && scripts/micro/dot2c
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// out[0..1]: x - bf16(x) by conversion + subtraction; out[2..3]: by dot2c, constant in an SGPR as src0;
// out[4..5]: by dot2c, constant in a VGPR as src1 and the pair as src0; out[6..7]: raw dot2c(pair, pair, 0)
__global__ void k(const float *x, float *out, unsigned *packed) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const f32x2 v = {x[2 * i], x[2 * i + 1]};
    const bf16x2 h = __builtin_convertvector(v, bf16x2);
    const unsigned hu = __builtin_bit_cast(unsigned, h);
    const f32x2 hf = __builtin_convertvector(h, f32x2);
    float *o = out + 8 * i;
    o[0] = v[0] - hf[0];
    o[1] = v[1] - hf[1];
    float a = v[0], b = v[1];
    asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(a) : "s"(0x0000bf80u), "v"(hu));
    asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(b) : "s"(0xbf800000u), "v"(hu));
    o[2] = a; o[3] = b;
    a = v[0]; b = v[1];
    unsigned c0 = 0x0000bf80u, c1 = 0xbf800000u;
    asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(a) : "v"(hu), "v"(c0));
    asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(b) : "v"(hu), "v"(c1));
    o[4] = a; o[5] = b;
    a = 0.f;
    asm volatile("v_dot2c_f32_bf16 %0, %1, %1" : "+v"(a) : "v"(hu));
    o[6] = a; o[7] = hf[0] * hf[0] + hf[1] * hf[1];
    packed[i] = hu;
}

int main() {
    const int n = 1 << 16;
    float *hx = (float *)malloc(n * 2 * 4), *ho = (float *)malloc(n * 8 * 4);
    srand(1);
    for (int i = 0; i < 2 * n; ++i) {
        const float m = (float)rand() / RAND_MAX * 2 - 1;
        hx[i] = ldexpf(m, rand() % 40 - 30);
    }
    float *dx, *d_o; unsigned *dp;
    hipMalloc(&dx, n * 8); hipMalloc(&d_o, n * 32); hipMalloc(&dp, n * 4);
    hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d_o, dp);
    hipMemcpy(ho, d_o, n * 32, hipMemcpyDeviceToHost);
    long bad_s = 0, bad_v = 0, bad_sq = 0;
    for (int i = 0; i < n; ++i) {
        const float *o = ho + 8 * i;
        bad_s += (o[2] != o[0]) + (o[3] != o[1]);
        bad_v += (o[4] != o[0]) + (o[5] != o[1]);
        bad_sq += fabsf(o[6] - o[7]) > 1e-6f * fabsf(o[7]);
    }
    printf("mismatch: sgpr-const form %ld, vgpr-const form %ld of %d; dot(pair,pair) off in %ld of %d\n", bad_s, bad_v, 2 * n, bad_sq, n);
    for (int i = 0; i < 4; ++i) {
        const float *o = ho + 8 * i;
        printf("x = (%.9g, %.9g): sub (%.9g, %.9g)  dot2c/s (%.9g, %.9g)  dot2c/v (%.9g, %.9g)  dot %.9g vs %.9g\n", hx[2 * i], hx[2 * i + 1],
               o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7]);
    }
    return 0;
}
