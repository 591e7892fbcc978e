// K3 fallback: direct (VALU) 3x3x3 convolution / transposed convolution with
// the fused BatchNorm-affine + ReLU + skip-add epilogue (module.py:26-33,
// mvsnet.py:48-93).  Any channel count, either layout.  This is the
// always-available HIP path for shapes the MFMA kernel does not cover (and the
// on-GPU cross-check for it); the MFMA implicit-GEMM kernel in conv3d_mfma.hip
// is the fast path.
#include "mvs_common.h"

namespace mvs {

struct ConvGeom {
    int B, Ci, Co, D, H, W;     // input dims
    int Do, Ho, Wo;             // output dims
    int stride, relu;
    // element strides
    int64_t i_b, i_c, i_z, i_y, i_x;
    int64_t o_b, o_c, o_z, o_y, o_x;
    int channels_last;
};

__device__ __forceinline__ void decode_out(const ConvGeom &g, int64_t idx, int &b, int &co, int &z,
                                           int &y, int &x) {
    if (g.channels_last) {
        co = (int)(idx % g.Co); idx /= g.Co;
        x = (int)(idx % g.Wo); idx /= g.Wo;
        y = (int)(idx % g.Ho); idx /= g.Ho;
        z = (int)(idx % g.Do); idx /= g.Do;
        b = (int)idx;
    } else {
        x = (int)(idx % g.Wo); idx /= g.Wo;
        y = (int)(idx % g.Ho); idx /= g.Ho;
        z = (int)(idx % g.Do); idx /= g.Do;
        co = (int)(idx % g.Co); idx /= g.Co;
        b = (int)idx;
    }
}

__device__ __forceinline__ float epilogue(float acc, int co, int64_t o, const float *scale,
                                          const float *shift, const float *residual, int relu) {
    float v = acc;
    if (scale) v = v * scale[co];
    if (shift) v = v + shift[co];
    if (relu) v = relu_nan(v);
    if (residual) v = residual[o] + v;
    return v;
}

// weight (Co,Ci,3,3,3)
__global__ __launch_bounds__(256) void conv3d_direct_kernel(
    const float *__restrict__ in, const float *__restrict__ w, const float *__restrict__ scale,
    const float *__restrict__ shift, const float *__restrict__ residual, ConvGeom g,
    float *__restrict__ out) {
    const int64_t total = (int64_t)g.B * g.Co * g.Do * g.Ho * g.Wo;
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    int b, co, z, y, x;
    decode_out(g, idx, b, co, z, y, x);
    float acc = 0.0f;
    const float *wp = w + (int64_t)co * g.Ci * 27;
    for (int ci = 0; ci < g.Ci; ++ci) {
        const float *ip = in + (int64_t)b * g.i_b + (int64_t)ci * g.i_c;
#pragma unroll
        for (int kz = 0; kz < 3; ++kz) {
            int iz = z * g.stride + kz - 1;
            if (iz < 0 || iz >= g.D) continue;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                int iy = y * g.stride + ky - 1;
                if (iy < 0 || iy >= g.H) continue;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    int ixx = x * g.stride + kx - 1;
                    if (ixx < 0 || ixx >= g.W) continue;
                    acc = fmaf(ip[iz * g.i_z + iy * g.i_y + ixx * g.i_x],
                               wp[ci * 27 + (kz * 3 + ky) * 3 + kx], acc);
                }
            }
        }
    }
    int64_t o = (int64_t)b * g.o_b + (int64_t)co * g.o_c + z * g.o_z + y * g.o_y + x * g.o_x;
    out[o] = epilogue(acc, co, o, scale, shift, residual, g.relu);
}

// One input channel, CO output channels, stride 1, channels-last (the input gradient of the
// `prob` layer, mvsnet.py:81: a 1 -> 8 convolution with the flipped weights): one thread per
// voxel keeps the CO sums, reads its 27 neighbours once (the general kernel reads them once
// per output channel) and writes CO contiguous floats.  Weights are uniform: scalar loads.
template <int CO>
__global__ __launch_bounds__(256) void conv3d_cin1_cl_kernel(
    const float *__restrict__ in, const float *__restrict__ w, const float *__restrict__ scale,
    const float *__restrict__ shift, const float *__restrict__ residual, ConvGeom g,
    float *__restrict__ out) {
    const int64_t total = (int64_t)g.B * g.D * g.H * g.W;
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int64_t vox = idx;
    const int x = (int)(idx % g.W); idx /= g.W;
    const int y = (int)(idx % g.H); idx /= g.H;
    const int z = (int)(idx % g.D);
    float acc[CO];
#pragma unroll
    for (int co = 0; co < CO; ++co) acc[co] = 0.0f;
#pragma unroll
    for (int kz = 0; kz < 3; ++kz) {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int iz = z + kz - 1, iy = y + ky - 1, ix = x + kx - 1;
                const bool ok = (unsigned)iz < (unsigned)g.D && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
                const int64_t off = vox + ((int64_t)(kz - 1) * g.H + (ky - 1)) * g.W + (kx - 1);
                const float v = ok ? in[off] : 0.0f;
#pragma unroll
                for (int co = 0; co < CO; ++co) acc[co] = fmaf(v, w[co * 27 + (kz * 3 + ky) * 3 + kx], acc[co]);
            }
        }
    }
    float o[CO];
#pragma unroll
    for (int co = 0; co < CO; ++co) o[co] = epilogue(acc[co], co, vox * CO + co, scale, shift, residual, g.relu);
#pragma unroll
    for (int q = 0; q < CO / 4; ++q)
        reinterpret_cast<float4 *>(out + vox * CO)[q] = make_float4(o[q * 4], o[q * 4 + 1], o[q * 4 + 2], o[q * 4 + 3]);
}

// weight (Ci,Co,3,3,3); out[o] += in[i] * w[k] for o = stride*i - 1 + k
__global__ __launch_bounds__(256) void deconv3d_direct_kernel(
    const float *__restrict__ in, const float *__restrict__ w, const float *__restrict__ scale,
    const float *__restrict__ shift, const float *__restrict__ residual, ConvGeom g,
    float *__restrict__ out) {
    const int64_t total = (int64_t)g.B * g.Co * g.Do * g.Ho * g.Wo;
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    int b, co, z, y, x;
    decode_out(g, idx, b, co, z, y, x);
    float acc = 0.0f;
    const int s = g.stride;
    for (int ci = 0; ci < g.Ci; ++ci) {
        const float *ip = in + (int64_t)b * g.i_b + (int64_t)ci * g.i_c;
        const float *wp = w + ((int64_t)ci * g.Co + co) * 27;
#pragma unroll
        for (int kz = 0; kz < 3; ++kz) {
            int tz = z + 1 - kz;
            if (tz < 0 || (tz % s) != 0 || tz / s >= g.D) continue;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                int ty = y + 1 - ky;
                if (ty < 0 || (ty % s) != 0 || ty / s >= g.H) continue;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    int tx = x + 1 - kx;
                    if (tx < 0 || (tx % s) != 0 || tx / s >= g.W) continue;
                    acc = fmaf(ip[(tz / s) * g.i_z + (ty / s) * g.i_y + (tx / s) * g.i_x],
                               wp[(kz * 3 + ky) * 3 + kx], acc);
                }
            }
        }
    }
    int64_t o = (int64_t)b * g.o_b + (int64_t)co * g.o_c + z * g.o_z + y * g.o_y + x * g.o_x;
    out[o] = epilogue(acc, co, o, scale, shift, residual, g.relu);
}

int conv3d_direct_launch(const float *in, const float *weight, const float *scale,
                         const float *shift, const float *residual, int relu, int transposed,
                         int B, int Cin, int Cout, int D, int H, int W, int stride, int layout,
                         float *out, hipStream_t st) {
    ConvGeom g;
    g.B = B; g.Ci = Cin; g.Co = Cout; g.D = D; g.H = H; g.W = W;
    g.stride = stride; g.relu = relu;
    if (transposed) {
        g.Do = D * stride; g.Ho = H * stride; g.Wo = W * stride;
    } else {
        g.Do = (D - 1) / stride + 1; g.Ho = (H - 1) / stride + 1; g.Wo = (W - 1) / stride + 1;
    }
    g.channels_last = layout == MVS_LAYOUT_NHWC;
    if (g.channels_last) {
        g.i_c = 1; g.i_x = Cin; g.i_y = (int64_t)W * Cin; g.i_z = (int64_t)H * g.i_y;
        g.i_b = (int64_t)D * g.i_z;
        g.o_c = 1; g.o_x = Cout; g.o_y = (int64_t)g.Wo * Cout; g.o_z = (int64_t)g.Ho * g.o_y;
        g.o_b = (int64_t)g.Do * g.o_z;
    } else {
        g.i_x = 1; g.i_y = W; g.i_z = (int64_t)H * W; g.i_c = (int64_t)D * g.i_z;
        g.i_b = (int64_t)Cin * g.i_c;
        g.o_x = 1; g.o_y = g.Wo; g.o_z = (int64_t)g.Ho * g.Wo; g.o_c = (int64_t)g.Do * g.o_z;
        g.o_b = (int64_t)Cout * g.o_c;
    }
    const int64_t total = (int64_t)B * Cout * g.Do * g.Ho * g.Wo;
    const int64_t blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffLL) {
        set_error("conv3d(direct): problem too large");
        return MVS_EINVAL;
    }
    if (!transposed && stride == 1 && Cin == 1 && Cout == 8 && g.channels_last) {
        const int64_t vblocks = ((int64_t)B * D * H * W + 255) / 256;
        hipLaunchKernelGGL((conv3d_cin1_cl_kernel<8>), dim3((unsigned)vblocks), dim3(256), 0, st, in, weight, scale,
                           shift, residual, g, out);
        return check_launch("mvs_conv3d_f32(direct, Cin=1)");
    }
    if (transposed)
        hipLaunchKernelGGL(deconv3d_direct_kernel, dim3((unsigned)blocks), dim3(256), 0, st, in,
                           weight, scale, shift, residual, g, out);
    else
        hipLaunchKernelGGL(conv3d_direct_kernel, dim3((unsigned)blocks), dim3(256), 0, st, in,
                           weight, scale, shift, residual, g, out);
    return check_launch("mvs_conv3d_f32(direct)");
}

}  // namespace mvs
