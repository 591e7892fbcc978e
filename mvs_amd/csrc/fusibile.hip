// Depth-map fusion of the step behind the path (second half): the `fusibile`
// kernel of the reference tree (fusibile/fusibile.cu:138-277) for gfx950.  One thread per pixel of
// the reference camera: lift the pixel with its depth to a 3D point, project it into every other
// view, sample that view's (normal, depth) map bilinearly, and count the views whose depth agrees
// in disparity space (|f b / d - f b / d'| < disp_thresh, b = distance of the camera centres) and
// whose normal agrees (angle < normal_thresh); a pixel with at least num_consistent agreeing views
// emits the average of the lifted points (and normals, colours).
//
// Parity of this row is UNPINNED: fusibile needs CUDA and OpenCV to build, neither is in the image,
// and the reference holds no fixture for it.  Two CUDA-specific behaviours are restated from their
// documentation: (a) tex2D with cudaFilterModeLinear on unnormalised coordinates = bilinear
// interpolation at (x - 0.5, y - 0.5) with the weights held in 1.8 fixed point and the indices
// clamped (cudaAddressModeWrap falls back to clamping for unnormalised coordinates); (b) nvcc
// contracts a * b + c into FMAs (this file enables contraction for the same reason).  The numpy
// restatement oracle/fusibile.py follows the same statements and is what the tests compare with.
#include "mvs_common.h"

namespace mvs {

struct FuseCam {
    float P[12];      // projection matrix, row major
    float Minv[9];    // inverse of P[:, :3]
    float Pcol[3];    // P[:, 3]
    float C[3];       // camera centre
    float f;          // focal length (K(0,0) of camera 0 for every camera, as the reference sets it)
};

struct FuseArgs {
    const float4 *nd;        // [N,H,W] (nx, ny, nz, depth)
    const float4 *color;     // [N,H,W] (b, g, r, a) floats 0..255, or null
    const FuseCam *cams;     // [N]
    float4 *out_point, *out_normal, *out_color;   // [H,W]
    int N, H, W, ref;
    float disp_thresh, normal_thresh;
    int num_consistent;
};

#pragma clang fp contract(fast)

__device__ __forceinline__ float4 tex_linear(const float4 *__restrict__ img, int H, int W, float x, float y) {
    // tex2D(tex, x, y), linear filter, unnormalised coordinates, clamped addressing
    const float xb = x - 0.5f, yb = y - 0.5f;
    const float fx = floorf(xb), fy = floorf(yb);
    const float a = floorf((xb - fx) * 256.0f + 0.5f) * (1.0f / 256.0f);   // 1.8 fixed point
    const float b = floorf((yb - fy) * 256.0f + 0.5f) * (1.0f / 256.0f);
    const int i0 = min(max((int)fx, 0), W - 1), i1 = min(max((int)fx + 1, 0), W - 1);
    const int j0 = min(max((int)fy, 0), H - 1), j1 = min(max((int)fy + 1, 0), H - 1);
    const float4 t00 = img[(size_t)j0 * W + i0], t10 = img[(size_t)j0 * W + i1];
    const float4 t01 = img[(size_t)j1 * W + i0], t11 = img[(size_t)j1 * W + i1];
    const float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    return make_float4(w00 * t00.x + w10 * t10.x + w01 * t01.x + w11 * t11.x,
                       w00 * t00.y + w10 * t10.y + w01 * t01.y + w11 * t11.y,
                       w00 * t00.z + w10 * t10.z + w01 * t01.z + w11 * t11.z,
                       w00 * t00.w + w10 * t10.w + w01 * t01.w + w11 * t11.w);
}

// get3Dpoint_cu (fusibile.cu:56-65)
__device__ __forceinline__ float3 lift(const FuseCam &c, int px, int py, float depth) {
    const float x = depth * (float)px - c.Pcol[0], y = depth * (float)py - c.Pcol[1], z = depth - c.Pcol[2];
    return make_float3(c.Minv[0] * x + c.Minv[1] * y + c.Minv[2] * z, c.Minv[3] * x + c.Minv[4] * y + c.Minv[5] * z,
                       c.Minv[6] * x + c.Minv[7] * y + c.Minv[8] * z);
}

__global__ __launch_bounds__(256) void fusibile_kernel(FuseArgs a) {
    const int px = blockIdx.x * 32 + (threadIdx.x & 31), py = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (px >= a.W || py >= a.H) return;
    const size_t center = (size_t)py * a.W + px, plane = (size_t)a.H * a.W;
    const FuseCam &rc = a.cams[a.ref];
    const float4 normal = a.nd[a.ref * plane + center];
    const float3 X = lift(rc, px, py, normal.w);
    float3 cX = X;
    float4 cN = normal;
    float4 cT = a.color ? a.color[a.ref * plane + center] : make_float4(0.f, 0.f, 0.f, 0.f);
    int consistent = 0;
    for (int i = 0; i < a.N; ++i) {
        if (i == a.ref) continue;
        const FuseCam &c = a.cams[i];
        // project_on_camera (fusibile.cu:126-132)
        const float tx = c.P[0] * X.x + c.P[1] * X.y + c.P[2] * X.z + c.P[3];
        const float ty = c.P[4] * X.x + c.P[5] * X.y + c.P[6] * X.z + c.P[7];
        const float tz = c.P[8] * X.x + c.P[9] * X.y + c.P[10] * X.z + c.P[11];
        const float qx = tx / tz, qy = ty / tz, depth = tz;
        if (!(qx >= 0.0f && qx < (float)a.W && qy >= 0.0f && qy < (float)a.H)) continue;
        const float4 nd = tex_linear(a.nd + i * plane, a.H, a.W, qx + 0.5f, qy + 0.5f);
        const float dx = rc.C[0] - c.C[0], dy = rc.C[1] - c.C[1], dz = rc.C[2] - c.C[2];
        const float baseline = sqrtf(dx * dx + dy * dy + dz * dz);
        const float disp0 = rc.f * baseline / depth, disp1 = rc.f * baseline / nd.w;
        if (!(fabsf(disp0 - disp1) < a.disp_thresh)) continue;
        float angle = acosf(nd.x * normal.x + nd.y * normal.y + nd.z * normal.z);
        if (angle != angle) angle = 0.0f;      // the dot product was 1 (or beyond): identical normals
        if (!(angle < a.normal_thresh)) continue;
        const float3 tX = lift(c, (int)qx, (int)qy, nd.w);
        cX.x += tX.x; cX.y += tX.y; cX.z += tX.z;
        cN.x += nd.x; cN.y += nd.y; cN.z += nd.z;
        if (a.color) {
            const float4 t = tex_linear(a.color + i * plane, a.H, a.W, qx + 0.5f, qy + 0.5f);
            cT.x += t.x; cT.y += t.y; cT.z += t.z;
        }
        ++consistent;
    }
    const float k = (float)consistent + 1.0f;
    if (consistent >= a.num_consistent) {
        a.out_point[center] = make_float4(cX.x / k, cX.y / k, cX.z / k, 0.f);
        a.out_normal[center] = make_float4(cN.x / k, cN.y / k, cN.z / k, 0.f);
        if (a.out_color) a.out_color[center] = make_float4(cT.x / k, cT.y / k, cT.z / k, 0.f);
    } else {
        a.out_point[center] = make_float4(0.f, 0.f, 0.f, 0.f);   // "no point": the host keeps X != 0 only
        a.out_normal[center] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.out_color) a.out_color[center] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

}  // namespace mvs

using namespace mvs;

extern "C" int mvs_fusibile_fuse_f32(const float *normals_depths, const float *colors, const float *cams, int N,
                                     int H, int W, int ref, float disp_thresh, float normal_thresh,
                                     int num_consistent, float *out_points, float *out_normals, float *out_colors,
                                     void *stream) {
    if (!normals_depths || !cams || !out_points || !out_normals || N < 1 || H < 1 || W < 1 || ref < 0 || ref >= N ||
        (colors && !out_colors)) {
        set_error("mvs_fusibile_fuse_f32: invalid argument");
        return MVS_EINVAL;
    }
    static_assert(sizeof(FuseCam) == 28 * sizeof(float), "camera record = 28 floats");
    FuseArgs a;
    a.nd = reinterpret_cast<const float4 *>(normals_depths);
    a.color = reinterpret_cast<const float4 *>(colors);
    a.cams = reinterpret_cast<const FuseCam *>(cams);
    a.out_point = reinterpret_cast<float4 *>(out_points);
    a.out_normal = reinterpret_cast<float4 *>(out_normals);
    a.out_color = colors ? reinterpret_cast<float4 *>(out_colors) : nullptr;
    a.N = N; a.H = H; a.W = W; a.ref = ref;
    a.disp_thresh = disp_thresh; a.normal_thresh = normal_thresh; a.num_consistent = num_consistent;
    hipLaunchKernelGGL(fusibile_kernel, dim3((W + 31) / 32, (H + 7) / 8), dim3(256), 0, as_stream(stream), a);
    return check_launch("mvs_fusibile_fuse_f32");
}
