// A caller of include/mvs_hip.h that is not Python ("a standalone C++ bench/test
// binary on the GPU box"): warp -> fused variance (persistent kernel, caller workspace) ->
// CostRegNet in one call -> softmax regression, on a dump of one of the reference-generated
// golden cases (tests/test_cpp_abi.py writes the dump from tests/golden/*.npz), compared with
// the reference's own outputs.  Test infrastructure.
//
//   hipcc --offload-arch=gfx950 -O2 -Iinclude tests/cpp/abi_chain.cpp -Lmvs_amd/csrc -lmvs_hip \
//         -Wl,-rpath,'$ORIGIN/../../mvs_amd/csrc' -o tests/cpp/abi_chain
//   tests/cpp/abi_chain <dump-dir>      -> one JSON line, exit status 0 iff every check passed
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#include "mvs_hip.h"

#define HIP_OK(x)                                                                              \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                       \
            std::exit(2);                                                                      \
        }                                                                                      \
    } while (0)
#define MVS_OK_(x)                                                                             \
    do {                                                                                       \
        int r_ = (x);                                                                          \
        if (r_ != MVS_OK) {                                                                    \
            std::fprintf(stderr, "%s -> %d: %s\n", #x, r_, mvs_last_error_string());           \
            std::exit(3);                                                                      \
        }                                                                                      \
    } while (0)

struct Dump {
    std::map<std::string, std::vector<float>> arr;
    std::map<std::string, long> dims;
    const std::vector<float> &get(const std::string &k) const {
        auto it = arr.find(k);
        if (it == arr.end()) { std::fprintf(stderr, "dump has no '%s'\n", k.c_str()); std::exit(2); }
        return it->second;
    }
};

static Dump load(const std::string &dir) {
    Dump d;
    FILE *m = std::fopen((dir + "/manifest.txt").c_str(), "r");
    FILE *b = std::fopen((dir + "/blob.bin").c_str(), "rb");
    if (!m || !b) { std::fprintf(stderr, "cannot open dump in %s\n", dir.c_str()); std::exit(2); }
    char name[256];
    long n;
    while (std::fscanf(m, "%255s %ld", name, &n) == 2) {
        if (name[0] == '#') { d.dims[name + 1] = n; continue; }   // "#B 1": a dimension, not an array
        std::vector<float> v((size_t)n);
        if (std::fread(v.data(), sizeof(float), (size_t)n, b) != (size_t)n) { std::fprintf(stderr, "short blob\n"); std::exit(2); }
        d.arr[name] = std::move(v);
    }
    std::fclose(m);
    std::fclose(b);
    return d;
}

static float *to_dev(const std::vector<float> &h) {
    float *p;
    HIP_OK(hipMalloc(&p, h.size() * sizeof(float)));
    HIP_OK(hipMemcpy(p, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    return p;
}
static float *dev_alloc(size_t n) {
    float *p;
    HIP_OK(hipMalloc(&p, (n ? n : 1) * sizeof(float)));
    return p;
}
static std::vector<float> to_host(const float *p, size_t n) {
    std::vector<float> h(n);
    HIP_OK(hipMemcpy(h.data(), p, n * sizeof(float), hipMemcpyDeviceToHost));
    return h;
}
static double maxabs(const std::vector<float> &a, const std::vector<float> &b) {
    double m = 0;
    for (size_t i = 0; i < a.size(); ++i) m = std::fmax(m, std::fabs((double)a[i] - (double)b[i]));
    return m;
}

int main(int argc, char **argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: %s <dump-dir>\n", argv[0]); return 2; }
    const Dump d = load(argv[1]);
    const int B = (int)d.dims.at("B"), V = (int)d.dims.at("V"), C = (int)d.dims.at("C"), D = (int)d.dims.at("D"),
              H = (int)d.dims.at("H"), W = (int)d.dims.at("W");
    const size_t plane = (size_t)H * W, vol = (size_t)D * plane;
    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));

    // ---- inputs: features [V,B,C,H,W] planar (the reference's layout), rot_trans [V-1,B,12], depth planes [B,D]
    float *feats = to_dev(d.get("features")), *rt = to_dev(d.get("rot_trans")), *dv = to_dev(d.get("depth_values"));
    const size_t fmap = (size_t)B * C * plane;

    // (1) un-fused warp of source view 1 (mvs_warp_fwd_f32) vs the reference's homo_warping
    float *warped = dev_alloc((size_t)B * C * vol);
    MVS_OK_(mvs_warp_fwd_f32(feats + fmap, rt, dv, 0, B, C, D, H, W, 0, warped, st));

    // (2) fused warp + variance through the workspace entry point: C16 features -> C8 volume
    float *f16 = dev_alloc((size_t)V * fmap);
    MVS_OK_(mvs_nchw_to_nhwc_f32(feats, f16, V * B * (C / 16), 16, (int64_t)plane, st));
    const size_t vws = mvs_costvol_variance_workspace_bytes(0, B, V, C, D, H, W, MVS_LAYOUT_C16);
    void *var_ws = nullptr;
    if (vws) HIP_OK(hipMalloc(&var_ws, vws));
    float *var8 = dev_alloc((size_t)B * C * vol);
    MVS_OK_(mvs_costvol_variance_fwd_ws_f32(f16, f16 + fmap, rt, dv, 0, B, V, C, D, H, W, 0, 0, MVS_LAYOUT_C16,
                                            MVS_LAYOUT_C8, 0, var8, var_ws, vws, st));
    // the same volume in the reference's planar layout, for the comparison
    float *var_pl = dev_alloc((size_t)B * C * vol);
    MVS_OK_(mvs_costvol_variance_fwd_f32(feats, feats + fmap, rt, dv, 0, B, V, C, D, H, W, 0, 0, MVS_LAYOUT_NCHW,
                                         MVS_LAYOUT_NCHW, var_pl, st));

    // (3) CostRegNet in one call
    static const char *names[11] = {"conv0", "conv1", "conv2", "conv3", "conv4", "conv5", "conv6", "conv7", "conv9", "conv11", "prob"};
    static const int cin[11] = {32, 8, 16, 16, 32, 32, 64, 64, 32, 16, 8}, cout[11] = {8, 16, 16, 32, 32, 64, 64, 32, 16, 8, 1};
    static const int stride[11] = {1, 2, 1, 2, 1, 2, 1, 2, 2, 2, 1}, transposed[11] = {0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 0};
    mvs_conv_layer layers[11];
    for (int i = 0; i < 11; ++i) {
        const std::string n = names[i];
        layers[i].weight = to_dev(d.get(n + ".weight"));
        layers[i].scale = d.arr.count(n + ".scale") ? to_dev(d.get(n + ".scale")) : nullptr;
        layers[i].shift = d.arr.count(n + ".shift") ? to_dev(d.get(n + ".shift")) : nullptr;
        const int64_t pf = mvs_conv3d_packed_weight_floats(transposed[i], cin[i], cout[i], stride[i]);
        float *packed = nullptr;
        if (pf > 0) {
            packed = dev_alloc((size_t)pf);
            MVS_OK_(mvs_conv3d_pack_weights_f32(layers[i].weight, transposed[i], cin[i], cout[i], stride[i], packed, st));
        }
        layers[i].packed = packed;
        // the split-operand bf16 kernels, as the Python host runs the network: conv0; the stride-1 16/32/64-channel
        // layers; the transposed layers
        layers[i].packed_split = nullptr;
        void *ps = nullptr;
        if (i == 0) {
            HIP_OK(hipMalloc(&ps, mvs_conv3d_bf16x6_packed_bytes(cin[0])));
            MVS_OK_(mvs_conv3d_pack_weights_bf16x6_f32(layers[0].weight, cin[0], ps, st));
        } else if (!transposed[i] && stride[i] == 1 && mvs_conv_split_supported(3, cin[i], cout[i], 1)) {
            HIP_OK(hipMalloc(&ps, mvs_conv_split_packed_bytes(3, cin[i], cout[i], 1)));
            MVS_OK_(mvs_conv_split_pack_weights_f32(layers[i].weight, 3, cin[i], cout[i], 1, ps, st));
        } else if (transposed[i] && mvs_deconv_split_supported(cin[i], cout[i])) {
            HIP_OK(hipMalloc(&ps, mvs_deconv_split_packed_bytes(cin[i], cout[i])));
            MVS_OK_(mvs_deconv_split_pack_weights_f32(layers[i].weight, cin[i], cout[i], ps, st));
        }
        layers[i].packed_split = ps;
    }
    const size_t cws = mvs_costreg_workspace_bytes(B, 8, D, H, W);
    if (!cws) { std::fprintf(stderr, "costreg workspace: unsupported size\n"); return 3; }
    void *cr_ws;
    HIP_OK(hipMalloc(&cr_ws, cws));
    float *cost = dev_alloc((size_t)B * vol);
    MVS_OK_(mvs_costreg_fwd_f32(var8, MVS_LAYOUT_C8, layers, B, C, 8, D, H, W, 0, cr_ws, cws, cost, st));

    // (3b) the same network on the two-piece fp16 kernels, as the Python host runs it by default: the sweep entry hands over the
    // absmax block of the volume (conv0's operand scale), the layers chain theirs through the workspace
    const void *packed_f16[11];
    for (int i = 0; i < 11; ++i) {
        void *pf = nullptr;
        if (i == 0) {
            HIP_OK(hipMalloc(&pf, mvs_conv3d_f16x3_packed_bytes(cin[0])));
            MVS_OK_(mvs_conv3d_pack_weights_f16x3_f32(layers[0].weight, cin[0], pf, st));
        } else if (!transposed[i] && mvs_conv_split_supported(3, cin[i], cout[i], stride[i]) && (stride[i] == 1 || cin[i] == 8)) {
            HIP_OK(hipMalloc(&pf, mvs_conv_split_f16_packed_bytes(3, cin[i], cout[i], stride[i])));
            MVS_OK_(mvs_conv_split_pack_weights_f16_f32(layers[i].weight, 3, cin[i], cout[i], stride[i], pf, st));
        } else if (transposed[i] && mvs_deconv_split_supported(cin[i], cout[i])) {
            HIP_OK(hipMalloc(&pf, mvs_deconv_split_f16_packed_bytes(cin[i], cout[i])));
            MVS_OK_(mvs_deconv_split_pack_weights_f16_f32(layers[i].weight, cin[i], cout[i], pf, st));
        }
        packed_f16[i] = pf;
    }
    void *amax;
    HIP_OK(hipMalloc(&amax, MVS_ABSMAX_WORDS * 4));
    float *var8b = dev_alloc((size_t)B * C * vol), *cost2 = dev_alloc((size_t)B * vol);
    MVS_OK_(mvs_costvol_variance_fwd_ws2_f32(f16, f16 + fmap, rt, dv, 0, B, V, C, D, H, W, 0, 0, MVS_LAYOUT_C16,
                                             MVS_LAYOUT_C8, 0, var8b, var_ws, vws, amax, st));
    MVS_OK_(mvs_costreg_fwd2_f32(var8b, MVS_LAYOUT_C8, layers, packed_f16, B, C, 8, D, H, W, 0, cr_ws, cws, amax, cost2, st));
    float *depth2 = dev_alloc((size_t)B * plane), *conf2 = dev_alloc((size_t)B * plane);
    MVS_OK_(mvs_softmax_regress_conf_f32(cost2, dv, 0, 0, B, D, H, W, depth2, conf2, nullptr, st));

    // (4) softmax + depth regression + photometric confidence
    float *depth = dev_alloc((size_t)B * plane), *conf = dev_alloc((size_t)B * plane);
    MVS_OK_(mvs_softmax_regress_conf_f32(cost, dv, 0, 0, B, D, H, W, depth, conf, nullptr, st));
    HIP_OK(hipStreamSynchronize(st));

    const double e_warp = d.arr.count("warped1") ? maxabs(to_host(warped, (size_t)B * C * vol), d.get("warped1")) : -1;
    const double e_var = maxabs(to_host(var_pl, (size_t)B * C * vol), d.get("variance"));
    const double e_cost = maxabs(to_host(cost, (size_t)B * vol), d.get("cost"));
    const double e_depth = maxabs(to_host(depth, (size_t)B * plane), d.get("depth"));
    const double e_conf = maxabs(to_host(conf, (size_t)B * plane), d.get("confidence"));
    const double e_cost2 = maxabs(to_host(cost2, (size_t)B * vol), d.get("cost"));
    const double e_depth2 = maxabs(to_host(depth2, (size_t)B * plane), d.get("depth"));
    const double e_conf2 = maxabs(to_host(conf2, (size_t)B * plane), d.get("confidence"));
    // the range guard of the two-piece layers (conv_guard.h) must have stayed silent on this volume
    unsigned long long guard = ~0ull;
    MVS_OK_(mvs_guard_fallback_count(&guard));
    const bool ok = e_warp < 1e-6 && e_var < 1e-6 && e_depth < 1e-3 && e_conf < 1e-3 && e_depth2 < 1e-3 && e_conf2 < 1e-3 && guard == 0;
    std::printf("{\"version\": %d, \"arch\": \"%s\", \"warp_maxabs\": %.3g, \"variance_maxabs\": %.3g, \"cost_maxabs\": %.3g, "
                "\"depth_maxabs_mm\": %.3g, \"confidence_maxabs\": %.3g, \"two_piece_cost_maxabs\": %.3g, "
                "\"two_piece_depth_maxabs_mm\": %.3g, \"two_piece_confidence_maxabs\": %.3g, \"variance_workspace_bytes\": %zu, "
                "\"costreg_workspace_bytes\": %zu, \"guard_fallbacks\": %llu, \"ok\": %s}\n",
                mvs_version(), mvs_arch(), e_warp, e_var, e_cost, e_depth, e_conf, e_cost2, e_depth2, e_conf2, vws, cws, guard,
                ok ? "true" : "false");
    return ok ? 0 : 1;
}
