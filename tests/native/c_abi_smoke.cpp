// A torch-free consumer of the C ABI: plain HIP runtime + include/*.h, linked against libsimplenerf_hip.so.
// Generates the rays of a small frame, their coarse depths, a shuffled epoch of pixel indices, Philox draws, one Adam
// step and a display conversion, runs a 4x128 MLP forward in the three arithmetic modes and its training forward +
// backward, the one-call render ops (snerf_render_forward / _backward) against the same stages issued one by one, and
// checks invariants on the host.  Built and run by tests/test_gpu_native.py.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "simplenerf_hip.h"
#include "simplenerf_train.h"

#define HIP_OK(x)                                                                      \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } \
    } while (0)
#define SNERF_OK_(x)                                                                   \
    do {                                                                               \
        if ((x) != 0) { std::printf("ABI error at %s:%d: %s\n", __FILE__, __LINE__, snerf_last_error()); return 3; } \
    } while (0)
#define CHECK(cond)                                                                    \
    do {                                                                               \
        if (!(cond)) { std::printf("check failed at %s:%d: %s\n", __FILE__, __LINE__, #cond); return 4; } \
    } while (0)

template <typename T>
static T* dev(size_t n) {
    void* p = nullptr;
    if (hipMalloc(&p, n * sizeof(T)) != hipSuccess) std::abort();
    return static_cast<T*>(p);
}
template <typename T>
static std::vector<T> host(const T* d, size_t n) {
    std::vector<T> h(n);
    if (hipMemcpy(h.data(), d, n * sizeof(T), hipMemcpyDeviceToHost) != hipSuccess) std::abort();
    return h;
}

int main() {
    CHECK(snerf_abi_version() == SNERF_ABI_VERSION);
    const int h = 24, w = 32;
    const long long n = (long long)h * w;
    const float intrinsic[9] = {40.f, 0.f, 16.f, 0.f, 40.f, 12.f, 0.f, 0.f, 1.f};
    const float pose[16] = {1, 0, 0, 0.1f, 0, 1, 0, -0.2f, 0, 0, 1, 0.3f, 0, 0, 0, 1};

    // K1 + K2
    float *rays_o = dev<float>(3 * n), *rays_d = dev<float>(3 * n), *dirs = dev<float>(3 * n), *o_ndc = dev<float>(3 * n),
          *d_ndc = dev<float>(3 * n);
    SNERF_OK_(snerf_generate_rays(h, w, intrinsic, pose, 0.f, 1, 1.f, 0, n, rays_o, rays_d, dirs, o_ndc, d_ndc, nullptr));
    float *near = dev<float>(n), *far = dev<float>(n), *depths = dev<float>(n * 16);
    std::vector<float> zeros(n, 0.f), ones(n, 1.f);
    HIP_OK(hipMemcpy(near, zeros.data(), n * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(far, ones.data(), n * 4, hipMemcpyHostToDevice));
    SNERF_OK_(snerf_coarse_depths(near, far, n, 16, 0, nullptr, depths, nullptr));
    HIP_OK(hipDeviceSynchronize());
    auto hd = host(dirs, 3 * n);
    auto ho = host(rays_o, 3 * n);
    auto hz = host(depths, n * 16);
    for (long long i = 0; i < n; ++i) {
        const float nrm = std::sqrt(hd[3 * i] * hd[3 * i] + hd[3 * i + 1] * hd[3 * i + 1] + hd[3 * i + 2] * hd[3 * i + 2]);
        CHECK(std::fabs(nrm - 1.f) < 1e-6f);
        CHECK(ho[3 * i] == 0.1f && ho[3 * i + 1] == -0.2f && ho[3 * i + 2] == 0.3f);
        CHECK(hz[16 * i] == 0.f && hz[16 * i + 15] == 1.f && hz[16 * i + 7] < hz[16 * i + 8]);
    }

    // B2: one epoch of the index stream is a permutation of the pixels
    long long* idx = dev<long long>(n);
    SNERF_OK_(snerf_shuffled_indices(42, 0, 0, n, n, nullptr, 1, h, w, 0, h, 0, w, idx, nullptr));
    HIP_OK(hipDeviceSynchronize());
    auto hi = host(idx, n);
    std::vector<int> seen(n, 0);
    for (long long v : hi) { CHECK(v >= 0 && v < n); seen[v]++; }
    for (int c : seen) CHECK(c == 1);

    // B3: Philox known answer -- counter (row 0, block 0, stream 0), key 0 -> 6627e8d5 e169c58d bc57ac4c 9b00dbd8
    float* u = dev<float>(4);
    SNERF_OK_(snerf_random_uniform(0, 0, 0, nullptr, 1, 4, u, nullptr));
    HIP_OK(hipDeviceSynchronize());
    auto hu = host(u, 4);
    const unsigned kat[4] = {0x6627e8d5u, 0xe169c58du, 0xbc57ac4cu, 0x9b00dbd8u};
    for (int k = 0; k < 4; ++k) CHECK(hu[k] == (float)(kat[k] >> 8) * 5.9604644775390625e-8f);

    // O1: one Adam step on two tensors (the second without a gradient is left alone)
    std::vector<float> p0(1000, 1.f), g0(1000, 0.5f), p1(10, 2.f);
    float *dp0 = dev<float>(1000), *dg0 = dev<float>(1000), *dm0 = dev<float>(1000), *dv0 = dev<float>(1000);
    float *dp1 = dev<float>(10), *dm1 = dev<float>(10), *dv1 = dev<float>(10);
    HIP_OK(hipMemcpy(dp0, p0.data(), 4000, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dg0, g0.data(), 4000, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dp1, p1.data(), 40, hipMemcpyHostToDevice));
    HIP_OK(hipMemset(dm0, 0, 4000)); HIP_OK(hipMemset(dv0, 0, 4000)); HIP_OK(hipMemset(dm1, 0, 40)); HIP_OK(hipMemset(dv1, 0, 40));
    float* params[2] = {dp0, dp1};
    const float* grads[2] = {dg0, nullptr};
    float* ms[2] = {dm0, dm1};
    float* vs[2] = {dv0, dv1};
    const long long sizes[2] = {1000, 10};
    SNERF_OK_(snerf_adam_step(params, grads, ms, vs, sizes, 2, 1, 1e-3, 0.9, 0.999, 1e-8, nullptr));
    HIP_OK(hipDeviceSynchronize());
    auto q0 = host(dp0, 1000);
    auto q1 = host(dp1, 10);
    for (float v : q0) CHECK(std::fabs(v - (1.f - 1e-3f)) < 1e-6f);   // first Adam step moves by lr * sign(g)
    for (float v : q1) CHECK(v == 2.f);

    // f3: display conversion rounds half to even and clips
    const float rgb[6] = {0.5f / 255.f, 1.5f / 255.f, 2.f, -1.f, 0.25f, 1.f};
    const float dep[2] = {-3.f, 7.f};
    float *drgb = dev<float>(6), *ddep = dev<float>(2), *ddep_out = dev<float>(2);
    unsigned char* img = dev<unsigned char>(6);
    HIP_OK(hipMemcpy(drgb, rgb, 24, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(ddep, dep, 8, hipMemcpyHostToDevice));
    SNERF_OK_(snerf_to_display(drgb, ddep, 2, img, ddep_out, nullptr));
    HIP_OK(hipDeviceSynchronize());
    auto himg = host(img, 6);
    auto hdep = host(ddep_out, 2);
    CHECK(himg[0] == 0 && himg[1] == 2 && himg[2] == 255 && himg[3] == 0 && himg[4] == 64 && himg[5] == 255);
    CHECK(hdep[0] == 0.f && hdep[1] == 7.f);

    // K3 / K7: a 4x128 view-dependent MLP on those rays -- the three arithmetic modes agree, gradients are finite
    {
        snerf_mlp_desc desc = {4, 128, 1, 64, 10, 4, -1, 1, 1};
        const int np = snerf_mlp_num_params(&desc);
        CHECK(np == 16);
        const int shape[16][2] = {{128, 63}, {128, 1}, {128, 128}, {128, 1}, {128, 128}, {128, 1}, {128, 128}, {128, 1},
                                  {1, 128},  {1, 1},   {128, 128}, {128, 1}, {64, 155},  {64, 1},  {3, 64},    {3, 1}};
        std::vector<float*> params(np), g32(np), g16(np);
        unsigned state = 12345u;
        for (int i = 0; i < np; ++i) {
            const size_t count = (size_t)shape[i][0] * shape[i][1];
            std::vector<float> hostw(count);
            const float scale = (i & 1) ? 0.05f : 1.5f / std::sqrt((float)shape[i][1]);
            for (float& v : hostw) {
                state = state * 1664525u + 1013904223u;
                v = scale * ((float)(state >> 8) / 8388608.f - 1.f);
            }
            params[i] = dev<float>(count); g32[i] = dev<float>(count); g16[i] = dev<float>(count);
            HIP_OK(hipMemcpy(params[i], hostw.data(), count * 4, hipMemcpyHostToDevice));
        }
        float* packed = dev<float>(snerf_mlp_packed_floats(&desc));
        SNERF_OK_(snerf_mlp_pack(&desc, params.data(), np, packed, nullptr));
        const int S = 16;
        float *sig[3], *col[3];
        for (int m = 0; m < 3; ++m) {
            sig[m] = dev<float>(n * S); col[m] = dev<float>(3 * n * S);
            SNERF_OK_(snerf_mlp_forward(&desc, packed, o_ndc, d_ndc, dirs, depths, n, S, nullptr, sig[m], col[m], m, nullptr));
        }
        HIP_OK(hipDeviceSynchronize());
        auto c0 = host(col[0], 3 * n * S), c1 = host(col[1], 3 * n * S), c2 = host(col[2], 3 * n * S);
        float e1 = 0.f, e2 = 0.f, lo = 1.f, hi = 0.f;
        for (size_t i = 0; i < c0.size(); ++i) {
            e1 = std::fmax(e1, std::fabs(c1[i] - c0[i])); e2 = std::fmax(e2, std::fabs(c2[i] - c0[i]));
            lo = std::fmin(lo, c0[i]); hi = std::fmax(hi, c0[i]);
        }
        CHECK(hi - lo > 0.05f);                 // not a constant field
        CHECK(e1 < 1e-5f && e2 < 5e-3f);        // f16x3: fp32-grade; f16: its stated tolerance
        // training forward + backward, fp32 and the 16-bit mode, with d(sigma) = d(rgb) = 1
        float* saved = dev<float>(snerf_mlp_saved_floats(&desc, n, S));
        float* work = dev<float>(snerf_mlp_backward_workspace_floats(&desc, n, S));
        float *dsig = dev<float>(n * S), *dcol = dev<float>(3 * n * S);
        std::vector<float> one(3 * n * S, 1.f);
        HIP_OK(hipMemcpy(dsig, one.data(), n * S * 4, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(dcol, one.data(), 3 * n * S * 4, hipMemcpyHostToDevice));
        const int modes[2] = {SNERF_PRECISION_FP32, SNERF_PRECISION_F16};
        for (int k = 0; k < 2; ++k) {
            std::vector<float*>& g = k == 0 ? g32 : g16;
            SNERF_OK_(snerf_mlp_forward_train(&desc, packed, o_ndc, d_ndc, dirs, depths, n, S, nullptr, sig[0], col[0], saved,
                                              modes[k], nullptr));
            SNERF_OK_(snerf_mlp_backward(&desc, packed, saved, sig[0], col[0], dsig, dcol, n, S, work, g.data(), np, modes[k],
                                         0, nullptr));
        }
        HIP_OK(hipDeviceSynchronize());
        for (int i = 0; i < np; ++i) {
            const size_t count = (size_t)shape[i][0] * shape[i][1];
            auto a = host(g32[i], count), b = host(g16[i], count);
            double num = 0.0, den = 0.0;
            for (size_t j = 0; j < count; ++j) {
                CHECK(std::isfinite(a[j]) && std::isfinite(b[j]));
                num += ((double)a[j] - b[j]) * ((double)a[j] - b[j]); den += (double)a[j] * a[j];
            }
            CHECK(den > 0.0 && std::sqrt(num / den) < 0.1);
        }
        CHECK(snerf_mlp_forward(&desc, packed, o_ndc, d_ndc, dirs, depths, n, S, nullptr, sig[0], col[0], 7, nullptr) ==
              SNERF_E_UNSUPPORTED);

        // ---- the one-call render ops: coarse + fine pass of that MLP (same weights at both levels) in ONE call, against
        // the same stages issued one by one; then render_backward twice, the second time accumulating
        const int Sc = 16, Sf = 16, Sm = Sc + Sf;
        snerf_render_config cfg = {1, 0, 0, Sc, Sf, SNERF_PRECISION_FP32, 1, 0};
        snerf_render_mlp mlps[SNERF_RENDER_LEVELS] = {};
        mlps[SNERF_LEVEL_MAIN_COARSE] = {&desc, packed};
        mlps[SNERF_LEVEL_MAIN_FINE] = {&desc, packed};
        snerf_render_rays rr = {};
        rr.rays_o = rays_o; rr.rays_d = rays_d; rr.view_dirs = dirs; rr.rays_o_ndc = o_ndc; rr.rays_d_ndc = d_ndc;
        rr.near = near; rr.far = far;
        snerf_render_outputs ro = {};
        ro.depths_coarse = dev<float>(n * Sc); ro.depths_fine = dev<float>(n * Sm);
        for (int l : {0, 3}) {
            const int s = l == 0 ? Sc : Sm;
            snerf_render_level_out& lo = ro.level[l];
            lo.rgb = dev<float>(3 * n); lo.acc = dev<float>(n); lo.depth = dev<float>(n); lo.depth_var = dev<float>(n);
            lo.depth_ndc = dev<float>(n); lo.depth_var_ndc = dev<float>(n); lo.alpha = dev<float>(n * s);
            lo.weights = l == 0 ? nullptr : dev<float>(n * s);     // level 0: weights go through the workspace
            lo.sigma = dev<float>(n * s); lo.raw_rgb = dev<float>(3 * n * s);
            lo.saved_acts = dev<float>(snerf_mlp_saved_floats(&desc, n, s));
        }
        float* rwork = dev<float>(snerf_render_workspace_floats(&cfg, n));
        SNERF_OK_(snerf_render_forward(&cfg, mlps, &rr, n, &ro, rwork, nullptr));
        // stage by stage
        float *zc = dev<float>(n * Sc), *zf = dev<float>(n * Sm), *s0 = dev<float>(n * Sc), *c0s = dev<float>(3 * n * Sc),
              *w0 = dev<float>(n * Sc), *s1 = dev<float>(n * Sm), *c1s = dev<float>(3 * n * Sm), *rgb1 = dev<float>(3 * n),
              *acc1 = dev<float>(n), *dd = dev<float>(n), *dv = dev<float>(n), *dn = dev<float>(n), *dvn = dev<float>(n);
        SNERF_OK_(snerf_coarse_depths(near, far, n, Sc, 0, nullptr, zc, nullptr));
        SNERF_OK_(snerf_mlp_forward(&desc, packed, o_ndc, d_ndc, dirs, zc, n, Sc, nullptr, s0, c0s, 0, nullptr));
        SNERF_OK_(snerf_composite(s0, c0s, zc, d_ndc, rays_o, rays_d, n, Sc, 1, 0, rgb1, acc1, nullptr, nullptr, w0, dd, dv, dn, dvn, nullptr));
        SNERF_OK_(snerf_resample_depths(zc, w0, n, Sc, Sf, nullptr, zf, nullptr));
        SNERF_OK_(snerf_mlp_forward(&desc, packed, o_ndc, d_ndc, dirs, zf, n, Sm, nullptr, s1, c1s, 0, nullptr));
        SNERF_OK_(snerf_composite(s1, c1s, zf, d_ndc, rays_o, rays_d, n, Sm, 1, 0, rgb1, acc1, nullptr, nullptr, nullptr, dd, dv, dn, dvn, nullptr));
        HIP_OK(hipDeviceSynchronize());
        auto za = host(ro.depths_fine, n * Sm), zb = host(zf, n * Sm);
        auto ra = host(ro.level[3].rgb, 3 * n), rb = host(rgb1, 3 * n);
        auto da = host(ro.level[3].depth, n), db = host(dd, n);
        for (size_t i = 0; i < za.size(); ++i) CHECK(za[i] == zb[i]);
        for (size_t i = 0; i < ra.size(); ++i) CHECK(ra[i] == rb[i]);
        for (size_t i = 0; i < da.size(); ++i) CHECK(da[i] == db[i]);
        for (long long i = 0; i < n; ++i)
            for (int k = 1; k < Sm; ++k) CHECK(za[i * Sm + k] >= za[i * Sm + k - 1]);
        // backward: d(rgb_fine) = 1, d(depth_coarse) = 0.1; once overwriting, once more accumulating -> exactly twice
        std::vector<float*> gc(np), gf(np);
        for (int i = 0; i < np; ++i) { gc[i] = dev<float>((size_t)shape[i][0] * shape[i][1]); gf[i] = dev<float>((size_t)shape[i][0] * shape[i][1]); }
        float *g_rgb = dev<float>(3 * n), *g_depth = dev<float>(n);
        std::vector<float> tenth(n, 0.1f);
        HIP_OK(hipMemcpy(g_rgb, one.data(), 3 * n * 4, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(g_depth, tenth.data(), n * 4, hipMemcpyHostToDevice));
        snerf_render_level_grads rg[SNERF_RENDER_LEVELS] = {};
        rg[0].depth = g_depth; rg[0].param_grads = gc.data(); rg[0].num_params = np;
        rg[3].rgb = g_rgb; rg[3].param_grads = gf.data(); rg[3].num_params = np;
        float* bwork = dev<float>(snerf_render_backward_workspace_floats(&cfg, mlps, n));
        SNERF_OK_(snerf_render_backward(&cfg, mlps, &rr, n, &ro, rg, bwork, nullptr));
        HIP_OK(hipDeviceSynchronize());
        std::vector<std::vector<float>> once;
        for (int i = 0; i < np; ++i) once.push_back(host(gf[i], (size_t)shape[i][0] * shape[i][1]));
        rg[0].accumulate = rg[3].accumulate = 1;
        SNERF_OK_(snerf_render_backward(&cfg, mlps, &rr, n, &ro, rg, bwork, nullptr));
        HIP_OK(hipDeviceSynchronize());
        double norm = 0.0;
        for (int i = 0; i < np; ++i) {
            auto twice = host(gf[i], (size_t)shape[i][0] * shape[i][1]);
            for (size_t j = 0; j < twice.size(); ++j) { CHECK(twice[j] == 2.f * once[i][j]); norm += (double)once[i][j] * once[i][j]; }
        }
        CHECK(norm > 0.0);
        auto gcoarse = host(gc[0], (size_t)shape[0][0] * shape[0][1]);
        double cn = 0.0;
        for (float v : gcoarse) { CHECK(std::isfinite(v)); cn += (double)v * v; }
        CHECK(cn > 0.0);
        // ---- the fused render kernel (snerf_render_config::fused): an eval-mode 64 + 128 render as ONE launch against the same
        // call taking the stage-by-stage path -- every output bit-identical
        {
            const int Fc = 64, Ff = 128, Fm = Fc + Ff;
            snerf_render_outputs fo[2] = {};
            for (int k = 0; k < 2; ++k) {
                fo[k].depths_coarse = dev<float>(n * Fc); fo[k].depths_fine = dev<float>(n * Fm);
                for (int l : {0, 3}) {
                    const int s = l == 0 ? Fc : Fm;
                    snerf_render_level_out& lo = fo[k].level[l];
                    lo.rgb = dev<float>(3 * n); lo.acc = dev<float>(n); lo.depth = dev<float>(n); lo.depth_var = dev<float>(n);
                    lo.depth_ndc = dev<float>(n); lo.depth_var_ndc = dev<float>(n); lo.alpha = dev<float>(n * s);
                    lo.weights = dev<float>(n * s); lo.sigma = dev<float>(n * s); lo.raw_rgb = dev<float>(3 * n * s);
                }
                snerf_render_config fcfg = {1, 0, 0, Fc, Ff, SNERF_PRECISION_FP32, 0, k};
                float* fwork = dev<float>(snerf_render_workspace_floats(&fcfg, n));
                SNERF_OK_(snerf_render_forward(&fcfg, mlps, &rr, n, &fo[k], fwork, nullptr));
            }
            HIP_OK(hipDeviceSynchronize());
            auto same = [&](const float* a, const float* b, size_t count) {
                auto ha = host(a, count), hb = host(b, count);
                for (size_t i = 0; i < count; ++i) CHECK(ha[i] == hb[i]);
            };
            same(fo[0].depths_coarse, fo[1].depths_coarse, n * Fc);
            same(fo[0].depths_fine, fo[1].depths_fine, n * Fm);
            for (int l : {0, 3}) {
                const size_t s = l == 0 ? Fc : Fm;
                same(fo[0].level[l].rgb, fo[1].level[l].rgb, 3 * n); same(fo[0].level[l].acc, fo[1].level[l].acc, n);
                same(fo[0].level[l].depth, fo[1].level[l].depth, n); same(fo[0].level[l].depth_var_ndc, fo[1].level[l].depth_var_ndc, n);
                same(fo[0].level[l].alpha, fo[1].level[l].alpha, n * s); same(fo[0].level[l].weights, fo[1].level[l].weights, n * s);
                same(fo[0].level[l].sigma, fo[1].level[l].sigma, n * s); same(fo[0].level[l].raw_rgb, fo[1].level[l].raw_rgb, 3 * n * s);
            }
            auto acc = host(fo[1].level[3].acc, n);
            double mean = 0.0;
            for (float v : acc) mean += v;
            CHECK(mean / n > 1e-3);      // not an empty render
        }
        rr.view_dirs = nullptr;
        CHECK(snerf_render_forward(&cfg, mlps, &rr, n, &ro, rwork, nullptr) == SNERF_E_INVALID);
    }

    // errors are status codes with a message, not crashes
    CHECK(snerf_coarse_depths(nullptr, nullptr, 4, 8, 0, nullptr, nullptr, nullptr) == SNERF_E_INVALID);
    CHECK(snerf_last_error()[0] != 0);
    std::printf("c_abi_smoke: OK\n");
    return 0;
}
