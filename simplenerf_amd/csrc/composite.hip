// K4 -- sigma -> alpha -> transmittance -> weights alpha compositing with per-ray reductions.
//
// Reference arithmetic: SimpleNeRF.volume_rendering (src/models/SimpleNeRF01.py:430-483) and
// convert_depth_from_ndc (:486-502).  One wavefront per ray.  Lane l owns the C = ceil(S/64) consecutive samples
// l*C .. l*C+C-1 (so loads are contiguous per lane and coalesced per wave); the exclusive product
// T_j = prod_{k<j}(1 - alpha_k + 1e-10) is a sequential product inside the lane followed by a 64-lane shuffle scan
// of the lane totals; sums are lane-local then butterfly-reduced.
//
// Bound: HBM.  Algorithmic bytes per sample: 20 read (sigma 4, rgb 12, depth 4) + 12 written when alpha,
// visibility and weights are all requested; per ray 12-36 read, 24-32 written.
#include "composite_device.h"

namespace {

template <int C, bool RESAMPLE = false>
__global__ void __launch_bounds__(256) composite_kernel(CompositeArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = snerf::lane_id();
    const long long ray = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= a.num_rays) return;
    const int s = a.s;
    float* mine = RESAMPLE ? lds + (size_t)(threadIdx.x >> 6) * (s + snerf::resample_scratch_floats(s, a.s_f)) : nullptr;
    composite_ray<C, RESAMPLE>(a, ray, a.z + ray * s, a.sigma + ray * s, a.rgb + ray * s * 3, mine,
                               RESAMPLE ? a.z_fine + ray * (s + a.s_f) : nullptr, lane);
}

template <int C, bool RESAMPLE = false>
void launch(const CompositeArgs& a, hipStream_t stream) {
    const long long blocks = (a.num_rays + 3) / 4;
    const size_t lds_bytes = RESAMPLE ? 4 * sizeof(float) * (size_t)(a.s + snerf::resample_scratch_floats(a.s, a.s_f)) : 0;
    hipLaunchKernelGGL((composite_kernel<C, RESAMPLE>), dim3((unsigned)blocks), dim3(256), lds_bytes, stream, a);
}

template <bool RESAMPLE>
void launch_for_samples(const CompositeArgs& a, hipStream_t st) {
    const int c = (a.s + 63) / 64;
    switch (c) {
        case 1: launch<1, RESAMPLE>(a, st); break;
        case 2: launch<2, RESAMPLE>(a, st); break;
        case 3: launch<3, RESAMPLE>(a, st); break;
        case 4: launch<4, RESAMPLE>(a, st); break;
        case 5: case 6: launch<6, RESAMPLE>(a, st); break;
        case 7: case 8: launch<8, RESAMPLE>(a, st); break;
        default: launch<16, RESAMPLE>(a, st); break;
    }
}


// ------------------------------------------------------------------------------------------------------------------
// K6 -- backward of the compositing: (d rgb, d acc, d depth, d depth_ndc) per ray -> d sigma, d rgb per sample.
// Autograd of volume_rendering (:446-460) in closed form:
//   g_j = dL/dw_j = g_rgb . c_j + g_acc + g_depth (zd_j - depth)/(acc+1e-6) + g_depth_ndc (z_j - depth_ndc)/(acc+1e-6)
//   dL/dalpha_j = g_j T_j - (sum_{k>j} g_k w_k) / (1 - alpha_j + 1e-10)        (reverse wavefront scan)
//   dL/dsigma_j = dL/dalpha_j . delta_j . (1 - alpha_j);      dL/dc_j = w_j g_rgb
// Same wave-per-ray, lane-blocked layout as the forward; the forward quantities are recomputed, not stored.
struct CompositeBwdArgs {
    const float* sigma; const float* rgb; const float* z; const float* march_dirs; const float* rays_o; const float* rays_d;
    const float* g_rgb; const float* g_acc; const float* g_depth; const float* g_depth_ndc;
    float* d_sigma; float* d_rgb;
    long long num_rays; int s; int ndc; int white;
};

template <int C>
__global__ void __launch_bounds__(256) composite_backward_kernel(CompositeBwdArgs a) {
    const int lane = snerf::lane_id();
    const long long ray = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= a.num_rays) return;
    const int s = a.s;
    const float* zr = a.z + ray * s;
    const float* sr = a.sigma + ray * s;
    const float* cr = a.rgb + ray * s * 3;
    const int j0 = lane * C;
    float z[C + 1], sg[C], col[C][3];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int j = j0 + c;
        const bool in = j < s;
        z[c] = in ? zr[j] : 0.0f;
        sg[c] = in ? sr[j] : 0.0f;
#pragma unroll
        for (int k = 0; k < 3; ++k) col[c][k] = in ? cr[3 * j + k] : 0.0f;
    }
    const float far_cap = a.ndc ? 1.0f : 1e10f;
    z[C] = __shfl_down(z[0], 1, 64);
    if (j0 + C >= s) z[C] = far_cap;
#pragma unroll
    for (int c = 0; c < C; ++c)
        if (j0 + c == s - 1) z[c + 1] = far_cap;
    const float* md = a.march_dirs + ray * 3;
    const float norm = sqrtf((md[0] * md[0] + md[1] * md[1]) + md[2] * md[2]);

    float alpha[C], delta[C], keep = 1.0f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const bool in = j0 + c < s;
        delta[c] = (z[c + 1] - z[c]) * norm;
        alpha[c] = in ? 1.0f - expf(-sg[c] * delta[c]) : 0.0f;
        keep *= in ? (1.0f - alpha[c]) + 1e-10f : 1.0f;
    }
    const float incl = snerf::wave_inclusive_mul(keep);
    float trans = __shfl_up(incl, 1, 64);
    if (lane == 0) trans = 1.0f;
    float T[C], w[C], acc = 0.0f, dzm = 0.0f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        T[c] = trans;
        w[c] = alpha[c] * trans;
        trans *= (1.0f - alpha[c]) + 1e-10f;
        acc += w[c];
        dzm += w[c] * z[c];
    }
    acc = snerf::wave_sum(acc);
    dzm = snerf::wave_sum(dzm);
    const float inv = __fdiv_rn(1.0f, acc + 1e-6f);
    const float depth_march = dzm * inv;
    // depth the loss sees as 'depth': world depth for NDC scenes
    float zd[C];
    float depth = depth_march;
    if (a.ndc) {
        const float oz = a.rays_o[ray * 3 + 2], dzw = a.rays_d[ray * 3 + 2];
        const float tn = __fdiv_rn(-(1.0f + oz), dzw);
        const float scale = __fdiv_rn(oz + tn * dzw, dzw);
        float dw = 0.0f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float cst = (z[c] == 1.0f) ? 1e-3f : 0.0f;
            zd[c] = scale * (__fdiv_rn(1.0f, (1.0f - z[c]) + cst) - 1.0f) + tn;
            dw += w[c] * zd[c];
        }
        depth = snerf::wave_sum(dw) * inv;
    } else {
#pragma unroll
        for (int c = 0; c < C; ++c) zd[c] = z[c];
    }
    const float gr = a.g_rgb ? a.g_rgb[ray * 3 + 0] : 0.0f;
    const float gg = a.g_rgb ? a.g_rgb[ray * 3 + 1] : 0.0f;
    const float gb = a.g_rgb ? a.g_rgb[ray * 3 + 2] : 0.0f;
    float gacc = a.g_acc ? a.g_acc[ray] : 0.0f;
    if (a.white) gacc -= (gr + gg) + gb;  // rgb += 1 - acc
    const float gdep = (a.g_depth ? a.g_depth[ray] : 0.0f) * inv;
    const float gdn = (a.ndc && a.g_depth_ndc ? a.g_depth_ndc[ray] : 0.0f) * inv;

    float g[C], local = 0.0f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        g[c] = ((gr * col[c][0] + gg * col[c][1]) + gb * col[c][2]) + gacc + gdep * (zd[c] - depth) + gdn * (z[c] - depth_march);
        local += g[c] * w[c];
    }
    // sum over strictly later samples of g_k w_k: later lanes via the suffix scan, later samples of this lane locally
    const float suffix_incl = snerf::wave_suffix_add(local);
    float after = __shfl_down(suffix_incl, 1, 64);  // everything owned by later lanes
    if (lane == 63) after = 0.0f;
#pragma unroll
    for (int c = C - 1; c >= 0; --c) {
        const int j = j0 + c;
        if (j < s) {
            const float u = (1.0f - alpha[c]) + 1e-10f;
            const float dalpha = g[c] * T[c] - __fdiv_rn(after, u);
            a.d_sigma[ray * s + j] = dalpha * (delta[c] * (1.0f - alpha[c]));
            // (the file is compiled without the SLP vectoriser -- build.py FILE_FLAGS: packed into v_pk_mul_f32, these products
            // took the operand selection described at opaque_pair(), mlp_device.h, and once in ~10^5 times sixteen lanes of
            // d rgb came out 0.0 when this kernel ran beside another level's MLP backward)
            a.d_rgb[(ray * s + j) * 3 + 0] = w[c] * gr;
            a.d_rgb[(ray * s + j) * 3 + 1] = w[c] * gg;
            a.d_rgb[(ray * s + j) * 3 + 2] = w[c] * gb;
        }
        after += g[c] * w[c];
    }
}

template <int C>
void launch_bwd(const CompositeBwdArgs& a, hipStream_t stream) {
    const long long blocks = (a.num_rays + 3) / 4;
    hipLaunchKernelGGL(composite_backward_kernel<C>, dim3((unsigned)blocks), dim3(256), 0, stream, a);
}

}  // namespace

extern "C" int snerf_composite(const float* sigma, const float* rgb, const float* depths, const float* march_dirs,
                               const float* rays_o, const float* rays_d, long long num_rays, int num_samples, int ndc,
                               int white_bkgd, float* out_rgb, float* out_acc, float* out_alpha, float* out_visibility,
                               float* out_weights, float* out_depth, float* out_depth_var, float* out_depth_ndc,
                               float* out_depth_var_ndc, snerf_stream_t stream) {
    SNERF_REQUIRE(sigma && rgb && depths && march_dirs, "composite: NULL input");
    SNERF_REQUIRE(out_rgb && out_acc && out_depth && out_depth_var, "composite: NULL required output");
    SNERF_REQUIRE(!ndc || (rays_o && rays_d && out_depth_ndc && out_depth_var_ndc),
                  "composite: ndc needs world rays and the *_ndc outputs");
    SNERF_REQUIRE(num_rays >= 0 && num_samples >= 1, "composite: bad sizes n=%lld S=%d", num_rays, num_samples);
    if (num_samples > 64 * 16)
        return snerf::fail(SNERF_E_UNSUPPORTED, "composite: at most 1024 samples per ray (got %d)", num_samples);
    if (num_rays == 0) return SNERF_OK;
    if ((num_rays + 3) / 4 > 0x7fffffffLL) return snerf::fail(SNERF_E_UNSUPPORTED, "composite: too many rays in one call");
    CompositeArgs a{sigma, rgb, depths, march_dirs, rays_o, rays_d, out_rgb, out_acc, out_alpha, out_visibility,
                    out_weights, out_depth, out_depth_var, out_depth_ndc, out_depth_var_ndc, num_rays, num_samples,
                    ndc, white_bkgd, 0, nullptr, nullptr};
    launch_for_samples<false>(a, (hipStream_t)stream);
    return snerf::check_launch("composite");
}

// K4 + K5 in one launch (render.hip: the main coarse level of a model with a fine pass): snerf_composite's arguments -- the
// weights output may be NULL, they stay in LDS -- plus snerf_resample_depths' (num_fine, u, depths_fine).  Same arithmetic in
// the same order as the two separate kernels (tests/native/c_abi_smoke.cpp compares them bit for bit).  Returns
// SNERF_E_UNSUPPORTED when the sample counts do not fit (the caller then issues the two kernels).
namespace snerf {
int composite_resample(const float* sigma, const float* rgb, const float* depths, const float* march_dirs, const float* rays_o,
                       const float* rays_d, long long num_rays, int num_samples, int ndc, int white_bkgd, float* out_rgb,
                       float* out_acc, float* out_alpha, float* out_visibility, float* out_weights, float* out_depth,
                       float* out_depth_var, float* out_depth_ndc, float* out_depth_var_ndc, int num_fine, const float* u,
                       float* depths_fine, hipStream_t stream) {
    SNERF_REQUIRE(sigma && rgb && depths && march_dirs && depths_fine, "composite_resample: NULL input");
    SNERF_REQUIRE(out_rgb && out_acc && out_depth && out_depth_var, "composite_resample: NULL required output");
    SNERF_REQUIRE(!ndc || (rays_o && rays_d && out_depth_ndc && out_depth_var_ndc),
                  "composite_resample: ndc needs world rays and the *_ndc outputs");
    if (num_samples < 3 || num_fine < 1 || num_samples > 64 * 16 || (num_rays + 3) / 4 > 0x7fffffffLL ||
        4 * sizeof(float) * (size_t)(num_samples + resample_scratch_floats(num_samples, num_fine)) > 64 * 1024)
        return fail(SNERF_E_UNSUPPORTED, "composite_resample: %d + %d samples not built as one kernel", num_samples, num_fine);
    if (num_rays == 0) return SNERF_OK;
    CompositeArgs a{sigma, rgb, depths, march_dirs, rays_o, rays_d, out_rgb, out_acc, out_alpha, out_visibility,
                    out_weights, out_depth, out_depth_var, out_depth_ndc, out_depth_var_ndc, num_rays, num_samples,
                    ndc, white_bkgd, num_fine, u, depths_fine};
    launch_for_samples<true>(a, stream);
    return check_launch("composite_resample");
}
}  // namespace snerf

extern "C" int snerf_composite_backward(const float* sigma, const float* rgb, const float* depths, const float* march_dirs,
                                        const float* rays_o, const float* rays_d, long long num_rays, int num_samples,
                                        int ndc, int white_bkgd, const float* grad_rgb, const float* grad_acc,
                                        const float* grad_depth, const float* grad_depth_ndc, float* d_sigma, float* d_rgb,
                                        snerf_stream_t stream) {
    SNERF_REQUIRE(sigma && rgb && depths && march_dirs && d_sigma && d_rgb, "composite_backward: NULL pointer");
    SNERF_REQUIRE(!ndc || (rays_o && rays_d), "composite_backward: ndc needs the world rays");
    SNERF_REQUIRE(num_rays >= 0 && num_samples >= 1, "composite_backward: bad sizes n=%lld S=%d", num_rays, num_samples);
    if (num_samples > 64 * 16)
        return snerf::fail(SNERF_E_UNSUPPORTED, "composite_backward: at most 1024 samples per ray (got %d)", num_samples);
    if (num_rays == 0) return SNERF_OK;
    if ((num_rays + 3) / 4 > 0x7fffffffLL) return snerf::fail(SNERF_E_UNSUPPORTED, "composite_backward: too many rays");
    CompositeBwdArgs a{sigma, rgb, depths, march_dirs, rays_o, rays_d, grad_rgb, grad_acc, grad_depth, grad_depth_ndc,
                       d_sigma, d_rgb, num_rays, num_samples, ndc, white_bkgd};
    hipStream_t st = (hipStream_t)stream;
    const int c = (num_samples + 63) / 64;
    switch (c) {
        case 1: launch_bwd<1>(a, st); break;
        case 2: launch_bwd<2>(a, st); break;
        case 3: launch_bwd<3>(a, st); break;
        case 4: launch_bwd<4>(a, st); break;
        case 5: case 6: launch_bwd<6>(a, st); break;
        case 7: case 8: launch_bwd<8>(a, st); break;
        default: launch_bwd<16>(a, st); break;
    }
    return snerf::check_launch("composite_backward");
}
