// The one-call render ops of the C ABI: SimpleNeRF.render_rays (src/models/SimpleNeRF01.py:108-270) and what autograd
// replays for it, enqueued stage by stage on the caller's stream from C++ -- one boundary crossing per forward / backward
// instead of one per stage.  Orchestration only: every stage is one of the library's own entry points (K2 depths.hip,
// K3 mlp_forward*.hip, K4/K6 composite.hip, K5 depths.hip, K7 mlp_backward*.hip).
#include <algorithm>

#include "snerf_common.h"

namespace snerf {   // render_fused.hip: the whole eval-mode render of a plain coarse + fine model as one launch
int render_forward_fused(const snerf_render_config* cfg, const snerf_render_mlp* mlps, const snerf_render_rays* rays, long long n,
                         const snerf_render_outputs* out, hipStream_t stream, int* eligible);
}

namespace snerf {   // composite.hip: K4 + K5 of the main coarse level as one launch
int composite_resample(const float* sigma, const float* rgb, const float* depths, const float* march_dirs, const float* rays_o,
                       const float* rays_d, long long num_rays, int num_samples, int ndc, int white_bkgd, float* out_rgb,
                       float* out_acc, float* out_alpha, float* out_visibility, float* out_weights, float* out_depth,
                       float* out_depth_var, float* out_depth_ndc, float* out_depth_var_ndc, int num_fine, const float* u,
                       float* depths_fine, hipStream_t stream);
}

namespace {

struct Marching {
    const float *origins, *dirs;
};

int check_common(const snerf_render_config* cfg, const snerf_render_mlp* mlps, const snerf_render_rays* rays, long long n,
                 const snerf_render_outputs* out, const char* who) {
    SNERF_REQUIRE(cfg && mlps && rays && out, "%s: NULL argument", who);
    SNERF_REQUIRE(n >= 0, "%s: negative ray count", who);
    SNERF_REQUIRE(cfg->num_coarse >= 1 && cfg->num_fine >= 0, "%s: bad sample counts %d + %d", who, cfg->num_coarse, cfg->num_fine);
    SNERF_REQUIRE(mlps[SNERF_LEVEL_MAIN_COARSE].desc, "%s: the main coarse MLP is required", who);
    SNERF_REQUIRE(rays->rays_o && rays->rays_d && rays->near && rays->far, "%s: rays_o / rays_d / near / far are required", who);
    SNERF_REQUIRE(!cfg->ndc || (rays->rays_o_ndc && rays->rays_d_ndc), "%s: NDC rays are required when ndc", who);
    SNERF_REQUIRE(out->depths_coarse, "%s: depths_coarse is required", who);
    const bool fine = mlps[3].desc || mlps[4].desc || mlps[5].desc;
    SNERF_REQUIRE(!fine || cfg->num_fine > 0, "%s: a fine-level MLP needs num_fine > 0", who);
    SNERF_REQUIRE(!fine || mlps[SNERF_LEVEL_MAIN_FINE].desc, "%s: fine-level augmentation MLPs need the main fine MLP", who);
    SNERF_REQUIRE(!fine || out->depths_fine || rays->depths_fine, "%s: depths_fine is required with a fine pass", who);
    for (int l = 0; l < SNERF_RENDER_LEVELS; ++l) {
        if (!mlps[l].desc) continue;
        const snerf_render_level_out& o = out->level[l];
        SNERF_REQUIRE(mlps[l].packed, "%s: level %d has no packed weights", who, l);
        SNERF_REQUIRE(o.rgb && o.acc && o.depth && o.depth_var, "%s: level %d: rgb / acc / depth / depth_var are required", who, l);
        SNERF_REQUIRE((o.sigma && o.raw_rgb) || (cfg->fused && !cfg->keep_activations),
                      "%s: level %d: sigma / raw_rgb are required (optional only for a fused eval-mode render)", who, l);
        SNERF_REQUIRE(!cfg->ndc || (o.depth_ndc && o.depth_var_ndc), "%s: level %d: NDC depth outputs are required", who, l);
        SNERF_REQUIRE(!cfg->keep_activations || o.saved_acts, "%s: level %d: saved_acts is required with keep_activations", who, l);
        SNERF_REQUIRE(!mlps[l].desc->use_view_dirs || rays->view_dirs, "%s: level %d uses view directions but view_dirs is NULL", who, l);
        if (mlps[l].desc->predict_visibility && rays->num_other > 0) {
            SNERF_REQUIRE(rays->rays_o2, "%s: rays_o2 is required with num_other > 0", who);
            SNERF_REQUIRE(o.raw_visibility2 && o.visibility2 && o.view_dirs2,
                          "%s: level %d: raw_visibility2 / visibility2 / view_dirs2 are required with secondary views", who, l);
        }
    }
    SNERF_REQUIRE(rays->num_other >= 0, "%s: negative num_other", who);
    return SNERF_OK;
}

__global__ void __launch_bounds__(256) add_kernel(float* __restrict__ dst, const float* __restrict__ src, long long count) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) dst[i] += src[i];
}

}  // namespace

extern "C" size_t snerf_render_workspace_floats(const snerf_render_config* cfg, long long num_rays) {
    if (!cfg || num_rays < 0 || cfg->num_coarse < 1) return 0;
    return (size_t)num_rays * (size_t)cfg->num_coarse;
}

extern "C" int snerf_render_forward(const snerf_render_config* cfg, const snerf_render_mlp* mlps, const snerf_render_rays* rays,
                                    long long num_rays, const snerf_render_outputs* out, float* workspace,
                                    snerf_stream_t stream) {
    int rc = check_common(cfg, mlps, rays, num_rays, out, "render_forward");
    if (rc != SNERF_OK) return rc;
    if (num_rays == 0) return SNERF_OK;
    const long long n = num_rays;
    if (cfg->fused) {
        int eligible = 0;
        rc = snerf::render_forward_fused(cfg, mlps, rays, n, out, (hipStream_t)stream, &eligible);
        if (eligible) return rc;
        for (int l = 0; l < SNERF_RENDER_LEVELS; ++l)       // not a call the fused kernel is built for: the stage-by-stage path
            SNERF_REQUIRE(!mlps[l].desc || (out->level[l].sigma && out->level[l].raw_rgb),
                          "render_forward: level %d: sigma / raw_rgb are required (this call is outside the fused kernel's scope)", l);
    }
    const Marching m = cfg->ndc ? Marching{rays->rays_o_ndc, rays->rays_d_ndc} : Marching{rays->rays_o, rays->rays_d};
    const bool fine = mlps[SNERF_LEVEL_MAIN_FINE].desc != nullptr;
    float* coarse_weights = out->level[0].weights;
    // the main coarse level of a model with a fine pass: compositing and resampling in ONE kernel, the ray's weights handed
    // over in LDS (no predict_visibility on that level: its visibility2 compositing reads the weights from memory)
#ifdef SNERF_PROBE_NO_K4K5_FUSION
    const bool fuse_resample = false;
#else
    const bool fuse_resample = fine && !rays->depths_fine && !mlps[0].desc->predict_visibility && cfg->num_coarse >= 3 &&
                               sizeof(float) * 4 * (size_t)(2 * cfg->num_coarse + cfg->num_fine + 2 * (cfg->num_coarse - 1)) <= 64 * 1024;
#endif
    if (fine && !rays->depths_fine && !coarse_weights && !fuse_resample) {
        SNERF_REQUIRE(workspace, "render_forward: workspace is required (main coarse weights feed the resampling)");
        coarse_weights = workspace;
    }

    auto shade = [&](int l, const float* depths, int samples) -> int {
        const snerf_render_level_out& o = out->level[l];
        const float* dirs = mlps[l].desc->use_view_dirs ? rays->view_dirs : nullptr;
        const bool vis = mlps[l].desc->predict_visibility != 0;
        const bool vis2 = vis && rays->num_other > 0;
        float* level_weights = l == 0 && coarse_weights ? coarse_weights : o.weights;
        if (vis && (o.raw_visibility || vis2)) {
            // predict_visibility: secondary view directions per sample, the visibility-aware forward, compositing, and the
            // per-ray visibility of each secondary view (:317-326, :646-649, :479-482)
            int st = SNERF_OK;
            if (vis2) {
                SNERF_REQUIRE(level_weights, "render_forward: level %d needs its weights output to composite visibility2", l);
                st = snerf_other_view_dirs(depths, rays->rays_o, rays->rays_d, rays->rays_o2, n, samples, rays->num_other,
                                           cfg->ndc, o.view_dirs2, stream);
                if (st != SNERF_OK) return st;
            }
            st = snerf_mlp_forward_visibility(mlps[l].desc, mlps[l].packed, m.origins, m.dirs, dirs, depths, n, samples,
                                              rays->sigma_noise[l], vis2 ? o.view_dirs2 : nullptr, vis2 ? rays->num_other : 0,
                                              o.sigma, o.raw_rgb, o.raw_visibility, vis2 ? o.raw_visibility2 : nullptr,
                                              cfg->keep_activations ? o.saved_acts : nullptr, cfg->precision, stream);
            if (st != SNERF_OK) return st;
            st = snerf_composite(o.sigma, o.raw_rgb, depths, m.dirs, cfg->ndc ? rays->rays_o : nullptr,
                                 cfg->ndc ? rays->rays_d : nullptr, n, samples, cfg->ndc, cfg->white_bkgd, o.rgb, o.acc, o.alpha,
                                 o.visibility, level_weights, o.depth, o.depth_var, o.depth_ndc, o.depth_var_ndc, stream);
            if (st != SNERF_OK || !vis2) return st;
            return snerf_composite_visibility2(level_weights, o.acc, o.raw_visibility2, n, samples, rays->num_other, o.visibility2,
                                               stream);
        }
        int st = cfg->keep_activations
                     ? snerf_mlp_forward_train(mlps[l].desc, mlps[l].packed, m.origins, m.dirs, dirs, depths, n, samples,
                                               rays->sigma_noise[l], o.sigma, o.raw_rgb, o.saved_acts, cfg->precision, stream)
                     : snerf_mlp_forward(mlps[l].desc, mlps[l].packed, m.origins, m.dirs, dirs, depths, n, samples,
                                         rays->sigma_noise[l], o.sigma, o.raw_rgb, cfg->precision, stream);
        if (st != SNERF_OK) return st;
        if (l == 0 && fuse_resample)
            return snerf::composite_resample(o.sigma, o.raw_rgb, depths, m.dirs, cfg->ndc ? rays->rays_o : nullptr,
                                             cfg->ndc ? rays->rays_d : nullptr, n, samples, cfg->ndc, cfg->white_bkgd, o.rgb,
                                             o.acc, o.alpha, o.visibility, level_weights, o.depth, o.depth_var, o.depth_ndc,
                                             o.depth_var_ndc, cfg->num_fine, rays->u, out->depths_fine, (hipStream_t)stream);
        return snerf_composite(o.sigma, o.raw_rgb, depths, m.dirs, cfg->ndc ? rays->rays_o : nullptr,
                               cfg->ndc ? rays->rays_d : nullptr, n, samples, cfg->ndc, cfg->white_bkgd, o.rgb, o.acc, o.alpha,
                               o.visibility, level_weights, o.depth, o.depth_var, o.depth_ndc, o.depth_var_ndc, stream);
    };

    rc = snerf_coarse_depths(rays->near, rays->far, n, cfg->num_coarse, cfg->lindisp, rays->t_rand, out->depths_coarse, stream);
    if (rc != SNERF_OK) return rc;
    for (int l = 0; l < 3; ++l) {
        if (!mlps[l].desc) continue;
        rc = shade(l, out->depths_coarse, cfg->num_coarse);
        if (rc != SNERF_OK) return rc;
    }
    if (!fine) return SNERF_OK;
    const float* depths_fine = rays->depths_fine;
    if (!depths_fine) {
        if (!fuse_resample) {     // (else the main coarse level's kernel has written them already)
            rc = snerf_resample_depths(out->depths_coarse, coarse_weights, n, cfg->num_coarse, cfg->num_fine, rays->u,
                                       out->depths_fine, stream);
            if (rc != SNERF_OK) return rc;
        }
        depths_fine = out->depths_fine;
    }
    for (int l = 3; l < SNERF_RENDER_LEVELS; ++l) {
        if (!mlps[l].desc) continue;
        rc = shade(l, depths_fine, cfg->num_coarse + cfg->num_fine);
        if (rc != SNERF_OK) return rc;
    }
    return SNERF_OK;
}

extern "C" size_t snerf_render_backward_workspace_floats(const snerf_render_config* cfg, const snerf_render_mlp* mlps,
                                                         long long num_rays) {
    if (!cfg || !mlps || num_rays < 0 || cfg->num_coarse < 1 || cfg->num_fine < 0) return 0;
    size_t samples = 0, inner = 0;
    for (int l = 0; l < SNERF_RENDER_LEVELS; ++l) {
        if (!mlps[l].desc) continue;
        const int s = l < 3 ? cfg->num_coarse : cfg->num_coarse + cfg->num_fine;
        samples = std::max(samples, (size_t)num_rays * (size_t)s);
        inner = std::max(inner, snerf_mlp_backward_workspace_floats(mlps[l].desc, num_rays, s));
    }
    return 4 * samples + inner;
}

extern "C" int snerf_render_backward(const snerf_render_config* cfg, const snerf_render_mlp* mlps, const snerf_render_rays* rays,
                                     long long num_rays, const snerf_render_outputs* out, const snerf_render_level_grads* grads,
                                     float* workspace, snerf_stream_t stream) {
    int rc = check_common(cfg, mlps, rays, num_rays, out, "render_backward");
    if (rc != SNERF_OK) return rc;
    SNERF_REQUIRE(grads && workspace, "render_backward: NULL argument");
    SNERF_REQUIRE(cfg->keep_activations, "render_backward: the forward must have kept the activations");
    if (num_rays == 0) return SNERF_OK;
    const long long n = num_rays;
    const float* march_d = cfg->ndc ? rays->rays_d_ndc : rays->rays_d;
    size_t samples_max = 0;
    for (int l = 0; l < SNERF_RENDER_LEVELS; ++l)
        if (mlps[l].desc) samples_max = std::max(samples_max, (size_t)n * (size_t)(l < 3 ? cfg->num_coarse : cfg->num_coarse + cfg->num_fine));
    float* d_sigma = workspace;
    float* d_rgb = workspace + samples_max;
    float* inner = workspace + 4 * samples_max;
    for (int l = 0; l < SNERF_RENDER_LEVELS; ++l) {
        if (!mlps[l].desc) continue;
        const snerf_render_level_grads& g = grads[l];
        if (!g.param_grads || !(g.rgb || g.acc || g.depth || g.depth_ndc || g.sigma || g.raw_rgb)) continue;
        const int s = l < 3 ? cfg->num_coarse : cfg->num_coarse + cfg->num_fine;
        const float* depths = l < 3 ? out->depths_coarse : (rays->depths_fine ? rays->depths_fine : out->depths_fine);
        const snerf_render_level_out& o = out->level[l];
        rc = snerf_composite_backward(o.sigma, o.raw_rgb, depths, march_d, cfg->ndc ? rays->rays_o : nullptr,
                                      cfg->ndc ? rays->rays_d : nullptr, n, s, cfg->ndc, cfg->white_bkgd, g.rgb, g.acc, g.depth,
                                      g.depth_ndc, d_sigma, d_rgb, stream);
        if (rc != SNERF_OK) return rc;
        if (g.sigma) {
            hipLaunchKernelGGL(add_kernel, dim3(snerf::stride_grid(n * s, 256)), dim3(256), 0, (hipStream_t)stream, d_sigma, g.sigma, n * s);
            rc = snerf::check_launch("render_backward(add sigma)");
            if (rc != SNERF_OK) return rc;
        }
        if (g.raw_rgb) {
            hipLaunchKernelGGL(add_kernel, dim3(snerf::stride_grid(3 * n * s, 256)), dim3(256), 0, (hipStream_t)stream, d_rgb, g.raw_rgb, 3 * n * s);
            rc = snerf::check_launch("render_backward(add rgb)");
            if (rc != SNERF_OK) return rc;
        }
        rc = snerf_mlp_backward(mlps[l].desc, mlps[l].packed, o.saved_acts, o.sigma, o.raw_rgb, d_sigma, d_rgb, n, s, inner,
                                g.param_grads, g.num_params, cfg->precision, g.accumulate, stream);
        if (rc != SNERF_OK) return rc;
    }
    return SNERF_OK;
}
