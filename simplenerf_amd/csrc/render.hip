// The one-call render ops of the C ABI: SimpleNeRF.render_rays (src/models/SimpleNeRF01.py:108-270) and what autograd
// replays for it, enqueued stage by stage on the caller's stream from C++ -- one boundary crossing per forward / backward
// instead of one per stage.  Orchestration only: every stage is one of the library's own entry points (K2 depths.hip,
// K3 mlp_forward*.hip, K4/K6 composite.hip, K5 depths.hip, K7 mlp_backward*.hip).
#include <algorithm>
#include <map>
#include <mutex>
#include <utility>

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

// ---- levels side by side (round 5) -----------------------------------------------------------------------------------------
// The MLP levels of one render are independent of each other except main coarse -> (resampling) -> fine: with few rays per call
// -- one rank's share of a batch that is strong-scaled over 8 GPUs: 256 rays = 16 384 coarse samples = 64 workgroups for 256 CUs
// -- each level's kernels leave most of the chip idle and a call costs the SUM of their latencies (a chain or storing-forward
// launch takes ~70 us whether it covers 16 k or 65 k samples).  Below kSideBySideSamples samples per coarse level the
// augmentation levels run on side streams beside the main levels (forward: {main coarse -> fine} | points-aug | views-aug;
// backward: all four side by side), forked from and joined to the caller's stream with events -- still enqueue-only, and a
// capture of the caller's stream captures the side streams with it (fork / join is what a HIP graph records as parallel
// branches).  Side streams and events are created once per (device, caller's stream) and kept.  Above the threshold every CU is
// busy with one level and the levels stay on the caller's stream, in order (two backward calls side by side measured the same
// as back to back there: DESIGN_LOG 12.4).
#ifndef SNERF_SIDE_BY_SIDE_SAMPLES          // (A/B builds: -DSNERF_SIDE_BY_SIDE_SAMPLES=n)
#define SNERF_SIDE_BY_SIDE_SAMPLES 65536
#endif
constexpr long long kSideBySideSamples = SNERF_SIDE_BY_SIDE_SAMPLES;
constexpr int kSideStreams = 3;

struct SideStreams {
    hipStream_t stream[kSideStreams] = {};
    hipEvent_t fork = nullptr, join[kSideStreams] = {};
    bool ok = false;
};
std::mutex g_side_mutex;
std::map<std::pair<int, hipStream_t>, SideStreams> g_side;

SideStreams* side_streams(hipStream_t main) {
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(g_side_mutex);
    // (streams and events are kept for the life of the process -- a captured graph may hold them; a caller that keeps creating
    // streams gets side streams for the first kMaxSideSets of them and plain in-order levels afterwards)
    constexpr size_t kMaxSideSets = 64;
    auto found = g_side.find({device, main});
    if (found == g_side.end() && g_side.size() >= kMaxSideSets) return nullptr;
    SideStreams& s = g_side[{device, main}];
    if (!s.ok) {
        bool good = hipEventCreateWithFlags(&s.fork, hipEventDisableTiming) == hipSuccess;
        for (int i = 0; i < kSideStreams && good; ++i)
            good = hipStreamCreateWithFlags(&s.stream[i], hipStreamNonBlocking) == hipSuccess &&
                   hipEventCreateWithFlags(&s.join[i], hipEventDisableTiming) == hipSuccess;
        if (!good) {                 // (the caller falls back to one stream; nothing half-made stays behind)
            (void)hipGetLastError();
            if (s.fork) (void)hipEventDestroy(s.fork);
            for (int i = 0; i < kSideStreams; ++i) {
                if (s.join[i]) (void)hipEventDestroy(s.join[i]);
                if (s.stream[i]) (void)hipStreamDestroy(s.stream[i]);
            }
            g_side.erase({device, main});
            return nullptr;
        }
        s.ok = true;
    }
    return &s;
}

// fork: the side streams wait for what the caller's stream holds so far; join: the caller's stream waits for them
int fork_streams(SideStreams* s, hipStream_t main, int count) {
    if (hipEventRecord(s->fork, main) != hipSuccess) return snerf::fail(SNERF_E_HIP, "render: hipEventRecord(fork) failed");
    for (int i = 0; i < count; ++i)
        if (hipStreamWaitEvent(s->stream[i], s->fork, 0) != hipSuccess) return snerf::fail(SNERF_E_HIP, "render: hipStreamWaitEvent(fork) failed");
    return SNERF_OK;
}
int join_streams(SideStreams* s, hipStream_t main, int count) {
    for (int i = 0; i < count; ++i) {
        if (hipEventRecord(s->join[i], s->stream[i]) != hipSuccess) return snerf::fail(SNERF_E_HIP, "render: hipEventRecord(join) failed");
        if (hipStreamWaitEvent(main, s->join[i], 0) != hipSuccess) return snerf::fail(SNERF_E_HIP, "render: hipStreamWaitEvent(join) failed");
    }
    return SNERF_OK;
}

bool side_by_side(const snerf_render_config* cfg, const snerf_render_mlp* mlps, long long num_rays) {
#ifdef SNERF_PROBE_NO_SIDE_BY_SIDE
    return false;      // A/B builds (tools/probes/share_ab.py)
#endif
    if (num_rays * cfg->num_coarse > kSideBySideSamples) return false;
    int levels = 0;
    for (int l = 0; l < SNERF_RENDER_LEVELS; ++l) levels += mlps[l].desc ? 1 : 0;
    return levels > 2 || (levels == 2 && !mlps[SNERF_LEVEL_MAIN_FINE].desc);
}

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

    auto shade = [&](int l, const float* depths, int samples, snerf_stream_t stream) -> int {
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
    // the coarse augmentation levels beside {main coarse -> fine} when the call is small (see side_by_side)
    SideStreams* side = side_by_side(cfg, mlps, n) && (mlps[1].desc || mlps[2].desc) ? side_streams((hipStream_t)stream) : nullptr;
    int forked = 0;
    // (from the fork on every return joins the side streams first)
    auto done = [&](int status) { return side ? (join_streams(side, (hipStream_t)stream, forked) == SNERF_OK ? status : (status != SNERF_OK ? status : SNERF_E_HIP)) : status; };
    if (side) {
        forked = (mlps[1].desc ? 1 : 0) + (mlps[2].desc ? 1 : 0);
        rc = fork_streams(side, (hipStream_t)stream, forked);
        if (rc != SNERF_OK) return rc;
    }
    // the main coarse level FIRST: the fine level waits for it, so its workgroups must not queue behind the augmentation levels'
    // (enqueued last it ran last: 140 us instead of 86 and the fine pass started that much later, r05 trace of a 512-row pass)
    rc = shade(0, out->depths_coarse, cfg->num_coarse, stream);
    if (rc != SNERF_OK) return done(rc);
    int used = 0;
    for (int l = 1; l < 3; ++l) {
        if (!mlps[l].desc) continue;
        rc = shade(l, out->depths_coarse, cfg->num_coarse, side ? (snerf_stream_t)side->stream[used++] : stream);
        if (rc != SNERF_OK) return done(rc);
    }
    if (!fine) return done(SNERF_OK);
    const float* depths_fine = rays->depths_fine;
    if (!depths_fine) {
        if (!fuse_resample) {     // (else the main coarse level's kernel has written them already)
            rc = snerf_resample_depths(out->depths_coarse, coarse_weights, n, cfg->num_coarse, cfg->num_fine, rays->u,
                                       out->depths_fine, stream);
            if (rc != SNERF_OK) return done(rc);
        }
        depths_fine = out->depths_fine;
    }
    for (int l = 3; l < SNERF_RENDER_LEVELS; ++l) {
        if (!mlps[l].desc) continue;
        rc = shade(l, depths_fine, cfg->num_coarse + cfg->num_fine, stream);
        if (rc != SNERF_OK) return done(rc);
    }
    return done(SNERF_OK);
}

extern "C" size_t snerf_render_backward_workspace_floats(const snerf_render_config* cfg, const snerf_render_mlp* mlps,
                                                         long long num_rays) {
    if (!cfg || !mlps || num_rays < 0 || cfg->num_coarse < 1 || cfg->num_fine < 0) return 0;
    size_t samples = 0, inner = 0, every = 0;
    for (int l = 0; l < SNERF_RENDER_LEVELS; ++l) {
        if (!mlps[l].desc) continue;
        const int s = l < 3 ? cfg->num_coarse : cfg->num_coarse + cfg->num_fine;
        const size_t level_inner = snerf_mlp_backward_workspace_floats(mlps[l].desc, num_rays, s);
        samples = std::max(samples, (size_t)num_rays * (size_t)s);
        inner = std::max(inner, level_inner);
        every += 4 * (size_t)num_rays * (size_t)s + (level_inner + 63) / 64 * 64;
    }
    // levels side by side (small calls): each level its own gradient columns and inner workspace
    return side_by_side(cfg, mlps, num_rays) ? every : 4 * samples + inner;
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
    // one level's backward on `s`: compositing backward (K6) -> raw-output gradients added -> MLP backward (K7)
    auto level_backward = [&](int l, float* d_sigma, float* d_rgb, float* inner, snerf_stream_t s_l) -> int {
        const snerf_render_level_grads& g = grads[l];
        const int s = l < 3 ? cfg->num_coarse : cfg->num_coarse + cfg->num_fine;
        const float* depths = l < 3 ? out->depths_coarse : (rays->depths_fine ? rays->depths_fine : out->depths_fine);
        const snerf_render_level_out& o = out->level[l];
        int st = snerf_composite_backward(o.sigma, o.raw_rgb, depths, march_d, cfg->ndc ? rays->rays_o : nullptr,
                                          cfg->ndc ? rays->rays_d : nullptr, n, s, cfg->ndc, cfg->white_bkgd, g.rgb, g.acc, g.depth,
                                          g.depth_ndc, d_sigma, d_rgb, s_l);
        if (st != SNERF_OK) return st;
        if (g.sigma) {
            hipLaunchKernelGGL(add_kernel, dim3(snerf::stride_grid(n * s, 256)), dim3(256), 0, (hipStream_t)s_l, d_sigma, g.sigma, n * s);
            st = snerf::check_launch("render_backward(add sigma)");
            if (st != SNERF_OK) return st;
        }
        if (g.raw_rgb) {
            hipLaunchKernelGGL(add_kernel, dim3(snerf::stride_grid(3 * n * s, 256)), dim3(256), 0, (hipStream_t)s_l, d_rgb, g.raw_rgb, 3 * n * s);
            st = snerf::check_launch("render_backward(add rgb)");
            if (st != SNERF_OK) return st;
        }
        return snerf_mlp_backward(mlps[l].desc, mlps[l].packed, o.saved_acts, o.sigma, o.raw_rgb, d_sigma, d_rgb, n, s, inner,
                                  g.param_grads, g.num_params, cfg->precision, g.accumulate, s_l);
    };
    auto wanted = [&](int l) {
        const snerf_render_level_grads& g = grads[l];
        return mlps[l].desc && g.param_grads && (g.rgb || g.acc || g.depth || g.depth_ndc || g.sigma || g.raw_rgb);
    };
    int levels = 0;
    for (int l = 0; l < SNERF_RENDER_LEVELS; ++l) levels += wanted(l) ? 1 : 0;
    SideStreams* side = side_by_side(cfg, mlps, n) && levels > 1 ? side_streams((hipStream_t)stream) : nullptr;
    if (!side) {
        float* d_sigma = workspace;
        float* d_rgb = workspace + samples_max;
        float* inner = workspace + 4 * samples_max;
        for (int l = 0; l < SNERF_RENDER_LEVELS; ++l) {
            if (!wanted(l)) continue;
            rc = level_backward(l, d_sigma, d_rgb, inner, stream);
            if (rc != SNERF_OK) return rc;
        }
        return SNERF_OK;
    }
    // side by side: the heaviest level (main fine, else the first wanted) stays on the caller's stream, the others take the side
    // streams in turn; each level has its own region of the workspace (snerf_render_backward_workspace_floats)
    const int forked = std::min(levels - 1, kSideStreams);
    rc = fork_streams(side, (hipStream_t)stream, forked);
    if (rc != SNERF_OK) return rc;
    int keep = wanted(SNERF_LEVEL_MAIN_FINE) ? SNERF_LEVEL_MAIN_FINE : -1;
    for (int l = 0; keep < 0 && l < SNERF_RENDER_LEVELS; ++l)
        if (wanted(l)) keep = l;
    // each level's region of the workspace, in level order
    float* d_sigma_of[SNERF_RENDER_LEVELS] = {};
    float* region = workspace;
    size_t samples_of[SNERF_RENDER_LEVELS] = {};
    for (int l = 0; l < SNERF_RENDER_LEVELS; ++l) {
        if (!mlps[l].desc) continue;
        const int s_l = l < 3 ? cfg->num_coarse : cfg->num_coarse + cfg->num_fine;
        samples_of[l] = (size_t)n * (size_t)s_l;
        d_sigma_of[l] = region;
        region += 4 * samples_of[l] + (snerf_mlp_backward_workspace_floats(mlps[l].desc, n, s_l) + 63) / 64 * 64;
    }
    // the kept (heaviest) level is enqueued first -- enqueued last, its kernels queued behind everybody else's and the call ended
    // with it -- then the others, each on its side stream
    int used = 0;
    rc = level_backward(keep, d_sigma_of[keep], d_sigma_of[keep] + samples_of[keep], d_sigma_of[keep] + 4 * samples_of[keep], stream);
    for (int l = 0; l < SNERF_RENDER_LEVELS && rc == SNERF_OK; ++l) {
        if (!wanted(l) || l == keep) continue;
        rc = level_backward(l, d_sigma_of[l], d_sigma_of[l] + samples_of[l], d_sigma_of[l] + 4 * samples_of[l],
                            (snerf_stream_t)side->stream[used++ % forked]);
    }
    const int joined = join_streams(side, (hipStream_t)stream, forked);
    return rc != SNERF_OK ? rc : joined;
}
