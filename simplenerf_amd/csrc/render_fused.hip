// The fused per-ray-tile render kernel: K2 -> K3 (coarse) -> K4 + K5 -> K3 (fine) -> K4 of SimpleNeRF.render_rays
// (src/models/SimpleNeRF01.py:108-270) for an eval-mode render of a plain coarse + fine model as ONE launch, with each ray
// group's sample tile -- depths, densities, colours, compositing weights, resampled depths -- resident in LDS from the coarse
// depths to the fine colour (north_star: "LDS staging of MLP weights and per-ray sample tiles").
//
// A 256-thread workgroup (4 waves) owns a group of RG rays, RG x num_coarse = 128 samples where num_coarse divides 128
// (1 ray of 128 samples, 2 rays of 64), and walks it through the stages:
//   1. coarse depths of the group -> LDS tile (the arithmetic of coarse_depths_kernel, depths.hip);
//   2. the fused PE + MLP forward (mlp_forward_body.h -- the very code of mlp_forward_kernel) over the tile in 128-sample passes,
//      weights of the COARSE MLP streamed L2 -> LDS, sigma / rgb written to the LDS tile;
//   3. wave w composites ray w of the group from the tile and goes straight on to sample_pdf + merge (composite_device.h /
//      resample_device.h -- the code of composite_kernel<C, true>), merged depths -> LDS tile;
//   4. the forward again with the FINE MLP's weights over the RG x (num_coarse + num_fine) merged samples;
//   5. wave w composites ray w's fine samples.
// Per-ray outputs and the per-sample outputs the caller asked for go to HBM; nothing else does: the unfused path round-trips
// 16 B per sample (sigma, rgb) + 4 B per depth + the weights through HBM and pays five kernel boundaries.
//
// Same device functions in the same order => results are BIT-IDENTICAL to snerf_render_forward's six launches
// (tests/test_gpu_fused.py).  What it buys is bounded by the non-MLP remainder of the step (0.6 % in fp32) and it
// costs L2 locality: coarse and fine weight streams (2 x 2.4 MB) are live at once against 4 MB of L2 per XCD, where the
// separate launches stream one MLP at a time.  Measured in DESIGN; the model uses it when configs['model']['hip_fused_render'].
//
// Bound: MFMA fp32 (as K3).  Built for SNERF_PRECISION_FP32, the view-dependent 8 x 256 / 4 x 128 layouts, and the sample
// counts of the shipped configurations (64 + 128 and 128 + 128); anything else returns SNERF_E_UNSUPPORTED and the caller
// takes the six-launch path.
#include "composite_device.h"
#include "mlp_forward_body.h"

namespace {

using snerf::unit_linspace;

struct FusedArgs {
    MlpArgs coarse, fine;          // packed streams + plan offsets of the two MLPs (depths / sigma / rgb are set per group)
    CompositeArgs comp_c, comp_f;  // per-ray / per-sample outputs of the two levels (sigma / rgb / z are set per group)
    const float* near; const float* far; const float* t_rand;
    float* depths_coarse; float* depths_fine;                 // (n, S_c), (n, S_c + S_f) outputs
    float* raw_sigma_c; float* raw_rgb_c; float* raw_sigma_f; float* raw_rgb_f;   // 'raw_sigma' / 'raw_rgb' outputs or NULL
    long long num_rays;
    int s_c, s_f, lindisp, rays_per_group;
};

__device__ __forceinline__ float fused_depth_at(float near, float far, int j, int steps, int lindisp) {   // = depths.hip depth_at
    const float t = unit_linspace(j, steps);
    if (!lindisp) return near * (1.0f - t) + far * t;
    return __fdiv_rn(1.0f, __fdiv_rn(1.0f, near) * (1.0f - t) + __fdiv_rn(1.0f, far) * t);
}

template <int WT, int VT, int CC, int CF>
__global__ void __launch_bounds__(256, 1) render_fused_kernel(FusedArgs f) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int s_c = f.s_c, s_m = f.s_c + f.s_f, rg = f.rays_per_group;
    const long long ray0 = (long long)blockIdx.x * rg;
    const int rays = (int)(f.num_rays - ray0 < rg ? f.num_rays - ray0 : rg);      // rays of this group (>= 1)

    // LDS: [weight slabs + constants of the MLP pass | z tile | sigma tile | rgb tile | per-wave compositing scratch]
    float* tile_z = lds + 2 * SlabStream<WT>::kBufFloats + kMaxConstFloats;
    float* tile_sigma = tile_z + rg * s_m;
    float* tile_rgb = tile_sigma + rg * s_m;
    // per wave: compositing weights [S_c] | resample_wave's scratch | the ray's merged depths [S_c + S_f]
    const int resample_floats = snerf::resample_scratch_floats(s_c, f.s_f);
    const int per_wave = s_c + resample_floats + s_m;
    float* scratch_base = tile_rgb + 3 * rg * s_m;
    float* scratch = scratch_base + (size_t)wave * per_wave;

    // ---- 1. coarse depths (get_z_vals_coarse :272-302) -------------------------------------------------------------------
    for (int i = threadIdx.x; i < rays * s_c; i += 256) {
        const int r = i / s_c, j = i - r * s_c;
        const float n = f.near[ray0 + r], fr = f.far[ray0 + r];
        float z = fused_depth_at(n, fr, j, s_c, f.lindisp);
        if (f.t_rand) {
            const float zp = fused_depth_at(n, fr, j > 0 ? j - 1 : 0, s_c, f.lindisp);
            const float zn = fused_depth_at(n, fr, j < s_c - 1 ? j + 1 : s_c - 1, s_c, f.lindisp);
            const float lower = (j > 0) ? 0.5f * (z + zp) : z;
            const float upper = (j < s_c - 1) ? 0.5f * (zn + z) : z;
            z = lower + (upper - lower) * f.t_rand[(ray0 + r) * s_c + j];
        }
        tile_z[i] = z;
        f.depths_coarse[(ray0 + r) * s_c + j] = z;
    }
    __syncthreads();

    // ---- 2 .. 5: two levels of [MLP passes over the tile | compositing per ray] ---------------------------------------------
#pragma unroll 1
    for (int level = 0; level < 2; ++level) {
        MlpArgs a = level == 0 ? f.coarse : f.fine;
        const int s = level == 0 ? s_c : s_m;
        a.origins += ray0 * 3; a.dirs += ray0 * 3; a.view_dirs += ray0 * 3;
        a.depths = tile_z; a.sigma = tile_sigma; a.rgb = tile_rgb;
        a.samples = s; a.total = (long long)rays * s;
        const int passes = (int)((a.total + 127) / 128);
#pragma unroll 1
        for (int pass = 0; pass < passes; ++pass) {
            mlp_forward_body<WT, VT, true, false, false>(a, pass, lds);
            __syncthreads();      // every wave is done with the slab buffers (and, after the last pass, has written its samples)
        }
        float* raw_sigma = level == 0 ? f.raw_sigma_c : f.raw_sigma_f;
        float* raw_rgb = level == 0 ? f.raw_rgb_c : f.raw_rgb_f;
        if (raw_sigma)
            for (int i = threadIdx.x; i < rays * s; i += 256) raw_sigma[ray0 * s + i] = tile_sigma[i];
        if (raw_rgb)
            for (int i = threadIdx.x; i < 3 * rays * s; i += 256) raw_rgb[ray0 * s * 3 + i] = tile_rgb[i];
        if (wave < rays) {
            const long long ray = ray0 + wave;
            if (level == 0) {
                CompositeArgs c = f.comp_c;
                // (the merged depths land in this wave's scratch first: the tile's coarse depths are still being read)
                float* merged = scratch + s_c + resample_floats;
                composite_ray<CC, true>(c, ray, tile_z + wave * s_c, tile_sigma + wave * s_c, tile_rgb + 3 * wave * s_c, scratch, merged, lane);
                snerf::wave_lds_sync();
                for (int j = lane; j < s_m; j += 64) f.depths_fine[ray * s_m + j] = merged[j];
            } else {
                composite_ray<CF, false>(f.comp_f, ray, tile_z + wave * s_m, tile_sigma + wave * s_m, tile_rgb + 3 * wave * s_m, nullptr, nullptr, lane);
            }
        }
        __syncthreads();
        if (level == 0) {
            // the tile becomes the fine tile: ray r's merged depths at r * (S_c + S_f) (every wave has finished with the coarse one)
            for (int r = 0; r < rays; ++r) {
                const float* merged = scratch_base + (size_t)r * per_wave + s_c + resample_floats;
                for (int j = threadIdx.x; j < s_m; j += 256) tile_z[r * s_m + j] = merged[j];
            }
            __syncthreads();
        }
    }
}

template <int WT, int VT, int CC, int CF>
int launch_fused(const FusedArgs& f, hipStream_t stream) {
    const long long groups = (f.num_rays + f.rays_per_group - 1) / f.rays_per_group;
    if (groups > 0x7fffffffLL) return snerf::fail(SNERF_E_UNSUPPORTED, "render_forward_fused: too many rays in one call");
    const int s_m = f.s_c + f.s_f;
    const size_t per_wave = (size_t)f.s_c + snerf::resample_scratch_floats(f.s_c, f.s_f) + s_m;      // + the merged depths
    const size_t lds_bytes = sizeof(float) * (2 * SlabStream<WT>::kBufFloats + kMaxConstFloats + 5 * (size_t)f.rays_per_group * s_m + 4 * per_wave);
    if (lds_bytes > 160 * 1024) return snerf::fail(SNERF_E_UNSUPPORTED, "render_forward_fused: %zu bytes of LDS", lds_bytes);
    auto kernel = render_fused_kernel<WT, VT, CC, CF>;
    static snerf::DeviceOnce configured;
    const int attr = snerf::raise_dynamic_lds(configured, reinterpret_cast<const void*>(kernel), 160 * 1024, "render_forward_fused");
    if (attr != SNERF_OK) return attr;
    hipLaunchKernelGGL(kernel, dim3((unsigned)groups), dim3(256), lds_bytes, stream, f);
    return snerf::check_launch("render_forward_fused");
}

void fill_mlp(MlpArgs& a, const snerf::MlpPlan& plan, const float* packed, const float* origins, const float* dirs, const float* view_dirs) {
    a = MlpArgs();
    a.packed = packed; a.origins = origins; a.dirs = dirs; a.view_dirs = view_dirs;
    a.noise = nullptr; a.depth = plan.depth; a.width = plan.width;
    a.bias_offset = plan.bias_offset; a.feature_bias = plan.feature_bias(); a.views_bias = plan.views_bias();
    a.const_floats = (int)((plan.dgrad_offset - plan.bias_offset + 3) / 4 * 4);
    a.pts_out_w = plan.pts_out_w(); a.pts_out_b = plan.pts_out_b();
    a.views_out_w = plan.views_out_w(); a.views_out_b = plan.views_out_b();
    a.acts = nullptr; a.visibility = nullptr; a.view_dirs2 = nullptr; a.visibility2 = nullptr; a.num_other = 0;
    a.range_flag = nullptr; a.weight_range = nullptr;
}

void fill_composite(CompositeArgs& c, const snerf_render_level_out& o, const float* march_dirs, const snerf_render_rays* rays,
                    const snerf_render_config* cfg, long long n, int s) {
    c = CompositeArgs();
    c.march_dirs = march_dirs;
    c.rays_o = cfg->ndc ? rays->rays_o : nullptr; c.rays_d = cfg->ndc ? rays->rays_d : nullptr;
    c.out_rgb = o.rgb; c.out_acc = o.acc; c.out_alpha = o.alpha; c.out_vis = o.visibility; c.out_weights = o.weights;
    c.out_depth = o.depth; c.out_depth_var = o.depth_var; c.out_depth_ndc = o.depth_ndc; c.out_depth_var_ndc = o.depth_var_ndc;
    c.num_rays = n; c.s = s; c.ndc = cfg->ndc; c.white = cfg->white_bkgd;
    c.s_f = cfg->num_fine; c.u = rays->u; c.z_fine = nullptr;
}

}  // namespace

namespace snerf {

// -> SNERF_OK (enqueued), SNERF_E_UNSUPPORTED with `*eligible = 0` when the call is outside what the kernel is built for
// (nothing enqueued, no error text: the caller takes the six-launch path), or an error.
int render_forward_fused(const snerf_render_config* cfg, const snerf_render_mlp* mlps, const snerf_render_rays* rays, long long n,
                         const snerf_render_outputs* out, hipStream_t stream, int* eligible) {
    *eligible = 0;
    if (cfg->precision != SNERF_PRECISION_FP32 || cfg->keep_activations || cfg->num_fine < 1) return SNERF_E_UNSUPPORTED;
    for (int l : {1, 2, 4, 5})
        if (mlps[l].desc) return SNERF_E_UNSUPPORTED;
    if (!mlps[0].desc || !mlps[3].desc || rays->depths_fine || rays->num_other > 0) return SNERF_E_UNSUPPORTED;
    for (int l : {0, 3})
        if (rays->sigma_noise[l] || mlps[l].desc->predict_visibility) return SNERF_E_UNSUPPORTED;
    MlpPlan pc, pf;
    int st = build_plan(mlps[0].desc, &pc);
    if (st != SNERF_OK) return st;
    st = build_plan(mlps[3].desc, &pf);
    if (st != SNERF_OK) return st;
    auto main_layout = [](const MlpPlan& p) { return p.view_dependent && !p.sigma_pe && p.views_out_rows == 3; };
    if (!main_layout(pc) || !main_layout(pf) || pc.wt != pf.wt || pc.vt != pf.vt) return SNERF_E_UNSUPPORTED;
    const int key = pc.wt * 10 + pc.vt;
    if (key != 84 && key != 42) return SNERF_E_UNSUPPORTED;
    const int s_c = cfg->num_coarse, s_f = cfg->num_fine;
    const bool counts_64_128 = s_c == 64 && s_f == 128, counts_128_128 = s_c == 128 && s_f == 128;
    if (!counts_64_128 && !counts_128_128) return SNERF_E_UNSUPPORTED;
    if (!out->depths_fine) return SNERF_E_UNSUPPORTED;
    for (int l : {0, 3}) {       // the fused kernel reads the fp32 K-segment slabs of both buffers (snerf_common.h)
        st = packed_formats_require(mlps[l].packed, kPackFp32, "render_forward (fused)");
        if (st != SNERF_OK) { *eligible = 1; return st; }
    }
    FusedArgs f;
    const float* origins = cfg->ndc ? rays->rays_o_ndc : rays->rays_o;
    const float* dirs = cfg->ndc ? rays->rays_d_ndc : rays->rays_d;
    fill_mlp(f.coarse, pc, mlps[0].packed, origins, dirs, rays->view_dirs);
    fill_mlp(f.fine, pf, mlps[3].packed, origins, dirs, rays->view_dirs);
    if (f.coarse.const_floats > kMaxConstFloats || f.fine.const_floats > kMaxConstFloats) return SNERF_E_UNSUPPORTED;
    fill_composite(f.comp_c, out->level[0], dirs, rays, cfg, n, s_c);
    fill_composite(f.comp_f, out->level[3], dirs, rays, cfg, n, s_c + s_f);
    f.near = rays->near; f.far = rays->far; f.t_rand = rays->t_rand;
    f.depths_coarse = out->depths_coarse; f.depths_fine = out->depths_fine;
    f.raw_sigma_c = out->level[0].sigma; f.raw_rgb_c = out->level[0].raw_rgb;
    f.raw_sigma_f = out->level[3].sigma; f.raw_rgb_f = out->level[3].raw_rgb;
    f.num_rays = n; f.s_c = s_c; f.s_f = s_f; f.lindisp = cfg->lindisp;
    f.rays_per_group = 128 / s_c;
    *eligible = 1;
    ProfileScope timed(SNERF_PROFILE_MLP_FORWARD, stream, n * (long long)(2 * s_c + s_f));
    if (key == 84) return counts_64_128 ? launch_fused<8, 4, 1, 3>(f, stream) : launch_fused<8, 4, 2, 4>(f, stream);
    return counts_64_128 ? launch_fused<4, 2, 1, 3>(f, stream) : launch_fused<4, 2, 2, 4>(f, stream);
}

}  // namespace snerf
