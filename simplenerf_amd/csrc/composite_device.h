// K4 (+K5) device code: one ray composited by one wavefront (see composite.hip for the design notes) -- shared by
// composite.hip's kernels and the fused render kernel (render_fused.hip), so that both run the same arithmetic in the same
// order.
#pragma once
#include "resample_device.h"
#include "snerf_common.h"
#include "wave.h"

namespace {

struct CompositeArgs {
    const float* sigma; const float* rgb; const float* z; const float* march_dirs; const float* rays_o; const float* rays_d;
    float* out_rgb; float* out_acc; float* out_alpha; float* out_vis; float* out_weights;
    float* out_depth; float* out_depth_var; float* out_depth_ndc; float* out_depth_var_ndc;
    long long num_rays; int s; int ndc; int white;
    // RESAMPLE: the fused K4 + K5 kernel of the main coarse level (round 3) -- the wave that composited a ray keeps its weights
    // in LDS and goes straight on to the inverse-CDF resampling + merge of that ray (resample_device.h), instead of a second
    // launch re-reading depths and weights from HBM: one kernel boundary (~5 us) less per render call
    int s_f; const float* u; float* z_fine;
};

// One ray by one wavefront.  `zr` / `sr` / `cr`: the ray's S depths, densities and colours (global rows in composite_kernel,
// the LDS-resident sample tile in the fused render kernel); every output is indexed by the global `ray`.  RESAMPLE: `mine` =
// this wave's LDS scratch (S + resample_scratch_floats), `z_fine` = where the ray's merged depths go (global row or LDS tile).
template <int C, bool RESAMPLE>
__device__ __forceinline__ void composite_ray(const CompositeArgs& a, long long ray, const float* zr, const float* sr,
                                              const float* cr, float* mine, float* z_fine, int lane) {
    const int s = a.s;
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
    // depth of the sample after this lane's block: next lane's first, or the far cap after the last sample
    const float far_cap = a.ndc ? 1.0f : 1e10f;
    z[C] = __shfl_down(z[0], 1, 64);
    if (j0 + C >= s) z[C] = far_cap;
#pragma unroll
    for (int c = 0; c < C; ++c)
        if (j0 + c == s - 1) z[c + 1] = far_cap;

    const float* md = a.march_dirs + ray * 3;
    const float norm = sqrtf((md[0] * md[0] + md[1] * md[1]) + md[2] * md[2]);

    float alpha[C], keep = 1.0f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const bool in = j0 + c < s;
        const float delta = (z[c + 1] - z[c]) * norm;
        alpha[c] = in ? 1.0f - expf(-sg[c] * delta) : 0.0f;
        keep *= in ? (1.0f - alpha[c]) + 1e-10f : 1.0f;
    }
    const float incl = snerf::wave_inclusive_mul(keep);
    float trans = __shfl_up(incl, 1, 64);  // exclusive: product over all earlier lanes
    if (lane == 0) trans = 1.0f;

    float w[C], acc = 0.0f, r = 0.0f, g = 0.0f, b = 0.0f, dz = 0.0f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const bool in = j0 + c < s;
        w[c] = alpha[c] * trans;
        if (in) {
            const int j = j0 + c;
            if (a.out_alpha) a.out_alpha[ray * s + j] = alpha[c];
            if (a.out_vis) a.out_vis[ray * s + j] = trans;
            if (a.out_weights) a.out_weights[ray * s + j] = w[c];
        }
        trans *= (1.0f - alpha[c]) + 1e-10f;
        acc += w[c];
        r += w[c] * col[c][0];
        g += w[c] * col[c][1];
        b += w[c] * col[c][2];
        dz += w[c] * z[c];
    }
    acc = snerf::wave_sum(acc);
    r = snerf::wave_sum(r);
    g = snerf::wave_sum(g);
    b = snerf::wave_sum(b);
    dz = snerf::wave_sum(dz);
    const float depth_march = __fdiv_rn(dz, acc + 1e-6f);
    float var = 0.0f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float d = z[c] - depth_march;
        var += w[c] * (d * d);
    }
    var = snerf::wave_sum(var);

    float depth = depth_march, depth_var = var;
    if (a.ndc) {
        // world depths of the NDC samples (:495-501); the reference hard-codes near = 1 here
        const float oz = a.rays_o[ray * 3 + 2], dzw = a.rays_d[ray * 3 + 2];
        const float tn = __fdiv_rn(-(1.0f + oz), dzw);
        const float scale = __fdiv_rn(oz + tn * dzw, dzw);
        float zw[C], dw = 0.0f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float cst = (z[c] == 1.0f) ? 1e-3f : 0.0f;
            zw[c] = scale * (__fdiv_rn(1.0f, (1.0f - z[c]) + cst) - 1.0f) + tn;
            dw += w[c] * zw[c];
        }
        dw = snerf::wave_sum(dw);
        depth = __fdiv_rn(dw, acc + 1e-6f);
        float vw = 0.0f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float d = zw[c] - depth;
            vw += w[c] * (d * d);
        }
        depth_var = snerf::wave_sum(vw);
    }
    if (lane == 0) {
        if (a.white) {
            const float bg = 1.0f - acc;
            r += bg; g += bg; b += bg;
        }
        a.out_rgb[ray * 3 + 0] = r;
        a.out_rgb[ray * 3 + 1] = g;
        a.out_rgb[ray * 3 + 2] = b;
        a.out_acc[ray] = acc;
        a.out_depth[ray] = depth;
        a.out_depth_var[ray] = depth_var;
        if (a.ndc) {
            a.out_depth_ndc[ray] = depth_march;
            a.out_depth_var_ndc[ray] = var;
        }
    }
    if constexpr (RESAMPLE) {
        // this ray's sample tile stays on chip: weights -> LDS, then sample_pdf + merge by the same wave (per-wave LDS:
        // weights[s] | merged[s + s_f] | cdf[s - 1] | bins[s - 1]); the coarse depths are re-read from the row just loaded
#pragma unroll
        for (int c = 0; c < C; ++c)
            if (j0 + c < s) mine[j0 + c] = w[c];
        snerf::wave_lds_sync();
        snerf::resample_wave(zr, mine, s, a.s_f, a.u ? a.u + ray * a.s_f : nullptr, z_fine, mine + s, lane);
    }
}

}  // namespace
