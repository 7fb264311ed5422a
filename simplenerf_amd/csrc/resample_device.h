// K5 device code: inverse-CDF resampling of one ray's coarse weights + sorted merge, executed by ONE wavefront.  Shared by
// resample_kernel (depths.hip) and the fused compositing + resampling kernel (composite.hip), so that both run the same
// arithmetic in the same order.  Reference: SimpleNeRF.get_z_vals_fine (src/models/SimpleNeRF01.py:304-315), sample_pdf
// (:329-361).
#pragma once
#include "wave.h"

namespace snerf {

// torch.linspace(0, 1, steps)[i] in fp32, bit-exact (checked for steps 2..300, 512, 1000, 1024).
__device__ __forceinline__ float unit_linspace(int i, int steps) {
    if (steps == 1) return 0.0f;
    const float step = __fdiv_rn(1.0f, (float)(steps - 1));
    return (i < steps / 2) ? fmaf(step, (float)i, 0.0f) : fmaf(-step, (float)(steps - 1 - i), 1.0f);
}

// floats of per-wave LDS scratch resample_wave needs
__host__ __device__ inline int resample_scratch_floats(int s_c, int s_f) { return (s_c + s_f) + 2 * (s_c - 1); }

// `zc`, `wc`: this ray's coarse depths and weights (s_c each; global or LDS); `u`: the ray's s_f uniform draws or NULL for the
// deterministic linspace; `out`: the ray's s_c + s_f merged depths (global); `scratch`: resample_scratch_floats of LDS owned
// by this wave.  Scan: lane-blocked sequential prefix + 64-lane shuffle scan of the lane totals.  Search: binary search of
// the LDS-resident CDF.  Merge: every element of [coarse | samples] is scattered to its rank.
__device__ __forceinline__ void resample_wave(const float* __restrict__ zc, const float* __restrict__ wc, int s_c, int s_f,
                                              const float* __restrict__ u, float* __restrict__ out, float* scratch, int lane) {
    const int total = s_c + s_f;
    const int nb = s_c - 1;  // bins / cdf entries
    const int m = s_c - 2;   // pdf entries
    float* merged = scratch;
    float* cdf = merged + total;
    float* bins = cdf + nb;

    for (int j = lane; j < s_c; j += 64) merged[j] = zc[j];
    snerf::wave_lds_sync();
    for (int j = lane; j < nb; j += 64) bins[j] = 0.5f * (merged[j + 1] + merged[j]);

    // pdf = (w[1:-1] + 1e-5) / sum ; cdf = [0, cumsum(pdf)]
    const int per = (m + 63) / 64;  // consecutive pdf entries per lane
    const int begin = lane * per;
    float local = 0.0f;
    for (int k = 0; k < per; ++k) {
        const int j = begin + k;
        if (j < m) local += wc[j + 1] + 1e-5f;
    }
    const float denom_sum = snerf::wave_sum(local);
    float run = 0.0f;
    for (int k = 0; k < per; ++k) {
        const int j = begin + k;
        if (j < m) run += __fdiv_rn(wc[j + 1] + 1e-5f, denom_sum);
    }
    const float incl = snerf::wave_inclusive_add(run);
    float prefix = incl - run;  // exclusive prefix of this lane's block
    if (lane == 0) cdf[0] = 0.0f;
    for (int k = 0; k < per; ++k) {
        const int j = begin + k;
        if (j < m) {
            prefix += __fdiv_rn(wc[j + 1] + 1e-5f, denom_sum);
            cdf[j + 1] = prefix;
        }
    }
    snerf::wave_lds_sync();

    // inverse CDF (:345-359)
    for (int k = lane; k < s_f; k += 64) {
        const float uk = u ? u[k] : unit_linspace(k, s_f);
        int lo = 0, hi = nb;  // first index with cdf[idx] > u  == searchsorted(right=True)
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cdf[mid] <= uk) lo = mid + 1; else hi = mid;
        }
        const int below = lo - 1 > 0 ? lo - 1 : 0;
        const int above = lo < nb - 1 ? lo : nb - 1;
        const float cb = cdf[below], ca = cdf[above];
        float den = ca - cb;
        if (den < 1e-5f) den = 1.0f;
        const float t = __fdiv_rn(uk - cb, den);
        const float bb = bins[below], ba = bins[above];
        merged[s_c + k] = bb + t * (ba - bb);
    }
    snerf::wave_lds_sync();

    // sort(cat(coarse, samples)) (:314) as a merge by rank: the coarse depths are ascending (linspace, or jittered
    // inside disjoint strata), so the rank of a sample among them is a binary search; the samples are ascending too
    // when u is the deterministic linspace (inverse CDF is monotone), otherwise their mutual order is counted.
    // Ties: coarse before samples, equal samples by index -- a valid total order, and equal values are interchangeable.
    const float* smp = merged + s_c;
    const bool sorted_samples = (u == nullptr);
    for (int j = lane; j < s_c; j += 64) {          // coarse j: j + #{samples < z_j}
        const float v = merged[j];
        int below;
        if (sorted_samples) {
            int lo = 0, hi = s_f;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (smp[mid] < v) lo = mid + 1; else hi = mid; }
            below = lo;
        } else {
            below = 0;
            for (int q = 0; q < s_f; ++q) below += smp[q] < v;
        }
        out[j + below] = v;
    }
    for (int k = lane; k < s_f; k += 64) {          // sample k: #{coarse <= s_k} + rank among the samples
        const float v = smp[k];
        int lo = 0, hi = s_c;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (merged[mid] <= v) lo = mid + 1; else hi = mid; }
        int among = k;
        if (!sorted_samples) {
            among = 0;
            for (int q = 0; q < s_f; ++q) { const float x = smp[q]; among += (x < v) || (x == v && q < k); }
        }
        out[lo + among] = v;
    }
}

}  // namespace snerf
