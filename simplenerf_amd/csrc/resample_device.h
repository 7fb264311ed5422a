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

// torch.sum(x + 1e-5, -1) of one contiguous float row of `n` entries, in the order ATen's CPU kernel adds them
// (aten/src/ATen/native/cpu/SumKernel.cpp: cascade_sum -> vectorized_inner_sum -> row_sum -> multi_row_sum), so that the
// normaliser of sample_pdf (src/models/SimpleNeRF01.py:333) comes out bit-identical to the reference's on the same weights.
// That kernel is built for 8-float vectors on every x86 host (sum_stub is not registered for AVX-512, so AVX-512 hosts run
// the AVX2 build; checked against torch 2.10 for n = 5 ... 5000 under both capabilities, tools/check_torch_sum_order.py):
//   n >= 8: vector lane l (0..7) adds x[8 i + l] over the n/8 whole vectors i -- four interleaved accumulators
//           (i mod 4) over the first 4*(n/32) vectors with a 16-step cascade, the remaining vectors onto accumulator 0, then
//           acc0 += acc1, acc2, acc3 -- the scalar tail x[8 (n/8) ...] is summed sequentially from 0, and the eight lane
//           sums are added to it in lane order;
//   n < 8:  the same scheme with one-element "vectors" (scalar_inner_sum).
// Executed redundantly by the whole wave (lanes >= 8 mirror lane l & 7); returns the sum in every lane.
__device__ __forceinline__ float torch_row_sum(const float* __restrict__ w, int n, int lane) {
    const int width = n >= 8 ? 8 : 1;
    const int l = width == 8 ? (lane & 7) : 0;
    const int vectors = n / width;
    const int groups = vectors / 4;                  // multi_row_sum's `size` (four rows = four interleaved accumulators)
    int level_power = 4;
    {   // max(4, ceil_log2(groups) / 4)
        int lg = 1;
        if (groups > 2) lg = 32 - __clz(groups - 1);
        if (lg / 4 > level_power) level_power = lg / 4;
    }
    const int level_step = 1 << level_power;
    const int level_mask = level_step - 1;
    float acc[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[j][k] = 0.0f;
    auto at = [&](int vec) { return w[vec * width + l] + 1e-5f; };
    int i = 0;
    for (; i + level_step <= groups;) {
        for (int j = 0; j < level_step; ++j, ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[0][k] += at(i * 4 + k);
#pragma unroll
        for (int j = 1; j < 4; ++j) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { acc[j][k] += acc[j - 1][k]; acc[j - 1][k] = 0.0f; }
            if ((i & (level_mask << (j * level_power))) != 0) break;
        }
    }
    for (; i < groups; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[0][k] += at(i * 4 + k);
#pragma unroll
    for (int j = 1; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[0][k] += acc[j][k];
    float lane_sum = acc[0][0];
    for (int v = groups * 4; v < vectors; ++v) lane_sum += at(v);
    lane_sum += acc[0][1];
    lane_sum += acc[0][2];
    lane_sum += acc[0][3];
    if (width == 1) return lane_sum;
    float total = 0.0f;
    for (int k = vectors * 8; k < n; ++k) total += w[k] + 1e-5f;
#pragma unroll
    for (int k = 0; k < 8; ++k) total += __shfl(lane_sum, k, 64);
    return total;
}

// floats of per-wave LDS scratch resample_wave needs
__host__ __device__ inline int resample_scratch_floats(int s_c, int s_f) { return (s_c + s_f) + 2 * (s_c - 1); }

// `zc`, `wc`: this ray's coarse depths and weights (s_c each; global or LDS); `u`: the ray's s_f uniform draws or NULL for the
// deterministic linspace; `out`: the ray's s_c + s_f merged depths (global); `scratch`: resample_scratch_floats of LDS owned
// by this wave.  CDF: the reference's own summation orders (see below; until round 3 a 64-lane fp32 shuffle scan, which
// moved 0.02-0.18 % of the samples on identical inputs).  Search: binary search of the LDS-resident CDF.  Merge: every element of [coarse | samples] is scattered to its rank.
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

    // pdf = (w[1:-1] + 1e-5) / sum ; cdf = [0, cumsum(pdf)]  (:332-335) in the reference's own summation orders, so that
    // identical inputs give identical bits: torch.sum's vectorised order for the normaliser (torch_row_sum) and
    // torch.cumsum's for the running sum -- the CPU kernel accumulates a float row SEQUENTIALLY IN DOUBLE and rounds
    // every entry to float (aten/src/ATen/native/cpu/ReduceOpsKernel.cpp, acc_type<float, false> = double).  Every
    // lane runs the same serial chain (62-126 dependent fp64 adds per ray), fed by readlane from the lane that holds the
    // entry; lane j % 64 keeps entry j.
    const float denom_sum = torch_row_sum(wc + 1, m, lane);
    if (lane == 0) cdf[0] = 0.0f;
    double run = 0.0;
    for (int base = 0; base < m; base += 64) {          // 64 pdf entries at a time, one per lane, held in a register
        const int j = base + lane;
        const float mine = j < m ? __fdiv_rn(wc[j + 1] + 1e-5f, denom_sum) : 0.0f;
        const int count = m - base < 64 ? m - base : 64;
        float entry = 0.0f;
        for (int k = 0; k < count; ++k) {               // the serial fp64 chain: readlane + add per entry, no memory traffic
            run += (double)__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine), k));
            if (lane == k) entry = (float)run;
        }
        if (j < m) cdf[j + 1] = entry;
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
