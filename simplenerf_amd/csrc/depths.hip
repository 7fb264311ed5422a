// K2 coarse depths and K5 hierarchical resampling (inverse-CDF + sorted merge).
//
// Reference arithmetic (src/models/SimpleNeRF01.py): get_z_vals_coarse :272-302, get_z_vals_fine :304-315,
// sample_pdf :329-361.  Built with -ffp-contract=off; explicit fmaf() only where torch itself fuses
// (torch.linspace's CPU kernel evaluates start + step*i / end - step*(n-1-i) with one rounding).
#include "snerf_common.h"
#include "wave.h"

namespace {

// torch.linspace(0, 1, steps)[i] in fp32, bit-exact (checked for steps 2..300, 512, 1000, 1024).
__device__ __forceinline__ float unit_linspace(int i, int steps) {
    if (steps == 1) return 0.0f;
    const float step = __fdiv_rn(1.0f, (float)(steps - 1));
    return (i < steps / 2) ? fmaf(step, (float)i, 0.0f) : fmaf(-step, (float)(steps - 1 - i), 1.0f);
}

__device__ __forceinline__ float depth_at(float near, float far, int j, int steps, int lindisp) {
    const float t = unit_linspace(j, steps);
    if (!lindisp) return near * (1.0f - t) + far * t;
    return __fdiv_rn(1.0f, __fdiv_rn(1.0f, near) * (1.0f - t) + __fdiv_rn(1.0f, far) * t);
}

// One thread per (ray, sample).  HBM: reads 8 B/ray (+4 B/sample with jitter), writes 4 B/sample.
__global__ void __launch_bounds__(256) coarse_depths_kernel(const float* __restrict__ near, const float* __restrict__ far,
                                                            long long total, int steps, int lindisp,
                                                            const float* __restrict__ t_rand, float* __restrict__ out) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const long long ray = i / steps;
        const int j = (int)(i - ray * steps);
        const float n = near[ray], f = far[ray];
        float z = depth_at(n, f, j, steps, lindisp);
        if (t_rand) {  // stratified jitter inside [lower, upper) (:293-301)
            const float zp = depth_at(n, f, j > 0 ? j - 1 : 0, steps, lindisp);
            const float zn = depth_at(n, f, j < steps - 1 ? j + 1 : steps - 1, steps, lindisp);
            const float lower = (j > 0) ? 0.5f * (z + zp) : z;
            const float upper = (j < steps - 1) ? 0.5f * (zn + z) : z;
            z = lower + (upper - lower) * t_rand[i];
        }
        out[i] = z;
    }
}

// One wavefront per ray; 4 rays per 256-thread block.  Per-wave LDS: merged[S_c+S_f] | cdf[S_c-1] | bins[S_c-1].
// Scan: lane-blocked sequential prefix + 64-lane shuffle scan of the lane totals.  Search: binary search of the
// LDS-resident CDF.  Merge: every element of [coarse | samples] is scattered to its rank (binary searches; a counting
// pass over the samples only when `u` is random and they are unsorted).
// HBM: reads 8 B per coarse sample (+4 B per fine sample with `u`), writes 4 B per merged sample.
__global__ void __launch_bounds__(256) resample_kernel(const float* __restrict__ z_coarse, const float* __restrict__ weights,
                                                       long long num_rays, int s_c, int s_f, const float* __restrict__ u_in,
                                                       float* __restrict__ z_fine) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = snerf::lane_id();
    const int wave = threadIdx.x >> 6;
    const int total = s_c + s_f;
    const int nb = s_c - 1;  // bins / cdf entries
    const int m = s_c - 2;   // pdf entries
    float* merged = lds + (size_t)wave * (total + 2 * nb);
    float* cdf = merged + total;
    float* bins = cdf + nb;
    const long long ray = (long long)blockIdx.x * 4 + wave;
    if (ray >= num_rays) return;  // whole wave exits together; no block-level barrier is used below
    const float* zc = z_coarse + ray * s_c;
    const float* wc = weights + ray * s_c;

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
        const float u = u_in ? u_in[ray * s_f + k] : unit_linspace(k, s_f);
        int lo = 0, hi = nb;  // first index with cdf[idx] > u  == searchsorted(right=True)
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
        }
        const int below = lo - 1 > 0 ? lo - 1 : 0;
        const int above = lo < nb - 1 ? lo : nb - 1;
        const float cb = cdf[below], ca = cdf[above];
        float den = ca - cb;
        if (den < 1e-5f) den = 1.0f;
        const float t = __fdiv_rn(u - cb, den);
        const float bb = bins[below], ba = bins[above];
        merged[s_c + k] = bb + t * (ba - bb);
    }
    snerf::wave_lds_sync();

    // sort(cat(coarse, samples)) (:314) as a merge by rank: the coarse depths are ascending (linspace, or jittered
    // inside disjoint strata), so the rank of a sample among them is a binary search; the samples are ascending too
    // when u is the deterministic linspace (inverse CDF is monotone), otherwise their mutual order is counted.
    // Ties: coarse before samples, equal samples by index -- a valid total order, and equal values are interchangeable.
    float* out = z_fine + ray * total;
    const float* smp = merged + s_c;
    const bool sorted_samples = (u_in == nullptr);
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

}  // namespace

extern "C" int snerf_coarse_depths(const float* near, const float* far, long long num_rays, int num_samples,
                                   int lindisp, const float* t_rand, float* depths, snerf_stream_t stream) {
    SNERF_REQUIRE(near && far && depths, "coarse_depths: NULL pointer");
    SNERF_REQUIRE(num_rays >= 0 && num_samples >= 1, "coarse_depths: bad sizes n=%lld S=%d", num_rays, num_samples);
    if (num_rays == 0) return SNERF_OK;
    const long long total = num_rays * num_samples;
    hipLaunchKernelGGL(coarse_depths_kernel, dim3(snerf::stride_grid(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       near, far, total, num_samples, lindisp, t_rand, depths);
    return snerf::check_launch("coarse_depths");
}

extern "C" int snerf_resample_depths(const float* depths_coarse, const float* weights_coarse, long long num_rays,
                                     int num_coarse, int num_fine, const float* u, float* depths_fine,
                                     snerf_stream_t stream) {
    SNERF_REQUIRE(depths_coarse && weights_coarse && depths_fine, "resample_depths: NULL pointer");
    SNERF_REQUIRE(num_rays >= 0, "resample_depths: negative ray count");
    SNERF_REQUIRE(num_coarse >= 3 && num_fine >= 1, "resample_depths: need >= 3 coarse and >= 1 fine samples (got %d, %d)",
                  num_coarse, num_fine);
    const size_t lds_bytes = 4 * sizeof(float) * (size_t)(num_coarse + num_fine + 2 * (num_coarse - 1));
    if (lds_bytes > 64 * 1024)
        return snerf::fail(SNERF_E_UNSUPPORTED, "resample_depths: %d+%d samples exceed the per-block LDS budget",
                           num_coarse, num_fine);
    if (num_rays == 0) return SNERF_OK;
    const long long blocks = (num_rays + 3) / 4;
    if (blocks > 0x7fffffffLL) return snerf::fail(SNERF_E_UNSUPPORTED, "resample_depths: too many rays in one call");
    hipLaunchKernelGGL(resample_kernel, dim3((unsigned)blocks), dim3(256), lds_bytes, (hipStream_t)stream,
                       depths_coarse, weights_coarse, num_rays, num_coarse, num_fine, u, depths_fine);
    return snerf::check_launch("resample_depths");
}
