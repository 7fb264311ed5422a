// K2 coarse depths and K5 hierarchical resampling (inverse-CDF + sorted merge).
//
// Reference arithmetic (src/models/SimpleNeRF01.py): get_z_vals_coarse :272-302, get_z_vals_fine :304-315,
// sample_pdf :329-361.  Built with -ffp-contract=off; explicit fmaf() only where torch itself fuses
// (torch.linspace's CPU kernel evaluates start + step*i / end - step*(n-1-i) with one rounding).
#include "resample_device.h"
#include "snerf_common.h"
#include "wave.h"

namespace {

using snerf::unit_linspace;

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

// One wavefront per ray; 4 rays per 256-thread block.  Per-wave LDS: merged[S_c+S_f] | cdf[S_c-1] | bins[S_c-1]
// (resample_device.h).  HBM: reads 8 B per coarse sample (+4 B per fine sample with `u`), writes 4 B per merged sample.
__global__ void __launch_bounds__(256) resample_kernel(const float* __restrict__ z_coarse, const float* __restrict__ weights,
                                                       long long num_rays, int s_c, int s_f, const float* __restrict__ u_in,
                                                       float* __restrict__ z_fine) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = snerf::lane_id();
    const int wave = threadIdx.x >> 6;
    const long long ray = (long long)blockIdx.x * 4 + wave;
    if (ray >= num_rays) return;  // whole wave exits together; no block-level barrier is used below
    snerf::resample_wave(z_coarse + ray * s_c, weights + ray * s_c, s_c, s_f, u_in ? u_in + ray * s_f : nullptr,
                         z_fine + ray * (s_c + s_f), lds + (size_t)wave * snerf::resample_scratch_floats(s_c, s_f), lane);
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
