// predict_visibility support around the MLP (off in every shipped configuration):
//   other_view_dirs_kernel        SimpleNeRF.compute_other_view_dirs (src/models/SimpleNeRF01.py:317-326)
//   composite_visibility2_kernel  'visibility2' of volume_rendering (:479-482)
// Both are small streaming kernels (HBM/latency-bound): 12 B written per (sample, view) resp. 4 B read per (sample, view).
#include "snerf_common.h"
#include "wave.h"

namespace {

__global__ void __launch_bounds__(256) other_view_dirs_kernel(const float* __restrict__ depths, const float* __restrict__ rays_o,
                                                              const float* __restrict__ rays_d, const float* __restrict__ rays_o2,
                                                              long long total, int samples, int num_other, int ndc,
                                                              float* __restrict__ out) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const long long ray = i / samples;
        const float ox = rays_o[3 * ray], oy = rays_o[3 * ray + 1], oz = rays_o[3 * ray + 2];
        const float dx = rays_d[3 * ray], dy = rays_d[3 * ray + 1], dz = rays_d[3 * ray + 2];
        float z = depths[i];
        if (ndc) {   // :319-321, near plane hard-coded to 1 as in the reference; operation order as written there
            const float tn = __fdiv_rn(-(1.0f + oz), dz);
            const float num = oz + tn * dz;
            const float den = (1.0f - z) + 1e-6f;
            z = __fdiv_rn(__fdiv_rn(num, den) - oz, dz);
        }
        const float px = ox + z * dx, py = oy + z * dy, pz = oz + z * dz;
        for (int k = 0; k < num_other; ++k) {
            const float* c = rays_o2 + (ray * num_other + k) * 3;
            const float vx = px - c[0], vy = py - c[1], vz = pz - c[2];
            const float norm = sqrtf(vx * vx + vy * vy + vz * vz);
            float* o = out + (i * num_other + k) * 3;
            o[0] = __fdiv_rn(vx, norm); o[1] = __fdiv_rn(vy, norm); o[2] = __fdiv_rn(vz, norm);
        }
    }
}

// one wave per ray: lanes stride over the samples, butterfly sum per secondary view
__global__ void __launch_bounds__(256) composite_visibility2_kernel(const float* __restrict__ weights, const float* __restrict__ acc,
                                                                    const float* __restrict__ vis2, long long num_rays, int samples,
                                                                    int num_other, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long long waves = (long long)gridDim.x * 4;
    for (long long ray = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); ray < num_rays; ray += waves) {
        const float denom = acc[ray] + 1e-6f;
        for (int k = 0; k < num_other; ++k) {
            float s = 0.0f;
            for (int j = lane; j < samples; j += 64) s += weights[ray * samples + j] * vis2[(ray * samples + j) * num_other + k];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
            if (lane == 0) out[ray * num_other + k] = __fdiv_rn(s, denom);
        }
    }
}

}  // namespace

extern "C" int snerf_other_view_dirs(const float* depths, const float* rays_o, const float* rays_d, const float* rays_o2,
                                     long long num_rays, int num_samples, int num_other, int ndc, float* view_dirs2,
                                     snerf_stream_t stream) {
    SNERF_REQUIRE(depths && rays_o && rays_d && rays_o2 && view_dirs2, "other_view_dirs: NULL pointer");
    SNERF_REQUIRE(num_rays >= 0 && num_samples >= 1 && num_other >= 1, "other_view_dirs: bad sizes n=%lld S=%d views=%d", num_rays,
                  num_samples, num_other);
    if (num_rays == 0) return SNERF_OK;
    const long long total = num_rays * num_samples;
    hipLaunchKernelGGL(other_view_dirs_kernel, dim3(snerf::stride_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, depths,
                       rays_o, rays_d, rays_o2, total, num_samples, num_other, ndc, view_dirs2);
    return snerf::check_launch("other_view_dirs");
}

extern "C" int snerf_composite_visibility2(const float* weights, const float* acc, const float* visibility2, long long num_rays,
                                           int num_samples, int num_other, float* out, snerf_stream_t stream) {
    SNERF_REQUIRE(weights && acc && visibility2 && out, "composite_visibility2: NULL pointer");
    SNERF_REQUIRE(num_rays >= 0 && num_samples >= 1 && num_other >= 1, "composite_visibility2: bad sizes n=%lld S=%d views=%d",
                  num_rays, num_samples, num_other);
    if (num_rays == 0) return SNERF_OK;
    hipLaunchKernelGGL(composite_visibility2_kernel, dim3(snerf::stride_grid((num_rays + 3) / 4 * 256, 256)), dim3(256), 0,
                       (hipStream_t)stream, weights, acc, visibility2, num_rays, num_samples, num_other, out);
    return snerf::check_launch("composite_visibility2");
}
