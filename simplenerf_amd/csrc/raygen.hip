// K1 -- pinhole ray generation (+ forward-facing NDC warp) for a pixel range of one frame.
//
// Reference arithmetic (src/data_preprocessors/DataPreprocessor01.py): get_rays :351-368, get_view_dirs :392-394,
// get_ndc_rays :371-389.  Every expression below keeps the reference's fp32 operation order (the library is built
// with -ffp-contract=off), which makes the result bit-identical to the numpy path: sample positions later feed
// sin(512 x), so a 1-ulp difference in a ray would surface as ~3e-5 in the top encoding frequency.
//
// Bound: HBM write, 36 B/ray (60 B/ray with NDC); no reads.  One thread per ray, grid-stride.
#include "snerf_common.h"
#include "raygen_device.h"

namespace {

__global__ void __launch_bounds__(256) raygen_kernel(snerf::Camera c, float near, float pixel_offset, int width, int ndc,
                                                     long long first, long long count, float* __restrict__ rays_o,
                                                     float* __restrict__ rays_d, float* __restrict__ view_dirs,
                                                     float* __restrict__ rays_o_ndc, float* __restrict__ rays_d_ndc) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const long long pix = first + i;
        const float x = (float)(pix % width) + pixel_offset;
        const float y = (float)(pix / width) + pixel_offset;
        snerf::pinhole_ray(c, x, y, near, ndc != 0, rays_o + 3 * i, rays_d + 3 * i, view_dirs ? view_dirs + 3 * i : nullptr,
                           ndc ? rays_o_ndc + 3 * i : nullptr, ndc ? rays_d_ndc + 3 * i : nullptr);
    }
}

}  // namespace

extern "C" int snerf_generate_rays(int height, int width, const float* intrinsic, const float* pose,
                                   float pixel_offset, int ndc, float near, long long first_ray, long long num_rays,
                                   float* rays_o, float* rays_d, float* view_dirs, float* rays_o_ndc,
                                   float* rays_d_ndc, snerf_stream_t stream) {
    SNERF_REQUIRE(height > 0 && width > 0, "generate_rays: bad resolution %dx%d", height, width);
    SNERF_REQUIRE(intrinsic && pose, "generate_rays: intrinsic/pose must be host pointers, got NULL");
    SNERF_REQUIRE(first_ray >= 0 && num_rays >= 0 && first_ray + num_rays <= (long long)height * width,
                  "generate_rays: pixel range [%lld, %lld) outside the %dx%d frame", first_ray, first_ray + num_rays,
                  height, width);
    SNERF_REQUIRE(rays_o && rays_d, "generate_rays: rays_o/rays_d output is NULL");
    SNERF_REQUIRE(!ndc || (rays_o_ndc && rays_d_ndc), "generate_rays: ndc requested but NDC outputs are NULL");
    if (num_rays == 0) return SNERF_OK;
    snerf::Camera c;
    if (!snerf::make_camera(intrinsic, pose, height, width, &c)) return snerf::fail(SNERF_E_INVALID, "generate_rays: singular intrinsic");
    hipLaunchKernelGGL(raygen_kernel, dim3(snerf::stride_grid(num_rays, 256)), dim3(256), 0, (hipStream_t)stream, c,
                       near, pixel_offset, width, ndc, first_ray, num_rays, rays_o, rays_d, view_dirs, rays_o_ndc, rays_d_ndc);
    return snerf::check_launch("generate_rays");
}
