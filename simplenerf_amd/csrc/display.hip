// Output post-processing on the device: what the reference's Tester does on the host after copying
// every fp32 output back -- post_process_image / post_process_depth (src/data_preprocessors/DataPreprocessor01.py
// :1106-1114): colour -> clip to [0,1] -> round(255 x) (round-half-to-even, as numpy.round) -> uint8; depth -> clip at 0.
// One pass, 16 B read and 7 B written per ray, so a frame leaves the GPU as 3 B/pixel instead of 12 (+ ~1 KB/ray of
// alpha the reference also ships).  Bound: HBM.
#include "snerf_common.h"

namespace {

__global__ void __launch_bounds__(256) to_display_kernel(const float* __restrict__ rgb, const float* __restrict__ depth,
                                                         long long n, unsigned char* __restrict__ image,
                                                         float* __restrict__ depth_out) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = rgb[3 * i + c];
            v = fminf(fmaxf(v, 0.0f), 1.0f);     // numpy.clip; NaN propagates to 0 through the integer cast below
            image[3 * i + c] = (unsigned char)rintf(v * 255.0f);
        }
        if (depth_out) depth_out[i] = fmaxf(depth[i], 0.0f);
    }
}

}  // namespace

extern "C" int snerf_to_display(const float* rgb, const float* depth, long long num_rays, unsigned char* image,
                                float* depth_out, snerf_stream_t stream) {
    SNERF_REQUIRE(rgb && image, "to_display: NULL pointer");
    SNERF_REQUIRE(!depth_out || depth, "to_display: depth_out requested without depth");
    SNERF_REQUIRE(num_rays >= 0, "to_display: negative ray count");
    if (num_rays == 0) return SNERF_OK;
    hipLaunchKernelGGL(to_display_kernel, dim3(snerf::stride_grid(num_rays, 256)), dim3(256), 0, (hipStream_t)stream, rgb,
                       depth, num_rays, image, depth_out);
    return snerf::check_launch("to_display");
}
