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
        for (int c = 0; c < 3 && image; ++c) {
            float v = rgb[3 * i + c];
            // numpy.clip, then round-half-to-even, then the uint8 cast.  NaN: numpy keeps it through clip and round and
            // its float->uint8 cast yields 0 on x86-64 (pinned by tests/golden/display.npz); fmaxf(NaN, 0) = 0 gives the
            // same byte here
            v = fminf(fmaxf(v, 0.0f), 1.0f);
            image[3 * i + c] = (unsigned char)rintf(v * 255.0f);
        }
        if (depth_out) {
            // numpy.clip(depth, 0, inf): comparisons, not fmaxf -- NaN stays NaN and -0.0 stays -0.0 as in the reference
            const float d = depth[i];
            depth_out[i] = d < 0.0f ? 0.0f : d;
        }
    }
}

}  // namespace

extern "C" int snerf_to_display(const float* rgb, const float* depth, long long num_rays, unsigned char* image,
                                float* depth_out, snerf_stream_t stream) {
    SNERF_REQUIRE((rgb && image) || (!image && depth_out), "to_display: NULL pointer");
    SNERF_REQUIRE(!depth_out || depth, "to_display: depth_out requested without depth");
    SNERF_REQUIRE(num_rays >= 0, "to_display: negative ray count");
    if (num_rays == 0) return SNERF_OK;
    hipLaunchKernelGGL(to_display_kernel, dim3(snerf::stride_grid(num_rays, 256)), dim3(256), 0, (hipStream_t)stream, rgb,
                       depth, num_rays, image, depth_out);
    return snerf::check_launch("to_display");
}
