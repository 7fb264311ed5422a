// K1 -- pinhole ray generation (+ forward-facing NDC warp) for a pixel range of one frame.
//
// Reference arithmetic (src/data_preprocessors/DataPreprocessor01.py): get_rays :351-368, get_view_dirs :392-394,
// get_ndc_rays :371-389.  Every expression below keeps the reference's fp32 operation order (the library is built
// with -ffp-contract=off), which makes the result bit-identical to the numpy path: sample positions later feed
// sin(512 x), so a 1-ulp difference in a ray would surface as ~3e-5 in the top encoding frequency.
//
// Bound: HBM write, 36 B/ray (60 B/ray with NDC); no reads.  One thread per ray, grid-stride.
#include "snerf_common.h"

namespace {

struct RaygenConst {
    float kinv[9];   // inverse intrinsic, fp32
    float rot[9];    // pose[:3,:3]
    float org[3];    // pose[:3,3]
    float near;
    float cx, cy;    // -1/(w/(2 fx)), -1/(h/(2 fy)) evaluated in fp32 steps like the reference expression
    float pixel_offset;
    int width;
    int ndc;
};

__global__ void __launch_bounds__(256) raygen_kernel(RaygenConst c, long long first, long long count,
                                                     float* __restrict__ rays_o, float* __restrict__ rays_d,
                                                     float* __restrict__ view_dirs, float* __restrict__ rays_o_ndc,
                                                     float* __restrict__ rays_d_ndc) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const long long pix = first + i;
        const float x = (float)(pix % c.width) + c.pixel_offset;
        const float y = (float)(pix / c.width) + c.pixel_offset;
        // dirs = Kinv @ [x, y, 1]; then flip y and z (:362-363)
        float dir[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) dir[r] = (c.kinv[3 * r] * x + c.kinv[3 * r + 1] * y) + c.kinv[3 * r + 2] * 1.0f;
        dir[1] = -dir[1];
        dir[2] = -dir[2];
        // rays_d[j] = sum_k dirs[k] * pose[j,k]  (:365)
        float d[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) d[j] = (dir[0] * c.rot[3 * j] + dir[1] * c.rot[3 * j + 1]) + dir[2] * c.rot[3 * j + 2];
        const float o[3] = {c.org[0], c.org[1], c.org[2]};
        float* po = rays_o + 3 * i;
        float* pd = rays_d + 3 * i;
        po[0] = o[0]; po[1] = o[1]; po[2] = o[2];
        pd[0] = d[0]; pd[1] = d[1]; pd[2] = d[2];
        if (view_dirs) {
            const float nrm = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
            float* pv = view_dirs + 3 * i;
            pv[0] = __fdiv_rn(d[0], nrm); pv[1] = __fdiv_rn(d[1], nrm); pv[2] = __fdiv_rn(d[2], nrm);
        }
        if (c.ndc) {
            // shift origin to the near plane (:375-376), then project (:379-385)
            const float t = __fdiv_rn(-(c.near + o[2]), d[2]);
            const float sx = o[0] + t * d[0], sy = o[1] + t * d[1], sz = o[2] + t * d[2];
            const float two_near = 2.0f * c.near;
            float* pon = rays_o_ndc + 3 * i;
            float* pdn = rays_d_ndc + 3 * i;
            pon[0] = __fdiv_rn(c.cx * sx, sz);
            pon[1] = __fdiv_rn(c.cy * sy, sz);
            pon[2] = 1.0f + __fdiv_rn(two_near, sz);
            pdn[0] = c.cx * (__fdiv_rn(d[0], d[2]) - __fdiv_rn(sx, sz));
            pdn[1] = c.cy * (__fdiv_rn(d[1], d[2]) - __fdiv_rn(sy, sz));
            pdn[2] = __fdiv_rn(-two_near, sz);
        }
    }
}

// 3x3 inverse in double (adjugate), rounded to fp32: reproduces numpy.linalg.inv(float32 intrinsic) for camera
// matrices (checked bit-exact against the reference fixtures).
bool invert3x3(const float* m, float* out) {
    double a[9];
    for (int i = 0; i < 9; ++i) a[i] = m[i];
    const double c00 = a[4] * a[8] - a[5] * a[7], c01 = a[5] * a[6] - a[3] * a[8], c02 = a[3] * a[7] - a[4] * a[6];
    const double det = a[0] * c00 + a[1] * c01 + a[2] * c02;
    if (det == 0.0) return false;
    const double inv[9] = {c00 / det, (a[2] * a[7] - a[1] * a[8]) / det, (a[1] * a[5] - a[2] * a[4]) / det,
                           c01 / det, (a[0] * a[8] - a[2] * a[6]) / det, (a[2] * a[3] - a[0] * a[5]) / det,
                           c02 / det, (a[1] * a[6] - a[0] * a[7]) / det, (a[0] * a[4] - a[1] * a[3]) / det};
    for (int i = 0; i < 9; ++i) out[i] = (float)inv[i] + 0.0f;  // +0 turns -0 into +0 like LAPACK's result
    return true;
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
    RaygenConst c;
    if (!invert3x3(intrinsic, c.kinv)) return snerf::fail(SNERF_E_INVALID, "generate_rays: singular intrinsic");
    for (int r = 0; r < 3; ++r) {
        for (int k = 0; k < 3; ++k) c.rot[3 * r + k] = pose[4 * r + k];
        c.org[r] = pose[4 * r + 3];
    }
    c.near = near;
    const float fx = intrinsic[0], fy = intrinsic[4];
    c.cx = -1.0f / ((float)width / (2.0f * fx));
    c.cy = -1.0f / ((float)height / (2.0f * fy));
    c.pixel_offset = pixel_offset;
    c.width = width;
    c.ndc = ndc;
    hipLaunchKernelGGL(raygen_kernel, dim3(snerf::stride_grid(num_rays, 256)), dim3(256), 0, (hipStream_t)stream, c,
                       first_ray, num_rays, rays_o, rays_d, view_dirs, rays_o_ndc, rays_d_ndc);
    return snerf::check_launch("generate_rays");
}
