// Per-ray pinhole arithmetic shared by K1 (a pixel range of one frame, raygen.hip) and the training-batch assembler
// (arbitrary pixels of many views, batch.hip).  Reference: DataPreprocessor.get_rays :351-368, get_view_dirs :392-394,
// get_ndc_rays :371-389 (src/data_preprocessors/DataPreprocessor01.py).  Every expression keeps the reference's fp32
// operation order (the library is built with -ffp-contract=off), which makes the result bit-identical to numpy.
#pragma once
#include <hip/hip_runtime.h>

namespace snerf {

struct Camera {      // 24 floats; also the row layout of the device camera table (SNERF_CAMERA_FLOATS)
    float kinv[9];   // inverse intrinsic, fp32
    float rot[9];    // pose[:3,:3]
    float org[3];    // pose[:3,3]
    float cx, cy;    // -1/(w/(2 fx)), -1/(h/(2 fy)) evaluated in fp32 steps like the reference expression
    float pad;
};

// 3x3 inverse in double (adjugate), rounded to fp32: reproduces numpy.linalg.inv(float32 intrinsic) for camera
// matrices (checked bit-exact against the reference fixtures).  Runs identically on host and device (IEEE doubles).
__host__ __device__ inline bool invert3x3(const float* m, float* out) {
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

__host__ __device__ inline bool make_camera(const float* intrinsic, const float* pose, int height, int width, Camera* c) {
    if (!invert3x3(intrinsic, c->kinv)) return false;
    for (int r = 0; r < 3; ++r) {
        for (int k = 0; k < 3; ++k) c->rot[3 * r + k] = pose[4 * r + k];
        c->org[r] = pose[4 * r + 3];
    }
    const float fx = intrinsic[0], fy = intrinsic[4];
    c->cx = -1.0f / ((float)width / (2.0f * fx));
    c->cy = -1.0f / ((float)height / (2.0f * fy));
    c->pad = 0.0f;
    return true;
}

// Writes 3 floats to each non-null output.  (x, y) = pixel coordinates (+ the optional half-pixel offset).
__device__ __forceinline__ void pinhole_ray(const Camera& c, float x, float y, float near, bool ndc, float* po, float* pd,
                                            float* pv, float* pon, float* pdn) {
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
    po[0] = o[0]; po[1] = o[1]; po[2] = o[2];
    pd[0] = d[0]; pd[1] = d[1]; pd[2] = d[2];
    if (pv) {
        const float nrm = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
        pv[0] = __fdiv_rn(d[0], nrm); pv[1] = __fdiv_rn(d[1], nrm); pv[2] = __fdiv_rn(d[2], nrm);
    }
    if (ndc) {
        // shift origin to the near plane (:375-376), then project (:379-385)
        const float t = __fdiv_rn(-(near + o[2]), d[2]);
        const float sx = o[0] + t * d[0], sy = o[1] + t * d[1], sz = o[2] + t * d[2];
        const float two_near = 2.0f * near;
        pon[0] = __fdiv_rn(c.cx * sx, sz);
        pon[1] = __fdiv_rn(c.cy * sy, sz);
        pon[2] = 1.0f + __fdiv_rn(two_near, sz);
        pdn[0] = c.cx * (__fdiv_rn(d[0], d[2]) - __fdiv_rn(sx, sz));
        pdn[1] = c.cy * (__fdiv_rn(d[1], d[2]) - __fdiv_rn(sy, sz));
        pdn[2] = __fdiv_rn(-two_near, sz);
    }
}

}  // namespace snerf
