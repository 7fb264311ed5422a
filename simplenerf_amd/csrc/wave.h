// Wavefront (64-lane) primitives shared by the scan kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace snerf {

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Inclusive scans across the 64 lanes (Hillis-Steele over __shfl_up); identity-padded at the low lanes.
__device__ __forceinline__ float wave_inclusive_add(float v) {
    const int lane = lane_id();
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float n = __shfl_up(v, off, 64);
        if (lane >= off) v += n;
    }
    return v;
}

__device__ __forceinline__ float wave_inclusive_mul(float v) {
    const int lane = lane_id();
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float n = __shfl_up(v, off, 64);
        if (lane >= off) v *= n;
    }
    return v;
}

// Inclusive suffix sum: lane l receives v[l] + v[l+1] + ... + v[63] (only later lanes are ever added, so a tiny tail
// is not contaminated by rounding of a large head -- needed by the compositing backward).
__device__ __forceinline__ float wave_suffix_add(float v) {
    const int lane = lane_id();
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float n = __shfl_down(v, off, 64);
        if (lane + off < 64) v += n;
    }
    return v;
}

// Order this wave's LDS writes before its later LDS reads from other lanes.  LDS operations of one wave execute
// in issue order, so only the compiler needs fencing (no s_barrier: a wave owns its LDS region).
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

}  // namespace snerf
