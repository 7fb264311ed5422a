// What does the single-product MLP kernels' inner loop cost beyond its MFMAs?  One dependent chain of v_mfma_f32_32x32x16_f16
// per wave, as in seg_mfma1 (mlp_device_f16.h), with optional extras per MFMA: (L) the 1-KiB weight fragment read from LDS
// four k-steps ahead with counted lgkmcnt waits, (V) K full-rate VALU instructions, (T) every 16th
// MFMA a fresh accumulator (a new tile: no dependence on the previous MFMA).  One or two waves per SIMD.
// Prints ns per MFMA and SIMD.   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/mfma_core_mix tools/probes/mfma_core_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f16x8 lds_read(unsigned addr, int off) {
    f16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(off) : "memory");
    return v;
}
template <int N> __device__ __forceinline__ void wait_but(f16x8& a) {
    if (N == 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(a)::"memory");
    if (N == 3) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(a)::"memory");
    if (N == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a)::"memory");
    if (N == 1) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(a)::"memory");
    if (N == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a)::"memory");
}

template <bool L, int KV, int WAVES>
__global__ void __launch_bounds__(WAVES * 256, WAVES) core_kernel(const f16x8* ops, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) f16x8 lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16 * 64; i += blockDim.x) lds[i] = ops[i & 127];
    __syncthreads();
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)lds + lane * 16;
    f16x8 b = ops[64 + lane], a0 = ops[lane];
    f32x16 acc = {0};
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)a0[i] + i;
    float sink = 0;
    for (int it = 0; it < iters; ++it) {
        f16x8 a[5];
        if (L) {
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = lds_read(base, i * 1024);
        }
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            if (L) {
                if (ks + 4 < 16) a[(ks + 4) % 5] = lds_read(base, (ks + 4) * 1024);
                const int newer = 15 - ks < 4 ? 15 - ks : 4;
                f16x8& cur = a[ks % 5];
                if (newer == 4) wait_but<4>(cur); else if (newer == 3) wait_but<3>(cur); else if (newer == 2) wait_but<2>(cur);
                else if (newer == 1) wait_but<1>(cur); else wait_but<0>(cur);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur, b, acc, 0, 0, 0);
            } else {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b, acc, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < KV; ++i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[i & 7]) : "v"(v[(i + 3) & 7]));
        }
        sink += acc[0];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;     // next tile: fresh accumulator
    }
    float s = sink;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 1234.5f) out[0] = s;
}

template <bool L, int KV, int WAVES>
void run(const char* label, const f16x8* ops, float* out) {
    const int iters = 256;
    auto k = core_kernel<L, KV, WAVES>;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(256), dim3(WAVES * 256), 16 * 1024, 0, ops, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k, dim3(256), dim3(WAVES * 256), 16 * 1024, 0, ops, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: WAVES waves x iters x 16 MFMAs
    printf("%-44s waves/SIMD %d: %.2f ns per MFMA and SIMD\n", label, WAVES, ms / 5 / (iters * 16.0 * WAVES) * 1e6);
}

int main() {
    std::vector<_Float16> h(128 * 8);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (_Float16)(0.001f * (i % 97));
    f16x8* ops; float* out;
    hipMalloc(&ops, h.size() * 2); hipMalloc(&out, 4);
    hipMemcpy(ops, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    run<false, 0, 1>("warm-up", ops, out);
    run<false, 0, 1>("MFMA chain only", ops, out);
    run<false, 0, 2>("MFMA chain only", ops, out);
    run<true, 0, 1>("+ LDS fragment reads", ops, out);
    run<true, 0, 2>("+ LDS fragment reads", ops, out);
    run<true, 4, 1>("+ LDS + 4 VALU", ops, out);
    run<true, 4, 2>("+ LDS + 4 VALU", ops, out);
    run<true, 7, 1>("+ LDS + 7 VALU", ops, out);
    run<true, 7, 2>("+ LDS + 7 VALU", ops, out);
    run<false, 7, 2>("(no LDS) 7 VALU", ops, out);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
