// fp32 counterpart of mfma_shape_power.hip: bare loops, weight values from LDS (one ds_read_b128 = four k-steps), operands in
// registers, accumulation as in mlp_forward.hip -- (a) v_mfma_f32_32x32x2_f32 (one 32x32 tile), (b) v_mfma_f32_16x16x4_f32
// (four 16x16 tiles: two row halves x two sample halves).  Same FLOPs and LDS bytes.
//   usage: mfma_shape_f32 32|16 seconds
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kFrags = 64;     // 64 KiB of f32x4-per-lane fragments
constexpr int kIters = 2048;   // fragments per wave and launch (x 4 k-steps)

template <int SHAPE>
__global__ void __launch_bounds__(256, 1) loop_kernel(const f32x4* __restrict__ weights, const float* __restrict__ operands, float* out) {
    extern __shared__ __attribute__((aligned(16))) f32x4 lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < kFrags * 64; i += blockDim.x) lds[i] = weights[i];
    __syncthreads();
    float b[64];
#pragma unroll
    for (int k = 0; k < 64; ++k) b[k] = operands[(k * 64 + lane) % 4096];
    f32x16 acc32[2] = {{0}, {0}};
    f32x4 acc16[2][4] = {{{0}, {0}, {0}, {0}}, {{0}, {0}, {0}, {0}}};
    for (int it = 0; it < kIters; it += 16) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const f32x4 a = lds[((it + k) % kFrags) * 64 + lane];
            if (SHAPE == 32) {
#pragma unroll
                for (int q = 0; q < 4; ++q) acc32[k & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], b[(4 * k + q) & 63], acc32[k & 1], 0, 0, 0);
            } else {
                // the fragment's four values: k-steps q = 0..1 of row half 0 and of row half 1; each feeds two sample halves
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = q >> 1;
                    acc16[k & 1][2 * r + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], b[(4 * k + q) & 63], acc16[k & 1][2 * r + 0], 0, 0, 0);
                    acc16[k & 1][2 * r + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], b[(4 * k + q + 32) & 63], acc16[k & 1][2 * r + 1], 0, 0, 0);
                }
            }
        }
    }
    float s = 0.0f;
    for (int t = 0; t < 2; ++t) for (int r = 0; r < 16; ++r) s += acc32[t][r];
    for (int t = 0; t < 2; ++t) for (int u = 0; u < 4; ++u) for (int r = 0; r < 4; ++r) s += acc16[t][u][r];
    if (s == 12345.678f) out[0] = s;
}

int main(int argc, char** argv) {
    const int shape = argc > 1 ? atoi(argv[1]) : 32;
    const double seconds = argc > 2 ? atof(argv[2]) : 2.0;
    std::vector<float> w(kFrags * 64 * 4), x(4096);
    srand(1);
    auto rnd = [] { float u = 0; for (int i = 0; i < 4; ++i) u += (float)rand() / (float)RAND_MAX - 0.5f; return u; };
    for (auto& v : w) v = rnd() * 0.2f;
    for (auto& v : x) v = rnd() > 0 ? rnd() * 0.7f : 0.0f;
    f32x4* dw; float *dx, *dout;
    (void)hipMalloc(&dw, w.size() * 4); (void)hipMalloc(&dx, x.size() * 4); (void)hipMalloc(&dout, 4);
    (void)hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
    auto k32 = loop_kernel<32>; auto k16 = loop_kernel<16>;
    const size_t ldsb = kFrags * 1024;
    (void)hipFuncSetAttribute((const void*)k32, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    (void)hipFuncSetAttribute((const void*)k16, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    const int blocks = 256 * 4;
    // per fragment: 4 values x (32x32x2 = 4096 FLOP)  |  4 values x 2 x (16x16x4 = 2048 FLOP)
    const double flop_per_launch = (double)blocks * 4 * kIters * 4 * 4096.0;
    auto t0 = std::chrono::steady_clock::now();
    long launches = 0;
    double elapsed = 0;
    while (elapsed < seconds) {
        for (int i = 0; i < 10; ++i) {
            if (shape == 32) hipLaunchKernelGGL(k32, dim3(blocks), dim3(256), ldsb, 0, dw, dx, dout);
            else hipLaunchKernelGGL(k16, dim3(blocks), dim3(256), ldsb, 0, dw, dx, dout);
        }
        (void)hipDeviceSynchronize();
        launches += 10;
        elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    printf("fp32 shape %d: %.1f TFLOP/s (%ld launches in %.2f s)\n", shape, flop_per_launch * launches / elapsed / 1e12, launches, elapsed);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
