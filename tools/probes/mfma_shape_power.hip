// Which fp16 MFMA shape does more work under the board's power cap?  Two bare loops over the same data: every wave reads a
// 1-KiB weight fragment from LDS (ds_read_b128) and multiplies it with operands it keeps in registers, accumulating like the
// fused MLP kernels do -- (a) one v_mfma_f32_32x32x16_f16 per fragment (32 out rows x 32 samples), (b) two
// v_mfma_f32_16x16x32_f16 per fragment (16 out rows x 32 k, two 16-sample halves).  Same FLOPs, same LDS bytes, same
// registers.  (c) shape 64 (round 3): a 64-SAMPLE wave tile -- every fragment feeds FOUR 16x16x32 MFMAs (four 16-sample
// quarters): half the LDS bytes per FLOP; is the 16-bit inference kernel's ceiling LDS read bandwidth?
// Prints TFLOP/s; run beside a power/clock sampler (the Python driver mfma_shape_power.py does both).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_shape_power tools/probes/mfma_shape_power.hip && /tmp/mfma_shape_power 32|16|64 seconds waves [workgroups per CU]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kFrags = 64;      // 64 KiB of fragments in LDS, walked round and round
constexpr int kIters = 4096;    // fragments per wave and launch

template <int SHAPE>
__global__ void __launch_bounds__(512, 2) loop_kernel(const f16x8* __restrict__ weights, const f16x8* __restrict__ operands, float* out) {
    extern __shared__ __attribute__((aligned(16))) f16x8 lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < kFrags * 64; i += blockDim.x) lds[i] = weights[i];
    __syncthreads();
    f16x8 b[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) b[k] = operands[(k * 64 + lane) % (16 * 64)];
    f32x16 acc32 = {0};
    f32x4 acc16[4] = {{0}, {0}, {0}, {0}};
    f32x4 acc64[8] = {{0}, {0}, {0}, {0}, {0}, {0}, {0}, {0}};
    for (int it = 0; it < kIters; it += 16) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const f16x8 a = lds[((it + k) % kFrags) * 64 + lane];
            if (SHAPE == 32) {
                acc32 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[k], acc32, 0, 0, 0);
            } else if (SHAPE == 64) {
                // fragment k: 16 rows x 32 k, multiplied with the operands of FOUR 16-sample quarters
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    acc64[(k & 1) * 4 + q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b[(k + 4 * q) & 15], acc64[(k & 1) * 4 + q], 0, 0, 0);
            } else {
                // fragment k: 16 rows x 32 k; rows alternate between the two row halves of the 32-row tile
                acc16[(k & 1) * 2 + 0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b[k], acc16[(k & 1) * 2 + 0], 0, 0, 0);
                acc16[(k & 1) * 2 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b[(k + 8) & 15], acc16[(k & 1) * 2 + 1], 0, 0, 0);
            }
        }
    }
    float s = 0.0f;
    for (int r = 0; r < 16; ++r) s += acc32[r];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 4; ++r) s += acc16[t][r];
    for (int t = 0; t < 8; ++t) for (int r = 0; r < 4; ++r) s += acc64[t][r];
    if (s == 12345.678f) out[0] = s;   // keeps the loop alive
}

int main(int argc, char** argv) {
    const int shape = argc > 1 ? atoi(argv[1]) : 32;
    const double seconds = argc > 2 ? atof(argv[2]) : 2.0;
    const int threads = argc > 3 ? atoi(argv[3]) * 64 : 512;
    const int wgs_per_cu = argc > 4 ? atoi(argv[4]) : 2 * 512 / threads;
    std::vector<_Float16> w(kFrags * 64 * 8), x(16 * 64 * 8);
    srand(1);
    auto rnd = [] { float u = 0; for (int i = 0; i < 4; ++i) u += (float)rand() / RAND_MAX - 0.5f; return u; };
    for (auto& v : w) v = (_Float16)(rnd() * 0.2f);
    for (auto& v : x) v = (_Float16)(rnd() > 0 ? rnd() * 0.7f : 0.0f);   // post-ReLU-like operands
    f16x8 *dw, *dx; float* dout;
    hipMalloc(&dw, w.size() * 2); hipMalloc(&dx, x.size() * 2); hipMalloc(&dout, 4);
    hipMemcpy(dw, w.data(), w.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dx, x.data(), x.size() * 2, hipMemcpyHostToDevice);
    auto k32 = loop_kernel<32>; auto k16 = loop_kernel<16>; auto k64 = loop_kernel<64>;
    const size_t ldsb = kFrags * 1024;
    hipFuncSetAttribute((const void*)k32, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipFuncSetAttribute((const void*)k16, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipFuncSetAttribute((const void*)k64, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    const int blocks = 256 * wgs_per_cu;          // default: two 8-wave (or four 4-wave) workgroups per CU
    const double flop_per_launch = (double)blocks * (threads / 64) * kIters * (shape == 64 ? 65536.0 : 32768.0);
    auto t0 = std::chrono::steady_clock::now();
    long launches = 0;
    double elapsed = 0;
    while (elapsed < seconds) {
        for (int i = 0; i < 20; ++i) {
            if (shape == 32) hipLaunchKernelGGL(k32, dim3(blocks), dim3(threads), ldsb, 0, dw, dx, dout);
            else if (shape == 64) hipLaunchKernelGGL(k64, dim3(blocks), dim3(threads), ldsb, 0, dw, dx, dout);
            else hipLaunchKernelGGL(k16, dim3(blocks), dim3(threads), ldsb, 0, dw, dx, dout);
        }
        hipDeviceSynchronize();
        launches += 20;
        elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    printf("shape %d waves/WG %d WGs/CU %d: %.1f TFLOP/s (%ld launches in %.2f s)\n", shape, threads / 64, wgs_per_cu, flop_per_launch * launches / elapsed / 1e12, launches, elapsed);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
