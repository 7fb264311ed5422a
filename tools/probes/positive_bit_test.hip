// Checks positive_bits (mlp_device.h: sign of 0 - v shifted into the ReLU mask word by v_alignbit) against the plain C expression
// on random data including both zeros and denormals of both signs (NaN is excluded: either bit is allowed there).   hipcc --offload-arch=gfx950 -O3 -I simplenerf_amd/csrc -I include ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned positive_bits(unsigned m, float a, float b) {
    const f32x2 t = f32x2{0.0f, 0.0f} - f32x2{a, b};
    m = __builtin_amdgcn_alignbit(m, __float_as_uint(t[0]), 31);
    return __builtin_amdgcn_alignbit(m, __float_as_uint(t[1]), 31);
}
__global__ void k(const float* in, unsigned* out_asm, unsigned* out_c, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned a = 0, c = 0;
#pragma unroll
    for (int r = 15; r >= 1; r -= 2) a = positive_bits(a, in[i * 16 + r], in[i * 16 + r - 1]);
#pragma unroll
    for (int r = 0; r < 16; ++r) c |= (in[i * 16 + r] > 0.0f ? 1u : 0u) << r;
    out_asm[i] = a; out_c[i] = c;
}
int main() {
    const int n = 1 << 16;
    std::vector<float> h(n * 16);
    srand(3);
    for (auto& v : h) {
        const int t = rand() % 16;
        v = t == 0 ? 0.0f : t == 1 ? -0.0f : t == 2 ? 1e-42f : t == 3 ? -1e-42f : (float)rand() / RAND_MAX - 0.5f;
    }
    float* d; unsigned *a, *c;
    hipMalloc(&d, h.size() * 4); hipMalloc(&a, n * 4); hipMalloc(&c, n * 4);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    // n is not a multiple of the block: the last wave has inactive lanes
    hipLaunchKernelGGL(k, dim3((n - 37 + 255) / 256), dim3(256), 0, 0, d, a, c, n - 37);
    std::vector<unsigned> ha(n), hc(n);
    hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hc.data(), c, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n - 37; ++i) bad += ha[i] != hc[i];
    printf("positive_bit: %d of %d words differ from the C expression\n", bad, n - 37);
    return bad != 0;
}
