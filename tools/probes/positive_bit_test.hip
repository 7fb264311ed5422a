// Checks positive_bit (mlp_device.h: compare + v_addc_co building ReLU mask words) against the plain C expression on random data,
// including zeros, negatives zeros, NaN and denormals.   hipcc --offload-arch=gfx950 -O3 -I simplenerf_amd/csrc -I include ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
__device__ __forceinline__ unsigned positive_bit(unsigned m, float v) {
    const unsigned long long positive = __builtin_amdgcn_fcmpf(v, 0.0f, 2);
    asm("v_addc_co_u32 %0, vcc, %0, %0, %1" : "+v"(m) : "s"(positive) : "vcc");
    return m;
}
__global__ void k(const float* in, unsigned* out_asm, unsigned* out_c, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned a = 0, c = 0;
#pragma unroll
    for (int r = 15; r >= 0; --r) a = positive_bit(a, in[i * 16 + r]);
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
        v = t == 0 ? 0.0f : t == 1 ? -0.0f : t == 2 ? 1e-42f : t == 3 ? -1e-42f : t == 4 ? __builtin_nanf("") : (float)rand() / RAND_MAX - 0.5f;
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
