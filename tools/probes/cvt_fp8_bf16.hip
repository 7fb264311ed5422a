// The bf16 side of the fp8 saved activations (SNERF_PRECISION_BF16S8): what v_cvt_scalef32_pk_fp8_bf16 does above e4m3's
// range, whether an UNSIGNED 16-bit minimum on the bit patterns clamps non-negative bf16 values (their order is the order of
// their bits), and the round trip through v_cvt_scalef32_pk_bf16_fp8.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/cvt_fp8_bf16 tools/probes/cvt_fp8_bf16.hip && /tmp/cvt_fp8_bf16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

__global__ void k(const float* in, float* raw, float* clamped, int n) {
    const int i = threadIdx.x;
    if (i >= n) return;
    const bf16x2 v = {(__bf16)in[i], (__bf16)in[i]};
    s16x2 p = {0, 0};
    p = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(p, v, 1.0f, false);
    const bf16x2 back = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(__builtin_bit_cast(unsigned, p), 1.0f, false);
    raw[i] = (float)back[0];
    const u16x2 top = {0x43E0, 0x43E0};     // 448.0 as bf16
    const u16x2 bits = __builtin_elementwise_min(__builtin_bit_cast(u16x2, v), top);
    s16x2 q = {0, 0};
    q = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(q, __builtin_bit_cast(bf16x2, bits), 1.0f, false);
    const bf16x2 back2 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(__builtin_bit_cast(unsigned, q), 1.0f, false);
    clamped[i] = (float)back2[0];
}

int main() {
    const float h[] = {0.0f, 0.0009f, 0.002f, 0.3f, 1.0f, 1.06f, 100.0f, 440.0f, 448.0f, 450.0f, 464.0f, 500.0f, 3000.0f, 1e30f};
    const int n = sizeof(h) / sizeof(h[0]);
    float *in, *raw, *cl;
    hipMalloc(&in, n * 4); hipMalloc(&raw, n * 4); hipMalloc(&cl, n * 4);
    hipMemcpy(in, h, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, in, raw, cl, n);
    float a[32], b[32];
    hipMemcpy(a, raw, n * 4, hipMemcpyDeviceToHost); hipMemcpy(b, cl, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) printf("%12g -> fp8 -> %12g   with the 16-bit minimum first: %12g\n", h[i], a[i], b[i]);
    return 0;
}
