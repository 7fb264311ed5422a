// What do v_cvt_scalef32_pk_fp8_f16 / v_cvt_scalef32_pk_f16_fp8 do with their scale operand, and where do they saturate?
// (ISA text not in the image.)   hipcc --offload-arch=gfx950 -O2 -w -o /tmp/cvt8 tools/probes/cvt_fp8_scale.hip && /tmp/cvt8
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, float* out, unsigned* raw, float scale) {
    const int t = threadIdx.x;
    f16x2 v = {(_Float16)in[2 * t], (_Float16)in[2 * t + 1]};
    s16x2 old = {0, 0};
    s16x2 packed = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(old, v, scale, false);
    unsigned word = (unsigned short)packed[0];
    raw[t] = __builtin_bit_cast(unsigned, packed);
    f16x2 back = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(word, scale, false);
    out[2 * t] = (float)back[0];
    out[2 * t + 1] = (float)back[1];
}
int main() {
    float h[16] = {0.f, 1.f, 1.0625f, 3.3f, 100.f, 447.f, 500.f, 1000.f, 0.001f, 0.01f, -2.5f, 17.f, 240.f, 256.f, 0.5f, 0.75f};
    float *din, *dout; unsigned* draw;
    (void)hipMalloc(&din, 64); (void)hipMalloc(&dout, 64); (void)hipMalloc(&draw, 32);
    (void)hipMemcpy(din, h, 64, hipMemcpyHostToDevice);
    for (float scale : {1.0f, 2.0f, 0.5f, 8.0f}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(8), 0, 0, din, dout, draw, scale);
        float o[16]; unsigned r[8];
        (void)hipMemcpy(o, dout, 64, hipMemcpyDeviceToHost); (void)hipMemcpy(r, draw, 32, hipMemcpyDeviceToHost);
        printf("scale %g:\n", scale);
        for (int i = 0; i < 16; ++i) printf("  %10.5f -> fp8 0x%02x -> %10.5f\n", h[i], (r[i / 2] >> (8 * (i & 1))) & 0xff, o[i]);
    }
    return 0;
}
