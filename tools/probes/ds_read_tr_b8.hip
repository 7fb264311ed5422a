// What does ds_read_b64_tr_b8 deliver?  (The ISA text is not in the image.)  One wave; LDS holds byte b = its own offset
// (mod 256) in a 16 x 8-byte block per 16-lane group; lane l passes address 8 * l (its "own" 8 bytes) and the 8 bytes it
// receives are printed, next to ds_read_b64_tr_b16 on the same image for comparison.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/tr8 tools/probes/ds_read_tr_b8.hip && /tmp/tr8
#include <hip/hip_runtime.h>
#include <cstdio>

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__global__ void probe(unsigned long long* out8, unsigned long long* out16) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[1024];
    const int lane = threadIdx.x;
    for (int i = lane; i < 1024; i += 64) lds[i] = (unsigned char)i;
    __syncthreads();
    const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds + lane * 8;
    u32x2 a, b;
    asm volatile("ds_read_b64_tr_b8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(a) : "v"(addr) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(b) : "v"(addr) : "memory");
    out8[lane] = ((unsigned long long)a[1] << 32) | a[0];
    out16[lane] = ((unsigned long long)b[1] << 32) | b[0];
}

int main() {
    unsigned long long *d8, *d16, h8[64], h16[64];
    (void)hipMalloc(&d8, 512); (void)hipMalloc(&d16, 512);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d8, d16);
    (void)hipMemcpy(h8, d8, 512, hipMemcpyDeviceToHost); (void)hipMemcpy(h16, d16, 512, hipMemcpyDeviceToHost);
    printf("lane: tr_b8 bytes (LDS offsets, element 0 first) | tr_b16 halfwords (LDS byte offsets of each 16-bit element)\n");
    for (int l = 0; l < 64; ++l) {
        printf("%2d:", l);
        for (int k = 0; k < 8; ++k) printf(" %3llu", (h8[l] >> (8 * k)) & 0xff);
        printf("  |");
        for (int k = 0; k < 4; ++k) printf(" %3llu", (h16[l] >> (16 * k)) & 0xff);   // low byte of each element = its offset
        printf("\n");
    }
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
