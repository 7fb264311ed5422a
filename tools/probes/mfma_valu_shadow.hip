// How much VALU work fits in the shadow of an MFMA issued by the SAME wave (one wave per SIMD)?  A dependent chain of
// v_mfma_f32_32x32x16_f16 (32 matrix-pipe cycles each) with K independent VALU instructions of one kind behind every MFMA;
// prints time per MFMA relative to K = 0.  If the VALU issue overlaps the matrix pipe the ratio stays ~1 until K x (cycles per
// instruction) reaches 32.  Kinds: and (v_and_b32), bfe (v_bfe_i32), pkmul (v_pk_mul_f32), cvtbf (v_cvt_pk_bf16_f32),
// cvtf16 (v_cvt_pk_f16_f32), max3 (v_maximum3_f32), accrd (v_accvgpr_read_b32 of accumulators no MFMA in flight writes),
// alignbit (v_alignbit_b32), pkadd (v_pk_add_f32).
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/mfma_valu_shadow tools/probes/mfma_valu_shadow.hip && gpurun_out/mfma_valu_shadow
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum { AND, BFE, PKMUL, CVTBF, CVTF16, MAX3, ACCRD, ALIGNBIT, PKADD, PKMAXH, FMA, LDEXP, STORE, KINDS };
static const char* kNames[KINDS] = {"and", "bfe", "pkmul", "cvtbf", "cvtf16", "max3", "accrd", "alignbit", "pkadd", "pkmaxf16", "fma", "ldexp", "store16B"};

typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int KIND>
__device__ __forceinline__ void one(float (&v)[16], f32x2 (&w)[8], const f32x16& spare, int i, f32x4* sink = nullptr) {
    if (KIND == PKMAXH) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(v[i & 15]) : "v"(v[(i + 5) & 15]));
    if (KIND == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i & 15]) : "v"(v[(i + 5) & 15]), "v"(v[(i + 6) & 15]));
    if (KIND == LDEXP) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(v[i & 15]) : "v"(3));
    if (KIND == STORE) __builtin_nontemporal_store(f32x4{v[0], v[1], v[2], v[3]}, sink + (i & 63) * 4096);
    float& x = v[i & 15];
    f32x2& p = w[i & 7];
    if (KIND == AND) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(v[(i + 5) & 15]));
    if (KIND == BFE) asm volatile("v_bfe_i32 %0, %1, 3, 1" : "=v"(x) : "v"(v[(i + 5) & 15]));
    if (KIND == PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p) : "v"(w[(i + 3) & 7]));
    if (KIND == PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p) : "v"(w[(i + 3) & 7]));
    if (KIND == CVTBF) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(x) : "v"(v[(i + 5) & 15]), "v"(v[(i + 6) & 15]));
    if (KIND == CVTF16) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(x) : "v"(v[(i + 5) & 15]), "v"(v[(i + 6) & 15]));
    if (KIND == MAX3) asm volatile("v_maximum3_f32 %0, %0, |%1|, |%2|" : "+v"(x) : "v"(v[(i + 5) & 15]), "v"(v[(i + 6) & 15]));
    if (KIND == ALIGNBIT) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(x) : "v"(v[(i + 5) & 15]));
    if (KIND == ACCRD) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(spare[i & 15]));
}

template <int KIND, int K>
__global__ void __launch_bounds__(256, 1) loop_kernel(const f16x8* ops, float* out, int iters) {
    const int lane = threadIdx.x & 63;
    f16x8 a = ops[lane], b = ops[64 + lane];
    f32x16 acc = {0}, spare;
    float v[16];
    f32x2 w[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) { v[i] = (float)a[i & 7] + i; spare[i] = v[i]; }
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] = f32x2{v[2 * i], v[2 * i + 1]};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < K; ++i) one<KIND>(v, w, spare, m * K + i, reinterpret_cast<f32x4*>(out) + 1024 + (blockIdx.x * 256 + threadIdx.x));
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i] + v[i] + w[i & 7][0] + w[i & 7][1];
    if (s == 1234.5f) out[0] = s;
}

template <int KIND, int K>
float run(const f16x8* ops, float* out) {
    const int iters = 2048;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((loop_kernel<KIND, K>), dim3(256), dim3(256), 0, 0, ops, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((loop_kernel<KIND, K>), dim3(256), dim3(256), 0, 0, ops, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 5 / (iters * 8) * 1e6f;   // ns per MFMA
}

template <int KIND>
void row(const f16x8* ops, float* out, float base) {
    const float t[] = {run<KIND, 2>(ops, out), run<KIND, 4>(ops, out), run<KIND, 6>(ops, out), run<KIND, 8>(ops, out), run<KIND, 12>(ops, out), run<KIND, 16>(ops, out)};
    printf("%-9s K=2 %.2f  K=4 %.2f  K=6 %.2f  K=8 %.2f  K=12 %.2f  K=16 %.2f   (time per MFMA / time per bare MFMA)\n", kNames[KIND],
           t[0] / base, t[1] / base, t[2] / base, t[3] / base, t[4] / base, t[5] / base);
}

int main() {
    std::vector<_Float16> h(128 * 8);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (_Float16)(0.001f * (i % 97));
    f16x8* ops; float* out;
    hipMalloc(&ops, h.size() * 2); hipMalloc(&out, (size_t)(1024 + 64 * 4096 + 65536) * 16);
    hipMemcpy(ops, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    run<AND, 0>(ops, out);
    const float base = run<AND, 0>(ops, out);
    printf("bare dependent MFMA chain: %.2f ns per MFMA (one wave per SIMD, 256 workgroups)\n", base);
    row<AND>(ops, out, base); row<BFE>(ops, out, base); row<PKMUL>(ops, out, base); row<PKADD>(ops, out, base); row<CVTBF>(ops, out, base);
    row<CVTF16>(ops, out, base); row<MAX3>(ops, out, base); row<ACCRD>(ops, out, base); row<ALIGNBIT>(ops, out, base);
    row<PKMAXH>(ops, out, base); row<FMA>(ops, out, base); row<LDEXP>(ops, out, base);
    {   // stores: K = 1 and 2 per MFMA only (16 B per lane = 1 KiB per instruction)
        const float t1 = run<STORE, 1>(ops, out), t2 = run<STORE, 2>(ops, out);
        printf("store16B  K=1 %.2f  K=2 %.2f   (one / two global_store_dwordx4 per MFMA)\n", t1 / base, t2 / base);
    }
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
