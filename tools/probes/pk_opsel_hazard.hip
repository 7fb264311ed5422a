// Stand-alone check of one instruction form beside matrix work (round 5, see DESIGN.md "Side by side: the packed multiply").
// "Victim" kernels repeat one packed fp32 instruction form each -- among them  v_pk_mul_f32 d, a, b op_sel:[0,1]  (both result
// halves read the HIGH register of b: the form the compiler chose for the compositing backward's  w * (g_r, g_g)) -- and compare
// both halves with the single-lane instruction; an "aggressor" kernel runs MFMAs (and streams memory) on another stream.
// Prints mismatches per form, alone and beside.  (Generated table of forms: every operand-selection form the library's code
// objects contain, round 5, plus their neighbours.)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/pk_opsel_hazard tools/probes/pk_opsel_hazard.hip && /tmp/pk_opsel_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));


__global__ void __launch_bounds__(256) victim_0(unsigned long long* bad, float* sink, int iters) {   // mul plain
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    f32x2 a = {1.0f + 0.001f * (t & 1023), 2.0f + 0.002f * (t & 511)};
    f32x2 b = {3.0f + 0.003f * (t & 255), 0.5f + 0.004f * (t & 127)};
    f32x2 c = {0.25f + 0.005f * (t & 63), 7.0f + 0.006f * (t & 31)};
    unsigned long long mine = 0;
    float keep = 0.0f;
    for (int i = 0; i < iters; ++i) {
        f32x2 d;
        float lo, hi;
        asm volatile("v_pk_mul_f32 %0, %1, %2 " : "=v"(d) : "v"(a), "v"(b));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(lo) : "v"(a.x), "v"(b.x));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(hi) : "v"(a.y), "v"(b.y));
        mine += (__float_as_uint(d.x) != __float_as_uint(lo)) + (__float_as_uint(d.y) != __float_as_uint(hi));
        keep += d.x + d.y;
        a.x += 0.000001f; b.y += 0.000001f; a.y -= 0.000001f; b.x += 0.000002f;
    }
    if (mine) atomicAdd(bad, mine);
    if (keep == 123.456f) sink[t] = keep;
}

__global__ void __launch_bounds__(256) victim_1(unsigned long long* bad, float* sink, int iters) {   // mul op_sel:[0,1]
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    f32x2 a = {1.0f + 0.001f * (t & 1023), 2.0f + 0.002f * (t & 511)};
    f32x2 b = {3.0f + 0.003f * (t & 255), 0.5f + 0.004f * (t & 127)};
    f32x2 c = {0.25f + 0.005f * (t & 63), 7.0f + 0.006f * (t & 31)};
    unsigned long long mine = 0;
    float keep = 0.0f;
    for (int i = 0; i < iters; ++i) {
        f32x2 d;
        float lo, hi;
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(a), "v"(b));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(lo) : "v"(a.x), "v"(b.y));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(hi) : "v"(a.y), "v"(b.y));
        mine += (__float_as_uint(d.x) != __float_as_uint(lo)) + (__float_as_uint(d.y) != __float_as_uint(hi));
        keep += d.x + d.y;
        a.x += 0.000001f; b.y += 0.000001f; a.y -= 0.000001f; b.x += 0.000002f;
    }
    if (mine) atomicAdd(bad, mine);
    if (keep == 123.456f) sink[t] = keep;
}

__global__ void __launch_bounds__(256) victim_2(unsigned long long* bad, float* sink, int iters) {   // mul op_sel:[1,0]
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    f32x2 a = {1.0f + 0.001f * (t & 1023), 2.0f + 0.002f * (t & 511)};
    f32x2 b = {3.0f + 0.003f * (t & 255), 0.5f + 0.004f * (t & 127)};
    f32x2 c = {0.25f + 0.005f * (t & 63), 7.0f + 0.006f * (t & 31)};
    unsigned long long mine = 0;
    float keep = 0.0f;
    for (int i = 0; i < iters; ++i) {
        f32x2 d;
        float lo, hi;
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(d) : "v"(a), "v"(b));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(lo) : "v"(a.y), "v"(b.x));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(hi) : "v"(a.y), "v"(b.y));
        mine += (__float_as_uint(d.x) != __float_as_uint(lo)) + (__float_as_uint(d.y) != __float_as_uint(hi));
        keep += d.x + d.y;
        a.x += 0.000001f; b.y += 0.000001f; a.y -= 0.000001f; b.x += 0.000002f;
    }
    if (mine) atomicAdd(bad, mine);
    if (keep == 123.456f) sink[t] = keep;
}

__global__ void __launch_bounds__(256) victim_3(unsigned long long* bad, float* sink, int iters) {   // mul op_sel:[1,1]
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    f32x2 a = {1.0f + 0.001f * (t & 1023), 2.0f + 0.002f * (t & 511)};
    f32x2 b = {3.0f + 0.003f * (t & 255), 0.5f + 0.004f * (t & 127)};
    f32x2 c = {0.25f + 0.005f * (t & 63), 7.0f + 0.006f * (t & 31)};
    unsigned long long mine = 0;
    float keep = 0.0f;
    for (int i = 0; i < iters; ++i) {
        f32x2 d;
        float lo, hi;
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1]" : "=v"(d) : "v"(a), "v"(b));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(lo) : "v"(a.y), "v"(b.y));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(hi) : "v"(a.y), "v"(b.y));
        mine += (__float_as_uint(d.x) != __float_as_uint(lo)) + (__float_as_uint(d.y) != __float_as_uint(hi));
        keep += d.x + d.y;
        a.x += 0.000001f; b.y += 0.000001f; a.y -= 0.000001f; b.x += 0.000002f;
    }
    if (mine) atomicAdd(bad, mine);
    if (keep == 123.456f) sink[t] = keep;
}

__global__ void __launch_bounds__(256) victim_4(unsigned long long* bad, float* sink, int iters) {   // mul op_sel:[1,0] op_sel_hi:[0,1]
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    f32x2 a = {1.0f + 0.001f * (t & 1023), 2.0f + 0.002f * (t & 511)};
    f32x2 b = {3.0f + 0.003f * (t & 255), 0.5f + 0.004f * (t & 127)};
    f32x2 c = {0.25f + 0.005f * (t & 63), 7.0f + 0.006f * (t & 31)};
    unsigned long long mine = 0;
    float keep = 0.0f;
    for (int i = 0; i < iters; ++i) {
        f32x2 d;
        float lo, hi;
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(lo) : "v"(a.y), "v"(b.x));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(hi) : "v"(a.x), "v"(b.y));
        mine += (__float_as_uint(d.x) != __float_as_uint(lo)) + (__float_as_uint(d.y) != __float_as_uint(hi));
        keep += d.x + d.y;
        a.x += 0.000001f; b.y += 0.000001f; a.y -= 0.000001f; b.x += 0.000002f;
    }
    if (mine) atomicAdd(bad, mine);
    if (keep == 123.456f) sink[t] = keep;
}

__global__ void __launch_bounds__(256) victim_5(unsigned long long* bad, float* sink, int iters) {   // mul op_sel:[0,1] op_sel_hi:[1,0]
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    f32x2 a = {1.0f + 0.001f * (t & 1023), 2.0f + 0.002f * (t & 511)};
    f32x2 b = {3.0f + 0.003f * (t & 255), 0.5f + 0.004f * (t & 127)};
    f32x2 c = {0.25f + 0.005f * (t & 63), 7.0f + 0.006f * (t & 31)};
    unsigned long long mine = 0;
    float keep = 0.0f;
    for (int i = 0; i < iters; ++i) {
        f32x2 d;
        float lo, hi;
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(lo) : "v"(a.x), "v"(b.y));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(hi) : "v"(a.y), "v"(b.x));
        mine += (__float_as_uint(d.x) != __float_as_uint(lo)) + (__float_as_uint(d.y) != __float_as_uint(hi));
        keep += d.x + d.y;
        a.x += 0.000001f; b.y += 0.000001f; a.y -= 0.000001f; b.x += 0.000002f;
    }
    if (mine) atomicAdd(bad, mine);
    if (keep == 123.456f) sink[t] = keep;
}

__global__ void __launch_bounds__(256) victim_6(unsigned long long* bad, float* sink, int iters) {   // mul op_sel_hi:[0,1]
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    f32x2 a = {1.0f + 0.001f * (t & 1023), 2.0f + 0.002f * (t & 511)};
    f32x2 b = {3.0f + 0.003f * (t & 255), 0.5f + 0.004f * (t & 127)};
    f32x2 c = {0.25f + 0.005f * (t & 63), 7.0f + 0.006f * (t & 31)};
    unsigned long long mine = 0;
    float keep = 0.0f;
    for (int i = 0; i < iters; ++i) {
        f32x2 d;
        float lo, hi;
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(lo) : "v"(a.x), "v"(b.x));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(hi) : "v"(a.x), "v"(b.y));
        mine += (__float_as_uint(d.x) != __float_as_uint(lo)) + (__float_as_uint(d.y) != __float_as_uint(hi));
        keep += d.x + d.y;
        a.x += 0.000001f; b.y += 0.000001f; a.y -= 0.000001f; b.x += 0.000002f;
    }
    if (mine) atomicAdd(bad, mine);
    if (keep == 123.456f) sink[t] = keep;
}

__global__ void __launch_bounds__(256) victim_7(unsigned long long* bad, float* sink, int iters) {   // mul op_sel_hi:[1,0]
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    f32x2 a = {1.0f + 0.001f * (t & 1023), 2.0f + 0.002f * (t & 511)};
    f32x2 b = {3.0f + 0.003f * (t & 255), 0.5f + 0.004f * (t & 127)};
    f32x2 c = {0.25f + 0.005f * (t & 63), 7.0f + 0.006f * (t & 31)};
    unsigned long long mine = 0;
    float keep = 0.0f;
    for (int i = 0; i < iters; ++i) {
        f32x2 d;
        float lo, hi;
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(lo) : "v"(a.x), "v"(b.x));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(hi) : "v"(a.y), "v"(b.x));
        mine += (__float_as_uint(d.x) != __float_as_uint(lo)) + (__float_as_uint(d.y) != __float_as_uint(hi));
        keep += d.x + d.y;
        a.x += 0.000001f; b.y += 0.000001f; a.y -= 0.000001f; b.x += 0.000002f;
    }
    if (mine) atomicAdd(bad, mine);
    if (keep == 123.456f) sink[t] = keep;
}

__global__ void __launch_bounds__(256) victim_8(unsigned long long* bad, float* sink, int iters) {   // mul op_sel_hi:[0,0]
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    f32x2 a = {1.0f + 0.001f * (t & 1023), 2.0f + 0.002f * (t & 511)};
    f32x2 b = {3.0f + 0.003f * (t & 255), 0.5f + 0.004f * (t & 127)};
    f32x2 c = {0.25f + 0.005f * (t & 63), 7.0f + 0.006f * (t & 31)};
    unsigned long long mine = 0;
    float keep = 0.0f;
    for (int i = 0; i < iters; ++i) {
        f32x2 d;
        float lo, hi;
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,0]" : "=v"(d) : "v"(a), "v"(b));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(lo) : "v"(a.x), "v"(b.x));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(hi) : "v"(a.x), "v"(b.x));
        mine += (__float_as_uint(d.x) != __float_as_uint(lo)) + (__float_as_uint(d.y) != __float_as_uint(hi));
        keep += d.x + d.y;
        a.x += 0.000001f; b.y += 0.000001f; a.y -= 0.000001f; b.x += 0.000002f;
    }
    if (mine) atomicAdd(bad, mine);
    if (keep == 123.456f) sink[t] = keep;
}

__global__ void __launch_bounds__(256) victim_9(unsigned long long* bad, float* sink, int iters) {   // add plain
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    f32x2 a = {1.0f + 0.001f * (t & 1023), 2.0f + 0.002f * (t & 511)};
    f32x2 b = {3.0f + 0.003f * (t & 255), 0.5f + 0.004f * (t & 127)};
    f32x2 c = {0.25f + 0.005f * (t & 63), 7.0f + 0.006f * (t & 31)};
    unsigned long long mine = 0;
    float keep = 0.0f;
    for (int i = 0; i < iters; ++i) {
        f32x2 d;
        float lo, hi;
        asm volatile("v_pk_add_f32 %0, %1, %2 " : "=v"(d) : "v"(a), "v"(b));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(lo) : "v"(a.x), "v"(b.x));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(hi) : "v"(a.y), "v"(b.y));
        mine += (__float_as_uint(d.x) != __float_as_uint(lo)) + (__float_as_uint(d.y) != __float_as_uint(hi));
        keep += d.x + d.y;
        a.x += 0.000001f; b.y += 0.000001f; a.y -= 0.000001f; b.x += 0.000002f;
    }
    if (mine) atomicAdd(bad, mine);
    if (keep == 123.456f) sink[t] = keep;
}

__global__ void __launch_bounds__(256) victim_10(unsigned long long* bad, float* sink, int iters) {   // add op_sel:[0,1]
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    f32x2 a = {1.0f + 0.001f * (t & 1023), 2.0f + 0.002f * (t & 511)};
    f32x2 b = {3.0f + 0.003f * (t & 255), 0.5f + 0.004f * (t & 127)};
    f32x2 c = {0.25f + 0.005f * (t & 63), 7.0f + 0.006f * (t & 31)};
    unsigned long long mine = 0;
    float keep = 0.0f;
    for (int i = 0; i < iters; ++i) {
        f32x2 d;
        float lo, hi;
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(a), "v"(b));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(lo) : "v"(a.x), "v"(b.y));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(hi) : "v"(a.y), "v"(b.y));
        mine += (__float_as_uint(d.x) != __float_as_uint(lo)) + (__float_as_uint(d.y) != __float_as_uint(hi));
        keep += d.x + d.y;
        a.x += 0.000001f; b.y += 0.000001f; a.y -= 0.000001f; b.x += 0.000002f;
    }
    if (mine) atomicAdd(bad, mine);
    if (keep == 123.456f) sink[t] = keep;
}

__global__ void __launch_bounds__(256) victim_11(unsigned long long* bad, float* sink, int iters) {   // add op_sel:[1,0] op_sel_hi:[0,0]
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    f32x2 a = {1.0f + 0.001f * (t & 1023), 2.0f + 0.002f * (t & 511)};
    f32x2 b = {3.0f + 0.003f * (t & 255), 0.5f + 0.004f * (t & 127)};
    f32x2 c = {0.25f + 0.005f * (t & 63), 7.0f + 0.006f * (t & 31)};
    unsigned long long mine = 0;
    float keep = 0.0f;
    for (int i = 0; i < iters; ++i) {
        f32x2 d;
        float lo, hi;
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,0]" : "=v"(d) : "v"(a), "v"(b));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(lo) : "v"(a.y), "v"(b.x));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(hi) : "v"(a.x), "v"(b.x));
        mine += (__float_as_uint(d.x) != __float_as_uint(lo)) + (__float_as_uint(d.y) != __float_as_uint(hi));
        keep += d.x + d.y;
        a.x += 0.000001f; b.y += 0.000001f; a.y -= 0.000001f; b.x += 0.000002f;
    }
    if (mine) atomicAdd(bad, mine);
    if (keep == 123.456f) sink[t] = keep;
}

__global__ void __launch_bounds__(256) victim_12(unsigned long long* bad, float* sink, int iters) {   // add op_sel:[1,0] op_sel_hi:[0,0] neg_lo:[1,0] neg_hi:[1,0]
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    f32x2 a = {1.0f + 0.001f * (t & 1023), 2.0f + 0.002f * (t & 511)};
    f32x2 b = {3.0f + 0.003f * (t & 255), 0.5f + 0.004f * (t & 127)};
    f32x2 c = {0.25f + 0.005f * (t & 63), 7.0f + 0.006f * (t & 31)};
    unsigned long long mine = 0;
    float keep = 0.0f;
    for (int i = 0; i < iters; ++i) {
        f32x2 d;
        float lo, hi;
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,0] neg_lo:[1,0] neg_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(lo) : "v"(-a.y), "v"(b.x));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(hi) : "v"(-a.x), "v"(b.x));
        mine += (__float_as_uint(d.x) != __float_as_uint(lo)) + (__float_as_uint(d.y) != __float_as_uint(hi));
        keep += d.x + d.y;
        a.x += 0.000001f; b.y += 0.000001f; a.y -= 0.000001f; b.x += 0.000002f;
    }
    if (mine) atomicAdd(bad, mine);
    if (keep == 123.456f) sink[t] = keep;
}

__global__ void __launch_bounds__(256) victim_13(unsigned long long* bad, float* sink, int iters) {   // add op_sel_hi:[1,0]
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    f32x2 a = {1.0f + 0.001f * (t & 1023), 2.0f + 0.002f * (t & 511)};
    f32x2 b = {3.0f + 0.003f * (t & 255), 0.5f + 0.004f * (t & 127)};
    f32x2 c = {0.25f + 0.005f * (t & 63), 7.0f + 0.006f * (t & 31)};
    unsigned long long mine = 0;
    float keep = 0.0f;
    for (int i = 0; i < iters; ++i) {
        f32x2 d;
        float lo, hi;
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(lo) : "v"(a.x), "v"(b.x));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(hi) : "v"(a.y), "v"(b.x));
        mine += (__float_as_uint(d.x) != __float_as_uint(lo)) + (__float_as_uint(d.y) != __float_as_uint(hi));
        keep += d.x + d.y;
        a.x += 0.000001f; b.y += 0.000001f; a.y -= 0.000001f; b.x += 0.000002f;
    }
    if (mine) atomicAdd(bad, mine);
    if (keep == 123.456f) sink[t] = keep;
}

__global__ void __launch_bounds__(256) victim_14(unsigned long long* bad, float* sink, int iters) {   // add op_sel_hi:[0,1]
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    f32x2 a = {1.0f + 0.001f * (t & 1023), 2.0f + 0.002f * (t & 511)};
    f32x2 b = {3.0f + 0.003f * (t & 255), 0.5f + 0.004f * (t & 127)};
    f32x2 c = {0.25f + 0.005f * (t & 63), 7.0f + 0.006f * (t & 31)};
    unsigned long long mine = 0;
    float keep = 0.0f;
    for (int i = 0; i < iters; ++i) {
        f32x2 d;
        float lo, hi;
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(lo) : "v"(a.x), "v"(b.x));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(hi) : "v"(a.x), "v"(b.y));
        mine += (__float_as_uint(d.x) != __float_as_uint(lo)) + (__float_as_uint(d.y) != __float_as_uint(hi));
        keep += d.x + d.y;
        a.x += 0.000001f; b.y += 0.000001f; a.y -= 0.000001f; b.x += 0.000002f;
    }
    if (mine) atomicAdd(bad, mine);
    if (keep == 123.456f) sink[t] = keep;
}

__global__ void __launch_bounds__(256) victim_15(unsigned long long* bad, float* sink, int iters) {   // fma plain
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    f32x2 a = {1.0f + 0.001f * (t & 1023), 2.0f + 0.002f * (t & 511)};
    f32x2 b = {3.0f + 0.003f * (t & 255), 0.5f + 0.004f * (t & 127)};
    f32x2 c = {0.25f + 0.005f * (t & 63), 7.0f + 0.006f * (t & 31)};
    unsigned long long mine = 0;
    float keep = 0.0f;
    for (int i = 0; i < iters; ++i) {
        f32x2 d;
        float lo, hi;
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 " : "=v"(d) : "v"(a), "v"(b), "v"(c));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(lo) : "v"(a.x), "v"(b.x), "v"(c.x));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(hi) : "v"(a.y), "v"(b.y), "v"(c.y));
        mine += (__float_as_uint(d.x) != __float_as_uint(lo)) + (__float_as_uint(d.y) != __float_as_uint(hi));
        keep += d.x + d.y;
        a.x += 0.000001f; b.y += 0.000001f; a.y -= 0.000001f; b.x += 0.000002f;
    }
    if (mine) atomicAdd(bad, mine);
    if (keep == 123.456f) sink[t] = keep;
}

__global__ void __launch_bounds__(256) victim_16(unsigned long long* bad, float* sink, int iters) {   // fma op_sel:[0,1,0]
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    f32x2 a = {1.0f + 0.001f * (t & 1023), 2.0f + 0.002f * (t & 511)};
    f32x2 b = {3.0f + 0.003f * (t & 255), 0.5f + 0.004f * (t & 127)};
    f32x2 c = {0.25f + 0.005f * (t & 63), 7.0f + 0.006f * (t & 31)};
    unsigned long long mine = 0;
    float keep = 0.0f;
    for (int i = 0; i < iters; ++i) {
        f32x2 d;
        float lo, hi;
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(lo) : "v"(a.x), "v"(b.y), "v"(c.x));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(hi) : "v"(a.y), "v"(b.y), "v"(c.y));
        mine += (__float_as_uint(d.x) != __float_as_uint(lo)) + (__float_as_uint(d.y) != __float_as_uint(hi));
        keep += d.x + d.y;
        a.x += 0.000001f; b.y += 0.000001f; a.y -= 0.000001f; b.x += 0.000002f;
    }
    if (mine) atomicAdd(bad, mine);
    if (keep == 123.456f) sink[t] = keep;
}

__global__ void __launch_bounds__(256) victim_17(unsigned long long* bad, float* sink, int iters) {   // fma op_sel:[1,0,0]
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    f32x2 a = {1.0f + 0.001f * (t & 1023), 2.0f + 0.002f * (t & 511)};
    f32x2 b = {3.0f + 0.003f * (t & 255), 0.5f + 0.004f * (t & 127)};
    f32x2 c = {0.25f + 0.005f * (t & 63), 7.0f + 0.006f * (t & 31)};
    unsigned long long mine = 0;
    float keep = 0.0f;
    for (int i = 0; i < iters; ++i) {
        f32x2 d;
        float lo, hi;
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(lo) : "v"(a.y), "v"(b.x), "v"(c.x));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(hi) : "v"(a.y), "v"(b.y), "v"(c.y));
        mine += (__float_as_uint(d.x) != __float_as_uint(lo)) + (__float_as_uint(d.y) != __float_as_uint(hi));
        keep += d.x + d.y;
        a.x += 0.000001f; b.y += 0.000001f; a.y -= 0.000001f; b.x += 0.000002f;
    }
    if (mine) atomicAdd(bad, mine);
    if (keep == 123.456f) sink[t] = keep;
}

__global__ void __launch_bounds__(256) victim_18(unsigned long long* bad, float* sink, int iters) {   // fma op_sel:[0,0,1]
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    f32x2 a = {1.0f + 0.001f * (t & 1023), 2.0f + 0.002f * (t & 511)};
    f32x2 b = {3.0f + 0.003f * (t & 255), 0.5f + 0.004f * (t & 127)};
    f32x2 c = {0.25f + 0.005f * (t & 63), 7.0f + 0.006f * (t & 31)};
    unsigned long long mine = 0;
    float keep = 0.0f;
    for (int i = 0; i < iters; ++i) {
        f32x2 d;
        float lo, hi;
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(lo) : "v"(a.x), "v"(b.x), "v"(c.y));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(hi) : "v"(a.y), "v"(b.y), "v"(c.y));
        mine += (__float_as_uint(d.x) != __float_as_uint(lo)) + (__float_as_uint(d.y) != __float_as_uint(hi));
        keep += d.x + d.y;
        a.x += 0.000001f; b.y += 0.000001f; a.y -= 0.000001f; b.x += 0.000002f;
    }
    if (mine) atomicAdd(bad, mine);
    if (keep == 123.456f) sink[t] = keep;
}

__global__ void __launch_bounds__(256) victim_19(unsigned long long* bad, float* sink, int iters) {   // fma op_sel_hi:[1,0,1]
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    f32x2 a = {1.0f + 0.001f * (t & 1023), 2.0f + 0.002f * (t & 511)};
    f32x2 b = {3.0f + 0.003f * (t & 255), 0.5f + 0.004f * (t & 127)};
    f32x2 c = {0.25f + 0.005f * (t & 63), 7.0f + 0.006f * (t & 31)};
    unsigned long long mine = 0;
    float keep = 0.0f;
    for (int i = 0; i < iters; ++i) {
        f32x2 d;
        float lo, hi;
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(lo) : "v"(a.x), "v"(b.x), "v"(c.x));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(hi) : "v"(a.y), "v"(b.x), "v"(c.y));
        mine += (__float_as_uint(d.x) != __float_as_uint(lo)) + (__float_as_uint(d.y) != __float_as_uint(hi));
        keep += d.x + d.y;
        a.x += 0.000001f; b.y += 0.000001f; a.y -= 0.000001f; b.x += 0.000002f;
    }
    if (mine) atomicAdd(bad, mine);
    if (keep == 123.456f) sink[t] = keep;
}

__global__ void __launch_bounds__(256) victim_20(unsigned long long* bad, float* sink, int iters) {   // fma op_sel_hi:[1,0,0]
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    f32x2 a = {1.0f + 0.001f * (t & 1023), 2.0f + 0.002f * (t & 511)};
    f32x2 b = {3.0f + 0.003f * (t & 255), 0.5f + 0.004f * (t & 127)};
    f32x2 c = {0.25f + 0.005f * (t & 63), 7.0f + 0.006f * (t & 31)};
    unsigned long long mine = 0;
    float keep = 0.0f;
    for (int i = 0; i < iters; ++i) {
        f32x2 d;
        float lo, hi;
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(lo) : "v"(a.x), "v"(b.x), "v"(c.x));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(hi) : "v"(a.y), "v"(b.x), "v"(c.x));
        mine += (__float_as_uint(d.x) != __float_as_uint(lo)) + (__float_as_uint(d.y) != __float_as_uint(hi));
        keep += d.x + d.y;
        a.x += 0.000001f; b.y += 0.000001f; a.y -= 0.000001f; b.x += 0.000002f;
    }
    if (mine) atomicAdd(bad, mine);
    if (keep == 123.456f) sink[t] = keep;
}

__global__ void __launch_bounds__(256) victim_21(unsigned long long* bad, float* sink, int iters) {   // fma op_sel_hi:[1,1,0]
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    f32x2 a = {1.0f + 0.001f * (t & 1023), 2.0f + 0.002f * (t & 511)};
    f32x2 b = {3.0f + 0.003f * (t & 255), 0.5f + 0.004f * (t & 127)};
    f32x2 c = {0.25f + 0.005f * (t & 63), 7.0f + 0.006f * (t & 31)};
    unsigned long long mine = 0;
    float keep = 0.0f;
    for (int i = 0; i < iters; ++i) {
        f32x2 d;
        float lo, hi;
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,1,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(lo) : "v"(a.x), "v"(b.x), "v"(c.x));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(hi) : "v"(a.y), "v"(b.y), "v"(c.x));
        mine += (__float_as_uint(d.x) != __float_as_uint(lo)) + (__float_as_uint(d.y) != __float_as_uint(hi));
        keep += d.x + d.y;
        a.x += 0.000001f; b.y += 0.000001f; a.y -= 0.000001f; b.x += 0.000002f;
    }
    if (mine) atomicAdd(bad, mine);
    if (keep == 123.456f) sink[t] = keep;
}

// KIND 0: MFMAs and a memory stream (the shape of the library's kernels); 1: MFMAs only; 2: the memory stream only; 3: fp32 VALU only;
// 4: packed fp32 VALU only
template <int KIND>
__global__ void __launch_bounds__(256, 2) aggressor(float* sink, const float* stream, size_t words, int iters) {
    f32x16 acc[4] = {};
    f16x8 x = {1, 2, 3, 4, 5, 6, 7, 8}, y = {8, 7, 6, 5, 4, 3, 2, 1};
    float s = 0.0f;
    f32x2 v = {1.0f, 2.0f}, u = {1.000001f, 0.999999f};
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0 || KIND == 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc[k], 0, 0, 0);
        }
        if (KIND == 0 || KIND == 2) s += stream[(t * 4 + (size_t)i * 262144) % words];
        if (KIND == 3) {
#pragma unroll
            for (int k = 0; k < 32; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v.x) : "v"(u.x), "v"(u.y));
        }
        if (KIND == 4) {
#pragma unroll
            for (int k = 0; k < 32; ++k) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(v) : "v"(u));
        }
    }
    float total = s + v.x + v.y;
    for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) total += acc[k][r];
    if (total == 123.456f) sink[t] = total;
}

// ONE kernel: waves 0-3 of every workgroup repeat  v_pk_mul_f32 op_sel:[0,1], waves 4-7 run MFMAs -- two waves per SIMD, as in the
// library's eight-wave kernels
__global__ void __launch_bounds__(512) siblings(unsigned long long* bad, float* sink, int iters) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (threadIdx.x >= 256) {
        f32x16 acc[4] = {};
        f16x8 x = {1, 2, 3, 4, 5, 6, 7, 8}, y = {8, 7, 6, 5, 4, 3, 2, 1};
        for (int i = 0; i < iters / 4; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc[k], 0, 0, 0);
        float total = 0.0f;
        for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) total += acc[k][r];
        if (total == 123.456f) sink[t] = total;
        return;
    }
    f32x2 a = {1.0f + 0.001f * (t & 1023), 2.0f + 0.002f * (t & 511)};
    f32x2 b = {3.0f + 0.003f * (t & 255), 0.5f + 0.004f * (t & 127)};
    unsigned long long mine = 0;
    float keep = 0.0f;
    for (int i = 0; i < iters; ++i) {
        f32x2 d;
        float lo, hi;
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(a), "v"(b));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(lo) : "v"(a.x), "v"(b.y));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(hi) : "v"(a.y), "v"(b.y));
        mine += (__float_as_uint(d.x) != __float_as_uint(lo)) + (__float_as_uint(d.y) != __float_as_uint(hi));
        keep += d.x + d.y;
        a.x += 0.000001f; b.y += 0.000001f;
    }
    if (mine) atomicAdd(bad, mine);
    if (keep == 123.456f) sink[t] = keep;
}


// v_pk_mov_b32: D.lo = op_sel[0] ? S0.hi : S0.lo;  D.hi = op_sel[1] ? S1.hi : S1.lo  (the library holds a dozen  op_sel:[1,0])
#define MOV_VICTIM(NAME, MODS, LO, HI)                                                                             \
    __global__ void __launch_bounds__(256) NAME(unsigned long long* bad, float* sink, int iters) {                \
        const int t = blockIdx.x * blockDim.x + threadIdx.x;                                                      \
        f32x2 a = {1.0f + 0.001f * (t & 1023), 2.0f + 0.002f * (t & 511)};                                         \
        f32x2 b = {3.0f + 0.003f * (t & 255), 0.5f + 0.004f * (t & 127)};                                          \
        unsigned long long mine = 0;                                                                              \
        float keep = 0.0f;                                                                                        \
        for (int i = 0; i < iters; ++i) {                                                                         \
            f32x2 d;                                                                                              \
            asm volatile("v_pk_mov_b32 %0, %1, %2 " MODS : "=v"(d) : "v"(a), "v"(b));                              \
            mine += (__float_as_uint(d.x) != __float_as_uint(LO)) + (__float_as_uint(d.y) != __float_as_uint(HI)); \
            keep += d.x + d.y;                                                                                    \
            a.x += 0.000001f; b.y += 0.000001f; a.y -= 0.000001f; b.x += 0.000002f;                               \
            asm volatile("" : "+v"(a), "+v"(b));                                                                  \
        }                                                                                                         \
        if (mine) atomicAdd(bad, mine);                                                                           \
        if (keep == 123.456f) sink[t] = keep;                                                                     \
    }
MOV_VICTIM(mov_plain, "", a.x, b.x)
MOV_VICTIM(mov_10, "op_sel:[1,0]", a.y, b.x)
MOV_VICTIM(mov_01, "op_sel:[0,1]", a.x, b.y)
MOV_VICTIM(mov_11, "op_sel:[1,1]", a.y, b.y)


// The other instructions of the library that select a register half: the SDWA half-word read of the fp16 -> fp32 conversions
// (6 287 sites), and the fp8 conversions of the s8 modes (word select on the source resp. the destination).
__global__ void __launch_bounds__(256) cvt_sdwa_word1(unsigned long long* bad, float* sink, int iters) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned src = 0x3c003800u + 0x00010001u * (unsigned)(t & 1023);     // two fp16 values near 1 and 0.5
    unsigned long long mine = 0;
    float keep = 0.0f;
    for (int i = 0; i < iters; ++i) {
        float got, want;
        unsigned hi;
        asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(got) : "v"(src));
        asm volatile("v_lshrrev_b32 %0, 16, %1" : "=v"(hi) : "v"(src));
        asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(want) : "v"(hi));
        mine += __float_as_uint(got) != __float_as_uint(want);
        keep += got;
        src += 0x00010001u;
        asm volatile("" : "+v"(src));
    }
    if (mine) atomicAdd(bad, mine);
    if (keep == 123.456f) sink[t] = keep;
}
__global__ void __launch_bounds__(256) cvt_f16_fp8_word1(unsigned long long* bad, float* sink, int iters) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned src = 0x38304048u + 0x01010101u * (unsigned)(t & 7);        // four e4m3 values
    const float scale = 1.0f;
    unsigned long long mine = 0;
    unsigned keep = 0;
    for (int i = 0; i < iters; ++i) {
        unsigned got, want, hi;
        asm volatile("v_cvt_scalef32_pk_f16_fp8 %0, %1, %2 op_sel:[1,0,0]" : "=v"(got) : "v"(src), "v"(scale));
        asm volatile("v_lshrrev_b32 %0, 16, %1" : "=v"(hi) : "v"(src));
        asm volatile("v_cvt_scalef32_pk_f16_fp8 %0, %1, %2" : "=v"(want) : "v"(hi), "v"(scale));
        mine += got != want;
        keep ^= got;
        src = (src + 0x01010101u) & 0x7f7f7f7fu;
        asm volatile("" : "+v"(src));
    }
    if (mine) atomicAdd(bad, mine);
    if (keep == 0x12345678u) sink[t] = 1.0f;
}
__global__ void __launch_bounds__(256) cvt_fp8_f16_word1(unsigned long long* bad, float* sink, int iters) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned src = 0x3c003800u + 0x00010001u * (unsigned)(t & 1023);     // two fp16 values
    const float scale = 1.0f;
    unsigned long long mine = 0;
    unsigned keep = 0;
    for (int i = 0; i < iters; ++i) {
        unsigned got = 0x0000beefu, low = 0;
        asm volatile("v_cvt_scalef32_pk_fp8_f16 %0, %1, %2 op_sel:[0,0,1]" : "+v"(got) : "v"(src), "v"(scale));
        asm volatile("v_cvt_scalef32_pk_fp8_f16 %0, %1, %2" : "+v"(low) : "v"(src), "v"(scale));
        const unsigned want = 0x0000beefu | (low << 16);
        mine += got != want;
        keep ^= got;
        src += 0x00010001u;
        asm volatile("" : "+v"(src));
    }
    if (mine) atomicAdd(bad, mine);
    if (keep == 0x12345678u) sink[t] = 1.0f;
}

struct Form { void (*kernel)(unsigned long long*, float*, int); const char* name; };
static const Form kForms[] = {
    {victim_0, "v_pk_mul_f32 "},
    {victim_1, "v_pk_mul_f32 op_sel:[0,1]"},
    {victim_2, "v_pk_mul_f32 op_sel:[1,0]"},
    {victim_3, "v_pk_mul_f32 op_sel:[1,1]"},
    {victim_4, "v_pk_mul_f32 op_sel:[1,0] op_sel_hi:[0,1]"},
    {victim_5, "v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0]"},
    {victim_6, "v_pk_mul_f32 op_sel_hi:[0,1]"},
    {victim_7, "v_pk_mul_f32 op_sel_hi:[1,0]"},
    {victim_8, "v_pk_mul_f32 op_sel_hi:[0,0]"},
    {victim_9, "v_pk_add_f32 "},
    {victim_10, "v_pk_add_f32 op_sel:[0,1]"},
    {victim_11, "v_pk_add_f32 op_sel:[1,0] op_sel_hi:[0,0]"},
    {victim_12, "v_pk_add_f32 op_sel:[1,0] op_sel_hi:[0,0] neg_lo:[1,0] neg_hi:[1,0]"},
    {victim_13, "v_pk_add_f32 op_sel_hi:[1,0]"},
    {victim_14, "v_pk_add_f32 op_sel_hi:[0,1]"},
    {victim_15, "v_pk_fma_f32 "},
    {victim_16, "v_pk_fma_f32 op_sel:[0,1,0]"},
    {victim_17, "v_pk_fma_f32 op_sel:[1,0,0]"},
    {victim_18, "v_pk_fma_f32 op_sel:[0,0,1]"},
    {victim_19, "v_pk_fma_f32 op_sel_hi:[1,0,1]"},
    {victim_20, "v_pk_fma_f32 op_sel_hi:[1,0,0]"},
    {victim_21, "v_pk_fma_f32 op_sel_hi:[1,1,0]"},
    {mov_plain, "v_pk_mov_b32"},
    {mov_10, "v_pk_mov_b32 op_sel:[1,0]"},
    {mov_01, "v_pk_mov_b32 op_sel:[0,1]"},
    {mov_11, "v_pk_mov_b32 op_sel:[1,1]"},
    {cvt_sdwa_word1, "v_cvt_f32_f16_sdwa src0_sel:WORD_1"},
    {cvt_f16_fp8_word1, "v_cvt_scalef32_pk_f16_fp8 op_sel:[1,0,0]"},
    {cvt_fp8_f16_word1, "v_cvt_scalef32_pk_fp8_f16 op_sel:[0,0,1]"},
};

int main() {
    unsigned long long* bad;
    float *sink, *stream;
    const size_t words = 64u << 20;
    CHECK(hipMalloc(&bad, 8)); CHECK(hipMalloc(&sink, 1u << 26)); CHECK(hipMalloc(&stream, words * 4));
    CHECK(hipMemset(stream, 0, words * 4));
    hipStream_t sa, sb;
    CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    for (int beside = 0; beside < 2; ++beside)
        for (const Form& form : kForms) {
            CHECK(hipMemset(bad, 0, 8));
            CHECK(hipDeviceSynchronize());
            for (int round = 0; round < 10; ++round) {
                if (beside) hipLaunchKernelGGL(aggressor<0>, dim3(1024), dim3(256), 0, sb, sink, stream, words, 2000);
                for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL(form.kernel, dim3(128), dim3(256), 0, sa, bad, sink, 2000);
                CHECK(hipDeviceSynchronize());
            }
            unsigned long long n = 0;
            CHECK(hipMemcpy(&n, bad, 8, hipMemcpyDeviceToHost));
            printf("%s  %-72s %10llu mismatches in %.1e instructions\n", beside ? "beside MFMA" : "alone      ", form.name, n,
                   10.0 * 10 * 128 * 256 * 2000);
        }
    // what the neighbour has to be doing: the one bad multiply form beside each kind of aggressor, and beside MFMA waves of its own kernel
    const char* kinds[] = {"MFMA + memory stream", "MFMA only", "memory stream only", "fp32 VALU only", "packed fp32 VALU only"};
    for (int kind = 0; kind < 5; ++kind) {
        CHECK(hipMemset(bad, 0, 8));
        CHECK(hipDeviceSynchronize());
        for (int round = 0; round < 10; ++round) {
            if (kind == 0) hipLaunchKernelGGL(aggressor<0>, dim3(1024), dim3(256), 0, sb, sink, stream, words, 2000);
            if (kind == 1) hipLaunchKernelGGL(aggressor<1>, dim3(1024), dim3(256), 0, sb, sink, stream, words, 2000);
            if (kind == 2) hipLaunchKernelGGL(aggressor<2>, dim3(1024), dim3(256), 0, sb, sink, stream, words, 2000);
            if (kind == 3) hipLaunchKernelGGL(aggressor<3>, dim3(1024), dim3(256), 0, sb, sink, stream, words, 2000);
            if (kind == 4) hipLaunchKernelGGL(aggressor<4>, dim3(1024), dim3(256), 0, sb, sink, stream, words, 2000);
            for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL(kForms[1].kernel, dim3(128), dim3(256), 0, sa, bad, sink, 2000);
            CHECK(hipDeviceSynchronize());
        }
        unsigned long long n = 0;
        CHECK(hipMemcpy(&n, bad, 8, hipMemcpyDeviceToHost));
        printf("neighbour kernel: %-24s %s %10llu mismatches in %.1e instructions\n", kinds[kind], kForms[1].name, n, 10.0 * 10 * 128 * 256 * 2000);
    }
    CHECK(hipMemset(bad, 0, 8));
    for (int round = 0; round < 10; ++round) {
        hipLaunchKernelGGL(siblings, dim3(1024), dim3(512), 0, sa, bad, sink, 2000);
        CHECK(hipDeviceSynchronize());
    }
    unsigned long long n = 0;
    CHECK(hipMemcpy(&n, bad, 8, hipMemcpyDeviceToHost));
    printf("sibling waves of ONE kernel (4 multiply + 4 MFMA waves per workgroup): %llu mismatches in %.1e instructions\n", n,
           10.0 * 1024 * 256 * 2000);
    return 0;
}
