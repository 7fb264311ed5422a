// Device-side building blocks shared by the fused MLP forward (mlp_forward.hip) and backward (mlp_backward.hip)
// kernels: the LDS slab stream, the register-resident transposed GEMM segment, operand conversion, positional
// encoding.  See mlp_layout.h for the operand algebra.
#pragma once
#include "mlp_plan.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct MlpArgs {
    const float* packed;
    const float* origins;
    const float* dirs;
    const float* view_dirs;
    const float* depths;
    const float* noise;
    float* sigma;
    float* rgb;
    long long total;  // rays * samples
    int samples;
    int depth;
    int width;
    long long bias_offset, feature_bias, views_bias, pts_out_w, pts_out_b, views_out_w, views_out_b;
    // training only: saved-activation tiles, [wave block][act_rows][32 samples] (mlp_plan.h)
    float* acts;
    int act_rows, act_pev, act_h1, act_feature, act_hv, act_mask;
    int const_floats;       // biases + head weights: packed[bias_offset, bias_offset + const_floats), staged in LDS
    // predict_visibility (fp32 kernels): per-sample visibility of the primary view direction and of `num_other` secondary ones
    float* visibility;        // (total) or NULL
    const float* view_dirs2;  // (total, num_other, 3) or NULL
    float* visibility2;       // (total, num_other)
    int num_other;
    int* range_flag;          // fp16 modes: pinned host word, OR-ed with kRangeActivation when an operand left the fp16 range
    const int* weight_range;  // fp16 modes: the packed buffer's weight-range word (snerf_mlp_pack), forwarded to range_flag
};

// Arguments of the backward chain kernels (fp32: mlp_backward.hip, f16x3: mlp_backward_f16.hip)
struct ChainArgs {
    const float* packed;
    const float* acts;      // saved activation tiles
    const float* sigma;     // (M)   post-ReLU density from the forward (mask of the density ReLU)
    const float* rgb;       // (M,3) post-sigmoid colour from the forward
    const float* d_sigma;   // (M)
    const float* d_rgb;     // (M,3)
    float* grads;           // dY tiles
    long long total;
    int depth, width;
    long long dgrad_offset, pts_out_w, views_out_w;
    int act_rows, act_h1, act_hv, act_mask;
    int grad_rows, grad_feature, grad_yv, grad_head;
    unsigned* dy_max;  // f16x3: per dY region (index = first tile row / 32) the max |dY| over the call, as float bits
};

// Power-of-two factor that brings a positive magnitude into [8, 16) (1 for zero / non-finite input).  The f16x3
// backward chain renormalises every sample's gradient vector with it before each hi/lo split: an fp16 pair only keeps
// ~22 bits for values above 2^-2 (the lo part bottoms out at the 2^-24 subnormal step), loss gradients are 1e-5 .. 1e-10
// and shrink or grow from layer to layer, and the chain is linear per sample -- so the factor is exact to apply and
// exact to undo when the sample's dY values are stored.
__host__ __device__ inline float renorm_factor(float max_abs) {
    if (!(max_abs > 0.0f) || !(max_abs < 3.0e38f)) return 1.0f;
    int e;
    frexpf(max_abs, &e);             // max_abs = m * 2^e, m in [0.5, 1)
    int shift = 4 - e;
    if (shift > 120) shift = 120;
    if (shift < -120) shift = -120;
    return ldexpf(1.0f, shift);
}

// ---- [feature][32-sample] tile stores/loads of one wave block (rows in natural feature order) -------------------
// accumulator-ordered registers: register 16u+r of lane (j, half) is feature 32u + (r&3) + 8(r>>2) + 4*half
template <int N>
__device__ __forceinline__ void store_acc_tile(const float (&h)[N], float* __restrict__ tile, int lane) {
    const int j = lane & 31, half = lane >> 5;
#pragma unroll
    for (int n = 0; n < N; ++n) {
        const int f = 32 * (n >> 4) + (n & 3) + 8 * ((n & 15) >> 2) + 4 * half;
        __builtin_nontemporal_store(h[n], tile + f * 32 + j);   // saved tiles are written once and read back GBs later
    }
}
template <int N>
__device__ __forceinline__ void load_acc_tile(float (&h)[N], const float* __restrict__ tile, int lane) {
    const int j = lane & 31, half = lane >> 5;
#pragma unroll
    for (int n = 0; n < N; ++n) {
        const int f = 32 * (n >> 4) + (n & 3) + 8 * ((n & 15) >> 2) + 4 * half;
        h[n] = tile[f * 32 + j];
    }
}
// v if bit `bit` of `word` is set, else +0: the bit becomes 0 / -1 by a signed one-bit field extract and masks the value's
// bits -- two VALU instructions, against three for test + compare + select (128 values per layer and lane in the chains).
__device__ __forceinline__ float keep_if_bit(float v, unsigned word, int bit) {
    return __uint_as_float(__float_as_uint(v) & (unsigned)__builtin_amdgcn_sbfe((int)word, bit, 1));
}
// The same, kept at two instructions: left alone, the optimiser rewrites `v & sext(bit)` into test + compare + select (three,
// plus the two wait states gfx950 wants between the compare's scalar result and the select).  The empty asm only makes the
// extracted field opaque; it contains no instruction.
__device__ __forceinline__ float keep_if_bit_2op(float v, unsigned word, int bit) {
    int k = __builtin_amdgcn_sbfe((int)word, bit, 1);
    asm("" : "+v"(k));
    return __uint_as_float(__float_as_uint(v) & (unsigned)k);
}
// ReLU sign bits (see MlpPlan::act_mask): bit r of a tile's 16-bit field = accumulator register r of this lane > 0.
// 1.5 VALU per value and no scalar register in between: the sign bit of (0 - v) IS the predicate v > 0 (0 - (+-0) = +0,
// positive denormals stay denormal), the subtraction is one packed instruction per two values, and v_alignbit_b32 shifts
// the bit into the word ({m, t} >> 31 = m << 1 | sign(t); registers taken from 15 down to 0).  The C expression
// `m |= (h[r] > 0) << r` compiles to compare + select + shift-or, three per value plus the wait states gfx950 needs between a
// VALU write of a scalar pair and the VALU that reads it as a mask.  (An inline-asm compare + add-with-carry pair -- two per
// value -- was tried first and is NOT safe: the compiler inserts neither those wait states nor the ones between an MFMA's
// write and a VALU read for instructions inside an asm; the 16-bit training gradients differed run to run.)  Exactly the
// predicate of the reference's ReLU derivative for every number; a NaN pre-activation may give either bit (the fp16 modes
// fail the call on one, "Range" in simplenerf_hip.h; in fp32 the outputs are NaN as well).
__device__ __forceinline__ unsigned positive_bits(unsigned m, float a, float b) {   // m <- m << 2 | [a > 0] << 1 | [b > 0]
    const f32x2 t = f32x2{0.0f, 0.0f} - f32x2{a, b};
    m = __builtin_amdgcn_alignbit(m, __float_as_uint(t[0]), 31);
    return __builtin_amdgcn_alignbit(m, __float_as_uint(t[1]), 31);
}
__device__ __forceinline__ unsigned relu_bits(const float* __restrict__ h) {
    unsigned m = 0;
#pragma unroll
    for (int r = 15; r >= 1; r -= 2) m = positive_bits(m, h[r], h[r - 1]);
    return m;
}
// registers h[16u + r] of U (even) tiles starting at tile index t0 (even) -> mask words of this wave block
template <int N>
__device__ __forceinline__ void store_relu_masks(const float (&h)[N], unsigned* __restrict__ masks, int t0, int lane) {
#pragma unroll
    for (int u = 0; u < N / 16; u += 2)
        __builtin_nontemporal_store(relu_bits(h + 16 * u) | (relu_bits(h + 16 * (u + 1)) << 16), masks + ((t0 + u) >> 1) * 64 + lane);
}
// positional-encoding registers -> rows in the reference's encoding order (pads skipped)
template <int PAIRS, int NREG>
__device__ __forceinline__ void store_pe_tile(const float (&pe)[NREG], float* __restrict__ tile, int lane) {
    const int j = lane & 31, half = lane >> 5;
#pragma unroll
    for (int n = 0; n < NREG; ++n) {
        const int e0 = snerf::pe_feature(n, 0, PAIRS, 16), e1 = snerf::pe_feature(n, 1, PAIRS, 16);
        const int e = half ? e1 : e0;
        if (e >= 0) __builtin_nontemporal_store(pe[n], tile + e * 32 + j);
    }
}

// ------------------------------------------------------------------------------------------------
// weight slab stream: L2 -> LDS by LDS-DMA, double buffered
// ------------------------------------------------------------------------------------------------
template <int WT>
struct SlabStream {
    static constexpr int kBufFloats = WT * 1024;  // 16 k-steps x WT tiles x 64 lanes x 4
    const float* cur;                             // global address of the slab about to be consumed
    float* lds;
    int parity;
    int lane, wave;

    __device__ __forceinline__ void fetch(const float* src, float* dst) const {
#pragma unroll
        for (int i = 0; i < WT; ++i) {
            const int chunk = i * 4 + wave;  // 1 KiB per wave-instruction
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + chunk * 256 + lane * 4),
                                             (__attribute__((address_space(3))) void*)(dst + chunk * 256), 16, 0, 0);
        }
    }
    __device__ __forceinline__ void start(const float* first, float* lds_base, int lane_, int wave_) {
        cur = first; lds = lds_base; parity = 0; lane = lane_; wave = wave_;
        fetch(cur, lds);
    }
    // Make the current slab (U tiles wide) readable and OPEN the request for the one after it.  The WT DMA instructions
    // of that request are issued one at a time from inside the MFMA loop (fetch_piece<i>), because a 1-KiB LDS-DMA costs
    // ~60-100 issue cycles and a burst of WT of them in front of the slab's first MFMA idles the matrix pipe.
    template <int U>
    __device__ __forceinline__ const float* acquire() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my share of the current slab has landed
        __syncthreads();                                  // everyone's has, and everyone left the other buffer
        const float* ready = lds + parity * kBufFloats;
        cur += U * 1024;
        parity ^= 1;
        return ready;
    }
    // piece I (0 .. WT-1) of the slab that follows the one being consumed; `cur`/`parity` already point at it
    template <int I>
    __device__ __forceinline__ void fetch_piece() const {
        const int chunk = I * 4 + wave;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(cur + chunk * 256 + lane * 4),
                                         (__attribute__((address_space(3))) void*)(lds + parity * kBufFloats + chunk * 256),
                                         16, 0, 0);
    }
};

template <int I, int WT>
__device__ __forceinline__ void fetch_piece_if(const SlabStream<WT>& st) {
    if constexpr (I >= 0 && I < WT) st.template fetch_piece<I>();
}
// slot = g*U+u of the unrolled loop (a constant after unrolling): issue piece slot/kEvery when slot % kEvery == 0
template <int U, int WT, int EVERY>
__device__ __forceinline__ void constexpr_fetch(const SlabStream<WT>& st, int slot) {
#pragma unroll
    for (int i = 0; i < WT; ++i)
        if (slot == i * EVERY) {
#ifdef SNERF_ABL_F32_NODMA       // ablation probe: no weight stream (the slabs hold whatever the first request left)
            continue;
#endif
            const int chunk = i * 4 + st.wave;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(st.cur + chunk * 256 + st.lane * 4),
                                             (__attribute__((address_space(3))) void*)(st.lds + st.parity * SlabStream<WT>::kBufFloats + chunk * 256),
                                             16, 0, 0);
        }
}
template <int WT, int DONE>
__device__ __forceinline__ void tail_fetch(const SlabStream<WT>& st) {
#pragma unroll
    for (int i = DONE; i < WT; ++i) {
        const int chunk = i * 4 + st.wave;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(st.cur + chunk * 256 + st.lane * 4),
                                         (__attribute__((address_space(3))) void*)(st.lds + st.parity * SlabStream<WT>::kBufFloats + chunk * 256),
                                         16, 0, 0);
    }
}

// acc[u] += W_segment[u-th 32 rows] . B, B = `b` (one register per k-step), NSLAB slabs of 16 k-steps.
template <int U, int NSLAB, int WT, int NB>
__device__ __forceinline__ void gemm_segment(f32x16 (&acc)[U], const float (&b)[NB], SlabStream<WT>& st) {
    static_assert(NB >= NSLAB * 16, "B operand array too short");
    constexpr int kSlots = 4 * U;                      // (k-group, tile) iterations per slab
    constexpr int kEvery = kSlots >= WT ? kSlots / WT : 1;
#pragma unroll
    for (int sl = 0; sl < NSLAB; ++sl) {
        const float* slab = st.template acquire<U>() + st.lane * 4;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(slab + (g * U + u) * 256);
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], b[sl * 16 + g * 4 + q], acc[u], 0, 0, 0);
                // spread the next slab's WT DMA pieces over this slab's iterations (all indices are compile-time)
                constexpr_fetch<U, WT, kEvery>(st, g * U + u);
            }
        }
        if (kSlots < WT) tail_fetch<WT, kSlots>(st);
    }
}

// acc[u] += W_slab[u-th 32 rows] . B over the 16 k-steps of ONE slab that is already in LDS (`slab` = its address for this
// lane): gemm_segment's inner loops without the stream bookkeeping, so that the same slab can be applied to several
// operands (the views head of a predict_visibility MLP is evaluated once per view direction).
template <int U, int NB>
__device__ __forceinline__ void gemm_resident_slab(f32x16 (&acc)[U], const float (&b)[NB], const float* slab) {
    static_assert(NB >= 16, "B operand array too short");
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(slab + (g * U + u) * 256);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], b[g * 4 + q], acc[u], 0, 0, 0);
        }
    }
}

template <int U>
__device__ __forceinline__ void load_bias(f32x16 (&acc)[U], const float* __restrict__ bias, int half) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(bias + 32 * u + 8 * g + 4 * half);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[u][4 * g + q] = v[q];
        }
    }
}

template <int U, bool RELU>
__device__ __forceinline__ void to_operand(const f32x16 (&acc)[U], float (&h)[U * 16]) {
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) h[16 * u + r] = RELU ? fmaxf(acc[u][r], 0.0f) : acc[u][r];
}

// sum_f w[f] * x[f] over all features of one sample (both lane halves), features in accumulator order.
template <int N>
__device__ __forceinline__ float head_dot(const float (&h)[N], const float* __restrict__ w, int half) {
    float s = 0.0f;
#pragma unroll
    for (int g = 0; g < N / 4; ++g) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(w + 8 * g + 4 * half);
#pragma unroll
        for (int q = 0; q < 4; ++q) s = fmaf(v[q], h[4 * g + q], s);
    }
    return s + __shfl_xor(s, 32, 64);
}

__device__ __forceinline__ float sigmoidf(float x) { return __fdiv_rn(1.0f, 1.0f + expf(-x)); }

// ------------------------------------------------------------------------------------------------
// positional encoding
// ------------------------------------------------------------------------------------------------
// Two copies of `v` in one register pair that the compiler has to treat as two unrelated values.
// Why: a packed fp32 instruction whose LOW result reads the HIGH register of its second source (`op_sel:[0,1]` on v_pk_mul_f32 /
// v_pk_add_f32, `op_sel:[0,1,0]` on v_pk_fma_f32 -- how the compiler broadcasts a scalar that happens to live in the odd register
// of a pair) returns a wrong low half for a quarter wave about once in 10^4..10^5 executions on MI355X WHILE waves of another
// kernel issue MFMAs on the same SIMD; never when the kernel runs alone, and no other operand selection does it
// (tools/probes/pk_opsel_hazard.hip, profiles/r05_pk_opsel_hazard.txt).  Found as parameter gradients that differed in the last
// bits between repetitions once the levels of a small pass ran side by side (render.hip).  With the scalar spelled as an opaque
// pair the packed instruction reads lo from lo and hi from hi -- no operand selection at all.  simplenerf_amd/build.py scans
// the built library for the form and refuses it.
__device__ __forceinline__ f32x2 opaque_pair(float v) {
    f32x2 p = {v, v};
    asm volatile("" : "+v"(p));
    return p;
}

// sin and cos of 2*pi*turns.  The frequencies are exact powers of two, so `turns` = x * 2^k / (2 pi) is formed once
// per coordinate in fp64 and the reduction to [-1/8, 1/8] turns is exact; only the final polynomial is fp32.
// PAIRED: the sine and the cosine polynomial each read their own copy of th^2 (opaque_pair): where the compiler packs the two
// polynomials' steps into v_pk_fma_f32 it then needs no broadcast of th^2 (mlp_forward_m16.hip: the packing halves the
// encoding's VALU count, and 30 of its 60 packed steps had taken the form described above).  Same arithmetic either way.
template <bool PAIRED = false>
__device__ __forceinline__ void sincos_turns(double turns, float& s, float& c) {
    const double f = turns - rint(turns);   // [-1/2, 1/2]
    const double q = rint(4.0 * f);         // quadrant, -2 .. 2
    const double g = f - 0.25 * q;          // [-1/8, 1/8]
    const float th = (float)(g * 6.283185307179586476925);
    float t2 = th * th, t2c = t2;
    if constexpr (PAIRED) {
        const f32x2 both = opaque_pair(t2);
        t2 = both.x; t2c = both.y;
    }
    float sp = fmaf(t2, 2.7557319e-6f, -1.9841270e-4f);
    sp = fmaf(sp, t2, 8.3333333e-3f);
    sp = fmaf(sp, t2, -1.6666667e-1f);
    sp = fmaf(th * t2, sp, th);
    float cp = fmaf(t2c, -2.7557319e-7f, 2.4801587e-5f);
    cp = fmaf(cp, t2c, -1.3888889e-3f);
    cp = fmaf(cp, t2c, 4.1666667e-2f);
    cp = fmaf(cp, t2c, -0.5f);
    cp = fmaf(cp, t2c, 1.0f);
    const int qi = ((int)q) & 3;
    const float s0 = (qi & 1) ? cp : sp;
    const float c0 = (qi & 1) ? sp : cp;
    s = (qi >= 2) ? -s0 : s0;
    c = (qi == 1 || qi == 2) ? -c0 : c0;
}

// Fill the PE operand registers of this lane half (layout: mlp_layout.h pe_feature()).
template <int PAIRS, int NREG, bool PAIRED = false>
__device__ __forceinline__ void encode(const float (&x)[3], int half, float (&pe)[NREG]) {
    constexpr double kInvTwoPi = 0.15915494309189533576888;
    const double r0 = (double)x[0] * kInvTwoPi, r1 = (double)x[1] * kInvTwoPi, r2 = (double)x[2] * kInvTwoPi;
#pragma unroll
    for (int m = 0; m < PAIRS / 2; ++m) {
        constexpr int dummy = 0;
        (void)dummy;
        const int c0 = 2 * m, c1 = 2 * m + 1;           // pair index for half 0 / half 1
        const int d0 = c0 % 3, d1 = c1 % 3;
        const double a0 = d0 == 0 ? r0 : (d0 == 1 ? r1 : r2);
        const double a1 = d1 == 0 ? r0 : (d1 == 1 ? r1 : r2);
        const double f0 = (double)(1 << (c0 / 3)), f1 = (double)(1 << (c1 / 3));
        const double turns = half ? a1 * f1 : a0 * f0;
        sincos_turns<PAIRED>(turns, pe[2 * m], pe[2 * m + 1]);
    }
    pe[PAIRS] = half ? x[2] : x[0];
    pe[PAIRS + 1] = half ? 0.0f : x[1];
#pragma unroll
    for (int n = PAIRS + 2; n < NREG; ++n) pe[n] = 0.0f;
}

