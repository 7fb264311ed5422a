// K3 -- fused sample-position + positional-encoding + NeRF-MLP forward on the fp32 matrix cores.
//
// Replaces, for one MLP, render_rays' point computation (src/models/SimpleNeRF01.py:139-142, :203-206),
// run_network/batchify (:363-428), PositionalEncoder.encode (:556-557) and MLP.forward (:626-715).
//
// Structure (see mlp_layout.h for the operand algebra):
//   * a 256-thread workgroup = 4 wavefronts, one per SIMD; each wavefront owns 32 consecutive samples
//     (lane&31 = sample, lane>>5 = which half of the k pair the lane feeds);
//   * every layer is computed transposed, Y^T = W . X^T, with v_mfma_f32_32x32x2_f32; the 32x32 accumulator tiles
//     of one layer ARE the B operands of the next, so activations stay in registers from the encoded inputs to the
//     colour head (the reference round-trips ~12 KB per sample per MLP through memory);
//   * the A operands (weights) stream L2 -> LDS in 16-k-step slabs (WT*4 KiB) by LDS-DMA (global_load_lds_dwordx4),
//     double-buffered, one barrier per slab, shared by the 4 waves; each lane reads them back with one
//     ds_read_b128 per 4 MFMAs (linear, conflict-free);
//   * the skip connection and the [feature | encoding | view encoding] concatenations are extra K-segments
//     accumulated into the same tiles -- no data movement;
//   * density / colour heads (1-4 output rows) are VALU dot products over the register-resident activations.
//
// Numerics: v_mfma_f32_32x32x2_f32 is an exact fp32 FMA chain (no reduced-precision path), sin/cos use an exact
// power-of-two range reduction in fp64 turns, exp/divide/sqrt are correctly rounded library calls: results match the
// reference's fp32 CPU path to ~1e-6 relative (tests/test_gpu_*.py).
//
// Bound: MFMA (fp32, 256 FLOP/clk/CU).  Algorithmic work 2*MAC of the Linear layers: 1 186 816 FLOP per sample for
// the 8x256 view-dependent MLP.  HBM traffic per sample: 4 B depth read + 16 B written; weights (2.4 MB) stay in L2.
#include "mlp_plan.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

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
};

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
    // Make the current slab (U tiles wide) readable and start fetching the one after it.
    template <int U>
    __device__ __forceinline__ const float* acquire() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my share of the current slab has landed
        __syncthreads();                                  // everyone's has, and everyone left the other buffer
        const float* ready = lds + parity * kBufFloats;
        fetch(cur + U * 1024, lds + (parity ^ 1) * kBufFloats);
        cur += U * 1024;
        parity ^= 1;
        return ready;
    }
};

// acc[u] += W_segment[u-th 32 rows] . B, B = `b` (one register per k-step), NSLAB slabs of 16 k-steps.
template <int U, int NSLAB, int WT, int NB>
__device__ __forceinline__ void gemm_segment(f32x16 (&acc)[U], const float (&b)[NB], SlabStream<WT>& st) {
    static_assert(NB >= NSLAB * 16, "B operand array too short");
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
            }
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
// sin and cos of 2*pi*turns.  The frequencies are exact powers of two, so `turns` = x * 2^k / (2 pi) is formed once
// per coordinate in fp64 and the reduction to [-1/8, 1/8] turns is exact; only the final polynomial is fp32.
__device__ __forceinline__ void sincos_turns(double turns, float& s, float& c) {
    const double f = turns - rint(turns);   // [-1/2, 1/2]
    const double q = rint(4.0 * f);         // quadrant, -2 .. 2
    const double g = f - 0.25 * q;          // [-1/8, 1/8]
    const float th = (float)(g * 6.283185307179586476925);
    const float t2 = th * th;
    float sp = fmaf(t2, 2.7557319e-6f, -1.9841270e-4f);
    sp = fmaf(sp, t2, 8.3333333e-3f);
    sp = fmaf(sp, t2, -1.6666667e-1f);
    sp = fmaf(th * t2, sp, th);
    float cp = fmaf(t2, -2.7557319e-7f, 2.4801587e-5f);
    cp = fmaf(cp, t2, -1.3888889e-3f);
    cp = fmaf(cp, t2, 4.1666667e-2f);
    cp = fmaf(cp, t2, -0.5f);
    cp = fmaf(cp, t2, 1.0f);
    const int qi = ((int)q) & 3;
    const float s0 = (qi & 1) ? cp : sp;
    const float c0 = (qi & 1) ? sp : cp;
    s = (qi >= 2) ? -s0 : s0;
    c = (qi == 1 || qi == 2) ? -c0 : c0;
}

// Fill the PE operand registers of this lane half (layout: mlp_layout.h pe_feature()).
template <int PAIRS, int NREG>
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
        sincos_turns(turns, pe[2 * m], pe[2 * m + 1]);
    }
    pe[PAIRS] = half ? x[2] : x[0];
    pe[PAIRS + 1] = half ? 0.0f : x[1];
#pragma unroll
    for (int n = PAIRS + 2; n < NREG; ++n) pe[n] = 0.0f;
}

// ------------------------------------------------------------------------------------------------
// the kernel
// ------------------------------------------------------------------------------------------------
template <int WT, int VT, bool VIEWDEP, bool SIGMA_PE>
__global__ void __launch_bounds__(256, 1) mlp_forward_kernel(MlpArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5;

    SlabStream<WT> st;
    st.start(a.packed, lds, lane, wave);

    // ---- inputs of this lane's sample -------------------------------------------------------------
    const long long first = ((long long)blockIdx.x * 4 + wave) * 32 + (lane & 31);
    const bool live = first < a.total;
    const long long g = live ? first : a.total - 1;
    const long long ray = g / a.samples;
    const float z = a.depths[g];
    float x[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) x[k] = a.origins[ray * 3 + k] + a.dirs[ray * 3 + k] * z;  // mul, then add (:140-142)

    float pe[snerf::kPointsKSteps];
    encode<snerf::kPointsPairs, snerf::kPointsKSteps>(x, half, pe);
    float pev[snerf::kViewsKSteps];
    if (VIEWDEP) {
        float v[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) v[k] = a.view_dirs[ray * 3 + k];
        encode<snerf::kViewsPairs, snerf::kViewsKSteps>(v, half, pev);
    }

    // ---- trunk ------------------------------------------------------------------------------------
    const float* bias = a.packed + a.bias_offset;
    f32x16 acc[WT];
    float h[WT * 16];
    load_bias<WT>(acc, bias, half);
    gemm_segment<WT, 2, WT>(acc, pe, st);
    to_operand<WT, true>(acc, h);
#pragma unroll 1
    for (int l = 1; l < a.depth; ++l) {
        load_bias<WT>(acc, bias + (long long)l * a.width, half);
        if (l == 5) gemm_segment<WT, 2, WT>(acc, pe, st);  // skip connection: [encoding | h] (:662-663)
        gemm_segment<WT, WT, WT>(acc, h, st);
        to_operand<WT, true>(acc, h);
    }

    // ---- density (and view-independent colour) head -------------------------------------------------
    const float* wout = a.packed + a.pts_out_w;
    const float* bout = a.packed + a.pts_out_b;
    float sigma = head_dot<WT * 16>(h, wout, half) + bout[0];
    if (a.noise) sigma += a.noise[g];
    sigma = fmaxf(sigma, 0.0f);
    float rgb[3];
    if (!VIEWDEP) {
#pragma unroll
        for (int c = 0; c < 3; ++c) rgb[c] = sigmoidf(head_dot<WT * 16>(h, wout + (c + 1) * WT * 32, half) + bout[c + 1]);
    } else {
        // feature = feature_linear(h), no activation (:683)
        load_bias<WT>(acc, a.packed + a.feature_bias, half);
        gemm_segment<WT, WT, WT>(acc, h, st);
        to_operand<WT, false>(acc, h);
        // views layer over [feature | rest of the point encoding (points-aug only) | view encoding] (:633, :695-699)
        f32x16 accv[VT];
        load_bias<VT>(accv, a.packed + a.views_bias, half);
        gemm_segment<VT, WT, WT>(accv, h, st);
        if (SIGMA_PE) gemm_segment<VT, 2, WT>(accv, pe, st);
        gemm_segment<VT, 1, WT>(accv, pev, st);
        float hv[VT * 16];
        to_operand<VT, true>(accv, hv);
        const float* wv = a.packed + a.views_out_w;
        const float* bv = a.packed + a.views_out_b;
#pragma unroll
        for (int c = 0; c < 3; ++c) rgb[c] = sigmoidf(head_dot<VT * 16>(hv, wv + c * VT * 32, half) + bv[c]);
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the run-ahead prefetch must land before the LDS is released
    if (live && half == 0) {
        a.sigma[first] = sigma;
        a.rgb[first * 3 + 0] = rgb[0];
        a.rgb[first * 3 + 1] = rgb[1];
        a.rgb[first * 3 + 2] = rgb[2];
    }
}

template <int WT, int VT, bool VIEWDEP, bool SIGMA_PE>
int launch(const MlpArgs& a, hipStream_t stream) {
    const long long blocks = (a.total + 127) / 128;
    if (blocks > 0x7fffffffLL) return snerf::fail(SNERF_E_UNSUPPORTED, "mlp_forward: too many samples in one call");
    const size_t lds_bytes = 2 * sizeof(float) * SlabStream<WT>::kBufFloats;
    auto kernel = mlp_forward_kernel<WT, VT, VIEWDEP, SIGMA_PE>;
    static bool configured = false;  // raising the dynamic-LDS cap is idempotent; racing threads only repeat it
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds_bytes);
        if (e != hipSuccess) return snerf::fail(SNERF_E_HIP, "mlp_forward: hipFuncSetAttribute: %s", hipGetErrorString(e));
        configured = true;
    }
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(256), lds_bytes, stream, a);
    return snerf::check_launch("mlp_forward");
}

}  // namespace

extern "C" int snerf_mlp_forward(const snerf_mlp_desc* desc, const float* packed, const float* origins,
                                 const float* dirs, const float* view_dirs, const float* depths, long long num_rays,
                                 int num_samples, const float* sigma_noise, float* sigma, float* rgb, int precision,
                                 snerf_stream_t stream) {
    snerf::MlpPlan plan;
    const int st = snerf::build_plan(desc, &plan);
    if (st != SNERF_OK) return st;
    SNERF_REQUIRE(packed && origins && dirs && depths && sigma && rgb, "mlp_forward: NULL pointer");
    SNERF_REQUIRE(!plan.view_dependent || view_dirs, "mlp_forward: this MLP needs view_dirs");
    SNERF_REQUIRE(num_rays >= 0 && num_samples >= 1, "mlp_forward: bad sizes n=%lld S=%d", num_rays, num_samples);
    if (precision != SNERF_PRECISION_FP32)
        return snerf::fail(SNERF_E_UNSUPPORTED, "mlp_forward: precision %d not built", precision);
    if (num_rays == 0) return SNERF_OK;
    MlpArgs a;
    a.packed = packed; a.origins = origins; a.dirs = dirs; a.view_dirs = view_dirs; a.depths = depths;
    a.noise = sigma_noise; a.sigma = sigma; a.rgb = rgb;
    a.total = num_rays * num_samples; a.samples = num_samples; a.depth = plan.depth; a.width = plan.width;
    a.bias_offset = plan.bias_offset; a.feature_bias = plan.feature_bias(); a.views_bias = plan.views_bias();
    a.pts_out_w = plan.pts_out_w(); a.pts_out_b = plan.pts_out_b();
    a.views_out_w = plan.views_out_w(); a.views_out_b = plan.views_out_b();
    hipStream_t s = (hipStream_t)stream;
    const int key = plan.wt * 100 + plan.vt * 10 + (plan.sigma_pe ? 1 : 0);
    switch (key) {
        case 840: return launch<8, 4, true, false>(a, s);
        case 841: return launch<8, 4, true, true>(a, s);
        case 800: return launch<8, 4, false, false>(a, s);
        case 420: return launch<4, 2, true, false>(a, s);
        case 421: return launch<4, 2, true, true>(a, s);
        case 400: return launch<4, 2, false, false>(a, s);
        default:
            return snerf::fail(SNERF_E_UNSUPPORTED, "mlp_forward: width %d / views width %d combination not built "
                               "(256/128, 128/64, or 256|128 without a views layer)", plan.width, plan.views_width);
    }
}
