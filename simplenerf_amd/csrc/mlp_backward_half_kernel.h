// K7b (split-precision variant) -- the backward "dgrad" chain of the fused MLP on the fp16 matrix cores.  Kernel template +
// launcher, included by mlp_backward_f16.hip (f16x3 and the fp16 single-product format) and mlp_backward_bf16.hip (the bf16
// single-product format, SNERF_PRECISION_BF16: bf16 operand conversion, v_mfma_f32_32x32x16_bf16, the compact W^T stream).
//
// Same computation as mlp_backward_chain_kernel (mlp_backward.hip): dX^T = W^T . dY^T layer by layer in reverse, ReLU
// masks from the forward's sign-bit words, every dY written as a [feature][32-sample] fp32 tile for the weight-gradient kernel.
// Same arithmetic as mlp_forward_f16.hip: each operand is an fp16 hi/lo pair and each product is three MFMAs
// (hi.hi + hi.lo + lo.hi) into an fp32 accumulator; W^T is pre-split at pack time (MlpPlan::half_dgrad_stages) and
// streamed through the 3-slot LDS ring, one 32-row IN-feature tile (all of its k-steps over the OUT features) per unit.
// P = 1 (SNERF_PRECISION_F16, the 16-bit training mode): one MFMA per product, bf16 dY pieces, and -- round 3 -- a trunk
// whose layer epilogues ride behind the next tile's MFMAs (two waves per SIMD, eight-wave workgroups, five-slot ring); see
// the comment at `if constexpr (P == 1)` below.  Bound: instruction issue + LDS fragment reads (45 ns per MFMA
// and SIMD against a 26-ns "MFMA + one LDS fragment" floor); algorithmic HBM bytes 4.6 KB per sample (dY stores).
#include <algorithm>
#include <type_traits>

// timing ablations of this kernel alone (tools/probes/build_variant.py; wrong results): no weight DMA / no workgroup barrier
#ifdef SNERF_ABL_CHAIN_NODMA
#define SNERF_ABL_NODMA
#endif
#ifdef SNERF_ABL_CHAIN_NOBARRIER
#define SNERF_ABL_NOBARRIER
#endif
#pragma once
#include "clock_stamp.h"
#include "mlp_device_f16.h"
#include "mlp_plan.h"

namespace {

SNERF_STAMP_DEFINE(chain_f16)

struct HalfChainArgs {
    ChainArgs c;
    long long half_dgrad_offset;
    int slot_floats;   // LDS ring slot size
};

template <int U>
__device__ __forceinline__ void zero_tiles(f32x16 (&acc)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[u][r] = 0.0f;
}

__device__ __forceinline__ void store_tile_rows_scaled(const f32x16& acc, float* __restrict__ rows, int lane, float k) {
    const int j = lane & 31, half = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) __builtin_nontemporal_store(acc[r] * k, rows + ((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + j);
}

// Rescale this sample's gradient vector (U tiles of this lane + the partner lane half) to a maximum in [8, 16).
// max over the wave of a non-negative value -> atomicMax into the region's word (non-negative floats order like their bits)
__device__ __forceinline__ void publish_max(unsigned* word, float v, int lane) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    if (lane == 0 && v > 0.0f) atomicMax(word, __float_as_uint(v));
}

// max(v of this lane, v of the lane 32 away) without the LDS crossbar: v_permlane32_swap exchanges the upper half of one
// register with the lower half of another (two copies of v in, [lo, lo] and [hi, hi] out)
__device__ __forceinline__ float lane_halves_max(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
// publish_max for a non-negative v with the wave maximum formed by DPP row operations (shifts within the 16-lane rows, then
// the row broadcasts; lanes outside a step's mask see 0, the identity here): a dozen VALU instructions where the six
// dependent __shfl_xor steps each cost an LDS-crossbar round trip.  The maximum lands in lane 63.
__device__ __forceinline__ void publish_max_dpp(unsigned* word, float v, int lane) {
#define SNERF_DPP_MAX(ctrl, rows) v = fmaxf(v, __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), ctrl, rows, 0xf, true)))
    SNERF_DPP_MAX(0x111, 0xf);   // row_shr:1
    SNERF_DPP_MAX(0x112, 0xf);   // row_shr:2
    SNERF_DPP_MAX(0x114, 0xf);   // row_shr:4
    SNERF_DPP_MAX(0x118, 0xf);   // row_shr:8   -> lane 15 of every row holds the row's maximum
    SNERF_DPP_MAX(0x142, 0xa);   // row_bcast:15 into rows 1 and 3
    SNERF_DPP_MAX(0x143, 0xc);   // row_bcast:31 into rows 2 and 3
#undef SNERF_DPP_MAX
    if (lane == 63 && v > 0.0f) atomicMax(word, __float_as_uint(v));
}

template <int U>
__device__ __forceinline__ float tiles_max(const f32x16 (&acc)[U]) {
    float m = 0.0f;
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) m = fmaxf(m, fabsf(acc[u][r]));
    return fmaxf(m, __shfl_xor(m, 32, 64));
}

// `region_max` (may be null): receives max |dY| of the region just stored, in UNSCALED units, for the f16x3 weight-gradient
// kernel, which has to pick one power-of-two scale per dY region (its contraction runs over the samples).
template <int U>
__device__ __forceinline__ void renormalise(f32x16 (&acc)[U], float& gscale, float& gback, unsigned* region_max, int lane) {
    const float m = tiles_max<U>(acc);
    if (region_max) publish_max(region_max, m * gback, lane);
    float f = renorm_factor(m);
    // keep the cumulative factor (and its reciprocal) finite: a sample whose gradient underflows to ~1e-38 simply
    // stops being rescaled -- its contribution is nil anyway
    if (!(gscale * f < 1.0e30f) || !(gscale * f > 1.0e-30f)) f = 1.0f;
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[u][r] *= f;
    gscale *= f;
    gback = 1.0f / gscale;
}

// dY tile(s) of this wave block -> the weight-gradient kernel's input.  P = 3: fp32 [feature][32-sample] rows, true scale.
// P = 1 (SNERF_PRECISION_F16): bf16 operand pieces (store_pieces layout; bf16 because its exponent range needs no
// per-region scale at this point -- the region maxima are only complete when the whole chain kernel has finished).
template <int P, int U>
__device__ __forceinline__ void store_dy(const f32x16 (&acc)[U], float* __restrict__ grads, int row0, int lane, float k) {
    if constexpr (P == 3) {
#pragma unroll
        for (int u = 0; u < U; ++u) store_tile_rows_scaled(acc[u], grads + (row0 + 32 * u) * 32, lane, k);
    } else {
        __bf16* rows = reinterpret_cast<__bf16*>(grads) + row0 * 32;
        const int slot = 2 * (lane & 31) + (lane >> 5);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#ifdef SNERF_PROBE_HALF_DY     // traffic ablation (tools/probes/build_variant.py; WRONG results): the bytes an fp8 dY tile would take
                if (s == 1) continue;
#endif
                bf16x8 v;
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    const f32x2 f = f32x2{acc[u][8 * s + e], acc[u][8 * s + e + 1]} * k;
                    const bf16x2 b = __builtin_convertvector(f, bf16x2);
                    v[e] = b[0]; v[e + 1] = b[1];
                }
                __builtin_nontemporal_store(v, reinterpret_cast<bf16x8*>(rows + (2 * u + s) * 512 + slot * 8));   // read once, by the weight-gradient kernel
            }
    }
}
template <int P>
constexpr int dy_stores(int tiles) { return P == 3 ? 16 * tiles : 2 * tiles; }

// f(integral_constant<I>), ..., f(integral_constant<N-1>): one inlined copy of the body per index
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// Workgroup shape of the chain.  Single-product chain of a view-dependent MLP (the 16-bit training mode's main kernel): EIGHT
// waves -- two per SIMD, 256 registers each -- sharing ONE weight stream (half the L2 -> LDS traffic and DMA instructions of
// two 4-wave workgroups per CU) through a five-slot ring (see UnitStreamT: the counted waits then tolerate three units of
// store latency).  Everything else: four waves, one per SIMD, three slots.
constexpr int chain_waves(int products, bool viewdep, int depth) { return products == 1 && (viewdep || depth > 0) ? 8 : 4; }
constexpr int chain_ring(int products, bool viewdep, int depth) { return products == 1 && (viewdep || depth > 0) ? 5 : kUnitBuffers; }

// DEPTH > 0: compile-time trunk depth, layer loop fully unrolled -- the unit schedule (k-steps of each unit and its two
// successors, DMA pieces, counted vmcnt immediates incl. the dY stores, ring slots) folds to constants instead of ~200
// scalar instructions per unit of 16 MFMAs (see mlp_forward_f16.hip).
template <int WT, int VT, bool VIEWDEP, int P, int DEPTH, bool BF = false>
// (P = 3 and the runtime-depth view-independent P = 1 instance: one 4-wave workgroup per CU, 512 registers per wave; the
// 8-wave single-product instances run at the 256-register budget -- round 2's serial epilogue spilled ~120 registers there,
// the pipelined trunk below none)
__global__ void __launch_bounds__(chain_waves(P, VIEWDEP, DEPTH) * 64, chain_waves(P, VIEWDEP, DEPTH) == 8 ? 2 : 1) mlp_backward_chain_f16x3_kernel(HalfChainArgs args) {
    static_assert(!BF || P == 1, "bf16 operands: single-product kernels only");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NW = chain_waves(P, VIEWDEP, DEPTH);
    constexpr int RING = chain_ring(P, VIEWDEP, DEPTH);
    const ChainArgs& a = args.c;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5;
    SNERF_STAMP_BEGIN();
    constexpr int HK = WT * 2;   // k-steps over a full-width dY
    constexpr int VK = VT * 2;   // k-steps over the views layer's dY

    // unit idx -> k-steps: [views^T: WT units of VK][feature^T: WT of HK] (view-dependent MLPs), then (depth-1) x WT of HK
    const int depth = DEPTH > 0 ? DEPTH : a.depth;
    const int head_units = VIEWDEP ? 2 * WT : 0;
    const int total_units = head_units + (depth - 1) * WT;
    auto ks_of = [&](int idx) {
        if (idx >= total_units) return 0;
        if (VIEWDEP && idx < WT) return VK;
        return HK;
    };
    UnitStreamT<P, NW, RING, BF ? 256 : 512> st;
    st.start(a.packed + args.half_dgrad_offset, lds, ks_of(0), ks_of(1), lane, wave, args.slot_floats);
#pragma unroll
    for (int u = 2; u < RING - 1; ++u) st.start_more(u, ks_of(u));
    int unit_idx = 0;
    auto next_unit = [&]() {
        const float* p = st.acquire(ks_of(unit_idx + 1), ks_of(unit_idx + RING - 1));
        ++unit_idx;
        return p + lane * 4;
    };
    const long long block = (long long)blockIdx.x * NW + wave;
    const long long first = block * 32 + (lane & 31);
    const bool live = first < a.total;
    // this workgroup's copy of the region-maximum table (64 copies of 128 words, see region_max in mlp_backward.hip)
    unsigned* const dy_max = a.dy_max ? a.dy_max + (blockIdx.x & 63) * 128 : nullptr;
    // (P = 1: both buffers hold 16-bit rows -- a.act_rows / a.grad_rows count rows of 64 bytes; `grads` is then only
    // ever used through store_dy, with row numbers)
    const unsigned* masks = P == 3
        ? reinterpret_cast<const unsigned*>(a.acts + (block * a.act_rows + a.act_mask) * 32)
        : reinterpret_cast<const unsigned*>(reinterpret_cast<const _Float16*>(a.acts) + (block * a.act_rows + a.act_mask) * 32);
    float* grads = P == 3 ? a.grads + block * a.grad_rows * 32
                          : reinterpret_cast<float*>(reinterpret_cast<_Float16*>(a.grads) + block * a.grad_rows * 32);

    // ---- head gradients (pre-activation) -------------------------------------------------------------------------
    float dhead[4];
    dhead[0] = live && a.sigma[first] > 0.0f ? a.d_sigma[first] : 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float col = live ? a.rgb[first * 3 + c] : 0.0f;
        dhead[c + 1] = live ? a.d_rgb[first * 3 + c] * (col * (1.0f - col)) : 0.0f;
    }
    if (P == 3) {
        if (half == 0) {
#pragma unroll
            for (int c = 0; c < 4; ++c) grads[(a.grad_head + c) * 32 + (lane & 31)] = dhead[c];
        }
    } else {
        // first piece of the head tile: features 0..3 = the four head gradients, the rest zero
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (__bf16)((half == 0 && e < 4) ? dhead[e & 3] : 0.0f);
        *reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(grads) + a.grad_head * 32 + (2 * (lane & 31) + half) * 8) = v;
    }
    // Per-sample power-of-two scaling (see renorm_factor): `gscale` is the factor currently applied to this sample's
    // gradients inside the chain, `gback` = 1/gscale is applied whenever one of its dY tiles is stored.
    const float head_max = fmaxf(fmaxf(fabsf(dhead[0]), fabsf(dhead[1])), fmaxf(fabsf(dhead[2]), fabsf(dhead[3])));
    if (dy_max) publish_max(dy_max + a.grad_head / 32, head_max, lane);   // region scale of the head-weight products
    float gscale = renorm_factor(head_max);
    if (!(gscale < 0x1p99f)) gscale = 0x1p99f;   // (a power of two, like every factor that follows: the 16-bit trunk scales by exponent)
    float gback = 1.0f / gscale;
    const float dsig_raw = dhead[0];  // re-enters below, after the views/feature products have been renormalised
#pragma unroll
    for (int c = 0; c < 4; ++c) dhead[c] *= gscale;

    f32x16 acc[WT];
    f16x8 xh[HK], xl[HK];
    if (VIEWDEP) {
        // d hv = W_rgb^T dpre masked by the views-layer ReLU -> stored, split into the operand of the first product
        const float* wv = a.packed + a.views_out_w;
        f16x8 vh[VK], vl[VK];
        f32x16 dyvs[VT];
#pragma unroll
        for (int u = 0; u < VT; ++u) {
            f32x16& dyv = dyvs[u];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(wv + 0 * VT * 32 + 32 * u + 8 * g + 4 * half);
                const f32x4 w1 = *reinterpret_cast<const f32x4*>(wv + 1 * VT * 32 + 32 * u + 8 * g + 4 * half);
                const f32x4 w2 = *reinterpret_cast<const f32x4*>(wv + 2 * VT * 32 + 32 * u + 8 * g + 4 * half);
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    dyv[4 * g + q] = fmaf(w2[q], dhead[3], fmaf(w1[q], dhead[2], w0[q] * dhead[1]));
            }
        }
        apply_relu_masks<VT>(dyvs, masks, depth * WT, lane);
        store_dy<P, VT>(dyvs, grads, a.grad_yv, lane, gback);
        st.note_vmem(dy_stores<P>(VT));
        renormalise<VT>(dyvs, gscale, gback, dy_max ? dy_max + a.grad_yv / 32 : nullptr, lane);
#pragma unroll
        for (int u = 0; u < VT; ++u) {
            if constexpr (P == 1) convert_tile<false, BF>(dyvs[u], vh[2 * u], vh[2 * u + 1]);     // (= split_tile's hi halves)
            else split_tile<false>(dyvs[u], vh[2 * u], vl[2 * u], vh[2 * u + 1], vl[2 * u + 1]);
        }
        // d feature = Wv[:, :width]^T dYv
#pragma unroll
        for (int u = 0; u < WT; ++u) {
            const float* unit = next_unit();
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[u][r] = 0.0f;
            seg_product<P, VK, BF>(acc[u], unit, VK, vh, vl, st);
            const f32x16(&one)[1] = reinterpret_cast<const f32x16(&)[1]>(acc[u]);
            store_dy<P, 1>(one, grads, a.grad_feature + 32 * u, lane, gback);
            st.note_vmem(dy_stores<P>(1));
        }
        renormalise<WT>(acc, gscale, gback, dy_max ? dy_max + a.grad_feature / 32 : nullptr, lane);
#pragma unroll
        for (int u = 0; u < WT; ++u) {
            if constexpr (P == 1) convert_tile<false, BF>(acc[u], xh[2 * u], xh[2 * u + 1]);
            else split_tile<false>(acc[u], xh[2 * u], xl[2 * u], xh[2 * u + 1], xl[2 * u + 1]);
        }
        // d h_depth = W_feature^T dfeature
#pragma unroll
        for (int u = 0; u < WT; ++u) {
            const float* unit = next_unit();
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[u][r] = 0.0f;
            seg_product<P, HK, BF>(acc[u], unit, HK, xh, xl, st);
        }
    } else {
        zero_tiles<WT>(acc);
    }
    // + density head (and, without a views layer, the colour rows of pts_output_linear): d h += W_out^T dhead
    {
        const float* wo = a.packed + a.pts_out_w;
#pragma unroll
        for (int g = 0; g < WT * 4; ++g) {
            // One tile's weight loads at a time: the pointer of tile t is made to depend on a finished value of tile t-1.
            // Left free, the compiler issued the four-row variant's 128 loads (512 registers) in front of the loop and the
            // 256-register instance spilled 270 of them -- 135 KB of scratch traffic per wave block, as much as the kernel's
            // dY stores (PMC: 8.8 KB of HBM traffic per sample against the main chain's 4.8).
            if (!VIEWDEP && g > 0 && (g & 3) == 0) asm volatile("" : "+s"(wo) : "v"(acc[(g >> 2) - 1][15]));
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(wo + 8 * g + 4 * half);
            f32x4 w1 = {0, 0, 0, 0}, w2 = {0, 0, 0, 0}, w3 = {0, 0, 0, 0};
            if (!VIEWDEP) {
                w1 = *reinterpret_cast<const f32x4*>(wo + 1 * WT * 32 + 8 * g + 4 * half);
                w2 = *reinterpret_cast<const f32x4*>(wo + 2 * WT * 32 + 8 * g + 4 * half);
                w3 = *reinterpret_cast<const f32x4*>(wo + 3 * WT * 32 + 8 * g + 4 * half);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v = w0[q] * (VIEWDEP ? dsig_raw * gscale : dhead[0]);  // at the scale `acc` currently carries
                if (!VIEWDEP) v = fmaf(w3[q], dhead[3], fmaf(w2[q], dhead[2], fmaf(w1[q], dhead[1], v)));
                acc[g >> 2][4 * (g & 3) + q] += v;
            }
        }
    }
    // ---- trunk, last layer first: dY_l = dh_{l+1} . [h_{l+1} > 0];  dh_l = W_l[:, h-columns]^T dY_l -------------
    if constexpr (P == 1) {
        // Single-product chain (round 3): one wave per SIMD, so VALU work placed BETWEEN the tiles' MFMA runs executes with
        // the matrix pipe idle -- the layer epilogue (mask, maximum, scale back + bf16 store, renormalise, fp16 operands: 6.5
        // VALU per value, 830 per layer) used to do exactly that and the pipe was 25 % busy (PMC).  Here every piece of it
        // that the data flow allows rides BEHIND an MFMA of the next tile (seg_mfma1_side):
        //   * tile u's mask + maximum + scaled bf16 store run during the product of tile u+1 (one register pair per two
        //     k-steps);
        //   * the operand fragments of the next layer's products (k-step t <- registers 8(t&1).. of tile t>>1, times the
        //     sample's renormalisation factor) are converted during the product of that layer's tile 0, two k-steps ahead
        //     of the MFMA that consumes them.
        // What is left between two layers' matrix work: the last tile's epilogue, the sample maximum (lane halves exchanged
        // with v_permlane32_swap, the region maximum reduced with DPP row operations instead of six LDS-crossbar shuffles) and
        // the first two fragments.  The arithmetic -- every value, every rounding -- is the one of the serial formulation.
        __bf16* const rows16 = reinterpret_cast<__bf16*>(grads);
        const int slot8 = (2 * (lane & 31) + half) * 8;
        float mrun = 0.0f;       // this lane's max |dY| over the tiles of the layer being finished, at the scale `acc` carries
        // exponents of the two power-of-two factors (gback = 2^gback_exp: gscale is a product of renorm_factor() values)
        auto exponent_of = [](float power_of_two) { return (int)((__float_as_uint(power_of_two) >> 23) & 0xffu) - 127; };
        int gback_exp = exponent_of(gback), fnext_exp = 0;
        bf16x8 stage;
        auto epilogue_pair = [&](int t, int pr, int row0, const unsigned (&words)[WT / 2]) __attribute__((always_inline)) {
#ifdef SNERF_ABL_CHAIN_NOEPI     // timing ablation (tools/probes/build_variant.py): no epilogue work at all -- wrong results
            return;
#endif
            const int r = 2 * pr, b0 = 16 * (t & 1) + r;
            const float x = keep_if_bit_2op(acc[t][r], words[t >> 1], b0);
            const float y = keep_if_bit_2op(acc[t][r + 1], words[t >> 1], b0 + 1);
            acc[t][r] = x; acc[t][r + 1] = y;
            // (the NaN-propagating maximum: v_maximum3_f32 with |x| modifiers, no quieting moves in front of it)
            mrun = __builtin_elementwise_maximum(mrun, __builtin_elementwise_maximum(__builtin_fabsf(x), __builtin_fabsf(y)));
            // (the two power-of-two scalings of this path are v_ldexp_f32, not multiplies: the compiler packs a pair of fp32
            // multiplies into one v_pk_mul_f32, and packed fp32 arithmetic is the one VALU class that does NOT issue in the
            // shadow of an MFMA -- r03_mfma_valu_shadow.txt; 571 -> 534 us per fine-size call, gradients bit-identical)
            const bf16x2 q = __builtin_convertvector(f32x2{__builtin_ldexpf(x, gback_exp), __builtin_ldexpf(y, gback_exp)}, bf16x2);
            stage[r & 7] = q[0]; stage[(r & 7) + 1] = q[1];
#ifdef SNERF_ABL_CHAIN_NOSTORE   // timing ablation: the epilogue's arithmetic, but nothing written
            if (pr == 7 && t == 0 && stage[0] == (__bf16)123.0f)
#else
            if ((pr & 3) == 3
#ifdef SNERF_PROBE_HALF_DY
                && pr == 3
#endif
                )   // eight values staged: one 16-byte store (read once, by the weight-gradient kernel)
#endif
            {
                __builtin_nontemporal_store(stage, reinterpret_cast<bf16x8*>(rows16 + (long long)row0 * 32 + (2 * t + (pr >> 2)) * 512 + slot8));
                st.note_vmem(1);
            }
        };
        auto convert_fragment = [&](int t) __attribute__((always_inline)) {
            const f32x16& src = acc[t >> 1];
            const int o = 8 * (t & 1);
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                const f16x2 h = pack_pair<BF, false>(f32x2{__builtin_ldexpf(src[o + j], fnext_exp), __builtin_ldexpf(src[o + j + 1], fnext_exp)});
                xh[t][j] = h[0]; xh[t][j + 1] = h[1];
            }
        };
        // mask words: the layer being finished, and the one below it (requested one whole layer of products ahead: HBM latency
        // is longer than one tile's product)
        unsigned words_now[WT / 2], words_next[WT / 2];
        load_relu_words<WT>(words_now, masks, (depth - 1) * WT, lane);
        if (depth > 1) load_relu_words<WT>(words_next, masks, (depth - 2) * WT, lane);
#pragma unroll
        for (int t = 0; t < WT; ++t)
#pragma unroll
            for (int pr = 0; pr < 8; ++pr) epilogue_pair(t, pr, (depth - 1) * a.width, words_now);
        auto phase = [&](int l) __attribute__((always_inline)) {   // products with W_l^T; epilogues of layer l - 1
            const float msample = lane_halves_max(mrun);
            if (dy_max) publish_max_dpp(dy_max + (l * a.width) / 32, msample * gback, lane);
            float fnext = renorm_factor(msample);
            if (!(gscale * fnext < 1.0e30f) || !(gscale * fnext > 1.0e-30f)) fnext = 1.0f;
            gscale *= fnext;
            gback = 1.0f / gscale;
            fnext_exp = exponent_of(fnext);
            gback_exp = exponent_of(gback);
            mrun = 0.0f;
#pragma unroll
            for (int i = 0; i < WT / 2; ++i) words_now[i] = words_next[i];
            if (l > 1) load_relu_words<WT>(words_next, masks, (l - 2) * WT, lane);
            convert_fragment(0);
            convert_fragment(1);
            const int row0 = (l - 1) * a.width;
#pragma unroll
            for (int u = 0; u < WT; ++u) {
                const float* unit = next_unit();
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[u][r] = 0.0f;
                if (u == 0) {
                    seg_mfma1_side<HK, BF>(acc[0], unit, xh, st, [&](int ks) __attribute__((always_inline)) {
                        if (ks + 2 < HK) convert_fragment(ks + 2);
                    });
                } else {
                    seg_mfma1_side<HK, BF>(acc[u], unit, xh, st, [&](int ks) __attribute__((always_inline)) {
#pragma unroll
                        for (int pr = ks * 8 / HK; pr < (ks + 1) * 8 / HK; ++pr) epilogue_pair(u - 1, pr, row0, words_now);
                    });
                }
            }
#pragma unroll
            for (int pr = 0; pr < 8; ++pr) epilogue_pair(WT - 1, pr, row0, words_now);
        };
        if constexpr (DEPTH > 0) {
            static_for<0, DEPTH - 1>([&](auto step) __attribute__((always_inline)) { phase(DEPTH - 1 - decltype(step)::value); });
        } else {
#pragma unroll 1
            for (int l = depth - 1; l >= 1; --l) phase(l);
        }
        if (dy_max) publish_max_dpp(dy_max, mrun * gback, lane);   // layer 0's region
    } else {
    unsigned relu_words[WT / 2];   // sign bits of the layer whose dY is formed next; requested one layer ahead
    load_relu_words<WT>(relu_words, masks, (depth - 1) * WT, lane);
    auto trunk_layer = [&](int l) __attribute__((always_inline)) {
        mask_with_words<WT>(acc, relu_words);
        if (l > 0) load_relu_words<WT>(relu_words, masks, (l - 1) * WT, lane);   // in flight during this layer's products
        store_dy<P, WT>(acc, grads, l * a.width, lane, gback);
        st.note_vmem(dy_stores<P>(WT));
        unsigned* region = dy_max ? dy_max + (l * a.width) / 32 : nullptr;
        if (l == 0) {
            if (region) publish_max(region, tiles_max<WT>(acc) * gback, lane);
            return;
        }
        renormalise<WT>(acc, gscale, gback, region, lane);
#pragma unroll
        for (int u = 0; u < WT; ++u) split_tile<false>(acc[u], xh[2 * u], xl[2 * u], xh[2 * u + 1], xl[2 * u + 1]);
#pragma unroll
        for (int u = 0; u < WT; ++u) {
            const float* unit = next_unit();
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[u][r] = 0.0f;
            seg_product<P, HK, BF>(acc[u], unit, HK, xh, xl, st);
        }
    };
    if constexpr (DEPTH > 0) {
        static_for<0, DEPTH>([&](auto step) __attribute__((always_inline)) { trunk_layer(DEPTH - 1 - decltype(step)::value); });
    } else {
#pragma unroll 1
        for (int l = depth - 1; l >= 0; --l) trunk_layer(l);
    }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SNERF_STAMP_END(chain_f16);
}

template <int WT, int VT, bool VIEWDEP, int P, int DEPTH = 0, bool BF = false>
int launch_chain_half(const HalfChainArgs& args, hipStream_t stream) {
    constexpr int NW = chain_waves(P, VIEWDEP, DEPTH), RING = chain_ring(P, VIEWDEP, DEPTH);
    // (the 16-bit workspace holds whole groups of eight wave blocks, plan_workspace; the padded blocks get zero gradients)
    const long long wave_blocks = P == 1 ? (args.c.total + 255) / 256 * 8 : (args.c.total + 127) / 128 * 4;
    const long long blocks = wave_blocks / NW;
    const size_t lds_bytes = sizeof(float) * (RING * (size_t)args.slot_floats + NW * 256);  // ring + DMA dump area
    auto kernel = mlp_backward_chain_f16x3_kernel<WT, VT, VIEWDEP, P, DEPTH, BF>;
    static snerf::DeviceOnce configured;   // per device: the attribute belongs to (kernel, device)
    const int attr = snerf::raise_dynamic_lds(configured, reinterpret_cast<const void*>(kernel), (int)(sizeof(float) * (RING * (P == 1 ? 24 * 256 : kUnitBufFloats) + NW * 256)), "mlp_backward");
    if (attr != SNERF_OK) return attr;
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(NW * 64), lds_bytes, stream, args);
    return snerf::check_launch("mlp_backward(chain, f16x3)");
}

// FORMAT: 3 = f16x3, 1 = fp16 single product, 2 = bf16 single product (a.act_rows / a.grad_rows describe 16-bit rows for 1, 2).
template <int FORMAT>
int dispatch_chain(const snerf::MlpPlan& plan, const ChainArgs& a, long long stream_offset, hipStream_t stream) {
    constexpr int P = FORMAT == 3 ? 3 : 1;
    constexpr bool BF = FORMAT == 2;
    HalfChainArgs args;
    args.c = a;
    args.half_dgrad_offset = stream_offset;
    int most_ks = 0;
    for (const snerf::MlpPlan::HalfStage& st : plan.half_dgrad_stages) most_ks = std::max(most_ks, st.unit_floats / 512);
    args.slot_floats = P == 3 ? kUnitBufFloats : (most_ks + 7) / 8 * 8 * 256;   // hi halves only, whole DMA rounds of up to eight waves
    const int key = plan.wt * 10 + plan.vt;
    if (key == 84 && a.depth == 8)   // the shipped 8 x 256 trunk with a views layer: compile-time unit schedule
        return launch_chain_half<8, 4, true, P, 8, BF>(args, stream);
    if constexpr (P == 1) {
        if (key == 80 && a.depth == 8)   // the augmentation MLPs of the shipped experiments (no views layer)
            return launch_chain_half<8, 4, false, 1, 8, BF>(args, stream);
    }
    switch (key) {
        case 84: return launch_chain_half<8, 4, true, P, 0, BF>(args, stream);
        case 80: return launch_chain_half<8, 4, false, P, 0, BF>(args, stream);
        case 42: return launch_chain_half<4, 2, true, P, 0, BF>(args, stream);
        case 40: return launch_chain_half<4, 2, false, P, 0, BF>(args, stream);
        default: return snerf::fail(SNERF_E_UNSUPPORTED, "mlp_backward(f16x3): width %d / views width %d not built", plan.width,
                                    plan.views_width);
    }
}

}  // namespace
