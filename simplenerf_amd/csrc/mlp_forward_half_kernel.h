// K3 (split-precision variant) -- the fused PE + MLP forward on the fp16 matrix cores with fp32-grade accuracy.  Kernel
// template + launchers, included by mlp_forward_f16.hip (f16x3 and the fp16 single-product format) and mlp_forward_bf16.hip
// (the bf16 single-product format): two translation units, so the instantiations compile side by side.
//
// Every fp32 operand v is split into two fp16 numbers, v = hi + lo with hi = fp16(v), lo = fp16(v - hi) (about 22
// significand bits together; lo may be an fp16 subnormal -- v_mfma_f32_32x32x16_f16 honours subnormal inputs, probed
// in tools/probes/mfma_f16_denorm.hip), and every product W.x is evaluated as three MFMAs accumulating in fp32:
//     W.x ~= Wh.xh + Wh.xl + Wl.xh                      (the dropped Wl.xl term is ~2^-22 relative)
// That is 3/16 of the fp32-MFMA issue time for the same algorithmic FLOPs (fp16 MFMA runs 16x the fp32 rate), at an
// accuracy that still meets north_star's 1e-4 / 1e-3 parity bar against the reference's fp32 CPU path -- which plain
// fp16/bf16 inputs (8-11 significand bits through 10 chained layers) do not.
//
// Structure: the same register-resident transposed chain as mlp_forward.hip (accumulator tile of one layer = B
// operand of the next; for the 16-deep fp16 MFMA, registers 8s..8s+7 of a 32x32 tile are the 8 elements of k-step s),
// but OUT-TILE-MAJOR: one 32-row output tile is accumulated over all of its k-steps before the next one starts, so a
// single 16-register accumulator is live, its ReLU + hi/lo split (VALU) overlaps the next tile's MFMAs, and the LDS
// staging unit is "all k-steps of one out tile" (hi and lo fragments interleaved per k-step, 2 KiB each).
//
// Bound: MFMA fp16 (dense peak 2.5 PFLOP/s; 3 MFMA passes per algorithmic product -> 833 TFLOP/s algorithmic ceiling),
// with the weight stream L2 -> LDS (2.3 MB per 128 samples) as the secondary limit.
#include <algorithm>
#include <type_traits>

#pragma once
#include "clock_stamp.h"
#include "mlp_device_f16.h"
#include "mlp_plan.h"

namespace {

SNERF_STAMP_DEFINE(forward_f16)

// f(integral_constant<I>), ..., f(integral_constant<N-1>): one inlined copy of the body per index (a `#pragma unroll` on a
// loop this large is refused by the optimiser)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

struct HalfArgs {
    MlpArgs m;
    long long half_offset;
    int const_floats;  // biases + head weights: packed[bias_offset, bias_offset + const_floats), kept in LDS
    int slot_floats;   // LDS ring slot size
};

// P = 3: split precision (SNERF_PRECISION_F16X3).  P = 1: single fp16 product (SNERF_PRECISION_F16); its training variant
// saves the activations as 16-bit operand fragments (store_piece) instead of fp32 rows.
// The single-product kernels size the ring to their (hi-only) units so that two workgroups fit a CU: at one wave per SIMD
// every LDS latency, barrier and operand conversion sits in front of the 32-cycle MFMAs; a second wave fills those gaps
// (inference: 1.47 -> 1.89 M rays/s on the headline step).  The training variant needs ~120 more registers for the
// mask words and fragment stores and would spill at the 256-register budget of that occupancy (measured: slower), so
// it keeps one workgroup per CU.
// DEPTH > 0: the trunk depth is a compile-time constant and the layer loop is fully unrolled, so that the whole unit
// schedule -- k-steps of every unit and of its two successors, DMA pieces per wave, counted vmcnt immediates, ring slots,
// the point where a request's pieces run out -- folds to constants.  With the generic (DEPTH = 0) loop that bookkeeping
// is ~250 scalar instructions, compare-and-branch trees included, around the 16 MFMAs of a 256-wide unit: at one or two
// waves per SIMD every one of them is an issue slot, and the matrix pipe waited on the scalar unit (r02).
// S8 (SNERF_PRECISION_F16S8, training forward of the fp16 single-product format): the trunk activations h_1 .. h_D-1 are saved
// as fp8 e4m3 tiles (store_pieces8, mlp_device_f16.h); everything else as with P = 1.
template <int WT, int VT, bool VIEWDEP, bool SIGMA_PE, bool STORE, int P, int DEPTH, bool BF = false, bool S8 = false>
__global__ void __launch_bounds__(P == 1 ? 512 : 256, P == 1 ? 2 : 1) mlp_forward_f16x3_kernel(HalfArgs args) {
    static_assert(!BF || P == 1, "bf16 operands: single-product kernels only");
    static_assert(!S8 || (P == 1 && STORE), "8-bit saved activations: the single-product training forwards");
    // (only the 256-wide trunk: its weight-gradient products are the large register-tile class, the one built to read fp8 tiles;
    // a 128-wide MLP's are folded into the small-job launch and keep fp16 activations -- for it this mode IS the fp16 mode)
    constexpr bool SAVE8 = S8 && WT == 8;
    constexpr int NW = P == 1 ? 8 : 4;   // waves per workgroup (see UnitStreamT)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const MlpArgs& a = args.m;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5;
    constexpr int HK = WT * 2;  // k-steps of a full-width activation
    const int depth = DEPTH > 0 ? DEPTH : a.depth;

    // k-steps of staging unit `idx` in stream order: trunk layers (WT units each), feature stage (WT), views layer (VT)
    constexpr int kViewsKs = HK + (SIGMA_PE ? 4 : 0) + 2;
    const int trunk_units = depth * WT;
    auto ks_of = [&](int idx) {
        if (idx < trunk_units) {
            const int l = idx / WT;
            return l == 0 ? 4 : (l == 5 ? 4 + HK : HK);
        }
        if (!VIEWDEP) return 0;
        const int v = idx - trunk_units;
        return v < WT ? HK : (v < WT + VT ? kViewsKs : 0);
    };
    UnitStreamT<P, NW, kUnitBuffers, BF ? 256 : 512> st;
    st.start(a.packed + args.half_offset, lds, ks_of(0), ks_of(1), lane, wave, args.slot_floats);
    int unit_idx = 0;
    auto next_unit = [&]() {
        const float* p = st.acquire(ks_of(unit_idx + 1), ks_of(unit_idx + 2));
        ++unit_idx;
        return p + lane * 4;
    };
    // Biases and head weights live in LDS for the whole kernel: an ordinary global load inside the tile loop would make
    // the compiler wait vmcnt(0), i.e. drain the weight prefetch (LDS-DMA) that is deliberately left in flight.
    float* consts = lds + kUnitBuffers * args.slot_floats + NW * 256;  // after the ring and the DMA dump area (1 KiB per wave)
    for (int i = threadIdx.x * 4; i < args.const_floats; i += NW * 64 * 4)
        *reinterpret_cast<f32x4*>(consts + i) = *reinterpret_cast<const f32x4*>(a.packed + a.bias_offset + i);
    __syncthreads();
    SNERF_STAMP_BEGIN();

    const long long first = ((long long)blockIdx.x * NW + wave) * 32 + (lane & 31);
    const bool live = first < a.total;
    const long long g = live ? first : a.total - 1;
    const long long ray = g / a.samples;
    const float z = a.depths[g];
    float x[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) x[k] = a.origins[ray * 3 + k] + a.dirs[ray * 3 + k] * z;

    constexpr bool STORE32 = STORE && P == 3, STORE16 = STORE && P == 1;
    float* tile = nullptr;  // training: this wave block's saved-activation tiles (fp32, same layout as the fp32 path)
    _Float16* tile16 = nullptr;  // ... or 16-bit fragment pieces, rows of 32 x 16 bit (a.act_rows counts those)
    unsigned* masks = nullptr;
    if (STORE32) {
        tile = a.acts + ((long long)blockIdx.x * NW + wave) * a.act_rows * 32;
        masks = reinterpret_cast<unsigned*>(tile + a.act_mask * 32);
    }
    if (STORE16) {
        tile16 = reinterpret_cast<_Float16*>(a.acts) + ((long long)blockIdx.x * NW + wave) * a.act_rows * 32;
        masks = reinterpret_cast<unsigned*>(tile16 + a.act_mask * 32);
    }
    f16x8 pe_h[4], pe_l[4], pev_h[2], pev_l[2];
    {
        float pe[snerf::kPointsKSteps];
        encode<snerf::kPointsPairs, snerf::kPointsKSteps>(x, half, pe);
        if (STORE32) store_pe_tile<snerf::kPointsPairs, snerf::kPointsKSteps>(pe, tile, lane);
        split_encoding<32, 4, BF>(pe, pe_h, pe_l);
        if (STORE16) store_pieces<4>(pe_h, tile16, lane);
    }
    if (VIEWDEP) {
        float v[3], pev[snerf::kViewsKSteps];
#pragma unroll
        for (int k = 0; k < 3; ++k) v[k] = a.view_dirs[ray * 3 + k];
        encode<snerf::kViewsPairs, snerf::kViewsKSteps>(v, half, pev);
        if (STORE32) store_pe_tile<snerf::kViewsPairs, snerf::kViewsKSteps>(pev, tile + a.act_pev * 32, lane);
        split_encoding<16, 2, BF>(pev, pev_h, pev_l);
        if (STORE16) store_pieces<2>(pev_h, tile16 + a.act_pev * 32, lane);
    }

    const float* bias = consts;
    const float* wout = consts + (a.pts_out_w - a.bias_offset);
    const float* bout = consts + (a.pts_out_b - a.bias_offset);
    f16x8 xh[HK], xl[HK];
    float head[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // density (and view-independent colour) pre-activations
    const bool single = depth == 1;

    // One accumulator tile per out tile of the layer (WT x 16 registers -- the matrix pipe's own AGPR file), and the
    // activations xh/xl in arch VGPRs where the MFMA reads them directly.  After the layer's last tile the accumulators
    // are ReLU'd and split into the next layer's operands in one VALU pass.  (An earlier variant overlapped that pass
    // with the next tile's MFMAs through a second operand buffer; the extra 128 registers pushed the B operands into
    // AGPRs and every MFMA then paid v_accvgpr_read moves -- slower overall.)
    f32x16 acc[WT];
    unsigned mask_bits = 0;
    RangeWatch watch;   // one pre-activation per layer for this lane's sample (mlp_device_f16.h)
    auto heads_from = [&](const f32x16& t, int u) {
        head[0] += tile_dot_relu(t, wout + 32 * u, half);
        if (!VIEWDEP) {
#pragma unroll
            for (int c = 1; c < 4; ++c) head[c] += tile_dot_relu(t, wout + c * WT * 32 + 32 * u, half);
        }
    };

    // ---- trunk layer 0: encoding -> h ---------------------------------------------------------------------------
#pragma unroll
    for (int u = 0; u < WT; ++u) {
        const float* unit = next_unit();
        tile_bias(acc[u], bias + 32 * u, half);
        seg_product<P, 4, BF>(acc[u], unit, 4, pe_h, pe_l, st);
        if (!BF && u == 0) watch.probe(acc[0][0]);       // non-finite iff an encoded input left the fp16 range
        if (single) heads_from(acc[u], u);
        if (STORE32) { store_tile_rows<true>(acc[u], tile + (a.act_h1 + 32 * u) * 32, lane); st.note_vmem(16); }
        if (STORE16) { relu_mask_tile(acc[u], u, mask_bits, masks, 0, lane); st.note_vmem(u & 1); }
    }
    if (STORE32) store_relu_masks<WT>(acc, masks, 0, lane);
#pragma unroll
    for (int u = 0; u < WT; ++u) {
        if constexpr (P == 1) convert_tile<true, BF>(acc[u], xh[2 * u], xh[2 * u + 1]);
        else split_tile<true>(acc[u], xh[2 * u], xl[2 * u], xh[2 * u + 1], xl[2 * u + 1]);
    }
    if (STORE16) {
        if (SAVE8 && depth > 1) { store_pieces8<HK, BF>(xh, tile16 + a.act_h1 * 32, lane); st.note_vmem(HK / 2); }   // h_1
        else { store_pieces<HK>(xh, tile16 + a.act_h1 * 32, lane); st.note_vmem(HK); }
    }

    // ---- trunk layers 1 .. depth-1 --------------------------------------------------------------------------------
    auto trunk_layer = [&](int l) __attribute__((always_inline)) {
        const float* bl = bias + (long long)l * a.width;
        const bool last = l == depth - 1;
#pragma unroll
        for (int u = 0; u < WT; ++u) {
            const float* unit = next_unit();
            tile_bias(acc[u], bl + 32 * u, half);
            const int unit_ks = l == 5 ? 4 + HK : HK;
            if (l == 5) seg_product<P, 4, BF>(acc[u], unit, unit_ks, pe_h, pe_l, st);  // skip connection [encoding | h]
            seg_product<P, HK, BF>(acc[u], unit, unit_ks, xh, xl, st);
            if (!BF && u == 0) watch.probe(acc[0][0]);   // non-finite iff an activation of layer l-1 left the fp16 range
            if (last) heads_from(acc[u], u);
            if (STORE32) { store_tile_rows<true>(acc[u], tile + (a.act_h1 + l * a.width + 32 * u) * 32, lane); st.note_vmem(16); }
            if (STORE16) { relu_mask_tile(acc[u], u, mask_bits, masks, l * WT, lane); st.note_vmem(u & 1); }
        }
        if (STORE32) store_relu_masks<WT>(acc, masks, l * WT, lane);
#pragma unroll
        for (int u = 0; u < WT; ++u) {
            if constexpr (P == 1) convert_tile<true, BF>(acc[u], xh[2 * u], xh[2 * u + 1]);
            else split_tile<true>(acc[u], xh[2 * u], xl[2 * u], xh[2 * u + 1], xl[2 * u + 1]);
        }
        if constexpr (!VIEWDEP && STORE16 && !BF) {
            // a view-independent MLP's last activations are saved for the head weight gradients and multiplied by nothing in
            // this pass: no later pre-activation would turn non-finite, so they are watched directly
            if (last) {
#pragma unroll
                for (int i = 0; i < HK; ++i) watch.see(xh[i]);
            }
        }
        if (STORE16) {
            // (h_D, the last trunk tensor, stays fp16: it also feeds the small head products)
            if (SAVE8 && !last) { store_pieces8<HK, BF>(xh, tile16 + a.act_h1 * 32 + l * (WT * 512), lane); st.note_vmem(HK / 2); }
            else { store_pieces<HK>(xh, tile16 + (a.act_h1 + l * a.width) * 32, lane); st.note_vmem(HK); }
        }
    };
    if constexpr (DEPTH > 0) {
        static_for<1, DEPTH>([&](auto layer) __attribute__((always_inline)) { trunk_layer(decltype(layer)::value); });
    } else {
#pragma unroll 1
        for (int l = 1; l < depth; ++l) trunk_layer(l);
    }

    float sigma = (head[0] + __shfl_xor(head[0], 32, 64)) + bout[0];
    if (a.noise) sigma += a.noise[g];
    sigma = fmaxf(sigma, 0.0f);
    float rgb[3];
    if (!VIEWDEP) {
#pragma unroll
        for (int c = 0; c < 3; ++c) rgb[c] = sigmoidf((head[c + 1] + __shfl_xor(head[c + 1], 32, 64)) + bout[c + 1]);
    } else {
        // feature = feature_linear(h): no activation
        const float* bf = consts + (a.feature_bias - a.bias_offset);
#pragma unroll
        for (int u = 0; u < WT; ++u) {
            const float* unit = next_unit();
            tile_bias(acc[u], bf + 32 * u, half);
            seg_product<P, HK, BF>(acc[u], unit, HK, xh, xl, st);
            if (!BF && u == 0) watch.probe(acc[0][0]);
            if (STORE32) { store_tile_rows<false>(acc[u], tile + (a.act_feature + 32 * u) * 32, lane); st.note_vmem(16); }
        }
#pragma unroll
        for (int u = 0; u < WT; ++u) {
            if constexpr (P == 1) convert_tile<false, BF>(acc[u], xh[2 * u], xh[2 * u + 1]);
            else split_tile<false>(acc[u], xh[2 * u], xl[2 * u], xh[2 * u + 1], xl[2 * u + 1]);
        }
        if (STORE16) { store_pieces<HK>(xh, tile16 + a.act_feature * 32, lane); st.note_vmem(HK); }
        // views layer over [feature | rest of the point encoding (points-aug) | view encoding], then the colour head
        const float* bv = consts + (a.views_bias - a.bias_offset);
        const float* wv = consts + (a.views_out_w - a.bias_offset);
        const float* bo = consts + (a.views_out_b - a.bias_offset);
        float col[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int u = 0; u < VT; ++u) {
            const float* unit = next_unit();
            tile_bias(acc[u], bv + 32 * u, half);
            seg_product<P, HK, BF>(acc[u], unit, kViewsKs, xh, xl, st);
            if (SIGMA_PE) seg_product<P, 4, BF>(acc[u], unit, kViewsKs, pe_h, pe_l, st);
            seg_product<P, 2, BF>(acc[u], unit, kViewsKs, pev_h, pev_l, st);
            if (!BF && u == 0) watch.probe(acc[0][0]);   // the feature vector and the view encoding
            if (STORE32) { store_tile_rows<true>(acc[u], tile + (a.act_hv + 32 * u) * 32, lane); st.note_vmem(16); }
            if (STORE16) {
                f16x8 vh[2];
                convert_tile<true, BF>(acc[u], vh[0], vh[1]);
                if (!BF) { watch.see(vh[0]); watch.see(vh[1]); }      // saved for the weight gradients only: nothing downstream multiplies them
                store_pieces<2>(vh, tile16 + (a.act_hv + 32 * u) * 32, lane);
                relu_mask_tile(acc[u], u, mask_bits, masks, depth * WT, lane);
                st.note_vmem(2 + (u & 1));
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) col[c] += tile_dot_relu(acc[u], wv + c * VT * 32 + 32 * u, half);
        }
        if (STORE32) store_relu_masks<VT>(acc, masks, depth * WT, lane);
#pragma unroll
        for (int c = 0; c < 3; ++c) rgb[c] = sigmoidf((col[c] + __shfl_xor(col[c], 32, 64)) + bo[c]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!BF) watch.report(a.range_flag, a.weight_range);
    SNERF_STAMP_END(forward_f16);
    if (live && half == 0) {
        a.sigma[first] = sigma;
        a.rgb[first * 3 + 0] = rgb[0];
        a.rgb[first * 3 + 1] = rgb[1];
        a.rgb[first * 3 + 2] = rgb[2];
    }
}

template <int WT, int VT, bool VIEWDEP, bool SIGMA_PE, bool STORE, int P, int DEPTH = 0, bool BF = false, bool S8 = false>
int launch_half(const HalfArgs& args, hipStream_t stream) {
    constexpr int NW = P == 1 ? 8 : 4;
    const long long blocks = (args.m.total + NW * 32 - 1) / (NW * 32);
    if (blocks > 0x7fffffffLL) return snerf::fail(SNERF_E_UNSUPPORTED, "mlp_forward: too many samples in one call");
    const size_t lds_bytes = sizeof(float) * (kUnitBuffers * (size_t)args.slot_floats + NW * 256 + (size_t)args.const_floats);
    auto kernel = mlp_forward_f16x3_kernel<WT, VT, VIEWDEP, SIGMA_PE, STORE, P, DEPTH, BF, S8>;
    static snerf::DeviceOnce configured;   // per device: the attribute belongs to (kernel, device)
    const int attr = snerf::raise_dynamic_lds(configured, reinterpret_cast<const void*>(kernel), (int)(sizeof(float) * (kUnitBuffers * kUnitBufFloats + 2048 + 5120)), "mlp_forward");
    if (attr != SNERF_OK) return attr;
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(NW * 64), lds_bytes, stream, args);
    return snerf::check_launch("mlp_forward(f16x3)");
}

// FORMAT: 3 = split precision (SNERF_PRECISION_F16X3), 1 = single fp16 product (SNERF_PRECISION_F16), 2 = single bf16 product
// (SNERF_PRECISION_BF16), 4 / 5 = the training forward of SNERF_PRECISION_F16S8 / _BF16S8 (fp16 / bf16 products, trunk
// activations saved as fp8).  One translation unit instantiates the format it serves (mlp_forward_f16x3.hip: 3;
// mlp_forward_f16.hip: 1; mlp_forward_bf16.hip: 2; mlp_forward_s8.hip: 4; mlp_forward_bs8.hip: 5), so that they compile side by side.
template <int WT, int VT, bool VIEWDEP, bool SIGMA_PE, int FORMAT>
int launch_variant(const HalfArgs& args, bool train, hipStream_t stream) {
    constexpr int P = FORMAT == 3 ? 3 : 1;
    constexpr bool BF = FORMAT == 2 || FORMAT == 5;
    if constexpr (FORMAT == 4 || FORMAT == 5) {        // training only (rendering in these modes IS the fp16 / bf16 mode)
        if constexpr (WT == 8 && VIEWDEP) {
            if (args.m.depth == 8) return launch_half<WT, VT, VIEWDEP, SIGMA_PE, true, 1, 8, BF, true>(args, stream);
        }
        return launch_half<WT, VT, VIEWDEP, SIGMA_PE, true, 1, 0, BF, true>(args, stream);
    } else {
        if constexpr (WT == 8 && VIEWDEP) {   // the shipped 8 x 256 trunk: compile-time unit schedule (the view-independent
                                              // layout spills ~300 registers when unrolled: it keeps the loop)
            if (args.m.depth == 8)
                return train ? launch_half<WT, VT, VIEWDEP, SIGMA_PE, true, P, 8, BF>(args, stream)
                             : launch_half<WT, VT, VIEWDEP, SIGMA_PE, false, P, 8, BF>(args, stream);
        }
        return train ? launch_half<WT, VT, VIEWDEP, SIGMA_PE, true, P, 0, BF>(args, stream)
                     : launch_half<WT, VT, VIEWDEP, SIGMA_PE, false, P, 0, BF>(args, stream);
    }
}

// Argument block + layout dispatch shared by the entry points (argument checks were done by snerf_mlp_forward).  `stages` /
// `offset`: the unit stream this format reads.  For the single-product formats in training mode m.act_* describe the 16-bit
// tile layout.
template <int FORMAT>
int dispatch_half(const snerf::MlpPlan& plan, const MlpArgs& m, bool train, long long offset, hipStream_t stream) {
    using snerf::fail;
    HalfArgs args;
    args.m = m;
    args.half_offset = offset;
    args.const_floats = (int)((plan.dgrad_offset - plan.bias_offset + 3) / 4 * 4);  // biases + heads (+ alignment padding)
    if (args.const_floats > 5120) return fail(SNERF_E_UNSUPPORTED, "mlp_forward(f16x3): bias/head block of %d floats exceeds its LDS area", args.const_floats);
    int most_ks = 0;
    for (const snerf::MlpPlan::HalfStage& st : plan.half_stages) {
        if (st.unit_floats > kUnitBufFloats)
            return fail(SNERF_E_UNSUPPORTED, "mlp_forward(f16x3): staging unit of %d KiB exceeds the LDS buffer", st.unit_floats / 256);
        most_ks = std::max(most_ks, st.unit_floats / 512);
    }
    // the single-product formats request the hi half of each unit only: k KiB-pieces rounded up to eight (one per wave of
    // their 8-wave workgroups)
    args.slot_floats = FORMAT == 3 ? kUnitBufFloats : (most_ks + 7) / 8 * 8 * 256;
    const int key = plan.wt * 100 + plan.vt * 10 + (plan.sigma_pe ? 1 : 0);
    switch (key) {
        case 840: return launch_variant<8, 4, true, false, FORMAT>(args, train, stream);
        case 841: return launch_variant<8, 4, true, true, FORMAT>(args, train, stream);
        case 800: return launch_variant<8, 4, false, false, FORMAT>(args, train, stream);
        case 420: return launch_variant<4, 2, true, false, FORMAT>(args, train, stream);
        case 421: return launch_variant<4, 2, true, true, FORMAT>(args, train, stream);
        case 400: return launch_variant<4, 2, false, false, FORMAT>(args, train, stream);
        default:
            return fail(SNERF_E_UNSUPPORTED, "mlp_forward(f16x3): width %d / views width %d combination not built", plan.width,
                        plan.views_width);
    }
}

}  // namespace
