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
#include "mlp_device.h"

namespace snerf {
int mlp_forward_f16x3(const MlpPlan& plan, const MlpArgs& m, bool train, int products, hipStream_t stream);  // mlp_forward_f16.hip
int mlp_forward_m16(const MlpPlan& plan, const MlpArgs& m, int products, hipStream_t stream, bool bf16);  // mlp_forward_m16.hip; -1 = layout not built there
int mlp_forward_bf16(const MlpPlan& plan, const MlpArgs& m, bool train, hipStream_t stream);              // mlp_forward_bf16.hip
}

namespace {

// ------------------------------------------------------------------------------------------------
// the kernel
// ------------------------------------------------------------------------------------------------
// VIS: predict_visibility -- the views head has a 4th row (sigmoid -> per-sample visibility) and is evaluated once more per
// secondary view direction: the views layer's pre-activation WITHOUT its view-encoding segment is kept, and that segment's
// slab (the last of the stream, still resident in LDS) is applied again to each secondary direction's encoding.
template <int WT, int VT, bool VIEWDEP, bool SIGMA_PE, bool STORE, bool VIS = false>
__global__ void __launch_bounds__(256, 1) mlp_forward_kernel(MlpArgs a) {
    static_assert(!VIS || VIEWDEP, "the visibility row belongs to the views head");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5;

    SlabStream<WT> st;
    st.start(a.packed, lds, lane, wave);
    // Biases and head weights are staged in LDS once per workgroup: as global loads at the top of every layer their L2
    // latency sat between two layers' MFMAs (and a global load in the layer loop makes the compiler drain the DMA).
    float* consts = lds + 2 * SlabStream<WT>::kBufFloats;
    for (int i = threadIdx.x * 4; i < a.const_floats; i += 256 * 4)
        *reinterpret_cast<f32x4*>(consts + i) = *reinterpret_cast<const f32x4*>(a.packed + a.bias_offset + i);
    __syncthreads();

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

    // training: every layer's input is kept for the backward pass as a [feature][32-sample] tile of this wave block
    float* tile = nullptr;
    unsigned* masks = nullptr;
    if (STORE) {
        tile = a.acts + ((long long)blockIdx.x * 4 + wave) * a.act_rows * 32;
        masks = reinterpret_cast<unsigned*>(tile + a.act_mask * 32);
        store_pe_tile<snerf::kPointsPairs, snerf::kPointsKSteps>(pe, tile, lane);
        if (VIEWDEP) store_pe_tile<snerf::kViewsPairs, snerf::kViewsKSteps>(pev, tile + a.act_pev * 32, lane);
    }

    // ---- trunk ------------------------------------------------------------------------------------
    const float* bias = consts;
    f32x16 acc[WT];
    float h[WT * 16];
    load_bias<WT>(acc, bias, half);
    gemm_segment<WT, 2, WT>(acc, pe, st);
    to_operand<WT, true>(acc, h);
    if (STORE) {
        store_acc_tile(h, tile + a.act_h1 * 32, lane);
        store_relu_masks(h, masks, 0, lane);
    }
#pragma unroll 1
    for (int l = 1; l < a.depth; ++l) {
        load_bias<WT>(acc, bias + (long long)l * a.width, half);
        if (l == 5) gemm_segment<WT, 2, WT>(acc, pe, st);  // skip connection: [encoding | h] (:662-663)
        gemm_segment<WT, WT, WT>(acc, h, st);
        to_operand<WT, true>(acc, h);
        if (STORE) {
            store_acc_tile(h, tile + (a.act_h1 + l * a.width) * 32, lane);
            store_relu_masks(h, masks, l * WT, lane);
        }
    }

    // ---- density (and view-independent colour) head -------------------------------------------------
    const float* wout = consts + (a.pts_out_w - a.bias_offset);
    const float* bout = consts + (a.pts_out_b - a.bias_offset);
    float sigma = head_dot<WT * 16>(h, wout, half) + bout[0];
    if (a.noise) sigma += a.noise[g];
    sigma = fmaxf(sigma, 0.0f);
    float rgb[3];
    if (!VIEWDEP) {
#pragma unroll
        for (int c = 0; c < 3; ++c) rgb[c] = sigmoidf(head_dot<WT * 16>(h, wout + (c + 1) * WT * 32, half) + bout[c + 1]);
    } else {
        // feature = feature_linear(h), no activation (:683)
        load_bias<WT>(acc, consts + (a.feature_bias - a.bias_offset), half);
        gemm_segment<WT, WT, WT>(acc, h, st);
        to_operand<WT, false>(acc, h);
        if (STORE) store_acc_tile(h, tile + a.act_feature * 32, lane);
        // views layer over [feature | rest of the point encoding (points-aug only) | view encoding] (:633, :695-699)
        f32x16 accv[VT];
        load_bias<VT>(accv, consts + (a.views_bias - a.bias_offset), half);
        gemm_segment<VT, WT, WT>(accv, h, st);
        if (SIGMA_PE) gemm_segment<VT, 2, WT>(accv, pe, st);
        f32x16 accv_base[VT];
        const float* view_slab = nullptr;
        if constexpr (VIS) {
#pragma unroll
            for (int u = 0; u < VT; ++u) accv_base[u] = accv[u];
            view_slab = st.template acquire<VT>() + lane * 4;   // nothing follows it in the stream: no further request
            gemm_resident_slab<VT>(accv, pev, view_slab);
        } else {
            gemm_segment<VT, 1, WT>(accv, pev, st);
        }
        float hv[VT * 16];
        to_operand<VT, true>(accv, hv);
        if (STORE) {
            store_acc_tile(hv, tile + a.act_hv * 32, lane);
            store_relu_masks(hv, masks, a.depth * WT, lane);
        }
        const float* wv = consts + (a.views_out_w - a.bias_offset);
        const float* bv = consts + (a.views_out_b - a.bias_offset);
#pragma unroll
        for (int c = 0; c < 3; ++c) rgb[c] = sigmoidf(head_dot<VT * 16>(hv, wv + c * VT * 32, half) + bv[c]);
        if constexpr (VIS) {
            const float vis = sigmoidf(head_dot<VT * 16>(hv, wv + 3 * VT * 32, half) + bv[3]);   // :708-712
            if (a.visibility && live && half == 0) a.visibility[first] = vis;
#pragma unroll 1
            for (int k = 0; k < a.num_other; ++k) {   // 'visibility2': the same head from each secondary direction (:646-649)
                float v2[3], pev2[snerf::kViewsKSteps];
#pragma unroll
                for (int c = 0; c < 3; ++c) v2[c] = a.view_dirs2[(g * a.num_other + k) * 3 + c];
                encode<snerf::kViewsPairs, snerf::kViewsKSteps>(v2, half, pev2);
                f32x16 acc2[VT];
#pragma unroll
                for (int u = 0; u < VT; ++u) acc2[u] = accv_base[u];
                gemm_resident_slab<VT>(acc2, pev2, view_slab);
                float hv2[VT * 16];
                to_operand<VT, true>(acc2, hv2);
                const float vis2 = sigmoidf(head_dot<VT * 16>(hv2, wv + 3 * VT * 32, half) + bv[3]);
                if (live && half == 0) a.visibility2[first * a.num_other + k] = vis2;
            }
        }
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the run-ahead prefetch must land before the LDS is released
    if (live && half == 0) {
        a.sigma[first] = sigma;
        a.rgb[first * 3 + 0] = rgb[0];
        a.rgb[first * 3 + 1] = rgb[1];
        a.rgb[first * 3 + 2] = rgb[2];
    }
}

constexpr int kMaxConstFloats = 5120;   // LDS reserved for the bias / head block (20 KB; the 8x256 main MLP needs 12.6 KB)

template <int WT, int VT, bool VIEWDEP, bool SIGMA_PE, bool STORE, bool VIS = false>
int launch(const MlpArgs& a, hipStream_t stream) {
    if (a.const_floats > kMaxConstFloats)
        return snerf::fail(SNERF_E_UNSUPPORTED, "mlp_forward: bias/head block of %d floats exceeds its LDS area", a.const_floats);
    const long long blocks = (a.total + 127) / 128;
    if (blocks > 0x7fffffffLL) return snerf::fail(SNERF_E_UNSUPPORTED, "mlp_forward: too many samples in one call");
    const size_t lds_bytes = sizeof(float) * (2 * SlabStream<WT>::kBufFloats + (size_t)a.const_floats);
    auto kernel = mlp_forward_kernel<WT, VT, VIEWDEP, SIGMA_PE, STORE, VIS>;
    static snerf::DeviceOnce configured;   // per device: the attribute belongs to (kernel, device)
    const int attr = snerf::raise_dynamic_lds(configured, reinterpret_cast<const void*>(kernel), (int)(sizeof(float) * (2 * SlabStream<WT>::kBufFloats + kMaxConstFloats)), "mlp_forward");
    if (attr != SNERF_OK) return attr;
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(256), lds_bytes, stream, a);
    return snerf::check_launch("mlp_forward");
}

}  // namespace

struct VisibilityIo {   // predict_visibility outputs / secondary directions (all NULL / 0 for the plain entry points)
    float* visibility = nullptr;
    const float* view_dirs2 = nullptr;
    float* visibility2 = nullptr;
    int num_other = 0;
    bool wanted() const { return visibility || num_other > 0; }
};

static int forward_impl(const snerf_mlp_desc* desc, const float* packed, const float* origins, const float* dirs,
                        const float* view_dirs, const float* depths, long long num_rays, int num_samples,
                        const float* sigma_noise, float* sigma, float* rgb, float* saved_acts, bool train, int precision,
                        snerf_stream_t stream, const VisibilityIo& vis = VisibilityIo()) {
    snerf::MlpPlan plan;
    const int st = snerf::build_plan(desc, &plan);
    if (st != SNERF_OK) return st;
    SNERF_REQUIRE(packed && origins && dirs && depths && sigma && rgb, "mlp_forward: NULL pointer");
    SNERF_REQUIRE(!plan.view_dependent || view_dirs, "mlp_forward: this MLP needs view_dirs");
    SNERF_REQUIRE(num_rays >= 0 && num_samples >= 1, "mlp_forward: bad sizes n=%lld S=%d", num_rays, num_samples);
    SNERF_REQUIRE(!train || saved_acts, "mlp_forward_train: saved_acts is NULL");
    if (precision != SNERF_PRECISION_FP32 && precision != SNERF_PRECISION_F16X3 && precision != SNERF_PRECISION_F16 &&
        precision != SNERF_PRECISION_BF16)
        return snerf::fail(SNERF_E_UNSUPPORTED, "mlp_forward: precision %d not built", precision);
    const bool bf16 = precision == SNERF_PRECISION_BF16;
    if (vis.wanted()) {
        SNERF_REQUIRE(desc->predict_visibility, "mlp_forward_visibility: the descriptor has predict_visibility = 0");
        SNERF_REQUIRE(vis.num_other >= 0 && vis.num_other <= 64, "mlp_forward_visibility: %d secondary views", vis.num_other);
        SNERF_REQUIRE(vis.num_other == 0 || (vis.view_dirs2 && vis.visibility2),
                      "mlp_forward_visibility: view_dirs2 / visibility2 are required with num_other > 0");
        if (precision != SNERF_PRECISION_FP32)
            return snerf::fail(SNERF_E_UNSUPPORTED, "mlp_forward_visibility: built for SNERF_PRECISION_FP32 only");
        if (plan.sigma_pe)
            return snerf::fail(SNERF_E_UNSUPPORTED, "mlp_forward_visibility: not built for the points-augmentation layout");
    }
    if (num_rays == 0) return SNERF_OK;
    MlpArgs a;
    a.range_flag = nullptr;
    a.weight_range = nullptr;
    if (precision != SNERF_PRECISION_FP32 && !bf16) {    // fp16 modes: report an earlier launch's range violation, then arm the watch
        const int range = snerf::report_range(train ? "mlp_forward_train" : "mlp_forward");
        if (range != SNERF_OK) return range;
        a.range_flag = snerf::range_flag();
        if (!a.range_flag) return SNERF_E_HIP;
        a.weight_range = reinterpret_cast<const int*>(packed + plan.weight_range_word);
    }
    a.visibility = vis.visibility; a.view_dirs2 = vis.view_dirs2; a.visibility2 = vis.visibility2; a.num_other = vis.num_other;
    a.packed = packed; a.origins = origins; a.dirs = dirs; a.view_dirs = view_dirs; a.depths = depths;
    a.noise = sigma_noise; a.sigma = sigma; a.rgb = rgb;
    a.total = num_rays * num_samples; a.samples = num_samples; a.depth = plan.depth; a.width = plan.width;
    a.bias_offset = plan.bias_offset; a.feature_bias = plan.feature_bias(); a.views_bias = plan.views_bias();
    a.const_floats = (int)((plan.dgrad_offset - plan.bias_offset + 3) / 4 * 4);
    a.pts_out_w = plan.pts_out_w(); a.pts_out_b = plan.pts_out_b();
    a.views_out_w = plan.views_out_w(); a.views_out_b = plan.views_out_b();
    a.acts = saved_acts; a.act_rows = plan.act_rows(); a.act_pev = plan.act_pev(); a.act_h1 = plan.act_h(1);
    a.act_feature = plan.act_feature(); a.act_hv = plan.act_hv(); a.act_mask = plan.act_mask();
    hipStream_t s = (hipStream_t)stream;
    snerf::ProfileScope timed(SNERF_PROFILE_MLP_FORWARD, s, a.total);
    if (precision != SNERF_PRECISION_FP32 && !train) {
        // rendering with the fp16 modes: the 16x16x32 MFMA layout where it is built (more work per joule)
        // (A/B probes build with -DSNERF_PROBE_NO_M16; no run-time switch decides which kernel renders)
#ifndef SNERF_PROBE_NO_M16
        const int st16 = snerf::mlp_forward_m16(plan, a, precision == SNERF_PRECISION_F16X3 ? 3 : 1, s, bf16);
        if (st16 != -1) return st16;
#endif
    }
    if (precision == SNERF_PRECISION_F16X3) return snerf::mlp_forward_f16x3(plan, a, train, 3, s);
    if (precision == SNERF_PRECISION_F16 || bf16) {
        a.act_rows = plan.act16_rows();  // 16-bit pieces: same row numbers, rows of 64 bytes (mlp_plan.h)
        return bf16 ? snerf::mlp_forward_bf16(plan, a, train, s) : snerf::mlp_forward_f16x3(plan, a, train, 1, s);
    }
    const int key = plan.wt * 100 + plan.vt * 10 + (plan.sigma_pe ? 1 : 0);
    if (vis.wanted()) {
        if (key == 840) return train ? launch<8, 4, true, false, true, true>(a, s) : launch<8, 4, true, false, false, true>(a, s);
        if (key == 420) return train ? launch<4, 2, true, false, true, true>(a, s) : launch<4, 2, true, false, false, true>(a, s);
        return snerf::fail(SNERF_E_UNSUPPORTED, "mlp_forward_visibility: width %d / views width %d not built", plan.width, plan.views_width);
    }
#define SNERF_DISPATCH(WT_, VT_, VD_, SP_) return train ? launch<WT_, VT_, VD_, SP_, true>(a, s) : launch<WT_, VT_, VD_, SP_, false>(a, s)
    switch (key) {
        case 840: SNERF_DISPATCH(8, 4, true, false);
        case 841: SNERF_DISPATCH(8, 4, true, true);
        case 800: SNERF_DISPATCH(8, 4, false, false);
        case 420: SNERF_DISPATCH(4, 2, true, false);
        case 421: SNERF_DISPATCH(4, 2, true, true);
        case 400: SNERF_DISPATCH(4, 2, false, false);
        default:
            return snerf::fail(SNERF_E_UNSUPPORTED, "mlp_forward: width %d / views width %d combination not built "
                               "(256/128, 128/64, or 256|128 without a views layer)", plan.width, plan.views_width);
    }
#undef SNERF_DISPATCH
}

extern "C" int snerf_mlp_forward(const snerf_mlp_desc* desc, const float* packed, const float* origins,
                                 const float* dirs, const float* view_dirs, const float* depths, long long num_rays,
                                 int num_samples, const float* sigma_noise, float* sigma, float* rgb, int precision,
                                 snerf_stream_t stream) {
    return forward_impl(desc, packed, origins, dirs, view_dirs, depths, num_rays, num_samples, sigma_noise, sigma, rgb,
                        nullptr, false, precision, stream);
}

extern "C" int snerf_mlp_forward_train(const snerf_mlp_desc* desc, const float* packed, const float* origins,
                                       const float* dirs, const float* view_dirs, const float* depths,
                                       long long num_rays, int num_samples, const float* sigma_noise, float* sigma,
                                       float* rgb, float* saved_acts, int precision, snerf_stream_t stream) {
    return forward_impl(desc, packed, origins, dirs, view_dirs, depths, num_rays, num_samples, sigma_noise, sigma, rgb,
                        saved_acts, true, precision, stream);
}

extern "C" int snerf_mlp_forward_visibility(const snerf_mlp_desc* desc, const float* packed, const float* origins,
                                            const float* dirs, const float* view_dirs, const float* depths, long long num_rays,
                                            int num_samples, const float* sigma_noise, const float* view_dirs2, int num_other,
                                            float* sigma, float* rgb, float* visibility, float* visibility2,
                                            float* saved_acts, int precision, snerf_stream_t stream) {
    VisibilityIo vis;
    vis.visibility = visibility; vis.view_dirs2 = view_dirs2; vis.visibility2 = visibility2; vis.num_other = num_other;
    SNERF_REQUIRE(vis.wanted(), "mlp_forward_visibility: neither visibility nor secondary views requested");
    return forward_impl(desc, packed, origins, dirs, view_dirs, depths, num_rays, num_samples, sigma_noise, sigma, rgb,
                        saved_acts, saved_acts != nullptr, precision, stream, vis);
}

extern "C" size_t snerf_mlp_saved_floats(const snerf_mlp_desc* desc, long long num_rays, int num_samples) {
    snerf::MlpPlan plan;
    if (snerf::build_plan(desc, &plan) != SNERF_OK || num_rays < 0 || num_samples < 1) return 0;
    const long long blocks = (num_rays * num_samples + 255) / 256 * 8;  // whole workgroups of up to 8 wave blocks
    return (size_t)(blocks * plan.act_rows() * 32);
}
