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
#include "mlp_forward_body.h"
#include "mlp_generic.h"

namespace snerf {
int mlp_forward_f16_split(const MlpPlan& plan, const MlpArgs& m, bool train, hipStream_t stream);    // mlp_forward_f16x3.hip
int mlp_forward_f16_single(const MlpPlan& plan, const MlpArgs& m, bool train, hipStream_t stream);   // mlp_forward_f16.hip
// products = 3: SNERF_PRECISION_F16X3; 1: SNERF_PRECISION_F16
inline int mlp_forward_f16x3(const MlpPlan& plan, const MlpArgs& m, bool train, int products, hipStream_t stream) {
    return products == 3 ? mlp_forward_f16_split(plan, m, train, stream) : mlp_forward_f16_single(plan, m, train, stream);
}
int mlp_forward_m16(const MlpPlan& plan, const MlpArgs& m, int products, hipStream_t stream, bool bf16);  // mlp_forward_m16.hip; -1 = layout not built there
int mlp_forward_bf16(const MlpPlan& plan, const MlpArgs& m, bool train, hipStream_t stream);              // mlp_forward_bf16.hip
int mlp_forward_s8_train(const MlpPlan& plan, const MlpArgs& m, hipStream_t stream);                      // mlp_forward_s8.hip
int mlp_forward_bs8_train(const MlpPlan& plan, const MlpArgs& m, hipStream_t stream);                     // mlp_forward_bs8.hip
}

namespace {

template <int WT, int VT, bool VIEWDEP, bool SIGMA_PE, bool STORE, bool VIS = false>
__global__ void __launch_bounds__(256, 1) mlp_forward_kernel(MlpArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    mlp_forward_body<WT, VT, VIEWDEP, SIGMA_PE, STORE, VIS>(a, blockIdx.x, lds);
}

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
    snerf::GenericPlan layered;
    if (st == SNERF_E_UNSUPPORTED && snerf::generic_takes(desc, &layered)) {     // a shape of the layered path (mlp_generic.hip)
        SNERF_REQUIRE(packed && origins && dirs && depths && sigma && rgb, "mlp_forward: NULL pointer");
        SNERF_REQUIRE(!layered.view_dep || view_dirs, "mlp_forward: this MLP needs view_dirs");
        SNERF_REQUIRE(num_rays >= 0 && num_samples >= 1, "mlp_forward: bad sizes n=%lld S=%d", num_rays, num_samples);
        SNERF_REQUIRE(!train || saved_acts, "mlp_forward_train: saved_acts is NULL");
        SNERF_REQUIRE(!vis.wanted(), "mlp_forward_visibility: not built for this MLP shape");
        if (num_rays == 0) return SNERF_OK;
        return snerf::generic_forward(layered, packed, origins, dirs, view_dirs, depths, num_rays, num_samples, sigma_noise, sigma,
                                      rgb, train ? saved_acts : nullptr, precision, (hipStream_t)stream);
    }
    if (st != SNERF_OK) return st;
    SNERF_REQUIRE(packed && origins && dirs && depths && sigma && rgb, "mlp_forward: NULL pointer");
    SNERF_REQUIRE(!plan.view_dependent || view_dirs, "mlp_forward: this MLP needs view_dirs");
    SNERF_REQUIRE(num_rays >= 0 && num_samples >= 1, "mlp_forward: bad sizes n=%lld S=%d", num_rays, num_samples);
    SNERF_REQUIRE(!train || saved_acts, "mlp_forward_train: saved_acts is NULL");
    if (precision != SNERF_PRECISION_FP32 && precision != SNERF_PRECISION_F16X3 && precision != SNERF_PRECISION_F16 &&
        precision != SNERF_PRECISION_BF16 && precision != SNERF_PRECISION_F16S8 && precision != SNERF_PRECISION_BF16S8)
        return snerf::fail(SNERF_E_UNSUPPORTED, "mlp_forward: precision %d not built", precision);
    // (the fp8 modes: only what the training forward SAVES differs from their 16-bit mode)
    if (precision == SNERF_PRECISION_F16S8 && (!train || plan.depth < 2)) precision = SNERF_PRECISION_F16;
    if (precision == SNERF_PRECISION_BF16S8 && (!train || plan.depth < 2)) precision = SNERF_PRECISION_BF16;
    const bool bf16 = precision == SNERF_PRECISION_BF16 || precision == SNERF_PRECISION_BF16S8;
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
    {   // the weight streams this call reads must be the ones the buffer's last pack wrote (snerf_common.h)
        const int formats = snerf::packed_formats_require(packed, snerf::packed_formats_needed(precision, train),
                                                          train ? "mlp_forward_train" : "mlp_forward");
        if (formats != SNERF_OK) return formats;
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
    if (precision == SNERF_PRECISION_F16 || bf16 || precision == SNERF_PRECISION_F16S8) {
        a.act_rows = plan.act16_rows();  // 16-bit pieces: same row numbers, rows of 64 bytes (mlp_plan.h)
        if (precision == SNERF_PRECISION_F16S8) return snerf::mlp_forward_s8_train(plan, a, s);
        if (precision == SNERF_PRECISION_BF16S8) return snerf::mlp_forward_bs8_train(plan, a, s);
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
    snerf::GenericPlan layered;
    const int st = snerf::build_plan(desc, &plan);
    if (num_rays < 0 || num_samples < 1) return 0;
    if (st == SNERF_E_UNSUPPORTED && snerf::generic_takes(desc, &layered)) return snerf::generic_saved_floats(layered, num_rays * num_samples);
    if (st != SNERF_OK) return 0;
    const long long blocks = (num_rays * num_samples + 255) / 256 * 8;  // whole workgroups of up to 8 wave blocks
    return (size_t)(blocks * plan.act_rows() * 32);
}
