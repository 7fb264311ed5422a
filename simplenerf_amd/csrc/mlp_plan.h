// Host-side plan of one MLP's packed parameter buffer: which K-segments exist, in stream order, and where the
// bias and head blocks live.  Built from the C-ABI descriptor; used by snerf_mlp_packed_floats, snerf_mlp_pack and
// snerf_mlp_forward so the three can never disagree.
#pragma once
#include <vector>

#include "mlp_layout.h"
#include "snerf_common.h"

namespace snerf {

struct MlpPlan {
    int depth = 0, width = 0, views_width = 0;
    int wt = 0, vt = 0;          // 32-row output tiles of the trunk / views layer
    bool view_dependent = false;  // feature + views layer + rgb head exist
    bool sigma_pe = false;        // points-augmentation layout: trunk sees a low-degree encoding
    int points_degree = 0, views_degree = 0;
    int full_pe = 0, pts_in = 0, extra = 0, views_pe = 0;
    int pts_out_rows = 1;
    int views_out_rows = 3;       // 4 with predict_visibility (rgb + visibility)
    int num_params = 0;
    std::vector<Segment> segments;
    long long weight_floats = 0;  // all slabs + one maximum-size slab of zero padding (the prefetcher runs one ahead)
    long long bias_offset = 0;    // trunk biases [depth][width], feature bias [width], views bias [views_width]
    long long head_offset = 0;    // pts_output W [rows][width], b [4]; views_output W [3][views_width], b [4]
    long long dgrad_offset = 0;   // W^T stream consumed by the backward chain (views, feature, trunk depth-1 .. 1) + runway
    std::vector<Segment> dgrad_segments;
    // ---- f16x3 stream (mlp_forward_f16.hip): out-tile-major "units", each = every k-step of one 32-row out tile ------
    struct HalfSegment {
        int param, ld, out_dim;      // source weight tensor (rows x ld)
        int kind;                    // SEG_ACC / SEG_POINTS_PE / SEG_VIEWS_PE
        int ksteps;                  // 16-feature k-steps: 16 (width 256), 8 (width 128), 4 (point encoding), 2 (view encoding)
        int col_offset, feat_lo, feat_hi, degree;
        int transposed;              // 1: dgrad operand W^T -- tile rows are IN features col_offset + 32u + i (< feat_hi of
                                     //    them), k-steps run over OUT features in accumulator order
    };
    struct HalfStage {
        int tiles;                   // out tiles
        int nseg;
        HalfSegment seg[3];
        long long dst;               // offset (floats) of the stage's first unit
        int unit_floats;             // 512 floats (2 KiB: hi + lo fragment) per k-step
        int bf16;                    // 1: a unit of the compact bf16 streams -- ONE KiB per k-step (bf16(w), no lo fragment),
                                     //    unit_floats / 2 floats in memory
    };
    std::vector<HalfStage> half_stages;
    long long half_offset = 0;       // start of the f16x3 stream inside the packed buffer
    std::vector<HalfStage> half_dgrad_stages;  // W^T stream of the f16x3 backward chain (views, feature, trunk depth-1 .. 1)
    long long half_dgrad_offset = 0;
    // The forward stages once more in the fragment layout of v_mfma_f32_16x16x32_f16 (mlp_forward_m16.hip, inference): same
    // units and sizes, fragment f = 2c + r of a unit = rows 16r .. 16r+15 of the out tile x the 32 inputs of k-block c.
    std::vector<HalfStage> m16_stages;
    long long m16_offset = 0;
    // SNERF_PRECISION_BF16: the three half streams once more with bf16 weights, compact (hi only): forward units, W^T units of
    // the backward chain, and the 16x16x32 fragment order for inference (same condition as m16_stages).
    std::vector<HalfStage> bf_stages, bf_dgrad_stages, bf_m16_stages;
    long long bf_offset = 0, bf_dgrad_offset = 0, bf_m16_offset = 0;
    long long total_floats = 0;
    long long weight_range_word = 0;      // float offset of the buffer's fp16 weight-range word (see build_plan)

    // ---- saved-activation / gradient tiles of one 32-sample wave block (backward only) -----------------------
    // Every tensor is stored [feature][32 samples] (feature-major inside the block), features in natural order.
    // acts:  pe[64] | pev[32] | h_1 .. h_depth [width each] | feature[width] | hv[views_width]
    // grads: dY_0 .. dY_{depth-1} [width each] | dfeature[width] | dYv[views_width] | dhead[32]
    int act_pe() const { return 0; }
    int act_pev() const { return 64; }
    int act_h(int l) const { return 96 + (l - 1) * width; }  // l = 1 .. depth
    int act_feature() const { return 96 + depth * width; }
    int act_hv() const { return 96 + (depth + 1) * width; }
    // ReLU sign bits of every hidden tile, kept next to the activations for the backward chain: tile t = l*(width/32)+u
    // for trunk layer l's output (then the views layer's tiles); 16 bits per lane per tile, two tiles per dword, one
    // dword-row of 64 lanes = 2 rows of 32 floats.  The chain reads these 1 KB per layer instead of 32 KB of activations.
    int mask_tiles() const { return depth * (width / 32) + (view_dependent ? views_width / 32 : 0); }
    int act_mask() const { return 96 + (depth + 1) * width + (view_dependent ? views_width : 0); }
    int act_rows() const { return act_mask() + 2 * ((mask_tiles() + 1) / 2); }
    // SNERF_PRECISION_F16 keeps the same tensors as 16-bit operand fragments ("pieces", mlp_device_f16.h store_pieces):
    // the row numbers above then count rows of 32 x 16 bit (64 bytes), a 32-feature tile = two 1 KiB pieces = its 32
    // rows, and a pair of mask words takes four such rows.  Block strides of the two 16-bit buffers, in those rows:
    int act16_rows() const { return act_mask() + 4 * ((mask_tiles() + 1) / 2); }
    int grad16_rows() const { return grad_rows(); }
    int grad_y(int l) const { return l * width; }             // l = 0 .. depth-1
    int grad_feature() const { return depth * width; }
    int grad_yv() const { return (depth + 1) * width; }
    int grad_head() const { return (depth + 1) * width + (view_dependent ? views_width : 0); }
    int grad_rows() const { return grad_head() + 32; }

    long long trunk_bias(int layer) const { return bias_offset + (long long)layer * width; }
    long long feature_bias() const { return bias_offset + (long long)depth * width; }
    long long views_bias() const { return bias_offset + (long long)(depth + 1) * width; }
    long long pts_out_w() const { return head_offset; }
    long long pts_out_b() const { return head_offset + (long long)pts_out_rows * width; }
    long long views_out_w() const { return pts_out_b() + 4; }
    long long views_out_b() const { return views_out_w() + (long long)views_out_rows * views_width; }
};

inline int build_plan(const snerf_mlp_desc* d, MlpPlan* p) {
    if (!d) return fail(SNERF_E_INVALID, "mlp: NULL descriptor");
    MlpPlan plan;
    plan.depth = d->points_net_depth;
    plan.width = d->points_net_width;
    plan.view_dependent = d->view_dependent_rgb != 0;
    plan.views_width = plan.view_dependent ? d->views_net_width : 0;
    plan.points_degree = d->points_pe_degree;
    plan.views_degree = plan.view_dependent ? d->views_pe_degree : 0;
    plan.sigma_pe = d->sigma_pe_degree >= 0;
    if (plan.width != 128 && plan.width != 256)
        return fail(SNERF_E_UNSUPPORTED, "mlp: points_net_width %d not built (128 or 256)", plan.width);
    if (plan.depth < 1 || plan.depth == 5 || plan.depth > 64)
        return fail(SNERF_E_UNSUPPORTED, "mlp: points_net_depth %d unsupported (the reference itself cannot build depth 5: "
                    "the skip concat would feed pts_output_linear)", plan.depth);
    if (plan.points_degree < 1 || plan.points_degree > kMaxPointsDegree)
        return fail(SNERF_E_UNSUPPORTED, "mlp: points positional-encoding degree %d outside 1..%d", plan.points_degree,
                    kMaxPointsDegree);
    if (plan.view_dependent) {
        if (!d->use_view_dirs) return fail(SNERF_E_UNSUPPORTED, "mlp: view_dependent_rgb without use_view_dirs");
        if (d->views_net_depth != 1)
            return fail(SNERF_E_UNSUPPORTED, "mlp: views_net_depth %d not built (1)", d->views_net_depth);
        if (plan.views_width != 64 && plan.views_width != 128)
            return fail(SNERF_E_UNSUPPORTED, "mlp: views_net_width %d not built (64 or 128)", plan.views_width);
        if (plan.views_degree < 1 || plan.views_degree > kMaxViewsDegree)
            return fail(SNERF_E_UNSUPPORTED, "mlp: views positional-encoding degree %d outside 1..%d", plan.views_degree,
                        kMaxViewsDegree);
    }
    plan.full_pe = 3 + 6 * plan.points_degree;
    plan.pts_in = plan.full_pe;
    if (plan.sigma_pe) {
        if (d->sigma_pe_degree > plan.points_degree)
            return fail(SNERF_E_UNSUPPORTED, "mlp: sigma encoding degree %d exceeds the points degree %d",
                        d->sigma_pe_degree, plan.points_degree);
        if (!plan.view_dependent)
            return fail(SNERF_E_UNSUPPORTED, "mlp: points_sigma_positional_encoding_degree needs view-dependent colour");
        plan.pts_in = (2 * d->sigma_pe_degree + 1) * 3;
        plan.extra = plan.full_pe - plan.pts_in;
    }
    if (d->predict_visibility) {
        if (!plan.view_dependent)
            return fail(SNERF_E_UNSUPPORTED, "mlp: predict_visibility is built for view_dependent_rgb MLPs only");
        plan.views_out_rows = 4;
    }
    plan.views_pe = plan.view_dependent ? 3 + 6 * plan.views_degree : 0;
    plan.wt = plan.width / 32;
    plan.vt = plan.views_width / 32;
    plan.pts_out_rows = plan.view_dependent ? 1 : 4;
    plan.num_params = 2 * plan.depth + 2 + (plan.view_dependent ? 6 : 0);

    long long off = 0;
    auto add = [&](int param, int ld, int out_dim, int tiles, int ksteps, int kind, int col_offset, int lo, int hi,
                   int degree) {
        Segment s{param, ld, out_dim, tiles, ksteps, kind, col_offset, lo, hi, degree, off, 0};
        plan.segments.push_back(s);
        off += (long long)ksteps * tiles * 64;
    };
    const int acc_ks = plan.width / 2;
    add(0, plan.pts_in, plan.width, plan.wt, kPointsKSteps, SEG_POINTS_PE, 0, 0, plan.pts_in, plan.points_degree);
    for (int l = 1; l < plan.depth; ++l) {
        const bool skip_in = (l == 5);
        const int ld = plan.width + (skip_in ? plan.pts_in : 0);
        if (skip_in)
            add(2 * l, ld, plan.width, plan.wt, kPointsKSteps, SEG_POINTS_PE, 0, 0, plan.pts_in, plan.points_degree);
        add(2 * l, ld, plan.width, plan.wt, acc_ks, SEG_ACC, skip_in ? plan.pts_in : 0, 0, 0, 0);
    }
    if (plan.view_dependent) {
        const int pf = 2 * plan.depth + 2, pv = pf + 2;
        add(pf, plan.width, plan.width, plan.wt, acc_ks, SEG_ACC, 0, 0, 0, 0);
        const int ldv = plan.width + plan.extra + plan.views_pe;
        add(pv, ldv, plan.views_width, plan.vt, acc_ks, SEG_ACC, 0, 0, 0, 0);
        if (plan.sigma_pe)
            add(pv, ldv, plan.views_width, plan.vt, kPointsKSteps, SEG_POINTS_PE, plan.width, plan.pts_in, plan.full_pe,
                plan.points_degree);
        add(pv, ldv, plan.views_width, plan.vt, kViewsKSteps, SEG_VIEWS_PE, plan.width + plan.extra, 0, 0,
            plan.views_degree);
    }
    off += (long long)kSlabKSteps * plan.wt * 64;  // prefetch runway (zeros)
    plan.weight_floats = off;
    plan.bias_offset = off;
    off += (long long)(plan.depth + 1) * plan.width + 128;
    plan.head_offset = off;
    off += (long long)plan.pts_out_rows * plan.width + 4 + 4LL * 128 + 4;
    off = (off + 1023) / 1024 * 1024;
    plan.dgrad_offset = off;
    {
        long long doff = off;
        auto addt = [&](int param, int ld, int out_dim, int in_tiles, int ksteps, int col_offset, int in_features) {
            Segment s{param, ld, out_dim, in_tiles, ksteps, SEG_ACC, col_offset, 0, in_features, 0, doff, 1};
            plan.dgrad_segments.push_back(s);
            doff += (long long)ksteps * in_tiles * 64;
        };
        if (plan.view_dependent) {
            const int pf = 2 * plan.depth + 2, pv = pf + 2;
            const int ldv = plan.width + plan.extra + plan.views_pe;
            addt(pv, ldv, plan.views_width, plan.wt, plan.views_width / 2, 0, plan.width);  // d feature  = Wv[:, :W]^T dYv
            addt(pf, plan.width, plan.width, plan.wt, acc_ks, 0, plan.width);               // d h_depth  = Wf^T dfeature
        }
        for (int l = plan.depth - 1; l >= 1; --l) {                                         // d h_l = W_l[:, hcols]^T dY_l
            const bool skip_in = (l == 5);
            addt(2 * l, plan.width + (skip_in ? plan.pts_in : 0), plan.width, plan.wt, acc_ks, skip_in ? plan.pts_in : 0,
                 plan.width);
        }
        doff += (long long)kSlabKSteps * plan.wt * 64;
        off = doff;
    }
    off = (off + 1023) / 1024 * 1024;
    plan.half_offset = off;
    {
        long long hoff = off;
        auto stage = [&](int tiles, std::initializer_list<MlpPlan::HalfSegment> segs) {
            MlpPlan::HalfStage st;
            st.tiles = tiles; st.nseg = 0; st.dst = hoff; st.bf16 = 0;
            int ks = 0;
            for (const auto& sg : segs) { st.seg[st.nseg++] = sg; ks += sg.ksteps; }
            st.unit_floats = ks * 512;
            hoff += (long long)tiles * st.unit_floats;
            plan.half_stages.push_back(st);
        };
        const int hk = plan.width / 16;
        const MlpPlan::HalfSegment pe_trunk0{0, plan.pts_in, plan.width, SEG_POINTS_PE, 4, 0, 0, plan.pts_in, plan.points_degree, 0};
        stage(plan.wt, {pe_trunk0});
        for (int l = 1; l < plan.depth; ++l) {
            const bool skip_in = (l == 5);
            const int ld = plan.width + (skip_in ? plan.pts_in : 0);
            const MlpPlan::HalfSegment hseg{2 * l, ld, plan.width, SEG_ACC, hk, skip_in ? plan.pts_in : 0, 0, 0, 0, 0};
            if (skip_in) {
                const MlpPlan::HalfSegment pseg{2 * l, ld, plan.width, SEG_POINTS_PE, 4, 0, 0, plan.pts_in, plan.points_degree, 0};
                stage(plan.wt, {pseg, hseg});
            } else {
                stage(plan.wt, {hseg});
            }
        }
        if (plan.view_dependent) {
            const int pf = 2 * plan.depth + 2, pv = pf + 2;
            const int ldv = plan.width + plan.extra + plan.views_pe;
            stage(plan.wt, {MlpPlan::HalfSegment{pf, plan.width, plan.width, SEG_ACC, hk, 0, 0, 0, 0, 0}});
            const MlpPlan::HalfSegment vfeat{pv, ldv, plan.views_width, SEG_ACC, hk, 0, 0, 0, 0, 0};
            const MlpPlan::HalfSegment vpe{pv, ldv, plan.views_width, SEG_POINTS_PE, 4, plan.width, plan.pts_in, plan.full_pe, plan.points_degree, 0};
            const MlpPlan::HalfSegment vview{pv, ldv, plan.views_width, SEG_VIEWS_PE, 2, plan.width + plan.extra, 0, 0, plan.views_degree, 0};
            if (plan.sigma_pe) stage(plan.vt, {vfeat, vpe, vview});
            else stage(plan.vt, {vfeat, vview});
        }
        hoff += 24 * 512;  // prefetch runway: one maximum-size unit of zeros
        plan.half_dgrad_offset = hoff;
        auto tstage = [&](int param, int ld, int out_dim, int ksteps, int col_offset) {
            MlpPlan::HalfStage st;
            st.tiles = plan.wt; st.nseg = 1; st.dst = hoff; st.bf16 = 0;
            st.seg[0] = MlpPlan::HalfSegment{param, ld, out_dim, SEG_ACC, ksteps, col_offset, 0, plan.width, 0, 1};
            st.unit_floats = ksteps * 512;
            hoff += (long long)st.tiles * st.unit_floats;
            plan.half_dgrad_stages.push_back(st);
        };
        if (plan.view_dependent) {
            const int pf = 2 * plan.depth + 2, pv = pf + 2;
            tstage(pv, plan.width + plan.extra + plan.views_pe, plan.views_width, plan.views_width / 16, 0);
            tstage(pf, plan.width, plan.width, hk, 0);
        }
        for (int l = plan.depth - 1; l >= 1; --l) {
            const bool skip_in = (l == 5);
            tstage(2 * l, plan.width + (skip_in ? plan.pts_in : 0), plan.width, hk, skip_in ? plan.pts_in : 0);
        }
        hoff += 24 * 512;
        plan.m16_offset = hoff;
        // only for the layout mlp_forward_m16.hip is built for (the view-dependent 8 x 256 main MLP): the other MLPs' copies
        // were never read, yet written by every re-pack -- after every optimiser step -- and paid for in stream memory
        // 
        const bool m16_layout = plan.view_dependent && !plan.sigma_pe && plan.depth == 8 && plan.wt == 8 && plan.vt == 4 &&
                                plan.views_out_rows == 3;
        if (m16_layout) {
            for (MlpPlan::HalfStage st : plan.half_stages) {
                st.dst = hoff;
                hoff += (long long)st.tiles * st.unit_floats;
                plan.m16_stages.push_back(st);
            }
            hoff += 24 * 512;  // prefetch runway
        }
        // the bf16 copies (compact: unit_floats / 2 per unit)
        auto compact = [&](const std::vector<MlpPlan::HalfStage>& from, std::vector<MlpPlan::HalfStage>& to) {
            const long long first = hoff;
            for (MlpPlan::HalfStage st : from) {
                st.dst = hoff;
                st.bf16 = 1;
                hoff += (long long)st.tiles * (st.unit_floats / 2);
                to.push_back(st);
            }
            hoff += 24 * 256;  // prefetch runway
            return first;
        };
        plan.bf_offset = compact(plan.half_stages, plan.bf_stages);
        plan.bf_dgrad_offset = compact(plan.half_dgrad_stages, plan.bf_dgrad_stages);
        plan.bf_m16_offset = hoff;
        if (m16_layout) plan.bf_m16_offset = compact(plan.m16_stages, plan.bf_m16_stages);
        off = hoff;
    }
    // one reserved block behind the streams: word 0 is OR-ed with kRangeWeight by snerf_mlp_pack when a weight of THIS buffer
    // does not fit fp16, and read by the fp16-mode kernels that consume the buffer (an fp32-mode model may hold such weights)
    plan.weight_range_word = (off + 63) / 64 * 64;
    plan.total_floats = plan.weight_range_word + 64;
    *p = plan;
    return SNERF_OK;
}

}  // namespace snerf
