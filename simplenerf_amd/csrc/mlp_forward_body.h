// K3's kernel body (see mlp_forward.hip for the design notes) -- shared by mlp_forward.hip and render_fused.hip.
#pragma once
#include "mlp_device.h"

namespace {

// ------------------------------------------------------------------------------------------------
// the kernel
// ------------------------------------------------------------------------------------------------
// VIS: predict_visibility -- the views head has a 4th row (sigmoid -> per-sample visibility) and is evaluated once more per
// secondary view direction: the views layer's pre-activation WITHOUT its view-encoding segment is kept, and that segment's
// slab (the last of the stream, still resident in LDS) is applied again to each secondary direction's encoding.
// The kernel body as a device function of (arguments, wave-block group `block`, LDS base): mlp_forward_kernel runs it once per
// workgroup with its block index; the fused render kernel (render_fused.hip) runs it once per 128-sample pass of a ray
// group, with `a.depths` / `a.sigma` / `a.rgb` pointing at the group's LDS-resident sample tile.  All four waves of the
// workgroup must call it together (it contains workgroup barriers).
template <int WT, int VT, bool VIEWDEP, bool SIGMA_PE, bool STORE, bool VIS = false>
__device__ __forceinline__ void mlp_forward_body(const MlpArgs& a, long long block, float* lds) {
    static_assert(!VIS || VIEWDEP, "the visibility row belongs to the views head");
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
    const long long first = (block * 4 + wave) * 32 + (lane & 31);
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
        tile = a.acts + (block * 4 + wave) * a.act_rows * 32;
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

}  // namespace
