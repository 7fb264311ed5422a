// K3 (fp16 modes, inference) on v_mfma_f32_16x16x32_f16 -- the same fused PE + MLP forward as mlp_forward_f16.hip, same
// arithmetic per product (P = 3: hi.hi + hi.lo + lo.hi; P = 1: hi.hi), same unit stream mechanics, different MFMA shape.
//
// Why a second shape.  Every fp16-mode kernel runs the board at its 1400 W cap, so what counts is work per
// joule.  Bare loops over the same data (tools/probes/mfma_shape_power.hip: fragment from LDS, operands in registers,
// accumulation as here) deliver 1.55 PFLOP/s with v_mfma_f32_32x32x16_f16 and 1.70 PFLOP/s with v_mfma_f32_16x16x32_f16 at
// the same 1355 W and the same 2.3 GHz: four independent 16-cycle accumulator chains keep the pipe fuller than one
// 32-cycle chain.  The training kernels stay on the 32x32x16 layout their saved tensors are written in.
//
// Layout.  A wave still owns 32 samples and accumulates one 32-row out tile over all its k-steps before the next; the
// tile is four 16x16 accumulators acc[r][s] (row half r, sample half s; lane (n = lane & 15, g = lane >> 4) holds rows
// 16r + 4g .. +3 of sample 16s + n).  A 1-KiB weight fragment is 16 out rows x 32 inputs (k-block c, row half r) and feeds
// two MFMAs, one per sample half.  The B operand of k-block c of the NEXT layer is, per lane, {acc[0][s] (4 values),
// acc[1][s] (4 values)} of out tile c -- again no data movement between layers; mlp_pack.hip orders the weights' k slots
// to match (MlpPlan::m16_stages).  The encodings are computed in the 32x32 lane mapping (one sample per lane, no
// redundancy) and redistributed once through LDS.
//
// Built for the view-dependent 8 x 256 main MLP (what rendering evaluates); everything else takes mlp_forward_f16.hip.
// Bound: MFMA fp16 under the board power cap.
#include <algorithm>
#include <type_traits>

#include "clock_stamp.h"
#include "mlp_device_f16.h"

namespace {

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

SNERF_STAMP_DEFINE(forward_m16)

struct M16Args {
    MlpArgs m;
    long long stream_offset;   // MlpPlan::m16_offset
    int const_floats;
    int slot_floats;
};

typedef f32x4 Tile16[2][2];   // [row half][sample half]

// Ring slots of the weight stream.  The single-product kernels (two waves per SIMD on one ring) run THREE units ahead since
// round 5 -- measured neutral (0.278 against 0.281 ms per 262 144 samples: a build that never waits for its DMA is no faster
// either, what costs 14 % is ISSUING it; tools/probes/m16_ablation.py, profiles/r05_m16_ablation.txt) and kept for what it
// frees: the encodings' hand-over scratch (48 KiB, dead after the prologue) now lies in slots three and four, 124 instead of
// 148 KiB of LDS per workgroup.
__host__ __device__ constexpr int m16_ring(int products) { return products == 1 ? 4 : 3; }

__device__ __forceinline__ f32x4 mfma16(const f16x8& a, const f16x8& b, const f32x4& c) {
    return mfma_16x16x32<false>(a, b, c);
}

// acc += W[tile rows, NB k-blocks] . X; `p` walks the unit's fragments (lane offset applied); fragment 2c + r.
// P = 3: fragments of k-step f+1 are requested before the MFMAs of fragment f (counted wait, as seg_mfma).
template <int NB, typename Stream>
__device__ __forceinline__ void seg3_m16(Tile16& t, const float*& p, int unit_ks, const f16x8 (&bh)[NB][2], const f16x8 (&bl)[NB][2],
                                         Stream& st) {
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
    const unsigned base_lo = base + unit_ks * 1024;
    f16x8 ah = lds_read_f16x8(base, 0);
    f16x8 al = lds_read_f16x8(base_lo, 0);
#pragma unroll
    for (int f = 0; f < 2 * NB; ++f) {
        f16x8 nah = ah, nal = al;
        if (f + 1 < 2 * NB) {
            nah = lds_read_f16x8(base, (f + 1) * 1024);
            nal = lds_read_f16x8(base_lo, (f + 1) * 1024);
            lds_wait_all_but_two(ah, al);
        } else {
            lds_wait_all(ah, al);
        }
        const int c = f >> 1, r = f & 1;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            t[r][s] = mfma16(ah, bh[c][s], t[r][s]);
            t[r][s] = mfma16(ah, bl[c][s], t[r][s]);
            t[r][s] = mfma16(al, bh[c][s], t[r][s]);
        }
        if ((f & 1) == 0) st.fetch_piece();
        ah = nah; al = nal;
    }
    p += 2 * NB * 256;
}
// P = 1: one product per fragment and sample half.  Fragments are handled in PAIRS (the two row halves of a k-block): the
// pair after next is requested before the wait for this one -- one counted wait per four MFMAs.
__device__ __forceinline__ void lds_pair_landed(f16x8& a, f16x8& b, int newer) {   // `newer` folds to a constant
    if (newer >= 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(a), "+v"(b)::"memory");
    else if (newer == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a), "+v"(b)::"memory");
    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b)::"memory");
}
template <int NB, typename Stream, bool BF>
__device__ __forceinline__ void seg1_m16(Tile16& t, const float*& p, const f16x8 (&bh)[NB][2], Stream& st) {
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
    f16x8 a[3][2];   // k-blocks c, c+1, c+2 in rotation
#pragma unroll
    for (int c = 0; c < 2 && c < NB; ++c)
#pragma unroll
        for (int r = 0; r < 2; ++r) a[c][r] = lds_read_f16x8(base, (2 * c + r) * 1024);
#pragma unroll
    for (int c = 0; c < NB; ++c) {
        if (c + 2 < NB) {
#pragma unroll
            for (int r = 0; r < 2; ++r) a[(c + 2) % 3][r] = lds_read_f16x8(base, (2 * (c + 2) + r) * 1024);
        }
        const int newer = 2 * ((NB - 1 - c) < 2 ? (NB - 1 - c) : 2);   // fragment reads issued after this pair's
        lds_pair_landed(a[c % 3][0], a[c % 3][1], newer);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            t[r][0] = mfma_16x16x32<BF>(a[c % 3][r], bh[c][0], t[r][0]);
            t[r][1] = mfma_16x16x32<BF>(a[c % 3][r], bh[c][1], t[r][1]);
        }
        // (Every wave issues its DMA instructions behind the same k-blocks.  Spreading them -- wave w behind k-blocks w / 2 and
        // w / 2 + 4, so that the eight waves do not queue on the CU's one vector-memory path -- needs a wave-dependent branch in
        // this unrolled loop and measured 15 % SLOWER (profiles/r05_m16_ablation.txt, 'staggered issue'): the loop must stay one
        // basic block, as UnitStreamT::fetch_piece says.)
        if (((2 * c) & (Stream::kWaves - 1)) == 0) st.fetch_piece();
    }
    p += 2 * NB * 256;
}
template <int P, int NB, bool BF, typename Stream>
__device__ __forceinline__ void seg_m16(Tile16& t, const float*& p, int unit_ks, const f16x8 (&bh)[NB][2], const f16x8 (&bl)[NB][2],
                                        Stream& st) {
    if constexpr (P == 3) seg3_m16<NB>(t, p, unit_ks, bh, bl, st);
    else seg1_m16<NB, Stream, BF>(t, p, bh, st);
}

// rows 32u + 16r + 4g .. +3 of a per-feature vector (bias, head weights) for this lane's group g
__device__ __forceinline__ void tile_bias16(Tile16& t, const float* __restrict__ bias, int grp) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(bias + 16 * r + 4 * grp);
        t[r][0] = v;
        t[r][1] = v;
    }
}
// sum over this lane's 8 rows of w[row] * relu(t[row]) for each sample half
__device__ __forceinline__ void tile_dot_relu16(const Tile16& t, const float* __restrict__ w, int grp, float (&sum)[2]) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(w + 16 * r + 4 * grp);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int q = 0; q < 4; ++q) sum[s] = fmaf(v[q], fmaxf(t[r][s][q], 0.0f), sum[s]);
    }
    // evaluated HERE: left alone, the scheduler sinks these chains to the end of the kernel (nothing needs the heads before)
    // and spills the tiles to scratch to get there
    asm volatile("" : "+v"(sum[0]), "+v"(sum[1]));
}
// finished out tile -> the operand fragments of k-block (= tile index) of the next layer, per sample half
template <bool RELU, int P, bool BF>
__device__ __forceinline__ void tile_to_operand16(const Tile16& t, f16x8 (&h)[2], f16x8 (&l)[2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int q = 0; q < 4; q += 2) {
                f32x2 a = {t[r][s][q], t[r][s][q + 1]};
                if (RELU && P == 3) a = {fmaxf(a[0], 0.0f), fmaxf(a[1], 0.0f)};
                // (P = 1, fp16: relu(fp16(v)) == fp16(relu(v)), one packed maximum per pair -- pack_pair)
                const f16x2 ah = P == 1 ? pack_pair<BF, RELU>(a) : __builtin_convertvector(a, f16x2);
                h[s][4 * r + q] = ah[0]; h[s][4 * r + q + 1] = ah[1];
                if constexpr (P == 3) {
                    const f16x2 al = __builtin_convertvector(a - __builtin_convertvector(ah, f32x2), f16x2);
                    l[s][4 * r + q] = al[0]; l[s][4 * r + q + 1] = al[1];
                }
            }
}

template <int P, int DEPTH, bool BF = false>
__global__ void __launch_bounds__(P == 1 ? 512 : 256, P == 1 ? 2 : 1) mlp_forward_m16_kernel(M16Args args) {
    static_assert(!BF || P == 1, "bf16 operands: single-product kernels only");
    constexpr int NW = P == 1 ? 8 : 4, WT = 8, VT = 4, HB = WT;   // HB = k-blocks of a full-width activation
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const MlpArgs& a = args.m;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n16 = lane & 15, grp = lane >> 4;

    // fragments (= KiB-pieces, the "k-steps" of UnitStreamT) of staging unit idx: trunk layers, feature stage, views layer
    constexpr int kViewsKs = 2 * HB + 2;
    constexpr int trunk_units = DEPTH * WT;
    auto ks_of = [&](int idx) {
        if (idx < trunk_units) {
            const int l = idx / WT;
            return l == 0 ? 4 : (l == 5 ? 4 + 2 * HB : 2 * HB);
        }
        const int v = idx - trunk_units;
        return v < WT ? 2 * HB : (v < WT + VT ? kViewsKs : 0);
    };
    constexpr int RING = m16_ring(P);
    UnitStreamT<P, NW, RING, BF ? 256 : 512> st;
    st.start(a.packed + args.stream_offset, lds, ks_of(0), ks_of(1), lane, wave, args.slot_floats);
    int unit_idx = 0;
    auto next_unit = [&]() {
        const float* p = st.acquire(ks_of(unit_idx + 1), ks_of(unit_idx + RING - 1));
        ++unit_idx;
        return p + lane * 4;
    };
    float* consts = lds + RING * args.slot_floats + NW * 256;  // after the ring and the DMA dump area (1 KiB per wave)
    for (int i = threadIdx.x * 4; i < args.const_floats; i += NW * 64 * 4)
        *reinterpret_cast<f32x4*>(consts + i) = *reinterpret_cast<const f32x4*>(a.packed + a.bias_offset + i);

    // ---- encodings: computed with one sample per lane (lane & 31; lane half = k half of the 32x32 layout), then handed to
    // the lanes that need them through a 6-KiB LDS scratch per wave ----------------------------------------------------------
    f16x8 pe_h[2][2], pe_l[2][2], pev_h[1][2], pev_l[1][2];
    const long long wave_base = ((long long)blockIdx.x * NW + wave) * 32;
#ifdef SNERF_PROBE_M16_NOENCODE
    // timing probe (wrong results): the operands of the encodings are lane-dependent constants -- no positions, no sin / cos, no
    // hand-over through LDS: what the 861-instruction prologue costs a pass
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                pe_h[c][s][j] = (_Float16)(0.01f * ((lane * 7 + c * 3 + s + j) & 63) - 0.3f);
                pe_l[c][s][j] = (_Float16)0.0f;
            }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) { pev_h[0][s][j] = (_Float16)(0.02f * ((lane + s + j) & 31) - 0.3f); pev_l[0][s][j] = (_Float16)0.0f; }
#else
    {
        const int i32 = lane & 31, half = lane >> 5;
        const long long gi = wave_base + i32 < a.total ? wave_base + i32 : a.total - 1;
        const long long ray = gi / a.samples;
        const float z = a.depths[gi];
        float x[3], v[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) x[k] = a.origins[ray * 3 + k] + a.dirs[ray * 3 + k] * z;
#pragma unroll
        for (int k = 0; k < 3; ++k) v[k] = a.view_dirs[ray * 3 + k];
        float pe[snerf::kPointsKSteps], pev[snerf::kViewsKSteps];
#ifdef SNERF_PROBE_M16_UNPAIRED     // A/B build: the encoding as before the opaque pair (contains the hazardous packed form)
        constexpr bool kPaired = false;
#else
        constexpr bool kPaired = true;     // (see opaque_pair, mlp_device.h)
#endif
        encode<snerf::kPointsPairs, snerf::kPointsKSteps, kPaired>(x, half, pe);
        encode<snerf::kViewsPairs, snerf::kViewsKSteps, kPaired>(v, half, pev);
        // register 8ks + j of lane half h = position p = 16 (ks & 1) + 8h + j of k-block ks / 2 -> lane group (p & 15) / 4,
        // slot (p < 16 ? 0 : 4) + p % 4 of the fragment of sample half i32 / 16
        // (the ring slots from the third on are idle until the hand-over is done: P = 3: the third slot (44 KiB for 4 waves x 6 KiB);
        // P = 1: slots three and four of its 4-slot ring, 2 x 24 KiB for 8 waves x 6 KiB -- launch_m16 checks the sizes)
        _Float16* scratch = reinterpret_cast<_Float16*>(lds + 2 * args.slot_floats) + wave * (6 * 512);
        auto place = [&](int block, int ks, int j) {
            const int p = 16 * (ks & 1) + 8 * half + j;
            const int g = (p & 15) >> 2, t = (p < 16 ? 0 : 4) + (p & 3);
            return ((block * 2 + (i32 >> 4)) * 64 + 16 * g + (i32 & 15)) * 8 + t;
        };
        auto hand_over = [&](bool low) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float val = pe[8 * ks + j];
                    const _Float16 hi = pack_one<BF>(val);
                    scratch[place(ks >> 1, ks, j)] = low ? (_Float16)(val - (float)hi) : hi;
                }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float val = pev[8 * ks + j];
                    const _Float16 hi = pack_one<BF>(val);
                    scratch[place(2, ks, j)] = low ? (_Float16)(val - (float)hi) : hi;
                }
        };
        const f16x8* frags = reinterpret_cast<const f16x8*>(scratch) + lane;
        hand_over(false);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int s = 0; s < 2; ++s) pe_h[c][s] = frags[(c * 2 + s) * 64];
#pragma unroll
        for (int s = 0; s < 2; ++s) pev_h[0][s] = frags[(4 + s) * 64];
        if constexpr (P == 3) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            hand_over(true);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int s = 0; s < 2; ++s) pe_l[c][s] = frags[(c * 2 + s) * 64];
#pragma unroll
            for (int s = 0; s < 2; ++s) pev_l[0][s] = frags[(4 + s) * 64];
        }
    }
#endif
    __syncthreads();   // consts visible; every wave is done with the scratch before the DMA is given its slots
    if constexpr (RING > 3) {      // the rest of the initial run-ahead, now that the scratch is free
#pragma unroll
        for (int u = 2; u < RING - 1; ++u) st.start_more(u, ks_of(u));
    }
    SNERF_STAMP_BEGIN();

    const float* bias = consts;
    const float* wout = consts + (a.pts_out_w - a.bias_offset);
    const float* bout = consts + (a.pts_out_b - a.bias_offset);
    f16x8 xh[HB][2], xl[HB][2];
    Tile16 acc[WT];
    float head[2] = {0.0f, 0.0f};
    RangeWatch watch;   // one pre-activation per layer for each of this lane's two samples (mlp_device_f16.h)
    auto probe_tile = [&](const Tile16& t) __attribute__((always_inline)) {
        if constexpr (!BF) { watch.probe(t[0][0][0]); watch.probe(t[0][1][0]); }      // (bf16 has fp32's range: nothing to watch)
    };

    // ---- trunk layer 0: encoding -> h ---------------------------------------------------------------------------------
#pragma unroll
    for (int u = 0; u < WT; ++u) {
        const float* unit = next_unit();
        tile_bias16(acc[u], bias + 32 * u, grp);
        seg_m16<P, 2, BF>(acc[u], unit, 4, pe_h, pe_l, st);
        if (u == 0) probe_tile(acc[0]);           // non-finite iff an encoded input left the fp16 range
    }
#pragma unroll
    for (int u = 0; u < WT; ++u) tile_to_operand16<true, P, BF>(acc[u], xh[u], xl[u]);

    // ---- trunk layers 1 .. DEPTH-1 ---------------------------------------------------------------------------------------
    auto trunk_layer = [&](int l) __attribute__((always_inline)) {
        const float* bl = bias + l * (WT * 32);
#pragma unroll
        for (int u = 0; u < WT; ++u) {
            const float* unit = next_unit();
            tile_bias16(acc[u], bl + 32 * u, grp);
            const int unit_ks = l == 5 ? 4 + 2 * HB : 2 * HB;
            if (l == 5) seg_m16<P, 2, BF>(acc[u], unit, unit_ks, pe_h, pe_l, st);   // skip connection [encoding | h]
            seg_m16<P, HB, BF>(acc[u], unit, unit_ks, xh, xl, st);
            if (u == 0) probe_tile(acc[0]);       // non-finite iff an activation of layer l-1 left the fp16 range
            if (l == DEPTH - 1) tile_dot_relu16(acc[u], wout + 32 * u, grp, head);
        }
#pragma unroll
        for (int u = 0; u < WT; ++u) tile_to_operand16<true, P, BF>(acc[u], xh[u], xl[u]);
    };
    static_for<1, DEPTH>([&](auto layer) __attribute__((always_inline)) { trunk_layer(decltype(layer)::value); });

    // ---- feature = feature_linear(h): no activation ------------------------------------------------------------------------
    const float* bf = consts + (a.feature_bias - a.bias_offset);
#pragma unroll
    for (int u = 0; u < WT; ++u) {
        const float* unit = next_unit();
        tile_bias16(acc[u], bf + 32 * u, grp);
        seg_m16<P, HB, BF>(acc[u], unit, 2 * HB, xh, xl, st);
        if (u == 0) probe_tile(acc[0]);
    }
#pragma unroll
    for (int u = 0; u < WT; ++u) tile_to_operand16<false, P, BF>(acc[u], xh[u], xl[u]);
    // ---- views layer over [feature | view encoding], then the colour head ----------------------------------------------------
    const float* bv = consts + (a.views_bias - a.bias_offset);
    const float* wv = consts + (a.views_out_w - a.bias_offset);
    const float* bo = consts + (a.views_out_b - a.bias_offset);
    float col[3][2] = {{0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}};
#pragma unroll
    for (int u = 0; u < VT; ++u) {
        const float* unit = next_unit();
        tile_bias16(acc[u], bv + 32 * u, grp);
        seg_m16<P, HB, BF>(acc[u], unit, kViewsKs, xh, xl, st);
        seg_m16<P, 1, BF>(acc[u], unit, kViewsKs, pev_h, pev_l, st);
        if (u == 0) probe_tile(acc[0]);           // the feature vector and the view encoding
#pragma unroll
        for (int c = 0; c < 3; ++c) tile_dot_relu16(acc[u], wv + c * VT * 32 + 32 * u, grp, col[c]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (a wave never ends with LDS-DMA in flight)
    if constexpr (!BF) watch.report(a.range_flag, a.weight_range);
    SNERF_STAMP_END(forward_m16);

    // ---- outputs: the four lane groups hold partial sums over their rows ----------------------------------------------------
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const long long first = wave_base + 16 * s + n16;
        const bool live = first < a.total;
        float sg = head[s];
        sg += __shfl_xor(sg, 16, 64);
        sg += __shfl_xor(sg, 32, 64);
        float sigma = sg + bout[0];
        if (a.noise) sigma += a.noise[live ? first : a.total - 1];
        sigma = fmaxf(sigma, 0.0f);
        float rgb[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = col[c][s];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            rgb[c] = sigmoidf(v + bo[c]);
        }
        if (live && grp == 0) {
            a.sigma[first] = sigma;
            a.rgb[first * 3 + 0] = rgb[0];
            a.rgb[first * 3 + 1] = rgb[1];
            a.rgb[first * 3 + 2] = rgb[2];
        }
    }
}

template <int P, bool BF = false>
int launch_m16(const M16Args& args, hipStream_t stream) {
    constexpr int NW = P == 1 ? 8 : 4;
    const long long blocks = (args.m.total + NW * 32 - 1) / (NW * 32);
    if (blocks > 0x7fffffffLL) return snerf::fail(SNERF_E_UNSUPPORTED, "mlp_forward: too many samples in one call");
    constexpr int RING = m16_ring(P);
    // the encodings' scratch (NW x 6 KiB) lies in the ring's slots from the third on
    if ((size_t)(RING - 2) * args.slot_floats < (size_t)NW * 6 * 256)
        return snerf::fail(SNERF_E_UNSUPPORTED, "mlp_forward(m16): ring slots of %d floats cannot hold the encoding scratch", args.slot_floats);
    const size_t lds_bytes = sizeof(float) * (RING * (size_t)args.slot_floats + NW * 256 + (size_t)args.const_floats);
    auto kernel = mlp_forward_m16_kernel<P, 8, BF>;
    static snerf::DeviceOnce configured;   // per device: the attribute belongs to (kernel, device)
    const int attr = snerf::raise_dynamic_lds(configured, reinterpret_cast<const void*>(kernel), (int)(sizeof(float) * (kUnitBuffers * kUnitBufFloats + 2048 + 5120)), "mlp_forward");   // (P = 3: 132 + 8 + 20 KiB; P = 1: 4 x 24 + 8 + 20 KiB)
    if (attr != SNERF_OK) return attr;
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(NW * 64), lds_bytes, stream, args);
    return snerf::check_launch("mlp_forward(m16)");
}

}  // namespace

namespace snerf {

// Inference with the fp16 modes for the layout this file builds; -1 = the caller uses mlp_forward_f16.hip.
// `bf16`: the single-product kernel on bf16 operands (SNERF_PRECISION_BF16), reading the compact bf16 copy of the stream.
int mlp_forward_m16(const MlpPlan& plan, const MlpArgs& m, int products, hipStream_t stream, bool bf16) {
    if (!plan.view_dependent || plan.sigma_pe || plan.depth != 8 || plan.wt != 8 || plan.vt != 4 || plan.views_out_rows != 3) return -1;
    M16Args args;
    args.m = m;
    args.stream_offset = bf16 ? plan.bf_m16_offset : plan.m16_offset;
    args.const_floats = (int)((plan.dgrad_offset - plan.bias_offset + 3) / 4 * 4);
    if (args.const_floats > 5120) return -1;
    int most_ks = 0;
    for (const MlpPlan::HalfStage& st : plan.m16_stages) most_ks = std::max(most_ks, st.unit_floats / 512);
    if (most_ks * 512 > kUnitBufFloats) return -1;
    // P = 3: the encodings' scratch (4 waves x 6 KiB) borrows the third ring slot (44 KiB)
    args.slot_floats = products == 3 ? kUnitBufFloats : (most_ks + 7) / 8 * 8 * 256;
    if (bf16) return launch_m16<1, true>(args, stream);
    return products == 3 ? launch_m16<3>(args, stream) : launch_m16<1>(args, stream);
}

}  // namespace snerf
