// K3 (split-precision variant) -- the fused PE + MLP forward on the fp16 matrix cores with fp32-grade accuracy.
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
#include "mlp_device.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

struct HalfArgs {
    MlpArgs m;
    long long half_offset;
    int const_floats;  // biases + head weights: packed[bias_offset, bias_offset + const_floats), kept in LDS
};

constexpr int kUnitBufFloats = 22 * 512;  // largest unit: 22 k-steps x 2 KiB (views layer of the points-aug MLP)
constexpr int kUnitBuffers = 3;

__device__ __forceinline__ void wait_vmcnt(int n) {  // n is wave-uniform; the count must be an immediate
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
        case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
        case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// Weight stream L2 -> LDS by LDS-DMA through a ring of three unit buffers, TWO units ahead of the one being consumed:
// at fp16 rates one tile's MFMAs (~0.65 us) are shorter than the DMA's issue-to-landing time, so a single unit of
// run-ahead leaves the matrix pipe waiting.  A unit of k k-steps is 2k KiB-pieces (hi + lo fragment per k-step), k even,
// so each of the 4 waves issues exactly k/2 DMA instructions per unit and can wait with a COUNTED vmcnt that leaves the
// younger unit in flight (a plain __syncthreads() would drain it: its fence waits vmcnt(0) while LDS-DMA is pending).
struct UnitStream {
    const float* fetch_ptr;  // global address of the next unit to request
    const float* stream_base;
    float* lds;
    int slot;                // ring slot of the unit about to be consumed
    int lane, wave;

    const float* pend_src;   // unit being requested piecewise (one DMA instruction per call of fetch_piece)
    float* pend_dst;
    int pend_left;           // DMA instructions this wave still has to issue for it
    int issued;              // DMA instructions issued for the youngest requested unit (>= its piece count)

    // Branch-free on purpose: a conditional here would cut the unrolled MFMA loop into basic blocks and the fragment reads
    // could no longer be scheduled a k-step ahead.  Once the unit's pieces are all requested the same (last) piece is simply
    // requested again -- idempotent, and it only happens where a unit has more k-step pairs than its second successor has
    // pieces (a few times per pass).  `issued` feeds the counted vmcnt of the next acquire().
    __device__ __forceinline__ void fetch_piece() {
#ifdef SNERF_ABL_NODMA
        pend_left -= pend_left > 0 ? 1 : 0;
        return;
#endif
        const int adv = pend_left > 0 ? 1024 : 0;
        pend_src += adv; pend_dst += adv;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pend_src + lane * 4),
                                         (__attribute__((address_space(3))) void*)pend_dst, 16, 0, 0);
        pend_left -= pend_left > 0 ? 1 : 0;
        ++issued;
    }
    __device__ __forceinline__ void finish_fetch() {
        while (pend_left > 0) fetch_piece();
    }
    // No further unit to request: the (branch-free) fetch_piece calls of the remaining k-steps re-read one valid KiB of
    // the stream into a per-wave dump area instead of touching a live buffer.
    __device__ __forceinline__ void issued_next_none() {
        pend_src = stream_base;
        pend_dst = lds + kUnitBuffers * kUnitBufFloats + wave * 256;  // dump: 4 KiB right after the ring
        pend_left = 0;
        issued = 0;
    }
    __device__ __forceinline__ void begin_fetch(int ksteps, int into_slot) {
        pend_src = fetch_ptr + wave * 256 - 1024;   // fetch_piece pre-increments
        pend_dst = lds + into_slot * kUnitBufFloats + wave * 256 - 1024;
        pend_left = ksteps >> 1;
        issued = 0;
        fetch_ptr += ksteps * 512;
    }
    __device__ __forceinline__ void fetch(int ksteps, int into_slot) {
        begin_fetch(ksteps, into_slot);
        finish_fetch();
    }
    __device__ __forceinline__ void start(const float* first, float* lds_base, int ks0, int ks1, int lane_, int wave_) {
        fetch_ptr = first; stream_base = first; lds = lds_base; slot = 0; lane = lane_; wave = wave_; pend_left = 0; issued = 0;
        fetch(ks0, 0);
        if (ks1 > 0) fetch(ks1, 1);
        if (ks1 <= 0) issued_next_none();
    }
    // Unit i becomes readable.  `next` = k-steps of unit i+1 (still in flight afterwards), `next2` = k-steps of unit i+2,
    // which is requested now into the slot unit i-1 just vacated (0 = no such unit).
    // The request for unit i+2 is only OPENED here; its DMA instructions are issued one per two k-steps from inside the
    // MFMA loop (fetch_piece) so that their issue cost (~60-100 cycles each) does not sit in front of the tile's MFMAs.
    __device__ __forceinline__ const float* acquire(int next, int next2) {
        finish_fetch();                                         // (units shorter than their successor's piece count)
        wait_vmcnt(next > 0 ? issued : 0);                      // everything older than unit i+1's requests has landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // my LDS reads of unit i-1 are complete
#ifndef SNERF_ABL_NOBARRIER
        __builtin_amdgcn_s_barrier();
#endif
        const float* ready = lds + slot * kUnitBufFloats;
        const int vacated = slot == 0 ? kUnitBuffers - 1 : slot - 1;
        if (next2 > 0) begin_fetch(next2, vacated); else issued_next_none();
        slot = slot == kUnitBuffers - 1 ? 0 : slot + 1;
        return ready;
    }
};

// Converts accumulator registers (2i, 2i+1) of a finished tile -- ReLU optional -- into the fp16 hi/lo pair they form
// in the next layer's operand: registers 8s..8s+7 are the 8 elements of k-step s.
template <bool RELU>
struct TileSplitter {
    const f32x16* src;
    f16x8 *h0, *l0, *h1, *l1;
    bool on;
    __device__ __forceinline__ void step(int i) const {  // i = 0..7 (compile-time after unrolling)
        if (!on) return;
#ifdef SNERF_ABL_NOSPLIT
        if (i > 0) return;
#endif
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int r = 2 * i + e;
            float v = (*src)[r];
            if (RELU) v = fmaxf(v, 0.0f);
            const _Float16 hi = (_Float16)v;
            const _Float16 lo = (_Float16)(v - (float)hi);
            if (r < 8) { (*h0)[r] = hi; (*l0)[r] = lo; } else { (*h1)[r - 8] = hi; (*l1)[r - 8] = lo; }
        }
    }
};

// acc += W[tile rows, segment columns] . X over NKS k-steps; `p` walks the unit (lane offset already applied).
// Fragments for k-step ks+1 are requested before the MFMAs of k-step ks are issued (LDS latency hides under 96 MFMA
// cycles), and `side.step()` slots one slice of the previous tile's ReLU + hi/lo split (VALU) behind each k-step's MFMAs.
template <int NKS, int NB, typename Side>
__device__ __forceinline__ void seg_mfma(f32x16& acc, const float*& p, const f16x8 (&bh)[NB], const f16x8 (&bl)[NB],
                                         const Side& side, int side_first, UnitStream& st) {
    static_assert(NB >= NKS, "operand array too short");
    f16x8 ah = *reinterpret_cast<const f16x8*>(p);
    f16x8 al = *reinterpret_cast<const f16x8*>(p + 256);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        f16x8 nah = ah, nal = al;
#ifndef SNERF_ABL_NOLDSREAD
        if (ks + 1 < NKS) {
            nah = *reinterpret_cast<const f16x8*>(p + (ks + 1) * 512);
            nal = *reinterpret_cast<const f16x8*>(p + (ks + 1) * 512 + 256);
        }
#endif
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[ks], acc, 0, 0, 0);
        if (side_first + ks < 8) side.step(side_first + ks);
        if ((ks & 1) == 0) st.fetch_piece();
        ah = nah; al = nal;
        // Pin the issue order per k-step: the two fragment reads of the NEXT k-step, then this k-step's three MFMAs, then
        // the VALU slice.  Left to itself the scheduler (at the 256-VGPR ceiling) serialises read -> wait -> MFMA through
        // one register quad and exposes the LDS latency on every k-step.
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // DS read
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);  // MFMA
        if ((ks & 1) == 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // VMEM read (the LDS-DMA piece)
        __builtin_amdgcn_sched_group_barrier(0x002, 12, 0); // VALU
    }
    p += NKS * 512;
}

struct NoSide {
    __device__ __forceinline__ void step(int) const {}
};

__device__ __forceinline__ void tile_bias(f32x16& acc, const float* __restrict__ bias, int half) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(bias + 8 * g + 4 * half);
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[4 * g + q] = v[q];
    }
}

// (ReLU and) split one finished accumulator tile into the two k-steps it feeds in the next layer.
template <bool RELU>
__device__ __forceinline__ void split_tile(const f32x16& acc, f16x8& h0, f16x8& l0, f16x8& h1, f16x8& l1) {
#ifdef SNERF_ABL_NOCONVERT
    h0[0] = (_Float16)acc[0]; l0[0] = (_Float16)acc[1]; h1[0] = (_Float16)acc[8]; l1[0] = (_Float16)acc[9];
    return;
#endif
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float a = acc[j], b = acc[8 + j];
        if (RELU) { a = fmaxf(a, 0.0f); b = fmaxf(b, 0.0f); }
        const _Float16 ah = (_Float16)a, bh = (_Float16)b;
        h0[j] = ah; l0[j] = (_Float16)(a - (float)ah);
        h1[j] = bh; l1[j] = (_Float16)(b - (float)bh);
    }
}

// sum over the tile's 32 features (this lane half's 16) of w[f] * relu(acc)
__device__ __forceinline__ float tile_dot_relu(const f32x16& acc, const float* __restrict__ w, int half) {
    float s = 0.0f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(w + 8 * g + 4 * half);
#pragma unroll
        for (int q = 0; q < 4; ++q) s = fmaf(v[q], fmaxf(acc[4 * g + q], 0.0f), s);
    }
    return s;
}

template <int NREG, int NKS>
__device__ __forceinline__ void split_encoding(const float (&pe)[NREG], f16x8 (&h)[NKS], f16x8 (&l)[NKS]) {
    static_assert(NREG == NKS * 8, "8 encoding registers per k-step");
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = pe[8 * ks + j];
            const _Float16 hi = (_Float16)v;
            h[ks][j] = hi;
            l[ks][j] = (_Float16)(v - (float)hi);
        }
}

template <int WT, int VT, bool VIEWDEP, bool SIGMA_PE>
__global__ void __launch_bounds__(256, 1) mlp_forward_f16x3_kernel(HalfArgs args) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const MlpArgs& a = args.m;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5;
    constexpr int HK = WT * 2;  // k-steps of a full-width activation

    // k-steps of staging unit `idx` in stream order: trunk layers (WT units each), feature stage (WT), views layer (VT)
    constexpr int kViewsKs = HK + (SIGMA_PE ? 4 : 0) + 2;
    const int trunk_units = a.depth * WT;
    auto ks_of = [&](int idx) {
        if (idx < trunk_units) {
            const int l = idx / WT;
            return l == 0 ? 4 : (l == 5 ? 4 + HK : HK);
        }
        if (!VIEWDEP) return 0;
        const int v = idx - trunk_units;
        return v < WT ? HK : (v < WT + VT ? kViewsKs : 0);
    };
    UnitStream st;
    st.start(a.packed + args.half_offset, lds, ks_of(0), ks_of(1), lane, wave);
    int unit_idx = 0;
    auto next_unit = [&]() {
        const float* p = st.acquire(ks_of(unit_idx + 1), ks_of(unit_idx + 2));
        ++unit_idx;
        return p + lane * 4;
    };
    // Biases and head weights live in LDS for the whole kernel: an ordinary global load inside the tile loop would make
    // the compiler wait vmcnt(0), i.e. drain the weight prefetch (LDS-DMA) that is deliberately left in flight.
    float* consts = lds + kUnitBuffers * kUnitBufFloats + 1024;  // after the ring and the 4-KiB DMA dump area
    for (int i = threadIdx.x * 4; i < args.const_floats; i += 256 * 4)
        *reinterpret_cast<f32x4*>(consts + i) = *reinterpret_cast<const f32x4*>(a.packed + a.bias_offset + i);
    __syncthreads();

    const long long first = ((long long)blockIdx.x * 4 + wave) * 32 + (lane & 31);
    const bool live = first < a.total;
    const long long g = live ? first : a.total - 1;
    const long long ray = g / a.samples;
    const float z = a.depths[g];
    float x[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) x[k] = a.origins[ray * 3 + k] + a.dirs[ray * 3 + k] * z;

    f16x8 pe_h[4], pe_l[4], pev_h[2], pev_l[2];
    {
        float pe[snerf::kPointsKSteps];
        encode<snerf::kPointsPairs, snerf::kPointsKSteps>(x, half, pe);
        split_encoding<32, 4>(pe, pe_h, pe_l);
    }
    if (VIEWDEP) {
        float v[3], pev[snerf::kViewsKSteps];
#pragma unroll
        for (int k = 0; k < 3; ++k) v[k] = a.view_dirs[ray * 3 + k];
        encode<snerf::kViewsPairs, snerf::kViewsKSteps>(v, half, pev);
        split_encoding<16, 2>(pev, pev_h, pev_l);
    }

    const float* bias = consts;
    const float* wout = consts + (a.pts_out_w - a.bias_offset);
    const float* bout = consts + (a.pts_out_b - a.bias_offset);
    f16x8 xh[HK], xl[HK];
    float head[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // density (and view-independent colour) pre-activations
    const bool single = a.depth == 1;

    // One accumulator tile per out tile of the layer (WT x 16 registers -- the matrix pipe's own AGPR file), and the
    // activations xh/xl in arch VGPRs where the MFMA reads them directly.  After the layer's last tile the accumulators
    // are ReLU'd and split into the next layer's operands in one VALU pass.  (An earlier variant overlapped that pass
    // with the next tile's MFMAs through a second operand buffer; the extra 128 registers pushed the B operands into
    // AGPRs and every MFMA then paid v_accvgpr_read moves -- slower overall.)
    f32x16 acc[WT];
    const NoSide none;
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
        seg_mfma<4>(acc[u], unit, pe_h, pe_l, none, 8, st);
        if (single) heads_from(acc[u], u);
    }
#pragma unroll
    for (int u = 0; u < WT; ++u) split_tile<true>(acc[u], xh[2 * u], xl[2 * u], xh[2 * u + 1], xl[2 * u + 1]);

    // ---- trunk layers 1 .. depth-1 --------------------------------------------------------------------------------
#pragma unroll 1
    for (int l = 1; l < a.depth; ++l) {
        const float* bl = bias + (long long)l * a.width;
        const bool last = l == a.depth - 1;
#pragma unroll
        for (int u = 0; u < WT; ++u) {
            const float* unit = next_unit();
            tile_bias(acc[u], bl + 32 * u, half);
            if (l == 5) seg_mfma<4>(acc[u], unit, pe_h, pe_l, none, 8, st);  // skip connection [encoding | h]
            seg_mfma<HK>(acc[u], unit, xh, xl, none, 8, st);
            if (last) heads_from(acc[u], u);
        }
#pragma unroll
        for (int u = 0; u < WT; ++u) split_tile<true>(acc[u], xh[2 * u], xl[2 * u], xh[2 * u + 1], xl[2 * u + 1]);
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
            seg_mfma<HK>(acc[u], unit, xh, xl, none, 8, st);
        }
#pragma unroll
        for (int u = 0; u < WT; ++u) split_tile<false>(acc[u], xh[2 * u], xl[2 * u], xh[2 * u + 1], xl[2 * u + 1]);
        // views layer over [feature | rest of the point encoding (points-aug) | view encoding], then the colour head
        const float* bv = consts + (a.views_bias - a.bias_offset);
        const float* wv = consts + (a.views_out_w - a.bias_offset);
        const float* bo = consts + (a.views_out_b - a.bias_offset);
        float col[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int u = 0; u < VT; ++u) {
            const float* unit = next_unit();
            tile_bias(acc[u], bv + 32 * u, half);
            seg_mfma<HK>(acc[u], unit, xh, xl, none, 8, st);
            if (SIGMA_PE) seg_mfma<4>(acc[u], unit, pe_h, pe_l, none, 8, st);
            seg_mfma<2>(acc[u], unit, pev_h, pev_l, none, 8, st);
#pragma unroll
            for (int c = 0; c < 3; ++c) col[c] += tile_dot_relu(acc[u], wv + c * VT * 32 + 32 * u, half);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) rgb[c] = sigmoidf((col[c] + __shfl_xor(col[c], 32, 64)) + bo[c]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (live && half == 0) {
        a.sigma[first] = sigma;
        a.rgb[first * 3 + 0] = rgb[0];
        a.rgb[first * 3 + 1] = rgb[1];
        a.rgb[first * 3 + 2] = rgb[2];
    }
}

template <int WT, int VT, bool VIEWDEP, bool SIGMA_PE>
int launch_half(const HalfArgs& args, hipStream_t stream) {
    const long long blocks = (args.m.total + 127) / 128;
    if (blocks > 0x7fffffffLL) return snerf::fail(SNERF_E_UNSUPPORTED, "mlp_forward: too many samples in one call");
    const size_t lds_bytes = sizeof(float) * (kUnitBuffers * kUnitBufFloats + 1024 + (size_t)args.const_floats);
    auto kernel = mlp_forward_f16x3_kernel<WT, VT, VIEWDEP, SIGMA_PE>;
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)(sizeof(float) * (kUnitBuffers * kUnitBufFloats + 1024 + 5120)));
        if (e != hipSuccess) return snerf::fail(SNERF_E_HIP, "mlp_forward: hipFuncSetAttribute: %s", hipGetErrorString(e));
        configured = true;
    }
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(256), lds_bytes, stream, args);
    return snerf::check_launch("mlp_forward(f16x3)");
}

}  // namespace

namespace snerf {

// Called by snerf_mlp_forward for SNERF_PRECISION_F16X3 (argument checks already done there).
int mlp_forward_f16x3(const MlpPlan& plan, const MlpArgs& m, hipStream_t stream) {
    HalfArgs args;
    args.m = m;
    args.half_offset = plan.half_offset;
    args.const_floats = (int)((plan.dgrad_offset - plan.bias_offset + 3) / 4 * 4);  // biases + heads (+ alignment padding)
    if (args.const_floats > 5120) return fail(SNERF_E_UNSUPPORTED, "mlp_forward(f16x3): bias/head block of %d floats exceeds its LDS area", args.const_floats);
    for (const MlpPlan::HalfStage& st : plan.half_stages)
        if (st.unit_floats > kUnitBufFloats)
            return fail(SNERF_E_UNSUPPORTED, "mlp_forward(f16x3): staging unit of %d KiB exceeds the LDS buffer", st.unit_floats / 256);
    const int key = plan.wt * 100 + plan.vt * 10 + (plan.sigma_pe ? 1 : 0);
    switch (key) {
        case 840: return launch_half<8, 4, true, false>(args, stream);
        case 841: return launch_half<8, 4, true, true>(args, stream);
        case 800: return launch_half<8, 4, false, false>(args, stream);
        case 420: return launch_half<4, 2, true, false>(args, stream);
        case 421: return launch_half<4, 2, true, true>(args, stream);
        case 400: return launch_half<4, 2, false, false>(args, stream);
        default:
            return fail(SNERF_E_UNSUPPORTED, "mlp_forward(f16x3): width %d / views width %d combination not built", plan.width,
                        plan.views_width);
    }
}

}  // namespace snerf
