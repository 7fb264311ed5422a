// The layered path of the NeRF MLP for shapes the register-resident fused kernels are not built for: any points_net_width
// (64, 96, 512, ...), any views_net_width, views_net_depth > 1, any depth -- everything MLP.__init__ can build
// (src/models/SimpleNeRF01.py:567-609) except predict_visibility.  Same C ABI, same parameter tensors, same semantics; the
// fused kernels keep every shape they cover (mlp_plan.h build_plan), this file takes the rest so that the reference's config
// keys are honoured instead of refused.
//
// Design.  A 512-wide layer does not fit the register-resident chain (16 accumulator tiles + 16 operand tiles per wave > the
// 512-register file), so activations live in memory here: one row per sample in an [N][row] fp32 matrix
//     [ encoding | view encoding | H_0 ... (the skip layer's input stored as ONE block [encoding | H_4]) ... H_D-1 |
//       views input block [feature | rest of the encoding | view encoding] | HV_0 ... | pts_output | views_output ]
// and every Linear layer is one launch of ONE strided GEMM kernel on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32
// FMA chains, the parity arithmetic): C(m,n) = sum_k A(m,k) B(k,n) with row / column strides for all three matrices, so the
// forward (X . W^T), the input gradient (dZ . W), the weight gradient (dZ^T . X, split over the samples with a fixed-order
// reduction) and the concatenations (column offsets into the row) are the same kernel.  64 x 64 output tile per 256-thread
// workgroup, K staged through LDS 32 at a time with the next stage's global loads in flight.  Bias, ReLU, the ReLU mask of the
// backward and accumulation are epilogue options.
//
// Bound: MFMA fp32 for the wide layers (2 x 4 B x width per sample and layer over HBM against 2 x width^2 FLOP: MFMA-bound from
// width ~100 up), at the efficiency of a plain LDS-tiled GEMM -- a fallback for generality, not the headline path.
// Always fp32 arithmetic: the fp16 / bf16 modes are refused for these shapes.  The inference entry point has no workspace
// argument in the ABI, so this path keeps a per-device scratch arena for it (allocated on first use, grown when needed: a
// FIRST generic inference call inside a graph capture fails with SNERF_E_HIP).
#include <algorithm>
#include <mutex>
#include <vector>

#include "mlp_device.h"
#include "mlp_generic.h"

namespace {

using snerf::GenericPlan;

// ---------------------------------------------------------------------------------------------------------------- GEMM
struct GemmArgs {
    const float* A; long long a_rs, a_cs;
    const float* B; long long b_rs, b_cs;
    float* C; long long c_rs, c_cs;
    const float* bias;                         // per column n, or NULL
    const float* mask; long long mask_rs;      // C(m,n) *= (mask[m * mask_rs + n] > 0), or NULL (ReLU gate of the backward)
    int M, N, K;
    int relu, accumulate;
    long long k_chunk;                         // split-K: blockIdx.z owns k in [z * k_chunk, min(K, (z+1) * k_chunk)) and writes
    long long split_stride;                    // C + z * split_stride (a partial-sum buffer); 0 = no split
};

constexpr int kBK = 32, kPad = 1;

// Accumulators -> C (+ bias, + C, ReLU, gate).  D layout: lane (i = column, h), register r -> row (r & 3) + 8 (r >> 2) + 4 h of the 32 x 32
// tile.  A tile that lies wholly inside the matrix takes the branch-free path: the sixteen values of a register tile are read (C to add
// to, the gate) in one batch, then computed, then stored -- with a bounds test per element every read was its own exec-masked block
// and waited for its own round trip (64 of them per thread in the input-gradient products of the backward).
template <int TM, int TN>
__device__ __forceinline__ void write_tile(const f32x16 (&acc)[TM][TN], const GemmArgs& g, float* C, int m0, int n0, int wm, int wn, int i,
                                           int h) {
    const bool whole = m0 + 64 * TM <= g.M && n0 + 64 * TN <= g.N;
    if (whole) {
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const long long n = n0 + wn + 32 * tn + i;
            const float bias = g.bias ? g.bias[n] : 0.0f;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const long long m_first = m0 + wm + 32 * tm + 4 * h;
                float* const dst = C + m_first * g.c_rs + n * g.c_cs;
                const float* const gate = g.mask ? g.mask + m_first * g.mask_rs + n : nullptr;
                float old[16], open[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long long dm = (r & 3) + 8 * (r >> 2);
                    old[r] = g.accumulate ? dst[dm * g.c_rs] : 0.0f;
                    open[r] = gate ? gate[dm * g.mask_rs] : 1.0f;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long long dm = (r & 3) + 8 * (r >> 2);
                    float v = acc[tm][tn][r] + bias;
                    if (g.accumulate) v += old[r];
                    if (g.relu) v = fmaxf(v, 0.0f);
                    if (g.mask) v = open[r] > 0.0f ? v : 0.0f;
                    dst[dm * g.c_rs] = v;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const long long n = n0 + wn + 32 * tn + i;
        if (n >= g.N) continue;
        const float bias = g.bias ? g.bias[n] : 0.0f;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long m = m0 + wm + 32 * tm + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m >= g.M) continue;
                float v = acc[tm][tn][r] + bias;
                float* dst = C + m * g.c_rs + n * g.c_cs;
                if (g.accumulate) v += *dst;
                if (g.relu) v = fmaxf(v, 0.0f);
                if (g.mask) v = g.mask[m * g.mask_rs + n] > 0.0f ? v : 0.0f;
                *dst = v;
            }
    }
}

// BM x BN output tile per 256-thread workgroup = a 2 x 2 grid of waves, each (BM / 2) x (BN / 2) = TM x TN MFMA tiles of 32 x 32.
// <64, 64>: one tile per wave (the round-4 kernel: 0.40 of the fp32 matrix peak at width 512 -- per staged byte and per barrier
// it does a quarter of the matrix work of) <128, 128> (round 5): four tiles per wave, every operand value read from LDS feeds
// two MFMAs, 64 KiB of LDS for the two stages; used wherever the product is at least 128 x 128.
template <int BM, int BN>
__global__ void __launch_bounds__(256, 2) gemm_kernel(GemmArgs g) {
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int LA = BM * kBK / 256, LB = BN * kBK / 256;      // elements of a stage per thread
    __shared__ float As[2][kBK][BM + kPad];
    __shared__ float Bs[2][kBK][BN + kPad];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // tile of this workgroup.  Workgroups go to the eight XCDs in turn (linear id mod 8) and every XCD has its own L2: with the
    // plain (x = n tile, y = m tile) order the n tiles of one row block -- which all read the same rows of A -- land on different
    // XCDs and A comes from HBM once per n tile (4 x at width 512).  When the m tiles divide by eight, XCD x takes the row blocks
    // congruent to x and walks their n tiles one after the other: A's rows are fetched once and hit in that XCD's L2 afterwards.
    int tile_m = blockIdx.y, tile_n = blockIdx.x;
#ifndef SNERF_PROBE_GEMM_PLAIN_ORDER
    if ((gridDim.y & 7u) == 0) {
        const unsigned linear = blockIdx.y * gridDim.x + blockIdx.x, xcd = linear & 7u, idx = linear >> 3;
        tile_m = (int)((idx / gridDim.x) * 8u + xcd);
        tile_n = (int)(idx % gridDim.x);
    }
#endif
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const long long k_lo = g.split_stride ? (long long)blockIdx.z * g.k_chunk : 0;
    const long long k_hi = g.split_stride ? (k_lo + g.k_chunk < g.K ? k_lo + g.k_chunk : g.K) : g.K;
    float* C = g.C + (g.split_stride ? (long long)blockIdx.z * g.split_stride : 0);

    // thread -> elements of a stage: LA of A (BM x kBK) and LB of B (kBK x BN), walking the contiguous dimension first.
    // Element e of a thread sits a FIXED step from element e - 1, in memory and in LDS: one 32-bit byte offset per operand and
    // thread (computed once), a wave-uniform base per stage and element (scalar arithmetic), immediate LDS offsets.  (Round 4
    // computed m * a_rs + k * a_cs in 64 bits per element and stage: 1 021 vector instructions beside 16 MFMAs, 2 785 beside 64
    // in the first 128 x 128 build -- the kernel was bound by its address arithmetic, not by the matrix pipe.)
    const bool a_m_major = g.a_rs == 1;        // A contiguous along m (the transposed operand of the weight gradient)
    const bool b_n_major = g.b_cs == 1;
    const int a_m = a_m_major ? (tid & (BM - 1)) : (tid / kBK), a_k = a_m_major ? (tid / BM) : (tid & (kBK - 1));
    const int a_dm = a_m_major ? 0 : 256 / kBK, a_dk = a_m_major ? 256 / BM : 0;            // element e + 1 relative to e
    const int b_n = b_n_major ? (tid & (BN - 1)) : (tid / kBK), b_k = b_n_major ? (tid / BN) : (tid & (kBK - 1));
    const int b_dn = b_n_major ? 0 : 256 / kBK, b_dk = b_n_major ? 256 / BN : 0;
    const unsigned a_byte = (unsigned)((long long)a_m * g.a_rs + (long long)a_k * g.a_cs) * 4u;   // (a_m < BM, a_k < 32: fits)
    const unsigned b_byte = (unsigned)((long long)b_k * g.b_rs + (long long)b_n * g.b_cs) * 4u;
    // (a thread's elements span < 16 steps of at most 8 rows or 4 k-columns: byte offsets inside the tile stay far below 2^31
    // for any row length the activation matrix can have)
    const unsigned a_step32 = (unsigned)(((long long)a_dm * g.a_rs + (long long)a_dk * g.a_cs) * 4);
    const unsigned b_step32 = (unsigned)(((long long)b_dk * g.b_rs + (long long)b_dn * g.b_cs) * 4);
    const int m_left = g.M - m0, n_left = g.N - n0;           // rows / columns of the tile inside the matrix
    const char* const a_tile = reinterpret_cast<const char*>(g.A + (long long)m0 * g.a_rs);
    const char* const b_tile = reinterpret_cast<const char*>(g.B + (long long)n0 * g.b_cs);
    float ra[LA], rb[LB];
    unsigned a_inside = 0, b_inside = 0;      // bit e: element e of the stage in flight lies inside the matrix
    auto load_stage = [&](long long k0) {
        const char* a_stage = a_tile + k0 * g.a_cs * 4;
        const char* b_stage = b_tile + k0 * g.b_rs * 4;
        const int k_left = (int)(k_hi - k0 < kBK ? k_hi - k0 : kBK);       // 32 except in the last stage of a ragged K
        // branch-free: an element outside the matrix reads the stage's first element (always inside) and is replaced by zero
        // WHEN IT IS STORED, after this stage's MFMAs (a predicated load per element compiles to one exec-masked basic block per
        // load, and a select right here makes the MFMAs wait for the loads they are meant to hide)
        a_inside = b_inside = 0;
#ifdef SNERF_PROBE_GEMM_NOLOAD       // timing ablation (wrong results): nothing fetched from memory
        (void)a_stage; (void)b_stage; (void)k_left;
        return;
#endif
#pragma unroll
        for (int e = 0; e < LA; ++e) {
            const bool inside = a_m + e * a_dm < m_left && a_k + e * a_dk < k_left;
            a_inside |= inside ? 1u << e : 0u;
            ra[e] = *reinterpret_cast<const float*>(a_stage + (inside ? a_byte + (unsigned)e * a_step32 : 0u));
        }
#pragma unroll
        for (int e = 0; e < LB; ++e) {
            const bool inside = b_n + e * b_dn < n_left && b_k + e * b_dk < k_left;
            b_inside |= inside ? 1u << e : 0u;
            rb[e] = *reinterpret_cast<const float*>(b_stage + (inside ? b_byte + (unsigned)e * b_step32 : 0u));
        }
    };
    float* const as0 = &As[0][a_k][a_m];
    float* const bs0 = &Bs[0][b_k][b_n];
    const int as_step = a_dk * (BM + kPad) + a_dm, bs_step = b_dk * (BN + kPad) + b_dn;
    auto store_stage = [&](int buf) {
#ifdef SNERF_PROBE_GEMM_NOSTORE      // timing ablation (wrong results): nothing staged into LDS
        return;
#endif
#pragma unroll
        for (int e = 0; e < LA; ++e) as0[buf * (kBK * (BM + kPad)) + e * as_step] = (a_inside >> e) & 1u ? ra[e] : 0.0f;
#pragma unroll
        for (int e = 0; e < LB; ++e) bs0[buf * (kBK * (BN + kPad)) + e * bs_step] = (b_inside >> e) & 1u ? rb[e] : 0.0f;
    };

    // wave (wm, wn) owns a (32 TM) x (32 TN) sub-tile; MFMA operands: A(m = lane & 31, k = lane >> 5), B(k = lane >> 5, n = lane & 31)
    const int wm = (wave >> 1) * (32 * TM), wn = (wave & 1) * (32 * TN), i = lane & 31, h = lane >> 5;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.0f;
    if (k_lo < k_hi) {
        load_stage(k_lo);
        store_stage(0);
        __syncthreads();
        int buf = 0;
        for (long long k0 = k_lo; k0 < k_hi; k0 += kBK) {
            const bool more = k0 + kBK < k_hi;
            if (more) load_stage(k0 + kBK);                     // in flight during this stage's MFMAs
            const float* a_rd = &As[buf][h][wm + i];
            const float* b_rd = &Bs[buf][h][wn + i];
            // the operands of k-pair p + 1 are requested BEFORE the MFMAs of pair p are issued (two register sets; the scheduling
            // barrier keeps the compiler from sinking the reads back to their use): left alone it emitted read, wait, four MFMAs,
            // read, wait ... with one register set, and the matrix pipe sat idle for an LDS round trip per pair -- 0.54 busy (PMC,
            // profiles/r05_pmc_layered_gemm.json), 0.50 of the peak
            float av[2][TM], bv[2][TN];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) av[0][tm] = a_rd[32 * tm];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) bv[0][tn] = b_rd[32 * tn];
#pragma unroll
            for (int p = 0; p < kBK / 2; ++p) {
                if (p + 1 < kBK / 2) {
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm) av[(p + 1) & 1][tm] = a_rd[2 * (p + 1) * (BM + kPad) + 32 * tm];
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) bv[(p + 1) & 1][tn] = b_rd[2 * (p + 1) * (BN + kPad) + 32 * tn];
                }
#ifndef SNERF_PROBE_GEMM_NO_PREFETCH
                __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[p & 1][tm], bv[p & 1][tn], acc[tm][tn], 0, 0, 0);
#ifndef SNERF_PROBE_GEMM_NO_PREFETCH
                __builtin_amdgcn_sched_barrier(0);
#endif
            }
            if (more) {
                store_stage(buf ^ 1);
                __syncthreads();
                buf ^= 1;
            }
        }
    }
    write_tile<TM, TN>(acc, g, C, m0, n0, wm, wn, i, h);
}


// How an operand tile (TILE rows of m resp. n  x  kBK of k) gets from memory into LDS.  `st` = stride (floats) between neighbouring
// m resp. n, `sk` = between neighbouring k.
//   (element by element, any strides: gemm_kernel above -- one dword load and one ds_write_b32 per element; kept as written in
//    rounds 4-5: the same staging expressed through this struct compiled to the same instructions and ran 3-4 % slower)
//   MODE 1: sk == 1 -- 16-byte groups along k: four k of one row per load, four ds_write_b32 (row stride TILE + 1: conflict-free);
//   MODE 2: st == 1 -- 16-byte groups along m / n: four neighbours of one k per load, ONE ds_write_b128 (row stride TILE + 4).
// Round 5: with MODE 0 everywhere the staging cost 50 % on top of the MFMA loop (profiles/r05_layered_ablation.txt); every product
// of the layered path has a contiguous direction in each operand, so launch_gemm picks MODE 1 / 2 whenever strides, sizes and
// addresses are multiples of four floats, and gemm_kernel otherwise (ragged K such as the 63-wide encoding, odd row counts).
template <int TILE, int MODE>
struct Stager {
    static_assert(MODE == 1 || MODE == 2, "16-byte staging only");
    static constexpr int kRow = TILE + (MODE == 2 ? 4 : 1);            // floats per k-row of the tile in LDS
    static constexpr int kElems = TILE * kBK / 256;                      // floats per thread and stage
    static constexpr int kGroups = kElems / 4;                           // 16-byte groups per thread and stage
    static constexpr int kStepK2 = 256 / (TILE / 4);                     // MODE 2: k rows covered per pass of the 256 threads
    const char* tile;            // first element of the tile at k = 0
    long long sk_bytes;
    int left;                    // rows (m resp. n) of the tile inside the matrix
    int t0, k0;                  // this thread's first group: tile index, k index of its first element
    int dt, dk;                  // step to the thread's next group
    unsigned byte0, step;        // byte offset of the first element / group inside a stage, and to the next one
    unsigned inside;             // bit e: element / group e of the stage in flight lies inside the matrix
    float r[kElems];
    int lds0, lds_step;          // float offset of the first element inside a stage buffer, and to the next one

    __device__ __forceinline__ void init(const float* first, long long st, long long sk, int left_, int tid) {
        tile = reinterpret_cast<const char*>(first);
        sk_bytes = sk * 4;
        left = left_;
        inside = 0;
        if constexpr (MODE == 1) {
            t0 = tid >> 3; k0 = (tid & 7) * 4; dt = 32; dk = 0;
            byte0 = (unsigned)((long long)t0 * st + k0) * 4u;
            step = (unsigned)(32 * st * 4);
            lds0 = k0 * kRow + t0;
            lds_step = 32;
        } else {
            t0 = (tid & (TILE / 4 - 1)) * 4; k0 = tid / (TILE / 4); dt = 0; dk = kStepK2;
            byte0 = (unsigned)((long long)k0 * sk + t0) * 4u;
            step = (unsigned)(kStepK2 * sk * 4);
            lds0 = k0 * kRow + t0;
            lds_step = kStepK2 * kRow;
        }
    }
    // branch-free: an element outside the matrix reads the stage's first element (always inside) and is replaced by zero WHEN IT
    // IS STORED, after this stage's MFMAs (a predicated load per element compiles to one exec-masked basic block per load, and a
    // select right here makes the MFMAs wait for the loads they are meant to hide)
    __device__ __forceinline__ void load(long long k_first, int k_left) {
        const char* stage = tile + k_first * sk_bytes;
        inside = 0;
#ifdef SNERF_PROBE_GEMM_NOLOAD       // timing ablation (wrong results): nothing fetched from memory
        (void)stage; (void)k_left;
        return;
#endif
#pragma unroll
        for (int e = 0; e < kGroups; ++e) {
            const bool in = t0 + e * dt < left && k0 + e * dk < k_left;      // (sizes are multiples of four: all four or none)
            inside |= in ? 1u << e : 0u;
            const f32x4 v = *reinterpret_cast<const f32x4*>(stage + (in ? byte0 + (unsigned)e * step : 0u));
#pragma unroll
            for (int q = 0; q < 4; ++q) r[4 * e + q] = v[q];
        }
    }
    __device__ __forceinline__ void store(float* buffer) {
#ifdef SNERF_PROBE_GEMM_NOSTORE      // timing ablation (wrong results): nothing staged into LDS
        return;
#endif
        if constexpr (MODE == 1) {
#pragma unroll
            for (int e = 0; e < kGroups; ++e)
#pragma unroll
                for (int q = 0; q < 4; ++q) buffer[lds0 + e * lds_step + q * kRow] = (inside >> e) & 1u ? r[4 * e + q] : 0.0f;
        } else {
#pragma unroll
            for (int e = 0; e < kGroups; ++e) {
                const bool in = (inside >> e) & 1u;
                const f32x4 v = {in ? r[4 * e] : 0.0f, in ? r[4 * e + 1] : 0.0f, in ? r[4 * e + 2] : 0.0f, in ? r[4 * e + 3] : 0.0f};
                *reinterpret_cast<f32x4*>(buffer + lds0 + e * lds_step) = v;
            }
        }
    }
};

// BM x BN output tile per 256-thread workgroup = a 2 x 2 grid of waves, each (BM / 2) x (BN / 2) = TM x TN MFMA tiles of 32 x 32.
// <64, 64>: one tile per wave (the round-4 kernel: 0.40 of the fp32 matrix peak at width 512 -- per staged byte and per barrier
// it does a quarter of the matrix work of) <128, 128> (round 5): four tiles per wave, every operand value read from LDS feeds
// two MFMAs, 64 KiB of LDS for the two stages; used wherever the product is at least 128 x 128.  AV / BV: Stager modes.
template <int BM, int BN, int AV, int BV>
__global__ void __launch_bounds__(256, 2) gemm_vec_kernel(GemmArgs g) {
    static_assert(AV > 0 && BV > 0, "element-by-element staging is gemm_kernel's");
    constexpr int TM = BM / 64, TN = BN / 64;
    using StageA = Stager<BM, AV>;
    using StageB = Stager<BN, BV>;
    constexpr int SA = StageA::kRow, SB = StageB::kRow;
    __shared__ __attribute__((aligned(16))) float As[2][kBK * SA];
    __shared__ __attribute__((aligned(16))) float Bs[2][kBK * SB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // PERSISTENT workgroups: the grid is two per CU (launch_gemm) and a workgroup walks tiles `blockIdx.x, + gridDim.x, ...` of the
    // tile sequence below -- the first stage of its NEXT tile is requested before the accumulators of this one are written out, so
    // that neither the first loads' round trip nor the epilogue's stores stand alone between two tiles' MFMAs (the two workgroups of
    // a CU run in step: what one tile spent outside its MFMA loop nobody covered -- 85 us per pair of tiles against 55 of matrix
    // work, profiles/r05_layered_ablation.txt).
    // Tile sequence.  Workgroups go to the eight XCDs in turn (linear id mod 8; the grid is a multiple of eight, so a workgroup stays
    // on its XCD's residue) and every XCD has its own L2: with the plain (n tile fastest) order the n tiles of one row block -- which
    // all read the same rows of A -- land on different XCDs and A comes from HBM once per n tile (4 x at width 512).  When the m tiles
    // divide by eight, XCD x takes the row blocks congruent to x and walks their n tiles one after the other: A's rows are fetched
    // once and hit in that XCD's L2 afterwards.
    const unsigned tiles_n = (unsigned)((g.N + BN - 1) / BN), tiles_m = (unsigned)((g.M + BM - 1) / BM), tiles = tiles_n * tiles_m;
    auto tile_of = [&](unsigned linear_id, int* m_first, int* n_first) {
        unsigned tile_m = linear_id / tiles_n, tile_n = linear_id % tiles_n;
#ifndef SNERF_PROBE_GEMM_PLAIN_ORDER
        if ((tiles_m & 7u) == 0) {
            const unsigned xcd = linear_id & 7u, idx = linear_id >> 3;
            tile_m = (idx / tiles_n) * 8u + xcd;
            tile_n = idx % tiles_n;
        }
#endif
        *m_first = (int)tile_m * BM;
        *n_first = (int)tile_n * BN;
    };
    unsigned linear = blockIdx.x;
    if (linear >= tiles) return;
    int m0, n0;
    tile_of(linear, &m0, &n0);
    const long long k_lo = g.split_stride ? (long long)blockIdx.z * g.k_chunk : 0;
    const long long k_hi = g.split_stride ? (k_lo + g.k_chunk < g.K ? k_lo + g.k_chunk : g.K) : g.K;
    float* C = g.C + (g.split_stride ? (long long)blockIdx.z * g.split_stride : 0);

    // thread -> elements of a stage, walking the contiguous dimension first: a thread's elements sit a FIXED step apart in memory and
    // in LDS -- one 32-bit byte offset per operand and thread (computed once), a wave-uniform base per stage (scalar arithmetic),
    // immediate LDS offsets.  (Round 4 computed m * a_rs + k * a_cs in 64 bits per element and stage: 1 021 vector instructions
    // beside 16 MFMAs -- the kernel was bound by its address arithmetic, not by the matrix pipe.)
    StageA sa;
    StageB sb;
    sa.init(g.A + (long long)m0 * g.a_rs, g.a_rs, g.a_cs, g.M - m0, tid);
    sb.init(g.B + (long long)n0 * g.b_cs, g.b_cs, g.b_rs, g.N - n0, tid);
    auto load_stage = [&](long long k0) {
        const int k_left = (int)(k_hi - k0 < kBK ? k_hi - k0 : kBK);       // 32 except in the last stage of a ragged K
        sa.load(k0, k_left);
        sb.load(k0, k_left);
    };
    auto store_stage = [&](int buf) {
        sa.store(As[buf]);
        sb.store(Bs[buf]);
    };

    // wave (wm, wn) owns a (32 TM) x (32 TN) sub-tile; MFMA operands: A(m = lane & 31, k = lane >> 5), B(k = lane >> 5, n = lane & 31)
    const int wm = (wave >> 1) * (32 * TM), wn = (wave & 1) * (32 * TN), i = lane & 31, h = lane >> 5;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.0f;
    const bool any_k = k_lo < k_hi;
    if (any_k) load_stage(k_lo);
    while (true) {
        if (any_k) {
            store_stage(0);
            __syncthreads();
            int buf = 0;
            for (long long k0 = k_lo; k0 < k_hi; k0 += kBK) {
                const bool more = k0 + kBK < k_hi;
                if (more) load_stage(k0 + kBK);                     // in flight during this stage's MFMAs
                const float* a_rd = &As[buf][h * SA + wm + i];
                const float* b_rd = &Bs[buf][h * SB + wn + i];
                // the operands of k-pair p + 1 are requested BEFORE the MFMAs of pair p are issued (two register sets; the scheduling
                // barrier keeps the compiler from sinking the reads back to their use): left alone it emitted read, wait, four MFMAs,
                // read, wait ... with one register set, and the matrix pipe sat idle for an LDS round trip per pair -- 0.54 busy (PMC,
                // profiles/r05_pmc_layered_gemm.json), 0.50 of the peak
                float av[2][TM], bv[2][TN];
    #pragma unroll
                for (int tm = 0; tm < TM; ++tm) av[0][tm] = a_rd[32 * tm];
    #pragma unroll
                for (int tn = 0; tn < TN; ++tn) bv[0][tn] = b_rd[32 * tn];
    #pragma unroll
                for (int p = 0; p < kBK / 2; ++p) {
                    if (p + 1 < kBK / 2) {
    #pragma unroll
                        for (int tm = 0; tm < TM; ++tm) av[(p + 1) & 1][tm] = a_rd[2 * (p + 1) * SA + 32 * tm];
    #pragma unroll
                        for (int tn = 0; tn < TN; ++tn) bv[(p + 1) & 1][tn] = b_rd[2 * (p + 1) * SB + 32 * tn];
                    }
    #ifndef SNERF_PROBE_GEMM_NO_PREFETCH
                    __builtin_amdgcn_sched_barrier(0);
    #endif
    #pragma unroll
                    for (int tm = 0; tm < TM; ++tm)
    #pragma unroll
                        for (int tn = 0; tn < TN; ++tn)
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[p & 1][tm], bv[p & 1][tn], acc[tm][tn], 0, 0, 0);
    #ifndef SNERF_PROBE_GEMM_NO_PREFETCH
                    __builtin_amdgcn_sched_barrier(0);
    #endif
                }
                if (more) {
                    store_stage(buf ^ 1);
                    __syncthreads();
                    buf ^= 1;
                }
            }
            __syncthreads();          // (every wave is done reading the buffers before the next tile's first stage overwrites buffer 0)
        }
        // the next tile's first stage, in flight while this tile's accumulators go out
        const unsigned next = linear + gridDim.x;
        int m_next = 0, n_next = 0;
        if (next < tiles) {
            tile_of(next, &m_next, &n_next);
            sa.init(g.A + (long long)m_next * g.a_rs, g.a_rs, g.a_cs, g.M - m_next, tid);
            sb.init(g.B + (long long)n_next * g.b_cs, g.b_cs, g.b_rs, g.N - n_next, tid);
            if (any_k) load_stage(k_lo);
        }
        write_tile<TM, TN>(acc, g, C, m0, n0, wm, wn, i, h);
        if (next >= tiles) break;
        linear = next; m0 = m_next; n0 = n_next;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.0f;
    }
}

// out[m][n] (+)= sum_z partial[z][m][n] in z order (bit-reproducible)
__global__ void __launch_bounds__(256) reduce_splits_kernel(const float* __restrict__ partial, long long split_stride, int splits,
                                                            long long count, float* __restrict__ out, int accumulate) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += stride) {
        float s = accumulate ? out[e] : 0.0f;
        for (int z = 0; z < splits; ++z) s += partial[z * split_stride + e];
        out[e] = s;
    }
}

// column sums of dZ (N x cols, row stride ld) over a chunk of rows: partial[z][col]
__global__ void __launch_bounds__(256) colsum_kernel(const float* __restrict__ dz, long long ld, long long rows, int cols,
                                                     long long rows_per_split, float* __restrict__ partial) {
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
    const long long lo = (long long)blockIdx.y * rows_per_split, hi = lo + rows_per_split < rows ? lo + rows_per_split : rows;
    float s = 0.0f;
    if (col < cols)
        for (long long r = lo + part; r < hi; r += 4) s += dz[r * ld + col];
    __shared__ float sh[4][64];
    sh[part][threadIdx.x & 63] = s;
    __syncthreads();
    if (part == 0 && col < cols) partial[(long long)blockIdx.y * cols + col] = ((sh[0][threadIdx.x] + sh[1][threadIdx.x]) + sh[2][threadIdx.x]) + sh[3][threadIdx.x];
}

// The same sums for rows of whole 16-byte groups (ld and cols multiples of 4, dz 16-byte aligned: every Linear layer but the
// heads).  Round 5: the scalar kernel above kept ONE 256-byte load per wave in flight -- 614 us per call at 262 144 x 512, 0.9 TB/s,
// 17 % of the layered backward (rocprofv3, profiles/r05_layered_kernel_stats.csv).  Here a lane owns four columns, a wave 256, the
// eight waves of a workgroup take the rows of their chunk in turn, eight rows (8 KiB per wave) in flight; a lane adds its rows in
// order, the waves' sums are added in wave order: fixed, whatever the timing.
__global__ void __launch_bounds__(512) colsum4_kernel(const float* __restrict__ dz, long long ld, long long rows, int cols,
                                                      long long rows_per_split, float* __restrict__ partial) {
    constexpr int kParts = 8, kAhead = 8;
    const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int col = (blockIdx.x * 64 + lane) * 4;
    const long long lo = (long long)blockIdx.y * rows_per_split, hi = lo + rows_per_split < rows ? lo + rows_per_split : rows;
    f32x4 s = {0.0f, 0.0f, 0.0f, 0.0f};
    if (col < cols) {
        const float* base = dz + col;
        long long r = lo + part;
        for (; r + (long long)(kAhead - 1) * kParts < hi; r += (long long)kAhead * kParts) {
            f32x4 v[kAhead];
#pragma unroll
            for (int e = 0; e < kAhead; ++e) v[e] = *reinterpret_cast<const f32x4*>(base + (r + (long long)e * kParts) * ld);
#pragma unroll
            for (int e = 0; e < kAhead; ++e) s += v[e];
        }
        for (; r < hi; r += kParts) s += *reinterpret_cast<const f32x4*>(base + r * ld);
    }
    __shared__ f32x4 sh[kParts][64];
    sh[part][lane] = s;
    __syncthreads();
    if (part == 0 && col < cols) {
        f32x4 t = sh[0][lane];
#pragma unroll
        for (int q = 1; q < kParts; ++q) t += sh[q][lane];
        *reinterpret_cast<f32x4*>(partial + (long long)blockIdx.y * cols + col) = t;
    }
}

// ------------------------------------------------------------------------------------------------------ encoding + heads
// [x, sin(2^0 x), cos(2^0 x), sin(2^1 x), ...] (PositionalEncoder :533-557) with the fused kernels' exact range reduction
struct EncodeArgs {
    const float *origins, *dirs, *view_dirs, *depths;
    float* acts; long long row;
    long long total; int samples;
    int points_degree, views_degree, pe_full, pts_in, views_pe;
    int c_pe, c_pev, c_x5, c_v0_extra, c_v0_views;       // -1 = block absent
};

// One thread per (sample, destination column): neighbouring lanes write neighbouring columns of one row.  (Round 4's kernel gave a
// thread a whole sample: every store of a wave went to 64 different rows -- 0.62 ms of the 8 x 512 forward at 262 144 samples,
// profiles/r05_layered_kernel_stats.csv.  The copies of the encoding a layer input needs -- skip layer, views layer -- are computed
// again instead of read back: same function of the same inputs, same bits.)
__device__ __forceinline__ float encode_column(const float (&x)[3], int c) {
    constexpr double kInvTwoPi = 0.15915494309189533576888;
    if (c < 3) return x[c];
    const int k = (c - 3) / 6, r = (c - 3) % 6, d = r % 3;
    float sn, cs;
    sincos_turns((double)x[d] * kInvTwoPi * (double)(1 << k), sn, cs);
    return r < 3 ? sn : cs;
}

__global__ void __launch_bounds__(256) encode_kernel(EncodeArgs a) {
    // destination blocks of a row, in order: [encoding | its first pts_in columns for the skip layer | its remaining columns for the
    // views layer | view encoding | view encoding for the views layer]
    const int n_pe = a.pe_full, n_x5 = a.c_x5 >= 0 ? a.pts_in : 0, n_extra = a.c_v0_extra >= 0 ? a.pe_full - a.pts_in : 0;
    const int n_pev = a.views_pe > 0 ? a.views_pe : 0;
    const int width = n_pe + n_x5 + n_extra + 2 * n_pev;
    const long long count = a.total * width, stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += stride) {
        const long long s = e / width;
        int j = (int)(e - s * width);
        const long long ray = s / a.samples;
        float* row = a.acts + s * a.row;
        int dest, c;
        bool views = false;
        if (j < n_pe) { dest = a.c_pe + j; c = j; }
        else if ((j -= n_pe) < n_x5) { dest = a.c_x5 + j; c = j; }
        else if ((j -= n_x5) < n_extra) { dest = a.c_v0_extra + j; c = a.pts_in + j; }
        else if ((j -= n_extra) < n_pev) { dest = a.c_pev + j; c = j; views = true; }
        else { j -= n_pev; dest = a.c_v0_views + j; c = j; views = true; }
        float x[3];
        if (views) {
            for (int k = 0; k < 3; ++k) x[k] = a.view_dirs[ray * 3 + k];
        } else {
            const float z = a.depths[s];
            for (int k = 0; k < 3; ++k) x[k] = a.origins[ray * 3 + k] + a.dirs[ray * 3 + k] * z;   // mul, then add (:140-142)
        }
        row[dest] = encode_column(x, c);
    }
}

// pts_output / views_output rows -> sigma (N), rgb (N,3) (:664-681, :703-706)
__global__ void __launch_bounds__(256) heads_kernel(const float* __restrict__ acts, long long row, int c_out, int c_vout, int view_dep,
                                                    const float* __restrict__ noise, long long total, float* __restrict__ sigma,
                                                    float* __restrict__ rgb) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x; s < total; s += stride) {
        const float* r = acts + s * row;
        float sg = r[c_out];
        if (noise) sg += noise[s];
        sigma[s] = fmaxf(sg, 0.0f);
        const float* col = view_dep ? r + c_vout : r + c_out + 1;
        for (int c = 0; c < 3; ++c) rgb[s * 3 + c] = sigmoidf(col[c]);
    }
}

// d sigma, d rgb -> gradients of the two head pre-activations: dout (N,4) and dvout (N,4), zero-padded
__global__ void __launch_bounds__(256) heads_backward_kernel(const float* __restrict__ sigma, const float* __restrict__ rgb,
                                                             const float* __restrict__ d_sigma, const float* __restrict__ d_rgb,
                                                             long long total, int view_dep, float* __restrict__ dout,
                                                             float* __restrict__ dvout) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x; s < total; s += stride) {
        float col[3];
        for (int c = 0; c < 3; ++c) {
            const float v = rgb[s * 3 + c];
            col[c] = d_rgb[s * 3 + c] * (v * (1.0f - v));
        }
        dout[s * 4] = sigma[s] > 0.0f ? d_sigma[s] : 0.0f;
        for (int c = 0; c < 3; ++c) {
            dout[s * 4 + 1 + c] = view_dep ? 0.0f : col[c];
            if (view_dep) dvout[s * 4 + c] = col[c];
        }
        if (view_dep) dvout[s * 4 + 3] = 0.0f;
    }
}

__global__ void __launch_bounds__(256) copy_kernel(float* __restrict__ dst, const float* __restrict__ src, long long count) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) dst[i] = src[i];
}

// ------------------------------------------------------------------------------------------------------------ host side
// two workgroups per CU of the current device, rounded down to a multiple of eight (512 on an MI355X)
unsigned persistent_workgroups() {
    int device = 0, cus = 256;
    if (hipGetDevice(&device) == hipSuccess) {
        int value = 0;
        if (hipDeviceGetAttribute(&value, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && value > 0) cus = value;
    }
    (void)hipGetLastError();
    const unsigned count = (unsigned)(2 * cus) & ~7u;
    return count ? count : 8u;
}

int launch_gemm(const GemmArgs& g, int splits, hipStream_t s) {
    if (g.M <= 0 || g.N <= 0) return SNERF_OK;
    // (the same arithmetic per output element either way: each is one fp32 FMA chain over k in order -- the tile only decides
    // which workgroup computes it)
    // ... the large tile only where it still fills the chip twice over: a 256 x 256 weight gradient split 32 ways is 128 workgroups
    // of 128 x 128 (22.6 -> 25.5 ms for the 8 x 256 / 2 x 128 backward before this condition)
    const long long large_tiles = (long long)((g.N + 127) / 128) * ((g.M + 127) / 128) * (splits > 0 ? splits : 1);
    if (g.M >= 128 && g.N >= 128 && large_tiles >= 512) {
        const dim3 grid((g.N + 127) / 128, (g.M + 127) / 128, splits > 0 ? splits : 1);
        // 16-byte staging (Stager MODE 1 / 2) where everything it touches is a multiple of four floats; the three combinations are
        // the path's three products: forward (A, B along k), input gradient (A along k, B along n), weight gradient (A along m,
        // B along n)
        auto aligned = [](const float* ptr) { return reinterpret_cast<uintptr_t>(ptr) % 16 == 0; };
        const bool k_fours = g.K % 4 == 0 && (splits <= 0 || g.k_chunk % 4 == 0);
        const bool a_along_k = g.a_cs == 1 && g.a_rs % 4 == 0 && aligned(g.A) && k_fours;
        const bool a_along_m = g.a_rs == 1 && g.a_cs % 4 == 0 && aligned(g.A) && g.M % 4 == 0;
        const bool b_along_k = g.b_rs == 1 && g.b_cs % 4 == 0 && aligned(g.B) && k_fours;
        const bool b_along_n = g.b_cs == 1 && g.b_rs % 4 == 0 && aligned(g.B) && g.N % 4 == 0;
#ifdef SNERF_PROBE_GEMM_SCALAR_STAGING
        hipLaunchKernelGGL((gemm_kernel<128, 128>), grid, dim3(256), 0, s, g);
#else
        // (persistent workgroups: two per CU of the current device, a multiple of eight so that a workgroup's tiles stay on its XCD)
        const unsigned tiles = grid.x * grid.y;
        const dim3 walkers(tiles < persistent_workgroups() ? tiles : persistent_workgroups(), 1, grid.z);
        if (a_along_k && b_along_k) hipLaunchKernelGGL((gemm_vec_kernel<128, 128, 1, 1>), walkers, dim3(256), 0, s, g);
        else if (a_along_k && b_along_n) hipLaunchKernelGGL((gemm_vec_kernel<128, 128, 1, 2>), walkers, dim3(256), 0, s, g);
        else if (a_along_m && b_along_n) hipLaunchKernelGGL((gemm_vec_kernel<128, 128, 2, 2>), walkers, dim3(256), 0, s, g);
        else hipLaunchKernelGGL((gemm_kernel<128, 128>), grid, dim3(256), 0, s, g);
#endif
    } else {
        const dim3 grid((g.N + 63) / 64, (g.M + 63) / 64, splits > 0 ? splits : 1);
        hipLaunchKernelGGL((gemm_kernel<64, 64>), grid, dim3(256), 0, s, g);
    }
    return snerf::check_launch("mlp_generic(gemm)");
}

// Y[rows, out] = act(X[rows, in] . W[out, in]^T + b): X, Y are column blocks of the activation matrix
int linear_forward(const float* acts_in, float* acts_out, long long row, long long rows, int in, int out, const float* w,
                   const float* b, bool relu, hipStream_t s) {
    GemmArgs g = {};
    g.A = acts_in; g.a_rs = row; g.a_cs = 1;
    g.B = w; g.b_rs = 1; g.b_cs = in;
    g.C = acts_out; g.c_rs = row; g.c_cs = 1;
    g.bias = b; g.M = (int)rows; g.N = out; g.K = in; g.relu = relu ? 1 : 0;
    return launch_gemm(g, 0, s);
}

// Scratch of the inference entry point (the ABI gives it no workspace argument): one block per (device, stream), so that two
// layered-shape forwards on different streams or threads of a device do not share an activation matrix (they used
// to).  A block is NEVER freed or moved once handed out -- an in-flight kernel or a captured graph may hold its address; a
// larger request on the same stream allocates another block and the old one stays (sizes are bounded by kInferenceChunk rows,
// so a stream grows its block a handful of times at most).  Growing inside a graph capture is refused: the allocation would be
// a synchronous call the capture cannot contain.
struct ArenaBlock { int device; hipStream_t stream; float* base; size_t floats; };
std::mutex g_arena_mutex;
std::vector<ArenaBlock> g_arena;

int arena(size_t floats, hipStream_t stream, float** out) {
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) return snerf::fail(SNERF_E_HIP, "mlp_forward(layered): no current HIP device");
    std::lock_guard<std::mutex> lock(g_arena_mutex);
    for (const ArenaBlock& b : g_arena)
        if (b.device == device && b.stream == stream && b.floats >= floats) {
            *out = b.base;
            return SNERF_OK;
        }
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &capturing) == hipSuccess && capturing != hipStreamCaptureStatusNone)
        return snerf::fail(SNERF_E_UNSUPPORTED, "mlp_forward(layered): the scratch block of this stream must grow to %zu MB, which a graph "
                                                "capture cannot contain -- run one call of this size on the stream before capturing",
                           floats * sizeof(float) >> 20);
    (void)hipGetLastError();
    void* p = nullptr;
    const hipError_t e = hipMalloc(&p, floats * sizeof(float));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return snerf::fail(SNERF_E_HIP, "mlp_forward(layered): scratch of %zu MB: %s", floats * sizeof(float) >> 20, hipGetErrorString(e));
    }
    g_arena.push_back({device, stream, static_cast<float*>(p), floats});
    *out = static_cast<float*>(p);
    return SNERF_OK;
}

constexpr long long kInferenceChunk = 65536;     // samples per pass of the inference entry point (bounds its scratch)

// forward over `total` consecutive samples starting at sample `first` of the call; acts has `total` rows
int forward_rows(const GenericPlan& p, const float* packed, const float* origins, const float* dirs, const float* view_dirs,
                 const float* depths, long long first, long long total, int samples, const float* noise, float* sigma, float* rgb,
                 float* acts, hipStream_t s) {
    EncodeArgs e = {};
    e.origins = origins + (first / samples) * 3; e.dirs = dirs + (first / samples) * 3;
    e.view_dirs = view_dirs ? view_dirs + (first / samples) * 3 : nullptr;
    // (the chunk starts at a ray boundary: kInferenceChunk is rounded to whole rays by the caller)
    e.depths = depths + first; e.acts = acts; e.row = p.row; e.total = total; e.samples = samples;
    e.points_degree = p.points_degree; e.views_degree = p.views_degree; e.pe_full = p.pe_full; e.pts_in = p.pts_in;
    e.views_pe = p.view_dep ? p.views_pe : 0;
    e.c_pe = p.c_pe; e.c_pev = p.c_pev; e.c_x5 = p.c_x5; e.c_v0_extra = p.view_dep && p.extra > 0 ? p.c_v0 + p.width : -1;
    e.c_v0_views = p.view_dep ? p.c_v0 + p.width + p.extra : -1;
    hipLaunchKernelGGL(encode_kernel, dim3(snerf::stride_grid(total * (p.pe_full + (p.view_dep ? p.views_pe : 0)), 256)), dim3(256), 0, s, e);
    int rc = snerf::check_launch("mlp_generic(encode)");
    if (rc != SNERF_OK) return rc;
    for (int l = 0; l < p.depth; ++l) {
        rc = linear_forward(acts + p.layer_in_col(l), acts + p.c_h[l], p.row, total, p.layer_in_dim(l), p.width,
                            packed + p.w_off[2 * l], packed + p.w_off[2 * l + 1], true, s);
        if (rc != SNERF_OK) return rc;
    }
    const int po = 2 * p.depth;
    rc = linear_forward(acts + p.c_h[p.depth - 1], acts + p.c_out, p.row, total, p.width, p.pts_out_rows, packed + p.w_off[po],
                        packed + p.w_off[po + 1], false, s);
    if (rc != SNERF_OK) return rc;
    if (p.view_dep) {
        rc = linear_forward(acts + p.c_h[p.depth - 1], acts + p.c_v0, p.row, total, p.width, p.width, packed + p.w_off[po + 2],
                            packed + p.w_off[po + 3], false, s);                                   // feature: no activation (:683)
        if (rc != SNERF_OK) return rc;
        for (int j = 0; j < p.views_depth; ++j) {
            rc = linear_forward(acts + (j == 0 ? p.c_v0 : p.c_hv[j - 1]), acts + p.c_hv[j], p.row, total,
                                j == 0 ? p.views_in : p.views_width, p.views_width, packed + p.w_off[po + 4 + 2 * j],
                                packed + p.w_off[po + 5 + 2 * j], true, s);
            if (rc != SNERF_OK) return rc;
        }
        const int pv = po + 4 + 2 * p.views_depth;
        rc = linear_forward(acts + p.c_hv[p.views_depth - 1], acts + p.c_vout, p.row, total, p.views_width, 3, packed + p.w_off[pv],
                            packed + p.w_off[pv + 1], false, s);
        if (rc != SNERF_OK) return rc;
    }
    hipLaunchKernelGGL(heads_kernel, dim3(snerf::stride_grid(total, 256)), dim3(256), 0, s, acts, p.row, p.c_out, p.c_vout,
                       p.view_dep ? 1 : 0, noise ? noise + first : nullptr, total, sigma + first, rgb + first * 3);
    return snerf::check_launch("mlp_generic(heads)");
}

}  // namespace

namespace snerf {

int generic_plan(const snerf_mlp_desc* d, GenericPlan* out) {
    if (!d) return fail(SNERF_E_INVALID, "mlp: NULL descriptor");
    GenericPlan p;
    p.depth = d->points_net_depth; p.width = d->points_net_width;
    p.view_dep = d->view_dependent_rgb != 0;
    p.views_depth = p.view_dep ? d->views_net_depth : 0;
    p.views_width = p.view_dep ? d->views_net_width : 0;
    p.points_degree = d->points_pe_degree; p.views_degree = p.view_dep ? d->views_pe_degree : 0;
    if (p.width < 1 || p.width > 4096) return fail(SNERF_E_UNSUPPORTED, "mlp: points_net_width %d outside 1..4096", p.width);
    if (p.depth < 1 || p.depth == 5 || p.depth > 64)
        return fail(SNERF_E_UNSUPPORTED, "mlp: points_net_depth %d unsupported (the reference itself cannot build depth 5: the skip "
                                         "concat would feed pts_output_linear)", p.depth);
    if (p.points_degree < 1 || p.points_degree > 16) return fail(SNERF_E_UNSUPPORTED, "mlp: points positional-encoding degree %d outside 1..16", p.points_degree);
    if (d->predict_visibility) return fail(SNERF_E_UNSUPPORTED, "mlp: predict_visibility is built for the fused shapes only (widths 128/256, views 64/128)");
    if (p.view_dep) {
        if (!d->use_view_dirs) return fail(SNERF_E_UNSUPPORTED, "mlp: view_dependent_rgb without use_view_dirs");
        if (p.views_depth < 1 || p.views_depth > 16) return fail(SNERF_E_UNSUPPORTED, "mlp: views_net_depth %d outside 1..16", p.views_depth);
        if (p.views_width < 1 || p.views_width > 4096) return fail(SNERF_E_UNSUPPORTED, "mlp: views_net_width %d outside 1..4096", p.views_width);
        if (p.views_degree < 1 || p.views_degree > 16) return fail(SNERF_E_UNSUPPORTED, "mlp: views positional-encoding degree %d outside 1..16", p.views_degree);
    }
    p.pe_full = 3 + 6 * p.points_degree;
    p.pts_in = p.pe_full;
    if (d->sigma_pe_degree >= 0) {
        if (d->sigma_pe_degree > p.points_degree) return fail(SNERF_E_UNSUPPORTED, "mlp: sigma encoding degree %d exceeds the points degree %d", d->sigma_pe_degree, p.points_degree);
        if (!p.view_dep) return fail(SNERF_E_UNSUPPORTED, "mlp: points_sigma_positional_encoding_degree needs view-dependent colour");
        p.pts_in = (2 * d->sigma_pe_degree + 1) * 3;
    }
    p.extra = p.pe_full - p.pts_in;
    p.views_pe = p.view_dep ? 3 + 6 * p.views_degree : 0;
    p.views_in = p.width + p.extra + p.views_pe;
    p.pts_out_rows = p.view_dep ? 1 : 4;
    p.num_params = 2 * p.depth + 2 + (p.view_dep ? 2 + 2 * p.views_depth + 2 : 0);
    // parameter offsets in the "packed" buffer (a plain concatenation in ABI order)
    long long off = 0;
    auto param = [&](long long count) { p.w_off.push_back(off); p.w_count.push_back(count); off += (count + 3) / 4 * 4; };
    for (int l = 0; l < p.depth; ++l) { param((long long)p.width * p.layer_in_dim(l)); param(p.width); }
    param((long long)p.pts_out_rows * p.width); param(p.pts_out_rows);
    if (p.view_dep) {
        param((long long)p.width * p.width); param(p.width);
        for (int j = 0; j < p.views_depth; ++j) { param((long long)p.views_width * (j == 0 ? p.views_in : p.views_width)); param(p.views_width); }
        param(3LL * p.views_width); param(3);
    }
    p.packed_floats = off;
    // activation row
    // (every block a Linear layer reads starts at a multiple of four floats, so that the GEMM can stage it 16 bytes at a time --
    // Stager MODE 1 / 2; H_4 follows the skip layer's encoding copy directly, as that layer reads [encoding | H_4] as ONE input)
    int c = 0;
#ifdef SNERF_PROBE_GENERIC_PACKED_ROW     // A/B builds: the row of rounds 4-5, blocks back to back
    auto four = [&]() {};
#else
    auto four = [&]() { c = (c + 3) / 4 * 4; };
#endif
    p.c_pe = c; c += p.pe_full;
    four(); p.c_pev = c; c += p.views_pe;
    p.c_x5 = -1;
    p.c_h.assign(p.depth, 0);
    for (int l = 0; l < p.depth; ++l) {
        four();
        if (l == 4 && p.depth > 5) { p.c_x5 = c; c += p.pts_in; }      // [encoding | H_4]: the skip layer's input (:662-663)
        p.c_h[l] = c; c += p.width;
    }
    four(); p.c_v0 = c; if (p.view_dep) c += p.views_in;
    p.c_hv.assign(p.views_depth, 0);
    for (int j = 0; j < p.views_depth; ++j) { four(); p.c_hv[j] = c; c += p.views_width; }
    four(); p.c_out = c; c += 4;
    p.c_vout = c; c += 4;
    p.row = (c + 3) / 4 * 4;
    *out = p;
    return SNERF_OK;
}

int generic_pack(const GenericPlan& p, const float* const* params, float* packed, hipStream_t s) {
    for (int i = 0; i < p.num_params; ++i) {
        hipLaunchKernelGGL(copy_kernel, dim3(stride_grid(p.w_count[i], 256)), dim3(256), 0, s, packed + p.w_off[i], params[i], p.w_count[i]);
        const int rc = check_launch("mlp_pack(layered)");
        if (rc != SNERF_OK) return rc;
    }
    return SNERF_OK;
}

size_t generic_saved_floats(const GenericPlan& p, long long total) { return (size_t)total * (size_t)p.row; }

int generic_forward(const GenericPlan& p, const float* packed, const float* origins, const float* dirs, const float* view_dirs,
                    const float* depths, long long num_rays, int num_samples, const float* noise, float* sigma, float* rgb,
                    float* saved_acts, int precision, hipStream_t s) {
    if (precision != SNERF_PRECISION_FP32)
        return fail(SNERF_E_UNSUPPORTED, "mlp_forward: this MLP shape (width %d, views %d x %d) runs on the layered fp32 path only; "
                                         "hip_precision must be 'fp32'", p.width, p.views_depth, p.views_width);
    const long long total = num_rays * num_samples;
    ProfileScope timed(SNERF_PROFILE_MLP_FORWARD, s, total);
    if (saved_acts)     // training: every layer's input is kept, the whole call in one pass
        return forward_rows(p, packed, origins, dirs, view_dirs, depths, 0, total, num_samples, noise, sigma, rgb, saved_acts, s);
    const long long rays_per_chunk = std::max(1LL, kInferenceChunk / num_samples);
    float* scratch = nullptr;
    const int rc = arena((size_t)std::min(num_rays, rays_per_chunk) * num_samples * p.row, s, &scratch);
    if (rc != SNERF_OK) return rc;
    for (long long ray = 0; ray < num_rays; ray += rays_per_chunk) {
        const long long rays = std::min(rays_per_chunk, num_rays - ray);
        const int st = forward_rows(p, packed, origins, dirs, view_dirs, depths, ray * num_samples, rays * num_samples, num_samples,
                                    noise, sigma, rgb, scratch, s);
        if (st != SNERF_OK) return st;
    }
    return SNERF_OK;
}

// workspace: dZ ping-pong (2 x N x widest) | d heads (2 x N x 4) | split-K partial sums
static int wgrad_splits(long long total) { return (int)std::min<long long>(64, std::max<long long>(1, total / 8192)); }

size_t generic_backward_workspace_floats(const GenericPlan& p, long long total) {
    const long long widest = std::max({p.width, p.views_width, p.views_in, p.pts_in + p.width});
    const long long biggest = std::max({(long long)p.width * (p.pts_in + p.width), (long long)p.views_width * p.views_in,
                                        (long long)p.views_width * p.views_width, (long long)p.width * p.width});
    return (size_t)(2 * total * widest + 8 * total + (long long)wgrad_splits(total) * (biggest + widest) + 64);
}

int generic_backward(const GenericPlan& p, const float* packed, const float* acts, const float* sigma, const float* rgb,
                     const float* d_sigma, const float* d_rgb, long long total, float* workspace, float* const* grads, int precision,
                     int accumulate, hipStream_t s) {
    if (precision != SNERF_PRECISION_FP32)
        return fail(SNERF_E_UNSUPPORTED, "mlp_backward: this MLP shape runs on the layered fp32 path only; hip_precision must be 'fp32'");
    ProfileScope timed(SNERF_PROFILE_MLP_BACKWARD, s, total);
    const long long widest = std::max({p.width, p.views_width, p.views_in, p.pts_in + p.width});
    float* ping = workspace;
    float* pong = workspace + total * widest;
    float* dout = workspace + 2 * total * widest;
    float* dvout = dout + 4 * total;
    float* partial = dvout + 4 * total;
    const int splits = wgrad_splits(total);
    const long long k_chunk = (total + splits - 1) / splits;

    hipLaunchKernelGGL(heads_backward_kernel, dim3(stride_grid(total, 256)), dim3(256), 0, s, sigma, rgb, d_sigma, d_rgb, total,
                       p.view_dep ? 1 : 0, dout, dvout);
    int rc = check_launch("mlp_generic(heads backward)");
    if (rc != SNERF_OK) return rc;

    // dW = dZ^T . X (split over the samples, fixed-order reduction), db = column sums of dZ
    auto weight_grad = [&](const float* dz, long long dz_ld, int out, const float* x, int in, float* gw, float* gb) -> int {
        GemmArgs g = {};
        g.A = dz; g.a_rs = 1; g.a_cs = dz_ld;           // A(m = out feature, k = sample) = dZ[k][m]
        g.B = x; g.b_rs = p.row; g.b_cs = 1;            // B(k = sample, n = in feature)
        g.C = partial; g.c_rs = in; g.c_cs = 1;
        g.M = out; g.N = in; g.K = (int)total; g.k_chunk = k_chunk; g.split_stride = (long long)out * in;
        int st = launch_gemm(g, splits, s);
        if (st != SNERF_OK) return st;
        hipLaunchKernelGGL(reduce_splits_kernel, dim3(stride_grid((long long)out * in, 256)), dim3(256), 0, s, partial,
                           (long long)out * in, splits, (long long)out * in, gw, accumulate);
        st = check_launch("mlp_generic(reduce)");
        if (st != SNERF_OK) return st;
        // bias gradient = column sums of dZ.  The weight partial sums above have just been folded, so their area is free again: the
        // column sums take four times as many row chunks as the GEMM had splits (32 chunks x two column blocks was 64 workgroups on
        // 256 CUs: 151 us per call at 262 144 x 512, 3.6 TB/s) and put their partial rows at its start.
        const int bsplits = in >= 4 ? splits * 4 : splits;        // (the area holds splits x out x in floats: room for splits x in rows of `out`)
        const long long b_chunk = (total + bsplits - 1) / bsplits;
        float* bpart = partial;
        if (out % 4 == 0 && dz_ld % 4 == 0 && reinterpret_cast<uintptr_t>(dz) % 16 == 0 && reinterpret_cast<uintptr_t>(bpart) % 16 == 0)
            hipLaunchKernelGGL(colsum4_kernel, dim3((out + 255) / 256, bsplits), dim3(512), 0, s, dz, dz_ld, total, out, b_chunk, bpart);
        else
            hipLaunchKernelGGL(colsum_kernel, dim3((out + 63) / 64, bsplits), dim3(256), 0, s, dz, dz_ld, total, out, b_chunk, bpart);
        st = check_launch("mlp_generic(bias sums)");
        if (st != SNERF_OK) return st;
        hipLaunchKernelGGL(reduce_splits_kernel, dim3(1), dim3(256), 0, s, bpart, (long long)out, bsplits, (long long)out, gb, accumulate);
        return check_launch("mlp_generic(reduce bias)");
    };
    // dX[:, cols] (+)= dZ . W[:, col0 : col0 + cols], then gated by the ReLU of the layer that produced X
    auto input_grad = [&](const float* dz, long long dz_ld, int out, const float* w, int w_ld, int col0, int cols, float* dx,
                          long long dx_ld, bool add, const float* gate) -> int {
        GemmArgs g = {};
        g.A = dz; g.a_rs = dz_ld; g.a_cs = 1;
        g.B = w + col0; g.b_rs = w_ld; g.b_cs = 1;
        g.C = dx; g.c_rs = dx_ld; g.c_cs = 1;
        g.M = (int)total; g.N = cols; g.K = out; g.accumulate = add ? 1 : 0;
        g.mask = gate; g.mask_rs = p.row;
        return launch_gemm(g, 0, s);
    };

    const int po = 2 * p.depth;
    float* dh = ping;        // gradient of the trunk's last activation H_D-1, then dZ of each trunk layer in turn
    float* other = pong;
    const float* h_last = acts + p.c_h[p.depth - 1];
    if (p.view_dep) {
        const int pv = po + 4 + 2 * p.views_depth;
        // views head and views layers, last first
        rc = weight_grad(dvout, 4, 3, acts + p.c_hv[p.views_depth - 1], p.views_width, grads[pv], grads[pv + 1]);
        if (rc != SNERF_OK) return rc;
        rc = input_grad(dvout, 4, 3, packed + p.w_off[pv], p.views_width, 0, p.views_width, other, p.views_width, false,
                        acts + p.c_hv[p.views_depth - 1]);
        if (rc != SNERF_OK) return rc;
        float* dzv = other; float* spare = dh;
        for (int j = p.views_depth - 1; j >= 0; --j) {
            const int in = j == 0 ? p.views_in : p.views_width;
            const float* x = acts + (j == 0 ? p.c_v0 : p.c_hv[j - 1]);
            rc = weight_grad(dzv, p.views_width, p.views_width, x, in, grads[po + 4 + 2 * j], grads[po + 5 + 2 * j]);
            if (rc != SNERF_OK) return rc;
            // j > 0: d HV_j-1, gated by its ReLU; j == 0: d feature = the first `width` columns of the views input (no activation)
            rc = input_grad(dzv, p.views_width, p.views_width, packed + p.w_off[po + 4 + 2 * j], in, 0, j == 0 ? p.width : p.views_width,
                            spare, j == 0 ? p.width : p.views_width, false, j == 0 ? nullptr : acts + p.c_hv[j - 1]);
            if (rc != SNERF_OK) return rc;
            std::swap(dzv, spare);
        }
        float* dfeature = dzv;          // (N, width)
        float* dlast = spare;
        rc = weight_grad(dfeature, p.width, p.width, h_last, p.width, grads[po + 2], grads[po + 3]);
        if (rc != SNERF_OK) return rc;
        rc = input_grad(dfeature, p.width, p.width, packed + p.w_off[po + 2], p.width, 0, p.width, dlast, p.width, false, nullptr);
        if (rc != SNERF_OK) return rc;
        dh = dlast; other = dfeature;
    }
    // density head (and the view-independent colour rows): dW_out, and its contribution to d H_D-1, then the ReLU gate
    rc = weight_grad(dout, 4, p.pts_out_rows, h_last, p.width, grads[po], grads[po + 1]);
    if (rc != SNERF_OK) return rc;
    rc = input_grad(dout, 4, p.pts_out_rows, packed + p.w_off[po], p.width, 0, p.width, dh, p.width, p.view_dep, h_last);
    if (rc != SNERF_OK) return rc;
    // trunk, last layer first: dh holds dZ_l
    for (int l = p.depth - 1; l >= 0; --l) {
        const int in = p.layer_in_dim(l);
        rc = weight_grad(dh, p.width, p.width, acts + p.layer_in_col(l), in, grads[2 * l], grads[2 * l + 1]);
        if (rc != SNERF_OK) return rc;
        if (l == 0) break;
        const int col0 = in - p.width;       // the skip layer's input is [encoding | H_l-1]: only the H columns carry on
        rc = input_grad(dh, p.width, p.width, packed + p.w_off[2 * l], in, col0, p.width, other, p.width, false, acts + p.c_h[l - 1]);
        if (rc != SNERF_OK) return rc;
        std::swap(dh, other);
    }
    return SNERF_OK;
}

}  // namespace snerf
