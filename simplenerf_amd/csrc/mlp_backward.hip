// K7 -- backward of the fused NeRF MLP: parameter gradients from d(sigma), d(rgb).
//
// Autograd of MLP.forward (src/models/SimpleNeRF01.py:626-715) restated as three kernel families, all on the fp32
// matrix cores (v_mfma_f32_32x32x2_f32):
//
//  B1  mlp_backward_chain_kernel   "dgrad": the same register-resident transposed chain as the forward, run in
//      reverse.  dX^T[in, sample] = W^T[in, out] . dY^T[out, sample]; dY^T lives in accumulator layout, so it is the
//      B operand of the next (earlier) layer's product without leaving registers.  ReLU masks come from the
//      activations saved by snerf_mlp_forward_train.  Every dY_l is also written as a [feature][32-sample] tile
//      for B2.  Heads (1-4 rows) are VALU work.
//  B2  wgrad_kernel<TPW>           dW[out, in] = sum_samples dY^T[out, s] . X^T[in, s] -- contraction over samples,
//      so both operands are re-read from their tiles through LDS (padded rows, conflict-free ds_read_b128) with the
//      feature index on the lanes.  One workgroup owns one (job, chunk of wave blocks); partial sums per chunk are
//      written out and
//  B3  reduce_kernel               sums the chunks in a fixed order (bit-reproducible; no float atomics) and scatters
//      into the reference-layout gradient tensors (weight (out,in) row-major, bias).
//
// Bound: MFMA fp32; algorithmic FLOPs = 2x the forward's (dgrad + wgrad).  HBM per sample: reads ~10 KB of saved
// activations + ~10 KB of dY tiles twice (B1 writes, B2 reads) -- about 40 KB/sample, i.e. ~2.5 TB/s at the full MFMA
// rate: under the HBM roofline, overlapped with the matrix work.
#include "mlp_generic.h"
#include <algorithm>
#include <type_traits>

#include "clock_stamp.h"
#include "mlp_device_f16.h"

namespace snerf {
int mlp_backward_chain_f16x3(const MlpPlan& plan, const ChainArgs& a, int products, hipStream_t stream);  // mlp_backward_f16.hip
int mlp_backward_chain_bf16(const MlpPlan& plan, const ChainArgs& a, hipStream_t stream);                   // mlp_backward_bf16.hip
}

namespace {

template <int U>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[u][r] = 0.0f;
}

// dy[n] = (the forward's activation of the same feature was > 0) ? acc : 0, in accumulator order.  The signs come
// from the mask words the training forward wrote (MlpPlan::act_mask): 4 bytes per lane per two tiles instead of
// re-reading the 32-sample x 32-feature activation tiles (1 KB instead of 32 KB per layer and wave block).
// The words of a layer are requested one layer ahead of their use (load_relu_words), so that their L2 latency hides under
// the layer's products instead of sitting in front of them.
template <int U>
__device__ __forceinline__ void load_relu_words(unsigned (&words)[U / 2], const unsigned* __restrict__ masks, int t0, int lane) {
#pragma unroll
    for (int p = 0; p < U / 2; ++p) words[p] = __builtin_nontemporal_load(masks + ((t0 >> 1) + p) * 64 + lane);   // read once
}
template <int U>
__device__ __forceinline__ void relu_backward(const f32x16 (&acc)[U], const unsigned (&words)[U / 2], float (&dy)[U * 16]) {
#pragma unroll
    for (int u = 0; u < U; u += 2)
#pragma unroll
        for (int r = 0; r < 32; ++r) dy[16 * u + r] = keep_if_bit(acc[u + (r >> 4)][r & 15], words[u >> 1], r);
}

template <int WT, int VT, bool VIEWDEP>
__global__ void __launch_bounds__(256, 1) mlp_backward_chain_kernel(ChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5;
    SlabStream<WT> st;
    st.start(a.packed + a.dgrad_offset, lds, lane, wave);

    const long long block = (long long)blockIdx.x * 4 + wave;
    const long long first = block * 32 + (lane & 31);
    const bool live = first < a.total;
    const unsigned* masks = reinterpret_cast<const unsigned*>(a.acts + (block * a.act_rows + a.act_mask) * 32);
    float* grads = a.grads + block * a.grad_rows * 32;

    // ---- head gradients (pre-activation), stored as rows 0..3 of the head tile -----------------------------------
    float dhead[4];
    dhead[0] = live && a.sigma[first] > 0.0f ? a.d_sigma[first] : 0.0f;     // ReLU of the density (:672)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float col = live ? a.rgb[first * 3 + c] : 0.0f;
        dhead[c + 1] = live ? a.d_rgb[first * 3 + c] * (col * (1.0f - col)) : 0.0f;  // sigmoid' = y (1 - y)
    }
    if (half == 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) grads[(a.grad_head + c) * 32 + (lane & 31)] = dhead[c];
    }

    f32x16 acc[WT];
    zero_acc<WT>(acc);
    float dy[WT * 16];
    if (VIEWDEP) {
        // d hv = W_rgb^T dpre, masked by the views-layer ReLU (:699-706)
        const float* wv = a.packed + a.views_out_w;
        unsigned hv_bits[VT / 2];
#pragma unroll
        for (int p = 0; p < VT / 2; ++p) hv_bits[p] = masks[((a.depth * WT) / 2 + p) * 64 + lane];
        float dyv[VT * 16];
#pragma unroll
        for (int g = 0; g < VT * 4; ++g) {
            f32x4 w0 = *reinterpret_cast<const f32x4*>(wv + 0 * VT * 32 + 8 * g + 4 * half);
            f32x4 w1 = *reinterpret_cast<const f32x4*>(wv + 1 * VT * 32 + 8 * g + 4 * half);
            f32x4 w2 = *reinterpret_cast<const f32x4*>(wv + 2 * VT * 32 + 8 * g + 4 * half);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float v = fmaf(w2[q], dhead[3], fmaf(w1[q], dhead[2], w0[q] * dhead[1]));
                const int n = 4 * g + q;   // register 16u + r of the views tile u
                dyv[n] = (hv_bits[n >> 5] >> (n & 31)) & 1u ? v : 0.0f;
            }
        }
        store_acc_tile(dyv, grads + a.grad_yv * 32, lane);
        // d feature = Wv[:, :width]^T dYv
        gemm_segment<WT, VT, WT>(acc, dyv, st);
        to_operand<WT, false>(acc, dy);
        store_acc_tile(dy, grads + a.grad_feature * 32, lane);
        // d h_depth = W_feature^T dfeature   (feature_linear has no activation, :683)
        zero_acc<WT>(acc);
        gemm_segment<WT, WT, WT>(acc, dy, st);
    }
    // + density head (and, without a views layer, the colour rows of pts_output_linear): d h += W_out^T dhead
    {
        const float* wo = a.packed + a.pts_out_w;
#pragma unroll
        for (int g = 0; g < WT * 4; ++g) {
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(wo + 8 * g + 4 * half);
            f32x4 w1 = {0, 0, 0, 0}, w2 = {0, 0, 0, 0}, w3 = {0, 0, 0, 0};
            if (!VIEWDEP) {
                w1 = *reinterpret_cast<const f32x4*>(wo + 1 * WT * 32 + 8 * g + 4 * half);
                w2 = *reinterpret_cast<const f32x4*>(wo + 2 * WT * 32 + 8 * g + 4 * half);
                w3 = *reinterpret_cast<const f32x4*>(wo + 3 * WT * 32 + 8 * g + 4 * half);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v = w0[q] * dhead[0];
                if (!VIEWDEP) v = fmaf(w3[q], dhead[3], fmaf(w2[q], dhead[2], fmaf(w1[q], dhead[1], v)));
                acc[g >> 2][4 * (g & 3) + q] += v;
            }
        }
    }
    // ---- trunk, last layer first: dY_l = dh_{l+1} . [h_{l+1} > 0];  dh_l = W_l[:, h-columns]^T dY_l -------------
    unsigned relu_words[WT / 2];
    load_relu_words<WT>(relu_words, masks, (a.depth - 1) * WT, lane);
#pragma unroll 1
    for (int l = a.depth - 1; l >= 0; --l) {
        relu_backward<WT>(acc, relu_words, dy);
        if (l > 0) load_relu_words<WT>(relu_words, masks, (l - 1) * WT, lane);   // in flight during this layer's products
        store_acc_tile(dy, grads + (l * a.width) * 32, lane);
        if (l == 0) break;
        zero_acc<WT>(acc);
        gemm_segment<WT, WT, WT>(acc, dy, st);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ------------------------------------------------------------------------------------------------
// B2: weight gradients
// ------------------------------------------------------------------------------------------------
struct WgradJob {
    int dy_row0, out_rows, out_tiles;   // rows of the grads tile
    int x_row0, in_rows, in_tiles;      // rows of the acts tile
    int grad_rows, act_rows;            // tile heights (rows per wave block)
    int chunks;
    long long blocks;                   // wave blocks in total
    long long partial_off;              // floats into the partial buffer: [chunk][out_tiles*32][in_tiles*32] then
    long long bias_off;                 //                                  [chunk][out_tiles*32]
    int w_param, w_ld, w_col;           // destination: grad of params[w_param] (out_rows x w_ld), columns from w_col
    int b_param;                        // bias destination or -1
    int half;                           // 1: computed by the f16x3 kernel with dY scaled by wgrad_scale(max of its region)
    // 16-bit tiles (SNERF_PRECISION_F16) are addressable per 16-row piece only:
    int dy_skip;                        // rows of the first dY tile that belong to another job (head tile: sigma | rgb rows)
    int x_kind;                         // SEG_ACC: X rows in natural order; SEG_POINTS_PE / SEG_VIEWS_PE: X is an encoding
                                        // tile in REGISTER order -- column p of the product is encoding index
                                        // pe_index(p) and only indices in [feat_lo, feat_hi) belong to this job
    int feat_lo, feat_hi;
    int x8;                             // SNERF_PRECISION_F16S8: X is an 8-bit (fp8 e4m3) tile image (mlp_device_f16.h store_pieces8)
    // 16-bit modes: a skinny product over the SAME dY rows rides in this job as extra X tiles -- the last x2_tiles of in_tiles
    // come from rows x2_row0 (a whole encoding tile) -- instead of streaming dY a second time from a job of its own ...
    int x2_row0, x2_tiles;
    // ... and that product's own entry stays in the table for the reduction only (launched = 0), as a VIEW of this job's
    // partial sums: partial_ld floats per row (0: in_tiles * 32), its columns from partial_col0
    int launched, partial_ld, partial_col0;
};

// Encoding index held at position p of a 16-bit encoding tile (forward: store_pieces of the pe_h fragments), or -1.
__host__ __device__ inline int pe_position_feature(int p, int kind) {
    const int c = p & 15;
    const int n = 8 * (p >> 4) + 4 * (c >> 3) + (c & 3), h = (c >> 2) & 1;
    return snerf::pe_feature(n, h, kind == snerf::SEG_POINTS_PE ? snerf::kPointsPairs : snerf::kViewsPairs, 16);
}

// Per-region max |dY| words, written by the chain kernels with atomicMax.  kMaxSlots copies of the table, 512 bytes apart,
// indexed by workgroup: with a single copy every wave of the launch hit the same few cache lines of one L2 channel and the
// atomics alone cost 5.6 % of the chain kernel.  Readers take the maximum over the copies (one word per lane).
constexpr int kRegionSlots = 64, kRegionWords = 128;      // 64 x 128 words at workspace floats [448, 448 + 8192)
constexpr int kRegionTableFloat0 = 448;
__device__ __forceinline__ float region_max(const float* workspace, int region) {
    const unsigned w = reinterpret_cast<const unsigned*>(workspace)[kRegionTableFloat0 + (threadIdx.x & (kRegionSlots - 1)) * kRegionWords + region];
    float v = __uint_as_float(w);     // non-negative floats order like their bits
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// One power-of-two scale per dY region for the f16x3 weight-gradient product (the contraction runs over the samples, so a
// per-sample factor cannot be pulled out): max |dY| -> [2^9, 2^10).  Samples whose gradients are < 6e-11 of the largest
// one underflow the fp16 pair; their contribution is below fp32 rounding of the sum anyway.
__host__ __device__ inline float wgrad_scale(float max_abs) {
    if (!(max_abs > 0.0f) || !(max_abs < 3.0e38f)) return 1.0f;
    int e;
    frexpf(max_abs, &e);
    int shift = 10 - e;
    if (shift > 120) shift = 120;
    if (shift < -120) shift = -120;
    return ldexpf(1.0f, shift);
}

// 8 consecutive samples of one tile row (two swizzled 16-byte chunks) -> fp16 hi/lo fragments; `k` scales (dY) or is 1 (X)
__device__ __forceinline__ void split_fragment(const float* __restrict__ row, int c0, int swz, float k, f16x8& hi, f16x8& lo,
                                               float& sum) {
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(row + ((c0 ^ swz) << 2));
    const f32x4 v1 = *reinterpret_cast<const f32x4*>(row + (((c0 + 1) ^ swz) << 2));
    sum += (v0[0] + v0[1]) + (v0[2] + v0[3]) + (v1[0] + v1[1]) + (v1[2] + v1[3]);
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    // two values at a time: every conversion is one packed instruction and the halves land in adjacent fragment lanes
#pragma unroll
    for (int j = 0; j < 4; j += 2) {
        const f32x2 a = f32x2{v0[j], v0[j + 1]} * k, b = f32x2{v1[j], v1[j + 1]} * k;
        const f16x2 ah = __builtin_convertvector(a, f16x2), bh = __builtin_convertvector(b, f16x2);
        const f16x2 al = __builtin_convertvector(a - __builtin_convertvector(ah, f32x2), f16x2);
        const f16x2 bl = __builtin_convertvector(b - __builtin_convertvector(bh, f32x2), f16x2);
        hi[j] = ah[0]; hi[j + 1] = ah[1]; lo[j] = al[0]; lo[j + 1] = al[1];
        hi[4 + j] = bh[0]; hi[5 + j] = bh[1]; lo[4 + j] = bl[0]; lo[5 + j] = bl[1];
    }
}

// One LDS-DMA instruction: every lane moves 16 bytes from its own global address to lds_base + 16 * lane (1 KiB per wave).
// Issued through inline assembly on purpose: for the builtin, the compiler's wait-count insertion cannot tell which LDS
// bytes an outstanding DMA will write (the ping-pong buffer is selected at run time), so it puts `s_waitcnt vmcnt(0)`
// in front of the first LDS read that follows -- i.e. it waits for the PREFETCH of the next block before computing on
// the current one, and the kernel runs as DMA time + compute time instead of their maximum (checked in the ISA and by
// ablation builds).  Here the only vmcnt wait is the explicit one at the top of each iteration.  m0 carries the LDS
// base address of the instruction; nothing else in these kernels uses it.  `nt`: the saved activations and layer gradients
// are read exactly once, by one workgroup (MI355X_MICROARCH.md "nt-weights": set nt on streams that ONE CU reads once);
// measured -2 % on the 16-bit backward, -1.5 % f16x3, -0.5 % fp32.
__device__ __forceinline__ void lds_dma_16(const float* src, const float* lds_dst) {
    const unsigned base = __builtin_amdgcn_readfirstlane(
        (unsigned)(size_t)(__attribute__((address_space(3))) const void*)lds_dst);   // 32-bit LDS byte address
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt" ::"s"(base), "v"(src) : "memory", "m0");
}

// The same with the global address split into a wave-uniform base (SGPR pair) and a 32-bit per-lane byte offset: the
// per-piece address arithmetic then runs on the scalar unit, a 64-bit add per piece and block.
__device__ __forceinline__ void lds_dma_16_base(const void* uniform_base, unsigned lane_byte_offset, unsigned lds_byte_address) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt" ::"s"(lds_byte_address), "v"(lane_byte_offset),
                 "s"(uniform_base)
                 : "memory", "m0");
}

constexpr int kMaxJobs = 16;
struct JobTable {
    int count;
    int wg_start[kMaxJobs + 1];  // workgroup range of job j in the single launch: [wg_start[j], wg_start[j+1])
    WgradJob jobs[kMaxJobs];
};
struct GradPointers {
    float* p[40];
};

// Per-wave register tile of the weight-gradient kernel: NO x NI accumulator tiles of 32x32.  The 4 waves of a workgroup
// form a (WO x WI) grid over the job's (out_tiles x in_tiles) product, WO = out_tiles / NO, WI = in_tiles / NI.
template <int NO, int NI, bool F16>
__global__ void __launch_bounds__(256, 1) wgrad_kernel(JobTable table, const float* __restrict__ grads,
                                                       const float* __restrict__ acts, float* __restrict__ partial,
                                                       const float* __restrict__ zeros) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    int ji = 0;
    while (ji + 1 < table.count && (int)blockIdx.x >= table.wg_start[ji + 1]) ++ji;
    const WgradJob& job = table.jobs[ji];
    const int chunk = blockIdx.x - table.wg_start[ji];
    const int rows_dy = job.out_tiles * 32, rows_x = job.in_tiles * 32, rows = rows_dy + rows_x;
    const int wgrid_i = job.in_tiles / NI;               // WI: 1, 2 or 4
    const int wo = wave / wgrid_i, wi = wave - wo * wgrid_i;
    const bool active = wo * NO < job.out_tiles;          // surplus waves only help with the staging
    const long long per = (job.blocks + job.chunks - 1) / job.chunks;
    const long long b0 = chunk * per, b1 = (b0 + per < job.blocks) ? b0 + per : job.blocks;

    f32x16 acc[NO][NI];
    float bsum[NO];
#pragma unroll
    for (int oo = 0; oo < NO; ++oo) {
        bsum[oo] = 0.0f;
#pragma unroll
        for (int ii = 0; ii < NI; ++ii)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[oo][ii][r] = 0.0f;
    }

    // Staging by LDS-DMA, 1 KiB (= 8 tile rows of 32 samples) per wave instruction.  LDS keeps plain 128-byte rows; the
    // 16-byte chunks of a row are XOR-swizzled by ((row >> 1) & 7) -- applied to the per-lane SOURCE address here and
    // to the read address below -- so that the 16 lanes of a ds_read_b128 group (16 rows, same logical chunk) hit 16
    // different bank quads.  Rows beyond the job's out_rows / in_rows are filled from a zero page.
    const int buf_floats = rows * 32;
    // Per piece of this wave (q = wave + 4k): source address of its first block, worked out ONCE -- per block the staging
    // is then a scalar 64-bit add and one DMA instruction per piece.  (Recomputing the per-lane addresses from the block
    // index inside the loop cost ~50 instructions, two divergent branches and several scalar loads of job fields per
    // piece: more issue time per block than its MFMAs take.)  The swizzle term (4q + lane/16) & 7 has 4q & 7 = 4(wave & 1)
    // for every piece of a wave, so one per-lane offset serves them all.  A piece whose 8 rows straddle the end of the
    // job's rows (head and encoding products) mixes real rows and zero-page lanes: per-lane pointers, rebuilt per block.
    constexpr int kMaxSlots = 16;                           // (8 + 8 tiles) * 4 pieces / 4 waves
    const int per_wave = rows >> 5;                         // pieces / 4 (rows is a multiple of 32)
    const char* slot_src[kMaxSlots];
    int slot_rows[kMaxSlots];                               // real rows in the piece: 8 all, 0 none (zero page), else ragged
    const char* zero_page = reinterpret_cast<const char*>(zeros + 192);      // 1 KiB
    const long long dy_stride = (long long)job.grad_rows * 128, x_stride = (long long)job.act_rows * 128;
#pragma unroll
    for (int k = 0; k < kMaxSlots; ++k) {
        const int r0 = (wave + 4 * k) * 8;
        int real;
        const char* src;
        if (r0 < rows_dy) {
            real = job.out_rows - r0;
            src = reinterpret_cast<const char*>(grads) + (b0 * job.grad_rows + job.dy_row0 + r0) * 128;
        } else {
            real = job.in_rows - (r0 - rows_dy);
            src = reinterpret_cast<const char*>(acts) + (b0 * job.act_rows + job.x_row0 + (r0 - rows_dy)) * 128;
        }
        real = real < 0 ? 0 : (real > 8 ? 8 : real);
        slot_rows[k] = real;
        slot_src[k] = real > 0 ? src : zero_page;
    }
    const unsigned lane_off = (lane >> 3) * 128 + (((lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7)) << 4);
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)lds;
    // Jobs without partial tiles (every 256 x 256 product): piece k of this wave lies k * 4 KiB behind the wave's first
    // piece of its region, so two running bases replace the per-slot state -- which does not fit the scalar registers
    // (the general path below keeps 48 values per wave and pays ~12 v_readlane per piece for it).
    const bool all_full = job.out_rows == rows_dy && job.in_rows == rows_x;
    const int dy_slots = (rows_dy / 8 - wave + 3) >> 2;     // pieces of this wave inside the dY rows
    const char* dy_run = reinterpret_cast<const char*>(grads) + (b0 * job.grad_rows + job.dy_row0 + wave * 8) * 128;
    const char* x_run = reinterpret_cast<const char*>(acts) + (b0 * job.act_rows + job.x_row0 + wave * 8 - rows_dy) * 128;
    // requests the NEXT block in line (blocks are staged strictly in order) into the buffer at `buffer_floats`
    auto stage_next = [&](int buffer_floats) {
        if (all_full) {
#pragma unroll
            for (int k = 0; k < kMaxSlots; ++k)
                if (k < per_wave)
                    lds_dma_16_base(k < dy_slots ? dy_run : x_run, lane_off + k * 4096,
                                    lds_base + (unsigned)buffer_floats * 4 + (wave + 4 * k) * 1024);
            dy_run += dy_stride;
            x_run += x_stride;
            return;
        }
#pragma unroll
        for (int k = 0; k < kMaxSlots; ++k) {
            if (k < per_wave) {
                const unsigned dst = lds_base + (unsigned)buffer_floats * 4 + (wave + 4 * k) * 1024;
                const int real = slot_rows[k];
                if (real == 8 || real == 0) {
                    lds_dma_16_base(slot_src[k], real ? lane_off : lane * 16, dst);
                } else {
                    const char* p = (lane >> 3) < real ? slot_src[k] + lane_off : zero_page + lane * 16;
                    lds_dma_16(reinterpret_cast<const float*>(p), reinterpret_cast<const float*>(lds) + buffer_floats + (wave + 4 * k) * 256);
                }
                if (real) slot_src[k] += (wave + 4 * k) * 8 < rows_dy ? dy_stride : x_stride;
            }
        }
    };

    const int swz = ((lane & 31) >> 1) & 7;
    const int row_off = (lane & 31) * 32;
    const int a_base = wo * NO * 1024;
    const int b_base = rows_dy * 32 + wi * NI * 1024;

    // (read once, before any DMA is in flight: a global load inside the loop makes the compiler wait for vmcnt(0),
    // i.e. for the prefetch of the next block as well)
    const float gk = F16 ? wgrad_scale(region_max(zeros, job.dy_row0 / 32)) : 1.0f;
    if constexpr (F16) {
        // fp16-split product, software-pipelined over the 16-sample k-steps (two per block): the LDS reads and the hi/lo
        // split of step t+1 (~260 VALU instructions) are issued together with the 48 MFMAs of step t, into the other
        // fragment set, so they run in the shadow of the matrix work instead of in front of it (the two used to add up:
        // ~2700 issue cycles per step for 1536 cycles of MFMAs).  Consequences for the staging: every LDS read of block
        // n is done after its FIRST half (the second step's fragments are converted there), so the barrier in the
        // middle of block n both releases its buffer for block n+2 and confirms that block n+1 has landed.
        struct SplitSet {
            f16x8 ah[NO], al[NO], bh[NI], bl[NI];
        };
        // (`counted` = 0 for the look-ahead past the last block, which re-converts valid memory and is never multiplied:
        // keeping the call unconditional keeps it in the MFMAs' basic block, where the scheduler interleaves the two)
        auto convert = [&](SplitSet& f, const float* buf, int kk, float counted) {
            const int c0 = 4 * kk + 2 * half;
            float unused = 0.0f;
#pragma unroll
            for (int oo = 0; oo < NO; ++oo) {
                float part = 0.0f;
                split_fragment(buf + a_base + oo * 1024 + row_off, c0, swz, gk, f.ah[oo], f.al[oo], part);
                bsum[oo] += part * counted;
            }
#pragma unroll
            for (int ii = 0; ii < NI; ++ii) split_fragment(buf + b_base + ii * 1024 + row_off, c0, swz, 1.0f, f.bh[ii], f.bl[ii], unused);
        };
        auto products = [&](const SplitSet& f) {
#pragma unroll
            for (int oo = 0; oo < NO; ++oo)
#pragma unroll
                for (int ii = 0; ii < NI; ++ii) {
                    acc[oo][ii] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[oo], f.bh[ii], acc[oo][ii], 0, 0, 0);
                    acc[oo][ii] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[oo], f.bl[ii], acc[oo][ii], 0, 0, 0);
                    acc[oo][ii] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[oo], f.bh[ii], acc[oo][ii], 0, 0, 0);
                }
        };
        const int nblocks = (int)(b1 - b0);
        SplitSet even, odd;
        if (nblocks > 0) {
            stage_next(0);
            if (nblocks > 1) stage_next(buf_floats);
            // block 0 must be in: everything but the second block's requests (per_wave DMA instructions per wave)
            if (nblocks > 1) wait_vmcnt(per_wave < 63 ? per_wave : 63); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (active) convert(even, lds, 0, 1.0f);
        }
        for (int n = 0; n < nblocks; ++n) {
            const float* cur = lds + (n & 1) * buf_floats;
            const float* nxt = lds + ((n + 1) & 1) * buf_floats;
            if (active) {
                products(even);
                convert(odd, cur, 1, 1.0f);
            }
            // all of this wave's reads of block n are complete once `odd` is converted (the compiler waits for the LDS
            // data before using it); block n+1 was requested at least half a block ago
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __syncthreads();
            if (n + 2 < nblocks) stage_next((n & 1) * buf_floats);
            if (active) {
                products(odd);
                convert(even, n + 1 < nblocks ? nxt : cur, 0, n + 1 < nblocks ? 1.0f : 0.0f);
            }
        }
    } else {
    if (b0 < b1) stage_next(0);
    for (long long b = b0; b < b1; ++b) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // block b has landed for every wave, and every wave is done reading the other buffer
        const float* cur = lds + ((b - b0) & 1) * buf_floats;
        if (b + 1 < b1) stage_next((int)((b - b0 + 1) & 1) * buf_floats);
        if (active) {
            // one 8-sample group at a time (not unrolled): keeps the live fragment registers at 4*(NO+NI) so that nothing
            // spills -- a scratch reload inside this loop would force vmcnt(0) and drain the prefetch DMA issued above
#pragma unroll 1
            for (int g = 0; g < 4; ++g) {
                const int off = (((2 * g + half) ^ swz) << 2) + row_off;
                f32x4 av[NO], bv[NI];
#pragma unroll
                for (int oo = 0; oo < NO; ++oo) av[oo] = *reinterpret_cast<const f32x4*>(cur + a_base + oo * 1024 + off);
#pragma unroll
                for (int ii = 0; ii < NI; ++ii) bv[ii] = *reinterpret_cast<const f32x4*>(cur + b_base + ii * 1024 + off);
#pragma unroll
                for (int oo = 0; oo < NO; ++oo) {
#pragma unroll
                    for (int ii = 0; ii < NI; ++ii)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            acc[oo][ii] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[oo][q], bv[ii][q], acc[oo][ii], 0, 0, 0);
                    bsum[oo] += (av[oo][0] + av[oo][1]) + (av[oo][2] + av[oo][3]);
                }
            }
        }
    }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!active) return;
    // partial[chunk][o*32 + row][i*32 + col], bias partial[chunk][o*32 + row]
    const int in_cols = job.in_tiles * 32;
    float* out = partial + job.partial_off + (long long)chunk * rows_dy * in_cols;
    float* bout = partial + job.bias_off + (long long)chunk * rows_dy;
#pragma unroll
    for (int oo = 0; oo < NO; ++oo) {
        const int o = wo * NO + oo;
#pragma unroll
        for (int ii = 0; ii < NI; ++ii) {
            const int i = wi * NI + ii;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                out[(long long)(o * 32 + row) * in_cols + i * 32 + (lane & 31)] = acc[oo][ii][r];
            }
        }
        if (wi == 0) {
            const float sum = bsum[oo] + __shfl_xor(bsum[oo], 32, 64);
            if (half == 0) bout[o * 32 + lane] = sum;
        }
    }
}


// ---- 16-bit weight gradients (SNERF_PRECISION_F16) ---------------------------------------------------------------
// dW = dY . X^T with X kept as fp16 and dY as bf16 operand PIECES (mlp_device_f16.h store_pieces: 1 KiB = 32 samples x
// 16 features, sample-major).  The contraction runs over the samples, so both operands are read back TRANSPOSED with
// ds_read_b64_tr_b16 (4 samples x 16 features per 16-lane group; two reads = the 8 samples of a lane's k-step fragment):
// X fragments go to the MFMA as they are, dY fragments are widened bf16 -> fp32, scaled by the region's power of two
// (wgrad_scale) and narrowed to fp16 -- 2 of the 10 fragments of the large products.  One fp16 MFMA per product.
// Bound: HBM (1 KiB per sample and 256x256 layer, against 32 MFMA cycles per wave and piece pair).
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// 8-bit variant: per 16-lane group the lanes' 8-byte chunks form an 8-row x 16-column block of bytes (row r = the chunks of lanes
// 2r, 2r+1) and lane j receives column j (tools/probes/ds_read_tr_b8.hip)
__device__ __forceinline__ u32x2 lds_read_tr8(unsigned lds_byte_address, int byte_offset) {
    u32x2 v;
    asm volatile("ds_read_b64_tr_b8 %0, %1 offset:%2" : "=v"(v) : "v"(lds_byte_address), "i"(byte_offset) : "memory");
    return v;
}
__device__ __forceinline__ u32x2 lds_read_tr16(unsigned lds_byte_address, int byte_offset) {
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(lds_byte_address), "i"(byte_offset) : "memory");
    return v;
}
template <typename T>
__device__ __forceinline__ void after_lds_wait(T& v) {  // orders every use of v behind the preceding s_waitcnt asm
    asm volatile("" : "+v"(v));
}
union Frag16 {
    u32x2 d[2];
    f16x8 h;
};
// f(integral_constant<I>), ..., f(integral_constant<N-1>): one inlined copy of the body per index
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

SNERF_STAMP_DEFINE(wgrad16)

constexpr int kPairBytes = 2304;   // LDS image of one 32-row tile: piece 0 at +0, piece 1 at +1152 (bank phase +32 dwords:
constexpr int kPieceGap = 1152;    // the two 16-lane groups of a half-wave read different pieces of the same tile)
constexpr int kWgrad16Buffers = 3;
#if defined(SNERF_PROBE_RING4)  // ablation: a fourth slot for the 8-bit instance (measured: +0.23 ms per config-5 iteration)
constexpr int wgrad16_ring(bool x8, bool) { return x8 ? 4 : kWgrad16Buffers; }
#elif defined(SNERF_PROBE_SMALL_RING)   // ablation: that many slots for the small-job launch (4: 107 -> 138 us, 5: 140 us)
constexpr int wgrad16_ring(bool, bool partial) { return partial ? SNERF_PROBE_SMALL_RING : kWgrad16Buffers; }
#else
constexpr int wgrad16_ring(bool, bool) { return kWgrad16Buffers; }
#endif

// PARTIAL: the small head / encoding jobs -- (8 x 2), (1 x 8), (4 x 1) and (1 x 4) tiles for the main MLP -- share ONE launch
// of the <2, 2> instance: a wave's register tile may then be only partly covered by the job (wave-uniform guards around its
// reads, MFMAs and stores).  As four launches of their own register-tile classes each of them was mostly launch, ramp and
// tail (20-50 us for a few MB of operands; 13 % of the 16-bit training iteration with the reduction, MFMA pipe 4-13 % busy).
// The per-tile arithmetic and the chunking are unchanged, so the partial sums -- and the gradients -- are bit-identical.
// BF (SNERF_PRECISION_BF16): X is saved as bf16 too, so both fragments go to v_mfma_f32_32x32x16_bf16 as they are -- no
// widening, no region scale (bf16 has the range), and the reduction applies no factor (WgradJob::half = 0).
// X8 (SNERF_PRECISION_F16S8; the large class only): a job with x8 set reads its X operand from fp8 tiles -- one KiB per
// tile and block instead of two, one ds_read_b64_tr_b8 per fragment instead of two tr_b16, four v_cvt_scalef32_pk_f16_fp8 --
// and stores its columns through the tile's byte order.
// (Two waves per SIMD for the 8-bit-X instance -- <2, 4> register tiles on a 4 x 2 wave grid, 128 accumulators, so that one
// wave's LDS-read latency and conversions run under the other's MFMAs -- take a 24-KiB block from 1.12 to 0.96 us in the kernel's
// skeleton, tools/probes/lds_dma_depth.hip, and measured the SAME as this instance in the kernel: 101 / 102 us per workgroup
// of one launch.  Not kept.)
template <int NO, int NI, bool PARTIAL = false, bool BF = false, bool X8 = false>
__global__ void __launch_bounds__(256, 1) wgrad16_kernel(JobTable table, const unsigned short* __restrict__ grads,
                                                         const unsigned short* __restrict__ acts, float* __restrict__ partial,
                                                         const float* __restrict__ zeros) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    int ji = 0;
    while (ji + 1 < table.count && (int)blockIdx.x >= table.wg_start[ji + 1]) ++ji;
    const WgradJob& job = table.jobs[ji];
    const int chunk = blockIdx.x - table.wg_start[ji];
    const int rows_dy = job.out_tiles * 32;
    const int wgrid_i = PARTIAL ? (job.in_tiles / NI > 1 ? job.in_tiles / NI : 1) : job.in_tiles / NI;
    const int wo = wave / wgrid_i, wi = wave - wo * wgrid_i;
    // tiles of this wave's register tile that lie inside the job (all of them unless PARTIAL)
    const int no_eff = PARTIAL ? (job.out_tiles - wo * NO < NO ? job.out_tiles - wo * NO : NO) : NO;
    const int ni_eff = PARTIAL ? (job.in_tiles - wi * NI < NI ? job.in_tiles - wi * NI : NI) : NI;
    const bool active = PARTIAL ? (no_eff > 0 && ni_eff > 0) : wo * NO < job.out_tiles;
    const long long per = (job.blocks + job.chunks - 1) / job.chunks;
    const long long b0 = chunk * per, b1 = (b0 + per < job.blocks) ? b0 + per : job.blocks;

    f32x16 acc[NO][NI];
    float bsum[NO];
#pragma unroll
    for (int oo = 0; oo < NO; ++oo) {
        bsum[oo] = 0.0f;
#pragma unroll
        for (int ii = 0; ii < NI; ++ii)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[oo][ii][r] = 0.0f;
    }

    // Staging: one LDS-DMA instruction per piece (lane-linear 1 KiB).  Every wave issues the same number of instructions
    // per block (surplus ones repeat the last piece) so that a counted vmcnt can leave later blocks in flight.
    const bool x8 = X8 && job.x8 != 0;
    const int pieces_dy = job.out_tiles * 2, pieces = pieces_dy + (x8 ? job.in_tiles : job.in_tiles * 2);
    const int per_wave = (pieces + 3) >> 2;
    const int dy_pieces_real = (job.dy_skip + job.out_rows + 15) >> 4;
    const int x_pieces_real = x8 ? (job.in_rows + 31) >> 5 : (job.in_rows + 15) >> 4;
    const int buf_floats = (job.out_tiles + job.in_tiles) * (kPairBytes / 4);
    // Source base, LDS offset and per-block stride of each of this wave's (at most 8) pieces, worked out once: per block
    // the staging then costs one scalar 64-bit add and one DMA instruction per piece.  (Recomputing the addresses from
    // the block index took ~75 scalar instructions per piece, 600 per block and wave -- more issue time than the
    // block's 32 MFMAs run; the kernel sat at 19 % MFMA-busy and 3.7 TB/s whatever the layout or the ring depth.)
    constexpr int kMaxPerWave = 8;                       // (8 + 8 tiles) * 2 pieces / 4 waves
    const char* piece_src[kMaxPerWave];
    unsigned piece_dst[kMaxPerWave];
    long long piece_stride[kMaxPerWave];
    const unsigned lane16 = lane * 16;
#pragma unroll
    for (int k = 0; k < kMaxPerWave; ++k) {
        int q = wave + 4 * k;
        if (q >= pieces) q = pieces - 1;
        const char* src = reinterpret_cast<const char*>(zeros + 192);   // 1 KiB zero page
        long long stride = 0;
        if (q < pieces_dy) {
            if (q < dy_pieces_real) {
                src = reinterpret_cast<const char*>(grads + ((b0 * job.grad_rows + job.dy_row0) * 32 + q * 512));
                stride = (long long)job.grad_rows * 64;
#ifdef SNERF_PROBE_HALF_DY     // traffic ablation (WRONG results): every second dY piece of a wide operand comes from the zero page
                if (job.out_tiles >= 4 && (q & 1)) { src = reinterpret_cast<const char*>(zeros + 192); stride = 0; }
#endif
            }
        } else if (!x8 && job.x2_tiles > 0 && q - pieces_dy >= 2 * (job.in_tiles - job.x2_tiles)) {
            // the rider's X tiles (WgradJob::x2_row0): whole encoding tiles
            src = reinterpret_cast<const char*>(acts + ((b0 * job.act_rows + job.x2_row0) * 32 +
                                                        (q - pieces_dy - 2 * (job.in_tiles - job.x2_tiles)) * 512));
            stride = (long long)job.act_rows * 64;
        } else if (q - pieces_dy < x_pieces_real) {
            // (8-bit tile images are one KiB each, packed: WgradJob::x_row0 already counts them so)
            src = reinterpret_cast<const char*>(acts + ((b0 * job.act_rows + job.x_row0) * 32 + (q - pieces_dy) * 512));
            stride = (long long)job.act_rows * 64;
#ifdef SNERF_PROBE_HALF_X      // traffic ablation (WRONG results): every second X piece of a wide operand comes from the zero page
            if (job.in_tiles >= 4 && ((q - pieces_dy) & 1)) { src = reinterpret_cast<const char*>(zeros + 192); stride = 0; }
#endif
        }
        piece_src[k] = src;
        piece_stride[k] = stride;
        piece_dst[k] = (x8 && q >= pieces_dy) ? (job.out_tiles + (q - pieces_dy)) * kPairBytes : (q >> 1) * kPairBytes + (q & 1) * kPieceGap;
    }
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)lds;
    // requests the NEXT block in line (blocks are staged strictly in order) into ring slot `slot`
    auto stage_next = [&](int slot) {
        const unsigned dst = lds_base + (unsigned)slot * (unsigned)(buf_floats * 4);
#pragma unroll
        for (int k = 0; k < kMaxPerWave; ++k) {
            if (k < per_wave) {
                lds_dma_16_base(piece_src[k], lane16, dst + piece_dst[k]);
                piece_src[k] += piece_stride[k];
            }
        }
    };

    // transposed-read address of this lane inside a tile image (T10): 16-lane group = (piece s, sample octet hh); lane
    // 4q+p of the group points at sample row q, feature columns 4p..4p+3 of the piece
    const int grp = lane >> 4, q4 = (lane >> 2) & 3, p4 = lane & 3;
    const unsigned lane_off = (grp & 1) * kPieceGap + 32 * (8 * (grp >> 1) + q4) + 16 * (p4 & 1) + 8 * (p4 >> 1);
    const float gk = BF ? 1.0f : wgrad_scale(region_max(zeros, job.dy_row0 / 32));
    // 8-bit X: lane (group g = lane >> 4, j = lane & 15) points at byte chunk j & 1 of slot 2 * (8 * (g >> 1) + (j >> 1)) + (g & 1),
    // i.e. sample 8 * (g >> 1) + (j >> 1) of the k-step, lane half g & 1, and receives byte j of its group's eight samples
    const unsigned lane_off8 = 16 * (2 * (8 * (grp >> 1) + ((lane & 15) >> 1)) + (grp & 1)) + 8 * (lane & 1);

    // Software pipeline over the 16-sample k-steps (two per block): the transposed reads of step t+1 are issued before the
    // MFMAs of step t, into the other fragment set, so the LDS latency and the bf16->fp16 conversion of the next step run
    // in the shadow of the current step's matrix work (issued one after the other, a block took ~4800 cycles for 1024
    // cycles of MFMAs: 19 % busy, PMC).  Block n+1 must therefore have landed -- for every wave -- in the MIDDLE of block
    // n: that is where the vmcnt wait and the barrier sit; the barrier also tells that everyone finished reading block
    // n-1 (those reads were waited for during k-step 1 of block n-1), whose buffer then receives block n+2.
    struct FragSet {
        Frag16 a[NO], bx[NI];
    };
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)lds + lane_off;
    const unsigned a_off = wo * NO * kPairBytes, b_off = (job.out_tiles + wi * NI) * kPairBytes;
    auto issue_reads = [&](FragSet& f, int slot, int kk) {
        const unsigned buf = lds0 + (unsigned)slot * (unsigned)(buf_floats * 4) + kk * 512;
        const unsigned a_kk = buf + a_off, b_kk = buf + b_off;
#pragma unroll
        for (int oo = 0; oo < NO; ++oo) {
            if (PARTIAL && oo >= no_eff) continue;      // (a tile outside the job would be read from the next buffer's bytes)
            f.a[oo].d[0] = lds_read_tr16(a_kk, oo * kPairBytes);
            f.a[oo].d[1] = lds_read_tr16(a_kk, oo * kPairBytes + 128);
        }
        if (x8) {
            const unsigned b8 = buf - lane_off + lane_off8 + b_off;
#pragma unroll
            for (int ii = 0; ii < NI; ++ii) f.bx[ii].d[0] = lds_read_tr8(b8, ii * kPairBytes);
            return;
        }
#pragma unroll
        for (int ii = 0; ii < NI; ++ii) {
            if (PARTIAL && ii >= ni_eff) continue;
            f.bx[ii].d[0] = lds_read_tr16(b_kk, ii * kPairBytes);
            f.bx[ii].d[1] = lds_read_tr16(b_kk, ii * kPairBytes + 128);
        }
    };
    auto wait_reads = [&](FragSet& f, f16x8 (&ah)[NO]) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int oo = 0; oo < NO; ++oo) { after_lds_wait(f.a[oo].d[0]); after_lds_wait(f.a[oo].d[1]); }
#pragma unroll
        for (int ii = 0; ii < NI; ++ii) { after_lds_wait(f.bx[ii].d[0]); after_lds_wait(f.bx[ii].d[1]); }
#pragma unroll
        for (int oo = 0; oo < NO; ++oo) {
            if (PARTIAL && oo >= no_eff) continue;
            // bf16 pairs -> fp32 (exact), bias sum in true units, x region scale -> fp16
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const unsigned word = f.a[oo].d[w >> 1][w & 1];
                const f32x2 v = {__uint_as_float(word << 16), __uint_as_float(word & 0xffff0000u)};
                bsum[oo] += v[0] + v[1];
                if constexpr (!BF) {
                    const f16x2 hcv = __builtin_convertvector(v * gk, f16x2);
                    ah[oo][2 * w] = hcv[0]; ah[oo][2 * w + 1] = hcv[1];
                }
            }
            if constexpr (BF) ah[oo] = f.a[oo].h;     // the bf16 fragment as read
        }
        if (x8) {       // eight fp8 bytes (samples 0..7 of this lane's k half) -> the fp16 operand
            const float sc = 1.0f;
#pragma unroll
            for (int ii = 0; ii < NI; ++ii) {
                const unsigned w0 = f.bx[ii].d[0][0], w1 = f.bx[ii].d[0][1];
                const f16x2 e01 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w0, sc, false), e23 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w0, sc, true);
                const f16x2 e45 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w1, sc, false), e67 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w1, sc, true);
                f.bx[ii].h = f16x8{e01[0], e01[1], e23[0], e23[1], e45[0], e45[1], e67[0], e67[1]};
            }
        }
    };
    auto products = [&](const FragSet& f, const f16x8 (&ah)[NO]) {
#pragma unroll
        for (int oo = 0; oo < NO; ++oo)
#pragma unroll
            for (int ii = 0; ii < NI; ++ii) {
                if (PARTIAL && (oo >= no_eff || ii >= ni_eff)) continue;
                acc[oo][ii] = mfma_32x32x16<BF>(ah[oo], f.bx[ii].h, acc[oo][ii]);
            }
    };

    const int nblocks = (int)(b1 - b0);
    SNERF_STAMP_BEGIN();
    FragSet even, odd;
    int stage_slot = 0, read_slot = 0;                       // ring positions of the next block to request / to read
    // RING slots, AHEAD blocks requested before the first is used.  What the large class is bound by (config 5, round 4;
    // tools/probes/lds_dma_depth.hip is this kernel's skeleton): 16-bit X, 32-KiB blocks -- the stream itself, 6.6 TB/s with
    // this addressing (244 us + the launch's fixed costs: ramp, 64 MiB of partial sums at the end = the measured 260); 8-bit X,
    // 24-KiB blocks -- one wave's chain of LDS-read latency and 32 MFMAs per block, ~1.1 us (215 us + fixed costs = 237), not
    // its bytes (181 us).  More blocks in flight are neutral in the skeleton and LOSE in this kernel: a fourth slot
    // (SNERF_PROBE_RING4) 256-260 us, the same three slots refilled one barrier earlier (SNERF_PROBE_DEEP) 243 us (255 for
    // 16-bit X), the small-job launch 107 -> 138 us with a fourth or fifth slot (SNERF_PROBE_SMALL_RING).  Not explained.
    constexpr int RING = wgrad16_ring(X8, PARTIAL);
    constexpr bool OVERLAP = !PARTIAL && NO * NI >= 8;       // (the schedule below)
    // The serial schedule reads block n's second k-step after the mid-block barrier, so block n's slot cannot be refilled
    // there: AHEAD = RING - 1.  In the overlapped schedule every read of block n has LANDED before that barrier (they are
    // waited for inside the first k-step), so the barrier would free block n's own slot: the probe requests one block more.
#ifdef SNERF_PROBE_DEEP
    constexpr int AHEAD = OVERLAP ? RING : RING - 1;
#else
    constexpr int AHEAD = RING - 1;
#endif
    auto advance = [](int& slot) { slot = slot == RING - 1 ? 0 : slot + 1; };
    auto first_block = [&]() {
        const int ahead = nblocks < AHEAD ? nblocks : AHEAD;
        for (int k = 0; k < ahead; ++k) { stage_next(stage_slot); advance(stage_slot); }
        wait_vmcnt((ahead - 1) * per_wave);
        __builtin_amdgcn_s_barrier();                       // block 0 is in for every wave
    };
    // block n+1 (requested half a block ago or earlier) is in for every wave, everyone is done with block n-1 (serial
    // schedule) / block n (overlapped): its slot is requested for block n + AHEAD
    auto next_block = [&](int n) {
        if (AHEAD == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else { const int behind = nblocks - (n + 2) < AHEAD - 2 ? nblocks - (n + 2) : AHEAD - 2; wait_vmcnt(behind * per_wave); }
        __builtin_amdgcn_s_barrier();
        if (n + AHEAD < nblocks) { stage_next(stage_slot); advance(stage_slot); }
    };

    // The large register tiles (one wave per SIMD, 16 or 8 MFMAs per k-step) run the conversions of k-step t+1 in the
    // SHADOW of k-step t's MFMAs: issued one after the other -- convert, then 16 MFMAs back to back -- the wave sat
    // issue-stalled behind the matrix pipe for a third of the time and issued VALU work with the pipe idle for another
    // 40 % (PMC, r04_pmc_train_f16.json): ~3200 cycles per block for 1024 cycles of MFMAs, and the 8-bit X operand --
    // fewer bytes, 64 more conversions per block -- came out SLOWER than the 16-bit one.  Order per k-step: the first
    // quarter of the MFMAs (the next step's transposed reads, issued at the end of the previous step, land meanwhile),
    // the LDS wait, then the remaining MFMAs with the next step's conversion tasks dealt out between them.  Same
    // arithmetic in the same order per accumulator and bias sum: results are bit-identical to the serial schedule.
    if constexpr (OVERLAP) {
        auto pipeline = [&](auto x8_job) __attribute__((always_inline)) {
            constexpr bool X8J = decltype(x8_job)::value;
            f16x8 ah_even[NO], ah_odd[NO];
            auto request = [&](FragSet& f, int slot, int kk) {
                const unsigned buf = lds0 + (unsigned)slot * (unsigned)(buf_floats * 4) + kk * 512;
                const unsigned a_kk = buf + a_off, b_kk = buf + b_off;
#pragma unroll
                for (int oo = 0; oo < NO; ++oo) {
                    f.a[oo].d[0] = lds_read_tr16(a_kk, oo * kPairBytes);
                    f.a[oo].d[1] = lds_read_tr16(a_kk, oo * kPairBytes + 128);
                }
                if constexpr (X8J) {
                    const unsigned b8 = buf - lane_off + lane_off8 + b_off;
#pragma unroll
                    for (int ii = 0; ii < NI; ++ii) f.bx[ii].d[0] = lds_read_tr8(b8, ii * kPairBytes);
                } else {
#pragma unroll
                    for (int ii = 0; ii < NI; ++ii) {
                        f.bx[ii].d[0] = lds_read_tr16(b_kk, ii * kPairBytes);
                        f.bx[ii].d[1] = lds_read_tr16(b_kk, ii * kPairBytes + 128);
                    }
                }
            };
            auto landed = [&](FragSet& f) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int oo = 0; oo < NO; ++oo) { after_lds_wait(f.a[oo].d[0]); after_lds_wait(f.a[oo].d[1]); }
#pragma unroll
                for (int ii = 0; ii < NI; ++ii) { after_lds_wait(f.bx[ii].d[0]); if (!X8J) after_lds_wait(f.bx[ii].d[1]); }
            };
            constexpr int TASKS = NO * 4 + (X8J ? NI : 0);
            // task t < 4 NO: word t & 3 of dY tile t >> 2 (bf16 pair -> fp32 (exact), bias sum in true units, x region scale
            // -> fp16); then, 8-bit X: tile t - 4 NO (eight fp8 bytes -> the fp16 operand)
            auto convert = [&](FragSet& f, f16x8 (&ah)[NO], auto task) __attribute__((always_inline)) {
                constexpr int t = decltype(task)::value;
                if constexpr (t < NO * 4) {
                    constexpr int oo = t >> 2, w = t & 3;
                    const unsigned word = f.a[oo].d[w >> 1][w & 1];
                    const f32x2 v = {__uint_as_float(word << 16), __uint_as_float(word & 0xffff0000u)};
                    bsum[oo] += v[0] + v[1];
                    if constexpr (!BF) {
                        const f16x2 hcv = __builtin_convertvector(v * gk, f16x2);
                        ah[oo][2 * w] = hcv[0]; ah[oo][2 * w + 1] = hcv[1];
                    } else if constexpr (w == 3) {
                        ah[oo] = f.a[oo].h;     // the bf16 fragment as read
                    }
                } else {
                    constexpr int ii = t - NO * 4;
                    const unsigned w0 = f.bx[ii].d[0][0], w1 = f.bx[ii].d[0][1];
                    if constexpr (BF) {       // (SNERF_PRECISION_BF16S8: the bf16 operand of the bf16 MFMA)
                        typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
                        const bf16x2_t e01 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w0, 1.0f, false), e23 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w0, 1.0f, true);
                        const bf16x2_t e45 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w1, 1.0f, false), e67 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w1, 1.0f, true);
                        f.bx[ii].d[0] = u32x2{__builtin_bit_cast(unsigned, e01), __builtin_bit_cast(unsigned, e23)};
                        f.bx[ii].d[1] = u32x2{__builtin_bit_cast(unsigned, e45), __builtin_bit_cast(unsigned, e67)};
                    } else {
                        const f16x2 e01 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w0, 1.0f, false), e23 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w0, 1.0f, true);
                        const f16x2 e45 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w1, 1.0f, false), e67 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w1, 1.0f, true);
                        f.bx[ii].h = f16x8{e01[0], e01[1], e23[0], e23[1], e45[0], e45[1], e67[0], e67[1]};
                    }
                }
            };
            auto convert_all = [&](FragSet& f, f16x8 (&ah)[NO]) {
                static_for<0, TASKS>([&](auto task) __attribute__((always_inline)) { convert(f, ah, task); });
            };
            // MFMAs of one k-step (f, ah); `with_next`: the next step's fragments g become operands (gh) meanwhile
            auto step = [&](const FragSet& f, const f16x8 (&ah)[NO], FragSet& g, f16x8 (&gh)[NO], auto with_next) __attribute__((always_inline)) {
                constexpr bool NEXT = decltype(with_next)::value;
                constexpr int M = NO * NI, LEAD = M / 4, SLOTS = M - LEAD + 1;
                static_for<0, M>([&](auto index) __attribute__((always_inline)) {
                    constexpr int m = decltype(index)::value, ii = m / NO, oo = m % NO;
                    acc[oo][ii] = mfma_32x32x16<BF>(ah[oo], f.bx[ii].h, acc[oo][ii]);
                    if constexpr (NEXT && m >= LEAD - 1) {
                        if constexpr (m == LEAD - 1) landed(g);
                        constexpr int t0 = (m - (LEAD - 1)) * TASKS / SLOTS, t1 = (m + 1 - (LEAD - 1)) * TASKS / SLOTS;
                        static_for<t0, t1>([&](auto task) __attribute__((always_inline)) { convert(g, gh, task); });
                    }
                });
            };
            if (nblocks > 0) {
                first_block();
                if (active) {
                    request(even, read_slot, 0);
                    landed(even);
                    convert_all(even, ah_even);
                    request(odd, read_slot, 1);
                }
            }
#ifdef SNERF_PROBE_WGRAD_NOMATH      // ablation (WRONG results): the block stream alone -- requests, waits and barriers, no LDS reads, no MFMAs
            for (int n = 0; n + 1 < nblocks; ++n) { advance(read_slot); next_block(n); }
            return;
#endif
            for (int n = 0; n + 1 < nblocks; ++n) {
                if (active) step(even, ah_even, odd, ah_odd, std::true_type{});
                advance(read_slot);
                next_block(n);
                if (active) {
                    request(even, read_slot, 0);
                    step(odd, ah_odd, even, ah_even, std::true_type{});
                    request(odd, read_slot, 1);
                }
            }
            if (nblocks > 0 && active) {                     // the last block: nothing behind its second k-step
                step(even, ah_even, odd, ah_odd, std::true_type{});
                step(odd, ah_odd, even, ah_even, std::false_type{});
            }
        };
        if (X8 && x8) pipeline(std::bool_constant<X8>{}); else pipeline(std::false_type{});
    } else {
        f16x8 ah[NO];
        if (nblocks > 0) {
            first_block();
            if (active) issue_reads(even, read_slot, 0);
        }
        for (int n = 0; n < nblocks; ++n) {
            if (active) {
                wait_reads(even, ah);
                issue_reads(odd, read_slot, 1);
                products(even, ah);
            }
            advance(read_slot);
            if (n + 1 < nblocks) next_block(n);
            if (active) {
                wait_reads(odd, ah);
                if (n + 1 < nblocks) issue_reads(even, read_slot, 0);
                products(odd, ah);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (NO == 2 && NI == 8) SNERF_STAMP_END(wgrad16);      // (the large class)
    if (!active) return;
    const int in_cols = job.in_tiles * 32;
    // 8-bit X: accumulator column n = (lane half hh = n >> 4, byte b = n & 15 of the slot) is feature 16 (b >> 3) + 8 ((b >> 2) & 1) + 4 hh + (b & 3)
    const int n31 = lane & 31;
    const int column = x8 ? 16 * ((n31 & 15) >> 3) + 8 * ((n31 >> 2) & 1) + 4 * (n31 >> 4) + (n31 & 3) : n31;
    float* out = partial + job.partial_off + (long long)chunk * rows_dy * in_cols;
    float* bout = partial + job.bias_off + (long long)chunk * rows_dy;
#pragma unroll
    for (int oo = 0; oo < NO; ++oo) {
        if (PARTIAL && oo >= no_eff) continue;
        const int o = wo * NO + oo;
#pragma unroll
        for (int ii = 0; ii < NI; ++ii) {
            if (PARTIAL && ii >= ni_eff) continue;
            const int i = wi * NI + ii;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                out[(long long)(o * 32 + row) * in_cols + i * 32 + column] = acc[oo][ii][r];
            }
        }
        if (wi == 0) {
            const float sum = bsum[oo] + __shfl_xor(bsum[oo], 32, 64);
            if (half == 0) bout[o * 32 + lane] = sum;
        }
    }
}

__global__ void __launch_bounds__(256) clear_words_kernel(unsigned* __restrict__ words, int count) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) words[i] = 0u;
}

// B3: fixed-order sum over chunks, scattered into the reference-layout gradient tensors.  A block owns 64 consecutive
// output elements; its four waves each fold a quarter of the chunks (coalesced 256-byte rows, eight independent loads
// in flight per lane -- the small head/encoding jobs have few outputs but up to 512 chunks, so a single serial loop
// per output would be pure HBM latency), then wave 0 adds the four quarter sums in order.  The summation tree depends
// only on the chunk count, so results are bit-reproducible run to run.
__global__ void __launch_bounds__(256) reduce_kernel(JobTable table, const float* __restrict__ partial, GradPointers ptrs,
                                                     int accumulate) {
    __shared__ float quarter[4][64];
    const WgradJob& job = table.jobs[blockIdx.y];
    const float unscale =
        job.half ? 1.0f / wgrad_scale(region_max(partial, job.dy_row0 / 32)) : 1.0f;
    float* __restrict__ grad_w = ptrs.p[job.w_param];
    float* __restrict__ grad_b = job.b_param >= 0 ? ptrs.p[job.b_param] : nullptr;
    const int in_cols = job.partial_ld ? job.partial_ld : job.in_tiles * 32, rows_dy = job.out_tiles * 32;
    const long long nw = (long long)job.out_rows * job.in_rows;
    const long long total = nw + (grad_b ? job.out_rows : 0);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c0 = (int)((long long)job.chunks * wave / 4), c1 = (int)((long long)job.chunks * (wave + 1) / 4);
    for (long long base = (long long)blockIdx.x * 64; base < total; base += (long long)gridDim.x * 64) {
        const long long idx = base + lane;
        float s = 0.0f;
        int o = 0, i = 0;
        if (idx < total) {
            const float* p;
            long long stride;
            if (idx < nw) {
                o = (int)(idx / job.in_rows);
                i = (int)(idx - (long long)o * job.in_rows);
                p = partial + job.partial_off + (long long)(o + job.dy_skip) * in_cols + job.partial_col0 + i;
                stride = (long long)rows_dy * in_cols;
            } else {
                o = (int)(idx - nw);
                p = partial + job.bias_off + o + job.dy_skip;
                stride = rows_dy;
            }
            int c = c0;
            for (; c + 8 <= c1; c += 8) {
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = __builtin_nontemporal_load(p + (long long)(c + k) * stride);   // partial sums: read once
#pragma unroll
                for (int k = 0; k < 8; ++k) s += v[k];
            }
            for (; c < c1; ++c) s += __builtin_nontemporal_load(p + (long long)c * stride);
        }
        quarter[wave][lane] = s;
        __syncthreads();
        if (wave == 0 && idx < total) {
            const float sum = ((quarter[0][lane] + quarter[1][lane]) + quarter[2][lane]) + quarter[3][lane];
            if (idx < nw) {
                int col = i;
                if (job.x_kind != snerf::SEG_ACC) {   // 16-bit encoding tile: position -> encoding index (or padding)
                    const int e = pe_position_feature(i, job.x_kind);
                    col = (e >= job.feat_lo && e < job.feat_hi) ? e - job.feat_lo : -1;
                }
                // accumulate: dL/dparam is added to what the tensor holds (the trainer's later sub-batches) -- one
                // read-modify-write per element by its only writer, so the result stays bit-reproducible
                if (col >= 0) {
                    float* dst = grad_w + (long long)o * job.w_ld + job.w_col + col;
                    *dst = accumulate ? *dst + sum * unscale : sum * unscale;
                }
            } else {
                grad_b[o] = accumulate ? grad_b[o] + sum : sum;
            }
        }
        __syncthreads();
    }
}

// Workgroups (= partial sums the reduction folds afterwards) of a small head / encoding job.  Every chunk ends with a partial
// of out_tiles x in_tiles x 4 KiB that is written once and read once, so the count trades staging parallelism against
// partial-sum traffic (the two 8x2-tile encoding jobs of the main MLP: 64 KB per chunk).  -DSNERF_SMALL_CHUNKS=n for A/B builds.
// 512 dated from one launch per register-tile class (one job per launch); with the four or five small jobs of an MLP in ONE launch
// it left a workgroup 8 blocks of a 64-samples-per-ray pass to stream and a 64 KB partial to write: config-5 backward per
// iteration 5.05 ms (384), 4.85 (256), 4.85 (128), 4.94 (64) against 5.25-5.49 at 512, same box; blocks / 32 clamped to
// 64 .. 384 measured the same as a fixed 192.
#ifndef SNERF_SMALL_CHUNKS
#define SNERF_SMALL_CHUNKS 192
#endif
#ifndef SNERF_MID_BLOCKS
#define SNERF_MID_BLOCKS 32
#endif
constexpr int kSmallJobChunks = SNERF_SMALL_CHUNKS;

// Register tile (NO, NI) per wave for a job of (out_tiles x in_tiles): the 4 waves must cover it as a WO x WI grid.
inline void wave_tile(const WgradJob& j, int* no, int* ni) {
    const int ot = j.out_tiles, it = j.in_tiles;
    if (ot >= 4) { *no = ot / 4; *ni = it; return; }          // waves split the out tiles
    if (ot == 2) { *no = 1; *ni = it >= 2 ? it / 2 : 1; return; }  // 2 x 2 wave grid (or 2 x 1)
    *no = 1; *ni = it >= 4 ? it / 4 : 1;                        // one out tile: waves split the in tiles
}

struct Workspace {
    long long grads_floats, partial_floats, total_floats;
    std::vector<WgradJob> jobs;
};

// `f16`: SNERF_PRECISION_F16 -- dY and X are 16-bit operand pieces (same row numbers; encoding tiles in register order)
// `s8`: SNERF_PRECISION_F16S8 -- the trunk activations h_1 .. h_D-1 are 8-bit tile images (jobs of trunk layers 1 .. D-1)
Workspace plan_workspace(const snerf::MlpPlan& p, long long total_samples, bool f16 = false, bool s8 = false) {
    Workspace w;
    // wave blocks the chain writes and the weight-gradient jobs contract over: whole workgroups of the chain kernel (eight
    // waves in the 16-bit mode, mlp_backward_f16.hip chain_waves; snerf_mlp_saved_floats sizes the saved tiles the same way)
    const long long blocks = f16 ? (total_samples + 255) / 256 * 8 : (total_samples + 127) / 128 * 4;
    w.grads_floats = blocks * p.grad_rows() * 32;   // (the 16-bit tiles use the first half)
    long long off = kRegionTableFloat0 + kRegionSlots * kRegionWords;
    // [0, 64): zero page for padded rows; [192, 448): 1 KiB zero page; [448, 8640): per-region max |dY| words (region_max)
    auto add = [&](int dy_row0, int out_rows, int x_row0, int in_rows, int w_param, int w_ld, int w_col, int b_param,
                   int x_kind = snerf::SEG_ACC) {
        WgradJob j;
        j.dy_row0 = dy_row0; j.out_rows = out_rows; j.dy_skip = 0;
        j.x_row0 = x_row0; j.in_rows = in_rows; j.x_kind = snerf::SEG_ACC; j.feat_lo = 0; j.feat_hi = 0;
        if (f16) {
            j.dy_skip = dy_row0 % 32;                 // the head tile's rgb rows
            j.dy_row0 = dy_row0 - j.dy_skip;
            if (x_kind != snerf::SEG_ACC) {           // the whole encoding tile; the reduction picks this job's indices
                const int tile_row0 = x_kind == snerf::SEG_POINTS_PE ? p.act_pe() : p.act_pev();
                j.x_kind = x_kind;
                j.feat_lo = x_row0 - tile_row0; j.feat_hi = j.feat_lo + in_rows;
                j.x_row0 = tile_row0;
                j.in_rows = x_kind == snerf::SEG_POINTS_PE ? 64 : 32;
            }
        }
        j.out_tiles = (j.dy_skip + out_rows + 31) / 32;
        j.in_tiles = (j.in_rows + 31) / 32;
        j.grad_rows = f16 ? p.grad16_rows() : p.grad_rows(); j.act_rows = f16 ? p.act16_rows() : p.act_rows();
        j.chunks = 1; j.blocks = blocks; j.partial_off = 0; j.bias_off = 0;
        j.w_param = w_param; j.w_ld = w_ld; j.w_col = w_col; j.b_param = b_param; j.half = f16 ? 1 : 0;
        // (256-wide trunks only: their products form the large class; mlp_forward_half_kernel.h stores fp8 under the same condition)
        j.x8 = (s8 && p.wt == 8 && x_kind == snerf::SEG_ACC && x_row0 >= p.act_h(1) && x_row0 < p.act_h(p.depth)) ? 1 : 0;
        // ... packed: layer l's eight KiB tiles follow layer l-1's, from where h_1's 16-bit rows would begin (64-byte rows)
        if (j.x8) j.x_row0 = p.act_h(1) + (x_row0 - p.act_h(1)) / 2;
        j.x2_row0 = 0; j.x2_tiles = 0; j.launched = 1; j.partial_ld = 0; j.partial_col0 = 0;
        w.jobs.push_back(j);
    };
    const int d = p.depth, wd = p.width;
    // trunk layer l: dY_l x [encoding | h_l]
    add(p.grad_y(0), wd, p.act_pe(), p.pts_in, 0, p.pts_in, 0, 1, snerf::SEG_POINTS_PE);
    for (int l = 1; l < d; ++l) {
        const bool skip_in = (l == 5);
        const int ld = wd + (skip_in ? p.pts_in : 0);
        add(p.grad_y(l), wd, p.act_h(l), wd, 2 * l, ld, skip_in ? p.pts_in : 0, 2 * l + 1);
        if (skip_in) add(p.grad_y(l), wd, p.act_pe(), p.pts_in, 2 * l, ld, 0, -1, snerf::SEG_POINTS_PE);
    }
    // heads: rows of the head tile are [d sigma_raw, d rgb_pre(3)]
    if (p.view_dependent) {
        add(p.grad_head(), 1, p.act_h(d), wd, 2 * d, wd, 0, 2 * d + 1);                      // pts_output_linear (1 row)
        add(p.grad_feature(), wd, p.act_h(d), wd, 2 * d + 2, wd, 0, 2 * d + 3);               // feature_linear
        const int ldv = wd + p.extra + p.views_pe;
        add(p.grad_yv(), p.views_width, p.act_feature(), wd, 2 * d + 4, ldv, 0, 2 * d + 5);    // views_linears.0 | feature
        if (p.sigma_pe)
            add(p.grad_yv(), p.views_width, p.act_pe() + p.pts_in, p.extra, 2 * d + 4, ldv, wd, -1, snerf::SEG_POINTS_PE);
        add(p.grad_yv(), p.views_width, p.act_pev(), p.views_pe, 2 * d + 4, ldv, wd + p.extra, -1, snerf::SEG_VIEWS_PE);
        add(p.grad_head() + 1, 3, p.act_hv(), p.views_width, 2 * d + 6, p.views_width, 0, 2 * d + 7);  // views_output_linear
    } else {
        add(p.grad_head(), 4, p.act_h(d), wd, 2 * d, wd, 0, 2 * d + 1);                      // pts_output_linear (4 rows)
    }
    // 16-bit modes: views_linears.0 | view-direction encoding (4 x 1 tiles: 8 KiB of dY and 2 KiB of X per block, in the
    // small-job launch) becomes a ninth X tile of views_linears.0 | feature, which streams the same dY.
    // (Measured and NOT kept: the density head (1 x 8 tiles over h_D) as a VALU side sum inside feature_linear's workgroups --
    // ~100 VALU per k-step there made the large launch 67 us slower for 17 us less in the small one.)
    if (f16) {
        for (WgradJob& host : w.jobs) {
            if (host.x_kind != snerf::SEG_ACC || host.out_tiles != 4 || host.in_tiles != 8 || host.dy_skip != 0) continue;
            for (WgradJob& rider : w.jobs) {
                if (rider.x_kind != snerf::SEG_VIEWS_PE || rider.dy_row0 != host.dy_row0 || rider.out_rows != host.out_rows ||
                    rider.dy_skip != 0 || rider.in_tiles != 1 || rider.launched == 0 || host.x2_tiles != 0)
                    continue;
                host.x2_row0 = rider.x_row0; host.x2_tiles = rider.in_tiles;
                rider.launched = 0; rider.partial_col0 = host.in_tiles * 32;
                host.in_tiles += rider.in_tiles;
                rider.partial_ld = host.in_tiles * 32;
            }
        }
    }
    // heaviest products first, so the tail of the single launch is made of the small head/encoding jobs
    std::stable_sort(w.jobs.begin(), w.jobs.end(), [](const WgradJob& a, const WgradJob& b) {
        return a.out_tiles * a.in_tiles > b.out_tiles * b.in_tiles;
    });
    // Chunks (= workgroups, = partial sums to fold afterwards) per job.  Jobs of one register-tile class share a launch.
    //  * Large products (>= 32 tiles) are MFMA-bound and hold one workgroup per CU (256 accumulator registers per
    //    lane): the class as a whole gets ~256 workgroups, however many samples there are.  Every chunk ends with a
    //    256 KB partial that the reduction has to read back, so more chunks than CUs only buys HBM traffic
    //    (128 chunks x 8 trunk layers was 256 MB per backward call; now 64 MB).
    //  * Small products (heads, encodings) are DMA-latency-bound with tiny epilogues: as many workgroups as there are
    //    8-block pieces, up to 512, so that every CU holds two of them.
    // Workgroups of the large class as a whole: one per CU, but not fewer than 16 blocks each (round 5, VERDICT r4 #3 "workgroup
    // counts scaled with rows").  Every workgroup ends with a 256 KB partial that is written once and read back by the reduction
    // whatever it contracted over: one rank's share of a strong-scaled batch (256-512 rows x 64 samples = 512-1024 blocks) paid the
    // full 64 MiB + reduction per level for 1/8 of the operands, and its levels run side by side (render.hip), so the CUs a level
    // leaves are not idle.  SNERF_LARGE_MIN_BLOCKS for A/B builds (0 = the fixed 256 of rounds 3-4).
#ifndef SNERF_LARGE_MIN_BLOCKS
#define SNERF_LARGE_MIN_BLOCKS 16
#endif
    const long long large_budget = SNERF_LARGE_MIN_BLOCKS > 0
        ? std::min<long long>(256, std::max<long long>(64, blocks / SNERF_LARGE_MIN_BLOCKS)) : 256;
    for (WgradJob& j : w.jobs) {
        int no, ni;
        wave_tile(j, &no, &ni);
        long long chunks, cap;
        if (j.out_tiles * j.in_tiles >= 32) {
            // ... shared among the jobs of the launch in proportion to the bytes a block of theirs streams (an 8-bit X
            // operand makes a block 24 KiB instead of 32): the launch ends with its slowest job, and with equal shares that
            // was feature_linear, whose X (h_D) stays 16-bit, 260 us, however fast the seven fp8 jobs ran
            long long peers_weight = 0;
            // (measured with in-kernel stamps, config 5, two boxes: a block of 24 KiB costs its workgroup 0.76 / 0.87 of what
            // one of 32 KiB does -- tools/probes/wgrad_wg_times.py; 0.80 here)
            auto weight = [](const WgradJob& k) { return 20 * k.out_tiles + (k.x8 ? 12 : 20) * k.in_tiles; };
            for (const WgradJob& k : w.jobs) {
                int ko, ki;
                wave_tile(k, &ko, &ki);
                if (ko == no && ki == ni && k.out_tiles * k.in_tiles >= 32 && k.launched) peers_weight += weight(k);
            }
            chunks = large_budget * weight(j) / peers_weight;   // rounded DOWN: 7 jobs x 37 chunks = 259 workgroups ran as 256 + a second round of 3
            // (a lone job of fewer than 64 tile products -- the views layer's 4 x 9 -- ends every workgroup with a 144 KB
            // partial: at least SNERF_MID_BLOCKS blocks of operands per workgroup; config 5: 128 instead of 256 workgroups
            // for a 64-samples-per-ray pass is worth 0.04 ms per iteration, 64 costs 0.25)
            // (round 5: ... but never fewer than 64 workgroups while a workgroup still gets four blocks -- one rank's share of a
            // strong-scaled batch, 256 rays x 64 samples = 512 blocks, left this job on 16 CUs: 33 us per launch for 0.3 us of
            // operand traffic, profiles/r05_train_share_512_kernel_stats.csv)
            if (j.out_tiles * j.in_tiles < 64 && chunks > blocks / SNERF_MID_BLOCKS)
                chunks = std::min(chunks, std::max<long long>({blocks / SNERF_MID_BLOCKS, std::min<long long>(64, blocks / 4), 1}));
            cap = blocks / 8;
        } else {
#ifndef SNERF_SMALL_CHUNKS_FP32
#define SNERF_SMALL_CHUNKS_FP32 256      // (512 -> 256: config-5 backward 29.06 -> 28.88 ms in fp32, 13.10 -> 13.02 in f16x3)
#endif
            chunks = f16 ? kSmallJobChunks : SNERF_SMALL_CHUNKS_FP32;     // (fp32 / f16x3: one launch per register-tile class, one or two jobs each)
            cap = blocks / 8;
        }
        if (chunks > cap) chunks = cap;
        if (chunks < 1) chunks = 1;
        j.chunks = (int)chunks;
    }
    // what the rounding left of the 256 goes, one workgroup each, to the large class's lightest jobs (first in the table
    // among equals): 7 fp8 jobs x 30 + 38 = 248 becomes 7 x 31 + 38 + 1
    {
        int used = 0, members = 0;
        for (const WgradJob& j : w.jobs) {
            int no, ni;
            wave_tile(j, &no, &ni);
            if (no == 2 && ni == 8 && j.launched && j.out_tiles * j.in_tiles >= 32) { used += j.chunks; ++members; }
        }
        for (int spare = (int)large_budget - used; members > 1 && spare > 0;) {
            bool given = false;
            for (int light = 1; light >= 0 && spare > 0; --light)
                for (WgradJob& j : w.jobs) {
                    int no, ni;
                    wave_tile(j, &no, &ni);
                    if (no != 2 || ni != 8 || !j.launched || j.out_tiles * j.in_tiles < 32 || (j.x8 != 0) != (light != 0)) continue;
                    if (spare > 0 && j.chunks < blocks / 8) { ++j.chunks; --spare; given = true; }
                }
            if (!given) break;
        }
    }
    for (WgradJob& j : w.jobs) {
        if (!j.launched) continue;        // (a view of its host's partial sums: below)
        j.partial_off = off;
        off += (long long)j.chunks * j.out_tiles * 32 * j.in_tiles * 32;
        j.bias_off = off;
        off += (long long)j.chunks * j.out_tiles * 32;
    }
    for (WgradJob& rider : w.jobs) {
        if (rider.launched) continue;
        for (const WgradJob& host : w.jobs)
            if (host.launched && host.x2_tiles > 0 && host.dy_row0 == rider.dy_row0 && host.x2_row0 == rider.x_row0) {
                rider.chunks = host.chunks; rider.partial_off = host.partial_off; rider.bias_off = host.bias_off;
            }
    }
    w.partial_floats = off;
    w.total_floats = w.grads_floats + w.partial_floats;
    return w;
}

template <int WT, int VT, bool VIEWDEP>
int launch_chain(const ChainArgs& a, hipStream_t stream) {
    const long long blocks = (a.total + 127) / 128;
    const size_t lds_bytes = 2 * sizeof(float) * SlabStream<WT>::kBufFloats;
    auto kernel = mlp_backward_chain_kernel<WT, VT, VIEWDEP>;
    static snerf::DeviceOnce configured;   // per device: the attribute belongs to (kernel, device)
    const int attr = snerf::raise_dynamic_lds(configured, reinterpret_cast<const void*>(kernel), (int)lds_bytes, "mlp_backward");
    if (attr != SNERF_OK) return attr;
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(256), lds_bytes, stream, a);
    return snerf::check_launch("mlp_backward(chain)");
}

template <int NO, int NI, bool F16 = false>
int launch_wgrad(const JobTable& table, const float* grads, const float* acts, float* partial, const float* zeros,
                 hipStream_t stream) {
    int max_rows = 0;
    for (int j = 0; j < table.count; ++j) {
        const int r = (table.jobs[j].out_tiles + table.jobs[j].in_tiles) * 32;
        if (r > max_rows) max_rows = r;
    }
    const size_t lds_bytes = 2 * sizeof(float) * 32 * (size_t)max_rows;  // double-buffered [rows][32 samples]
    auto kernel = wgrad_kernel<NO, NI, F16>;
    static snerf::DeviceOnce configured;   // per device: the attribute belongs to (kernel, device)
    const int attr = snerf::raise_dynamic_lds(configured, reinterpret_cast<const void*>(kernel), 2 * 4 * 32 * 512, "mlp_backward");
    if (attr != SNERF_OK) return attr;
    hipLaunchKernelGGL(kernel, dim3((unsigned)table.wg_start[table.count]), dim3(256), lds_bytes, stream, table, grads, acts,
                       partial, zeros);
    return snerf::check_launch("mlp_backward(wgrad)");
}

}  // namespace

template <int NO, int NI, bool PARTIAL = false, bool BF = false, bool X8 = false>
int launch_wgrad16(const JobTable& table, const float* grads, const float* acts, float* partial, hipStream_t stream) {
    int max_tiles = 0;
    for (int j = 0; j < table.count; ++j) max_tiles = std::max(max_tiles, table.jobs[j].out_tiles + table.jobs[j].in_tiles);
    const size_t lds_bytes = (size_t)wgrad16_ring(X8, PARTIAL) * max_tiles * kPairBytes;
    auto kernel = wgrad16_kernel<NO, NI, PARTIAL, BF, X8>;
    static snerf::DeviceOnce configured;   // per device: the attribute belongs to (kernel, device)
    const int attr = snerf::raise_dynamic_lds(configured, reinterpret_cast<const void*>(kernel), wgrad16_ring(X8, PARTIAL) * (PARTIAL ? 10 : 16) * kPairBytes, "mlp_backward");
    if (attr != SNERF_OK) return attr;
    hipLaunchKernelGGL(kernel, dim3((unsigned)table.wg_start[table.count]), dim3(256), lds_bytes, stream, table,
                       reinterpret_cast<const unsigned short*>(grads), reinterpret_cast<const unsigned short*>(acts), partial,
                       partial);
    return snerf::check_launch("mlp_backward(wgrad16)");
}

extern "C" size_t snerf_mlp_backward_workspace_floats(const snerf_mlp_desc* desc, long long num_rays, int num_samples) {
    snerf::MlpPlan plan;
    snerf::GenericPlan layered;
    const int plan_status = snerf::build_plan(desc, &plan);
    if (num_rays < 0 || num_samples < 1) return 0;
    if (plan_status == SNERF_E_UNSUPPORTED && snerf::generic_takes(desc, &layered))
        return snerf::generic_backward_workspace_floats(layered, num_rays * num_samples);
    if (plan_status != SNERF_OK) return 0;
    // every plan snerf_mlp_backward can build for this shape: fp32 / f16x3 tiles, the 16-bit tiles, and the 16-bit tiles with
    // fp8 saved activations (its per-job chunk distribution differs: block costs of 0.8, spare redistribution -- ADVICE r4)
    const long long total = num_rays * num_samples;
    return (size_t)std::max({plan_workspace(plan, total, false).total_floats, plan_workspace(plan, total, true).total_floats,
                             plan_workspace(plan, total, true, true).total_floats});
}

extern "C" int snerf_mlp_backward(const snerf_mlp_desc* desc, const float* packed, const float* saved_acts,
                                  const float* sigma, const float* rgb, const float* d_sigma, const float* d_rgb,
                                  long long num_rays, int num_samples, float* workspace, float* const* param_grads,
                                  int num_params, int precision, int accumulate, snerf_stream_t stream) {
    snerf::MlpPlan plan;
    const int st = snerf::build_plan(desc, &plan);
    snerf::GenericPlan layered;
    if (st == SNERF_E_UNSUPPORTED && snerf::generic_takes(desc, &layered)) {     // a shape of the layered path (mlp_generic.hip)
        SNERF_REQUIRE(packed && saved_acts && sigma && rgb && d_sigma && d_rgb && workspace && param_grads, "mlp_backward: NULL pointer");
        SNERF_REQUIRE(num_params == layered.num_params, "mlp_backward: expected %d gradient tensors, got %d", layered.num_params, num_params);
        for (int i = 0; i < num_params; ++i) SNERF_REQUIRE(param_grads[i], "mlp_backward: gradient tensor %d is NULL", i);
        SNERF_REQUIRE(num_rays >= 1 && num_samples >= 1, "mlp_backward: bad sizes n=%lld S=%d", num_rays, num_samples);
        return snerf::generic_backward(layered, packed, saved_acts, sigma, rgb, d_sigma, d_rgb, num_rays * num_samples, workspace,
                                       param_grads, precision, accumulate, (hipStream_t)stream);
    }
    if (st != SNERF_OK) return st;
    SNERF_REQUIRE(packed && saved_acts && sigma && rgb && d_sigma && d_rgb && workspace && param_grads,
                  "mlp_backward: NULL pointer");
    SNERF_REQUIRE(num_params == plan.num_params, "mlp_backward: expected %d gradient tensors, got %d", plan.num_params,
                  num_params);
    for (int i = 0; i < num_params; ++i) SNERF_REQUIRE(param_grads[i], "mlp_backward: gradient tensor %d is NULL", i);
    SNERF_REQUIRE(num_rays >= 1 && num_samples >= 1, "mlp_backward: bad sizes n=%lld S=%d", num_rays, num_samples);
    if (precision != SNERF_PRECISION_FP32 && precision != SNERF_PRECISION_F16X3 && precision != SNERF_PRECISION_F16 &&
        precision != SNERF_PRECISION_BF16 && precision != SNERF_PRECISION_F16S8 && precision != SNERF_PRECISION_BF16S8)
        return snerf::fail(SNERF_E_UNSUPPORTED, "mlp_backward: precision %d not built", precision);
    const bool bf16 = precision == SNERF_PRECISION_BF16 || precision == SNERF_PRECISION_BF16S8;
    const bool s8 = precision == SNERF_PRECISION_F16S8 || precision == SNERF_PRECISION_BF16S8;   // trunk activations saved as fp8
    const bool f16 = precision == SNERF_PRECISION_F16 || bf16 || s8;   // the 16-bit tile layouts; saved_acts must come from forward_train at the same precision
    if (precision != SNERF_PRECISION_FP32 && !bf16) {   // the forward that saved these activations may have left the fp16 range
        const int range = snerf::report_range("mlp_backward");
        if (range != SNERF_OK) return range;
    }
    {   // the backward chain reads the transposed weight stream of the training layout (snerf_common.h)
        const int formats = snerf::packed_formats_require(packed, snerf::packed_formats_needed(precision, true), "mlp_backward");
        if (formats != SNERF_OK) return formats;
    }
    const long long total = num_rays * num_samples;
    if ((total + 127) / 128 > 0x7fffffffLL) return snerf::fail(SNERF_E_UNSUPPORTED, "mlp_backward: too many samples");
    hipStream_t s = (hipStream_t)stream;
    const Workspace ws = plan_workspace(plan, total, f16, s8);
    // every limit is checked BEFORE anything is enqueued: the chain kernel publishes one maximum per 32-row dY region
    // into a kRegionWords-word slot, and the reduction's job table holds kMaxJobs entries
    if ((int)ws.jobs.size() > kMaxJobs) return snerf::fail(SNERF_E_UNSUPPORTED, "mlp_backward: too many weight-gradient jobs");
    if (plan.grad_head() / 32 + 1 > kRegionWords)
        return snerf::fail(SNERF_E_UNSUPPORTED, "mlp_backward: %d gradient regions exceed the region-maximum table (%d)",
                           plan.grad_head() / 32 + 1, kRegionWords);
    float* grads = workspace;
    float* partial = workspace + ws.grads_floats;
    snerf::ProfileScope timed(SNERF_PROFILE_MLP_BACKWARD, s, total);

    ChainArgs a;
    a.packed = packed; a.acts = saved_acts; a.sigma = sigma; a.rgb = rgb; a.d_sigma = d_sigma; a.d_rgb = d_rgb;
    a.grads = grads; a.total = total; a.depth = plan.depth; a.width = plan.width;
    a.dgrad_offset = plan.dgrad_offset; a.pts_out_w = plan.pts_out_w(); a.views_out_w = plan.views_out_w();
    a.act_rows = plan.act_rows(); a.act_h1 = plan.act_h(1); a.act_hv = plan.act_hv(); a.act_mask = plan.act_mask();
    a.grad_rows = plan.grad_rows(); a.grad_feature = plan.grad_feature(); a.grad_yv = plan.grad_yv();
    a.grad_head = plan.grad_head();
    if (f16) { a.act_rows = plan.act16_rows(); a.grad_rows = plan.grad16_rows(); }
    // partial[0, 64): zero page for padded staging rows; [192, 448): 1 KiB zero page; [448, 448 + 64 x 128): the region
    // maxima |dY| of the f16 modes, in 64 copies (kRegionSlots) of kRegionWords words, indexed by workgroup
    // (a kernel, not hipMemsetAsync: captured in a HIP graph, the memset node did not clear the table on later replays --
    // from the third replay on the region maxima only ever grew, and once the true gradients had shrunk by a few powers of
    // two the fp16 weight-gradient operands lost that many bits: 30 % errors in a graphed fp16 training run, exact again with
    // this kernel; found by tools/probes/graph_vs_eager.py, r02.  The library uses no memset nodes any more.)
    hipLaunchKernelGGL(clear_words_kernel, dim3(8), dim3(256), 0, s, reinterpret_cast<unsigned*>(partial),
                       kRegionTableFloat0 + kRegionSlots * kRegionWords);
    {
        const int zrc = snerf::check_launch("mlp_backward(clear)");
        if (zrc != SNERF_OK) return zrc;
    }
    int rc;
    const int key = precision != SNERF_PRECISION_FP32 ? -1 : plan.wt * 10 + plan.vt;
    a.dy_max = nullptr;
    if (precision != SNERF_PRECISION_FP32) {
        a.dy_max = reinterpret_cast<unsigned*>(partial) + kRegionTableFloat0;
        rc = bf16 ? snerf::mlp_backward_chain_bf16(plan, a, s) : snerf::mlp_backward_chain_f16x3(plan, a, f16 ? 1 : 3, s);
        if (rc != SNERF_OK) return rc;
    }
    switch (key) {
        case -1: rc = SNERF_OK; break;
        case 84: rc = launch_chain<8, 4, true>(a, s); break;
        case 80: rc = launch_chain<8, 4, false>(a, s); break;
        case 42: rc = launch_chain<4, 2, true>(a, s); break;
        case 40: rc = launch_chain<4, 2, false>(a, s); break;
        default: return snerf::fail(SNERF_E_UNSUPPORTED, "mlp_backward: width %d / views width %d not built", plan.width,
                                    plan.views_width);
    }
    if (rc != SNERF_OK) return rc;
    // f16x3: the large products (8 in-tiles wide) run on the fp16 pipe with one power-of-two scale per dY region; the
    // small head/encoding products are DMA-bound and stay on the fp32 kernel
    std::vector<WgradJob> jobs = ws.jobs;
    for (WgradJob& job : jobs) {
        int no, ni;
        wave_tile(job, &no, &ni);
        job.half = ((f16 && !bf16) || (precision == SNERF_PRECISION_F16X3 && ni == 8)) ? 1 : 0;   // (bf16: no region scale)
    }
    // (the zero page and the gradient-max word were cleared before the chain kernel)
    JobTable table;  // all jobs, for the reduction
    table.count = (int)jobs.size();
    table.wg_start[0] = 0;
    long long max_work = 0;
    for (int j = 0; j < table.count; ++j) {
        table.jobs[j] = jobs[j];
        table.wg_start[j + 1] = table.wg_start[j] + jobs[j].chunks;
        const long long work = (long long)jobs[j].out_rows * jobs[j].in_rows + jobs[j].out_rows;
        if (work > max_work) max_work = work;
    }
    static const int classes[][2] = {{2, 8}, {2, 2}, {2, 1}, {1, 9}, {1, 8}, {1, 4}, {1, 2}, {1, 1}};   // ({1, 9}: 16-bit modes only)
    // 16-bit mode: every small job (fewer than 32 tile products: heads, encodings) rides in ONE launch of the <2, 2> instance
    // with partial register tiles (wgrad16_kernel PARTIAL) instead of one launch per register-tile class
#ifdef SNERF_PROBE_NO_SMALL_FOLD     // A/B probe builds: one launch per register-tile class, as before round 3
    auto small16 = [&](const WgradJob&) { return false; };
#else
    // ... provided the four waves of that instance COVER the job: wave (wo, wi) of a wgrid_o x wgrid_i grid owns out tiles
    // [2 wo, 2 wo + 2) x in tiles [2 wi, 2 wi + 2) (wgrad16_kernel: wgrid_i = max(in_tiles / 2, 1)); a shape they do not
    // cover (3 in-tiles, 16 x 1, ...) goes to its own register-tile class below instead of silently losing tiles
    auto small16 = [&](const WgradJob& job) {
        if (!f16 || job.out_tiles * job.in_tiles >= 32) return false;
        const int wgrid_i = std::max(job.in_tiles / 2, 1);
        if (wgrid_i > 4 || 4 % wgrid_i != 0) return false;
        const int wgrid_o = 4 / wgrid_i;
        return wgrid_i * 2 >= job.in_tiles && wgrid_o * 2 >= job.out_tiles;
    };
#endif
    if (f16) {
        JobTable sub;
        sub.count = 0;
        sub.wg_start[0] = 0;
        for (const WgradJob& job : jobs) {
            if (!small16(job) || !job.launched) continue;
            sub.jobs[sub.count] = job;
            sub.wg_start[sub.count + 1] = sub.wg_start[sub.count] + job.chunks;
            ++sub.count;
        }
        if (sub.count > 0) {
            rc = bf16 ? launch_wgrad16<2, 2, true, true>(sub, grads, saved_acts, partial, s)
                      : launch_wgrad16<2, 2, true>(sub, grads, saved_acts, partial, s);
            if (rc != SNERF_OK) return rc;
        }
    }
    for (const auto& cls : classes) {
        JobTable sub;
        sub.count = 0;
        sub.wg_start[0] = 0;
        for (const WgradJob& job : jobs) {
            int no, ni;
            wave_tile(job, &no, &ni);
            if (no != cls[0] || ni != cls[1] || small16(job) || !job.launched) continue;
            sub.jobs[sub.count] = job;
            sub.wg_start[sub.count + 1] = sub.wg_start[sub.count] + job.chunks;
            ++sub.count;
        }
        if (sub.count == 0) continue;
        const bool x3 = precision == SNERF_PRECISION_F16X3;
        if (bf16) {
            if (cls[0] == 2 && cls[1] == 8) rc = s8 ? launch_wgrad16<2, 8, false, true, true>(sub, grads, saved_acts, partial, s)
                                                    : launch_wgrad16<2, 8, false, true>(sub, grads, saved_acts, partial, s);
            else if (cls[0] == 2 && cls[1] == 2) rc = launch_wgrad16<2, 2, false, true>(sub, grads, saved_acts, partial, s);
            else if (cls[0] == 2 && cls[1] == 1) rc = launch_wgrad16<2, 1, false, true>(sub, grads, saved_acts, partial, s);
            else if (cls[0] == 1 && cls[1] == 9) rc = launch_wgrad16<1, 9, false, true>(sub, grads, saved_acts, partial, s);
            else if (cls[0] == 1 && cls[1] == 8) rc = launch_wgrad16<1, 8, false, true>(sub, grads, saved_acts, partial, s);
            else if (cls[0] == 1 && cls[1] == 4) rc = launch_wgrad16<1, 4, false, true>(sub, grads, saved_acts, partial, s);
            else if (cls[0] == 1 && cls[1] == 2) rc = launch_wgrad16<1, 2, false, true>(sub, grads, saved_acts, partial, s);
            else rc = launch_wgrad16<1, 1, false, true>(sub, grads, saved_acts, partial, s);
        }
        else if (f16) {
            if (cls[0] == 2 && cls[1] == 8) rc = s8 ? launch_wgrad16<2, 8, false, false, true>(sub, grads, saved_acts, partial, s)
                                                    : launch_wgrad16<2, 8>(sub, grads, saved_acts, partial, s);
            else if (cls[0] == 2 && cls[1] == 2) rc = launch_wgrad16<2, 2>(sub, grads, saved_acts, partial, s);
            else if (cls[0] == 2 && cls[1] == 1) rc = launch_wgrad16<2, 1>(sub, grads, saved_acts, partial, s);
            else if (cls[0] == 1 && cls[1] == 9) rc = launch_wgrad16<1, 9>(sub, grads, saved_acts, partial, s);
            else if (cls[0] == 1 && cls[1] == 8) rc = launch_wgrad16<1, 8>(sub, grads, saved_acts, partial, s);
            else if (cls[0] == 1 && cls[1] == 4) rc = launch_wgrad16<1, 4>(sub, grads, saved_acts, partial, s);
            else if (cls[0] == 1 && cls[1] == 2) rc = launch_wgrad16<1, 2>(sub, grads, saved_acts, partial, s);
            else rc = launch_wgrad16<1, 1>(sub, grads, saved_acts, partial, s);
        }
        else if (cls[0] == 2 && cls[1] == 8 && x3) rc = launch_wgrad<2, 8, true>(sub, grads, saved_acts, partial, partial, s);
        else if (cls[0] == 1 && cls[1] == 8 && x3) rc = launch_wgrad<1, 8, true>(sub, grads, saved_acts, partial, partial, s);
        else if (cls[0] == 2 && cls[1] == 8) rc = launch_wgrad<2, 8>(sub, grads, saved_acts, partial, partial, s);
        else if (cls[0] == 2 && cls[1] == 2) rc = launch_wgrad<2, 2>(sub, grads, saved_acts, partial, partial, s);
        else if (cls[0] == 2 && cls[1] == 1) rc = launch_wgrad<2, 1>(sub, grads, saved_acts, partial, partial, s);
        else if (cls[0] == 1 && cls[1] == 8) rc = launch_wgrad<1, 8>(sub, grads, saved_acts, partial, partial, s);
        else if (cls[0] == 1 && cls[1] == 4) rc = launch_wgrad<1, 4>(sub, grads, saved_acts, partial, partial, s);
        else if (cls[0] == 1 && cls[1] == 2) rc = launch_wgrad<1, 2>(sub, grads, saved_acts, partial, partial, s);
        else rc = launch_wgrad<1, 1>(sub, grads, saved_acts, partial, partial, s);
        if (rc != SNERF_OK) return rc;
    }
    for (const WgradJob& job : ws.jobs) {  // every job must have found its class
        int no, ni;
        wave_tile(job, &no, &ni);
        bool ok = false;
        for (const auto& cls : classes) ok = ok || (no == cls[0] && ni == cls[1]);
        if (!ok) return snerf::fail(SNERF_E_UNSUPPORTED, "mlp_backward: no weight-gradient kernel for a %dx%d-tile product",
                                    job.out_tiles, job.in_tiles);
    }
    if (plan.views_out_rows == 4 && !accumulate) {
        // predict_visibility: the visibility row of views_output_linear receives no gradient (no shipped loss reads the
        // visibility outputs; they are returned without a gradient path) -- an overwritten tensor holds zeros there
        const int d = plan.depth;
        hipLaunchKernelGGL(clear_words_kernel, dim3(1), dim3(256), 0, s,
                           reinterpret_cast<unsigned*>(param_grads[2 * d + 6] + 3LL * plan.views_width), plan.views_width);
        hipLaunchKernelGGL(clear_words_kernel, dim3(1), dim3(64), 0, s, reinterpret_cast<unsigned*>(param_grads[2 * d + 7] + 3), 1);
        const int zrc = snerf::check_launch("mlp_backward(clear visibility row)");
        if (zrc != SNERF_OK) return zrc;
    }
    GradPointers ptrs;
    for (int i = 0; i < num_params; ++i) ptrs.p[i] = param_grads[i];
    hipLaunchKernelGGL(reduce_kernel, dim3((unsigned)std::min<long long>((max_work + 63) / 64, 1024), table.count), dim3(256), 0, s, table, partial,
                       ptrs, accumulate ? 1 : 0);
    rc = snerf::check_launch("mlp_backward(reduce)");
    if (rc != SNERF_OK) return rc;
    return SNERF_OK;
}
