// Weight packer: reference-layout parameters (nn.Linear weight (out,in) row-major + bias) -> the MFMA-operand-
// ordered stream described in mlp_layout.h.  A pure permutation with zero fill; HBM-bound and tiny (2.4 MB per
// 8x256 MLP), run once per parameter update.
#include "mlp_generic.h"
#include <algorithm>

#include "mlp_plan.h"

namespace {

typedef float f32x4_t __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) zero_fill_kernel(f32x4_t* __restrict__ dst, long long count) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    const f32x4_t zero = {0.0f, 0.0f, 0.0f, 0.0f};
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) dst[i] = zero;
}

// Launch batching: the stream of one MLP has ~40 segments and ~36 f16x3 stages of a few KB each; one launch per piece
// made a re-pack (needed after every optimiser step) ~95 tiny GPU operations.  Pieces are grouped into kernarg-sized
// tables instead (blockIdx.y = piece), ~10 launches per re-pack.
constexpr int kSegmentsPerLaunch = 16;
struct SegmentTable {
    const float* w[kSegmentsPerLaunch];
    snerf::Segment seg[kSegmentsPerLaunch];
};
// (20 since round 5, 8 before: the 19 stages a 16-bit training re-pack writes -- forward + transposed stream of an 8 x 256 MLP -- go in ONE
// launch instead of three; a re-pack is launch-latency-bound, ~4.6 us per dependent launch, and a training iteration re-packs
// four MLPs.  The table travels as a kernel argument: it must stay under the 4-KiB limit.)
constexpr int kStagesPerLaunch = 20;
struct StageTable {
    const float* w[kStagesPerLaunch][3];
    snerf::MlpPlan::HalfStage stage[kStagesPerLaunch];
    int m16[kStagesPerLaunch];   // 1: fragment layout of the 16x16x32 MFMA (mlp_forward_m16.hip)
    int* weight_range;           // word of the packed buffer: kRangeWeight is OR-ed in when a weight does not fit fp16
};
static_assert(sizeof(StageTable) <= 3968, "StageTable is passed by value: kernel arguments are limited to 4 KiB");
constexpr int kCopiesPerLaunch = 16;
struct CopyTable {
    const float* src[kCopiesPerLaunch];
    long long dst[kCopiesPerLaunch];
    int count[kCopiesPerLaunch];
};

__global__ void __launch_bounds__(256) copy_rows_kernel(CopyTable t, float* __restrict__ packed) {
    const float* __restrict__ src = t.src[blockIdx.y];
    float* __restrict__ dst = packed + t.dst[blockIdx.y];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < t.count[blockIdx.y]; i += gridDim.x * blockDim.x) dst[i] = src[i];
}

__global__ void __launch_bounds__(256) pack_segment_kernel(SegmentTable table, float* __restrict__ packed) {
    const snerf::Segment& s = table.seg[blockIdx.y];
    const float* __restrict__ w = table.w[blockIdx.y];
    const long long total = (long long)s.ksteps * s.tiles * 64;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        // idx = ((g * tiles + u) * 64 + lane) * 4 + q
        const int q = (int)(idx & 3);
        const int lane = (int)((idx >> 2) & 63);
        const long long gu = idx >> 8;
        const int u = (int)(gu % s.tiles);
        const int g = (int)(gu / s.tiles);
        float v = 0.0f;
        if (s.transposed) {
            const long long src = snerf::transposed_index(s, u, lane & 31, 4 * g + q, lane >> 5);
            if (src >= 0) v = w[src];
        } else {
            const int row = 32 * u + (lane & 31);
            const int col = snerf::segment_column(s, 4 * g + q, lane >> 5);
            if (row < s.out_dim && col >= 0 && col < s.ld) v = w[(long long)row * s.ld + col];
        }
        packed[s.dst + idx] = v;
    }
}

// f16x3 stream: for every (stage, out tile u): the hi fragments [64 lanes][8 fp16] of all k-steps, then the lo fragments, where
// hi = fp16(w), lo = fp16(w - hi) (lo may be subnormal: the MFMA honours fp16 subnormals, measured).  Element j of lane
// (i, h) multiplies input feature  16*ks + 8*(j>>2) + 4*h + (j&3)  of an accumulator-sourced segment (the order in which
// a 32x32 accumulator tile turns into the next B operand), or encoding register n = 8*ks + j of the lane half.
// unit layout (fp16 elements): [ks][hi: 64 lanes x 8] for every k-step of the unit, then [ks][lo: 64 lanes x 8]
__device__ __forceinline__ void out_store(_Float16* out, long long unit, int unit_ks, int ks, int lane, int j, _Float16 hi,
                                          _Float16 lo) {
    const long long base = unit * unit_ks * 1024 + ks * 512 + lane * 8 + j;
    out[base] = hi;
    out[base + unit_ks * 512] = lo;
}
// compact bf16 unit (HalfStage::bf16): [ks][64 lanes x 8 bf16], one KiB per k-step, nothing else
__device__ __forceinline__ void out_store_bf16(_Float16* out, long long unit, int unit_ks, int ks, int lane, int j, float v) {
    out[unit * unit_ks * 512 + ks * 512 + lane * 8 + j] = __builtin_bit_cast(_Float16, (__bf16)v);
}

__global__ void __launch_bounds__(256) pack_half_stage_kernel(StageTable table, float* __restrict__ packed) {
    const snerf::MlpPlan::HalfStage& st = table.stage[blockIdx.y];
    const float* __restrict__ w0 = table.w[blockIdx.y][0];
    const float* __restrict__ w1 = table.w[blockIdx.y][1];
    const float* __restrict__ w2 = table.w[blockIdx.y][2];
    const int unit_ks = st.unit_floats / 512;
    const long long total = (long long)st.tiles * unit_ks * 512;  // one thread per (u, ks, lane, j): 512 elements per k-step
    const long long stride = (long long)gridDim.x * blockDim.x;
    _Float16* dst16 = reinterpret_cast<_Float16*>(packed + st.dst);
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        int j = (int)(idx & 7);
        const int lane = (int)((idx >> 3) & 63);
        const long long uk = idx >> 9;
        int ks = (int)(uk % unit_ks);
        const int u = (int)(uk / unit_ks);
        const int ks_unit = ks;
        int h = lane >> 5, row = 32 * u + (lane & 31);
        const int slot = j;
        if (table.m16[blockIdx.y]) {
            // Fragment f = 2c + r: rows 16r + (lane & 15) of the out tile, k-block c (32 inputs).  Lane group g = lane >> 4
            // holds in slot t the input at position p = (t < 4 ? 4g + t : 16 + 4g + t - 4) of the block -- the order in which
            // the four 16x16 accumulator tiles (row half t / 4) of the previous layer's out tile become this operand.
            // Position p is k-step 2c + p / 16, lane half (p % 16) / 8, element p % 8 of the 32x32x16 layout below.
            const int c = ks >> 1, r = ks & 1, g = lane >> 4;
            const int p = slot < 4 ? 4 * g + slot : 16 + 4 * g + (slot - 4);
            row = 32 * u + 16 * r + (lane & 15);
            ks = 2 * c + (p >> 4);
            h = (p & 15) >> 3;
            j = p & 7;
        }
        int si = 0;
        while (si + 1 < st.nseg && ks >= st.seg[si].ksteps) { ks -= st.seg[si].ksteps; ++si; }
        const snerf::MlpPlan::HalfSegment& sg = st.seg[si];
        const float* w = si == 0 ? w0 : (si == 1 ? w1 : w2);
        int col = -1;
        if (sg.transposed) {
            // dgrad operand: tile row = IN feature, k = OUT feature in accumulator order
            const int out = 16 * ks + 8 * (j >> 2) + 4 * h + (j & 3);
            float tv = 0.0f;
            if (out < sg.out_dim && row < sg.feat_hi) tv = w[(long long)out * sg.ld + sg.col_offset + row];
            if (st.bf16) { out_store_bf16(dst16, u, unit_ks, ks_unit, lane, slot, tv); continue; }
            const _Float16 thi = (_Float16)tv;
            out_store(dst16, u, unit_ks, ks_unit, lane, slot, thi, (_Float16)(tv - (float)thi));
            if (!(fabsf(tv) <= 65504.0f)) __hip_atomic_fetch_or(table.weight_range, snerf::kRangeWeight, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            continue;
        }
        if (sg.kind == snerf::SEG_ACC) {
            col = table.m16[blockIdx.y] ? sg.col_offset + 16 * ks + 8 * h + j      // (position p in natural feature order)
                                        : sg.col_offset + 16 * ks + 8 * (j >> 2) + 4 * h + (j & 3);
        } else if (sg.kind == snerf::SEG_POINTS_PE) {
            const int e = snerf::pe_feature(8 * ks + j, h, snerf::kPointsPairs, sg.degree);
            if (e >= sg.feat_lo && e < sg.feat_hi) col = sg.col_offset + (e - sg.feat_lo);
        } else {
            const int e = snerf::pe_feature(8 * ks + j, h, snerf::kViewsPairs, sg.degree);
            if (e >= 0) col = sg.col_offset + e;
        }
        float v = 0.0f;
        if (row < sg.out_dim && col >= 0 && col < sg.ld) v = w[(long long)row * sg.ld + col];
        if (st.bf16) { out_store_bf16(dst16, u, unit_ks, ks_unit, lane, slot, v); continue; }     // (bf16 has fp32's range)
        const _Float16 hi = (_Float16)v;
        const _Float16 lo = (_Float16)(v - (float)hi);
        out_store(dst16, u, unit_ks, ks_unit, lane, slot, hi, lo);
        // (NaN included.  Recorded in the packed buffer itself and reported only by an fp16-mode kernel that consumes THIS
        // buffer: an fp32-mode model may hold such weights legitimately, and must not trip another model's fp16-mode call)
        if (!(fabsf(v) <= 65504.0f)) __hip_atomic_fetch_or(table.weight_range, snerf::kRangeWeight, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace

// (a shape the fused kernels are not built for goes to the layered path, mlp_generic.hip: its "packed" buffer is a plain
// concatenation of the parameters)
extern "C" int snerf_mlp_num_params(const snerf_mlp_desc* desc) {
    snerf::MlpPlan plan;
    snerf::GenericPlan layered;
    const int st = snerf::build_plan(desc, &plan);
    if (st == SNERF_E_UNSUPPORTED && snerf::generic_takes(desc, &layered)) return layered.num_params;
    if (st != SNERF_OK) return 0;
    return plan.num_params;
}

extern "C" size_t snerf_mlp_packed_floats(const snerf_mlp_desc* desc) {
    snerf::MlpPlan plan;
    snerf::GenericPlan layered;
    const int st = snerf::build_plan(desc, &plan);
    if (st == SNERF_E_UNSUPPORTED && snerf::generic_takes(desc, &layered)) return (size_t)layered.packed_floats;
    if (st != SNERF_OK) return 0;
    return (size_t)plan.total_floats;
}

namespace {
// which 16-bit copies of the weights a pack writes besides the fp32 segments, biases and heads (everything else is zeroed):
// snerf::kPackF16 ... kPackAll (snerf_common.h); the mask of every packed buffer is recorded for the consumers' check
using snerf::kPackF16; using snerf::kPackF16Eval; using snerf::kPackBf16; using snerf::kPackBf16Eval; using snerf::kPackFp32; using snerf::kPackAll;

int pack_impl(const snerf_mlp_desc* desc, const float* const* params, int num_params, float* packed, snerf_stream_t stream,
              unsigned formats) {
    snerf::MlpPlan plan;
    const int st = snerf::build_plan(desc, &plan);
    snerf::GenericPlan layered;
    if (st == SNERF_E_UNSUPPORTED && snerf::generic_takes(desc, &layered)) {
        SNERF_REQUIRE(params && packed, "mlp_pack: NULL pointer");
        SNERF_REQUIRE(num_params == layered.num_params, "mlp_pack: expected %d parameter tensors, got %d", layered.num_params, num_params);
        for (int i = 0; i < num_params; ++i) SNERF_REQUIRE(params[i], "mlp_pack: parameter %d is NULL", i);
        return snerf::generic_pack(layered, params, packed, (hipStream_t)stream);
    }
    if (st != SNERF_OK) return st;
    SNERF_REQUIRE(params && packed, "mlp_pack: NULL pointer");
    SNERF_REQUIRE(num_params == plan.num_params, "mlp_pack: expected %d parameter tensors, got %d", plan.num_params,
                  num_params);
    for (int i = 0; i < num_params; ++i) SNERF_REQUIRE(params[i], "mlp_pack: parameter %d is NULL", i);
    hipStream_t s = (hipStream_t)stream;
    // zero fill by a kernel rather than hipMemsetAsync: the re-pack is part of the captured training graph, and memset nodes
    // proved unreliable on replay (see snerf_mlp_backward); total_floats is a multiple of 64
    hipLaunchKernelGGL(zero_fill_kernel, dim3(512), dim3(256), 0, s, reinterpret_cast<f32x4_t*>(packed), plan.total_floats / 4);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return snerf::fail(SNERF_E_HIP, "mlp_pack: zero fill: %s", hipGetErrorString(e));
    {
        SegmentTable table;
        int n = 0;
        long long most = 0;
        auto flush = [&]() {
            if (n == 0) return;
            hipLaunchKernelGGL(pack_segment_kernel, dim3(snerf::stride_grid(most, 256), n), dim3(256), 0, s, table, packed);
            n = 0;
            most = 0;
        };
        for (const std::vector<snerf::Segment>* list : {&plan.segments, &plan.dgrad_segments}) {
            if (!(formats & kPackFp32)) continue;          // (the fp32 K-segment slabs: read by the fp32 kernels only)
            for (const snerf::Segment& seg : *list) {
                table.w[n] = params[seg.param];
                table.seg[n] = seg;
                most = std::max(most, (long long)seg.ksteps * seg.tiles * 64);
                if (++n == kSegmentsPerLaunch) flush();
            }
        }
        flush();
    }
    {
        StageTable table;
        table.weight_range = reinterpret_cast<int*>(packed + plan.weight_range_word);     // zeroed by the fill above
        int n = 0;
        long long most = 0;
        auto flush = [&]() {
            if (n == 0) return;
            hipLaunchKernelGGL(pack_half_stage_kernel, dim3(snerf::stride_grid(most, 256), n), dim3(256), 0, s, table, packed);
            n = 0;
            most = 0;
        };
        for (const std::vector<snerf::MlpPlan::HalfStage>* list : {&plan.half_stages, &plan.half_dgrad_stages, &plan.m16_stages,
                                                                   &plan.bf_stages, &plan.bf_dgrad_stages, &plan.bf_m16_stages}) {
            const unsigned bit = (list == &plan.half_stages || list == &plan.half_dgrad_stages) ? kPackF16
                                 : list == &plan.m16_stages ? kPackF16Eval
                                 : list == &plan.bf_m16_stages ? kPackBf16Eval : kPackBf16;
            if (!(formats & bit)) continue;
            for (const snerf::MlpPlan::HalfStage& st : *list) {
                for (int k = 0; k < 3; ++k) table.w[n][k] = params[st.seg[k < st.nseg ? k : 0].param];
                table.stage[n] = st;
                table.m16[n] = (list == &plan.m16_stages || list == &plan.bf_m16_stages) ? 1 : 0;
                most = std::max(most, (long long)st.tiles * (st.unit_floats / 512) * 512);
                if (++n == kStagesPerLaunch) flush();
            }
        }
        flush();
    }
    CopyTable copies;
    int ncopies = 0;
    auto copy = [&](long long dst, const float* src, long long n) {
        copies.src[ncopies] = src;
        copies.dst[ncopies] = dst;
        copies.count[ncopies] = (int)n;
        ++ncopies;
    };
    const int d = plan.depth;
    SNERF_REQUIRE(d + 6 <= kCopiesPerLaunch, "mlp_pack: depth %d exceeds the copy table", d);
    for (int l = 0; l < d; ++l) copy(plan.trunk_bias(l), params[2 * l + 1], plan.width);
    copy(plan.pts_out_w(), params[2 * d], (long long)plan.pts_out_rows * plan.width);
    copy(plan.pts_out_b(), params[2 * d + 1], plan.pts_out_rows);
    if (plan.view_dependent) {
        copy(plan.feature_bias(), params[2 * d + 3], plan.width);
        copy(plan.views_bias(), params[2 * d + 5], plan.views_width);
        copy(plan.views_out_w(), params[2 * d + 6], (long long)plan.views_out_rows * plan.views_width);
        copy(plan.views_out_b(), params[2 * d + 7], plan.views_out_rows);
    }
    hipLaunchKernelGGL(copy_rows_kernel, dim3(4, ncopies), dim3(256), 0, s, copies, packed);
    if (e != hipSuccess) return snerf::fail(SNERF_E_HIP, "mlp_pack: copy: %s", hipGetErrorString(e));
    const int launched = snerf::check_launch("mlp_pack");
    if (launched == SNERF_OK) snerf::packed_formats_record(packed, formats);
    return launched;
}
}  // namespace

extern "C" int snerf_mlp_pack(const snerf_mlp_desc* desc, const float* const* params, int num_params, float* packed,
                              snerf_stream_t stream) {
    return pack_impl(desc, params, num_params, packed, stream, kPackAll);
}

extern "C" int snerf_mlp_pack_for(const snerf_mlp_desc* desc, const float* const* params, int num_params, float* packed,
                                  int precision, int training, snerf_stream_t stream) {
    const unsigned formats = snerf::packed_formats_needed(precision, training != 0);
    if (!formats) return snerf::fail(SNERF_E_UNSUPPORTED, "mlp_pack_for: unknown precision %d", precision);
    return pack_impl(desc, params, num_params, packed, stream, formats);
}
