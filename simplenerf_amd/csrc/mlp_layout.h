// Packed weight-stream layout shared by the packer (mlp_pack.hip) and the fused MLP kernel (mlp_forward.hip).
//
// The fused kernel computes every Linear layer TRANSPOSED on the fp32 matrix cores:
//     Y^T[out, sample] = W[out, in] . X^T[in, sample]
// with v_mfma_f32_32x32x2_f32: A = a 32(out) x 2(k) slice of W, B = a 2(k) x 32(sample) slice of the
// activations, D = a 32(out) x 32(sample) accumulator tile.  D holds, in lane (j = lane&31, h = lane>>5),
// register r, the value for sample j and out-feature (r&3) + 8*(r>>2) + 4*h of the tile -- which is exactly the
// shape of a B operand (one k value per lane half) of the NEXT layer.  So activations never leave registers:
// accumulator register ks = 16*t + r of a lane becomes the B operand of k-step ks of the next GEMM, and that
// k-step contracts the feature pair
//     feat(ks, h) = 8*(ks>>2) + 4*h + (ks&3),  h = 0, 1.
// Weights are therefore stored k-step-major so that one ds_read_b128 per lane yields the A operands of four
// consecutive k-steps:
//     packed[g][u][lane][q] = W[32*u + (lane&31)][column(feat(4g+q, lane>>5))]      (zero outside W)
// g = k-step group, u = 32-row output tile.  A "slab" is 4 groups = 16 k-steps of all U tiles = U*4 KiB, the
// unit that is staged through LDS.
//
// Positional-encoding inputs are generated in registers by the lanes themselves, in an order chosen so that
// each lane half evaluates whole (sin, cos) pairs:
//   pair c = 3*k + dim (frequency 2^k of coordinate dim) lives in half h = c&1, registers n = 2*(c>>1) (sin) and
//   n + 1 (cos); the raw coordinates use the two registers after the pairs: (x | z) then (y | zero).
// Register n is k-step n of the PE segment; pe_feature() maps (n, h) to the reference's encoding index
// [x, sin(2^0 x), cos(2^0 x), ...] (src/models/SimpleNeRF01.py:533-557), -1 for padding.
#pragma once

namespace snerf {

constexpr int kPointsPairs = 30;   // 10 frequencies x 3 coordinates
constexpr int kPointsKSteps = 32;  // 30 pair registers + 2 coordinate registers
constexpr int kViewsPairs = 12;    // 4 frequencies x 3 coordinates
constexpr int kViewsKSteps = 16;   // 12 pair registers + 2 coordinate registers + 2 zero registers
constexpr int kMaxPointsDegree = 10;
constexpr int kMaxViewsDegree = 4;
constexpr int kSlabKSteps = 16;

// Reference encoding index of PE register n in lane half h; pairs = 3*max_degree of that encoder.
__host__ __device__ inline int pe_feature(int n, int h, int pairs, int degree) {
    if (n < pairs) {
        const int c = (n >> 1) * 2 + h;  // pair index
        if (c >= pairs) return -1;
        const int k = c / 3, dim = c % 3;
        if (k >= degree) return -1;
        return 3 + 6 * k + 3 * (n & 1) + dim;
    }
    const int slot = (n - pairs) * 2 + h;  // 0:x 1:z 2:y 3:pad  -> (x|z), (y|pad)
    if (slot == 0) return 0;
    if (slot == 1) return 2;
    if (slot == 2) return 1;
    return -1;
}

// Input feature (row of X^T) contracted by k-step ks in lane half h when X^T is a previous accumulator.
__host__ __device__ inline int acc_feature(int ks, int h) { return 8 * (ks >> 2) + 4 * h + (ks & 3); }

enum SegmentKind { SEG_ACC = 0, SEG_POINTS_PE = 1, SEG_VIEWS_PE = 2 };

// One K-segment of one Linear layer in the packed stream.
struct Segment {
    int param;       // index of the weight tensor in the C-ABI `params` array
    int ld;          // its row length (in_features)
    int out_dim;     // its number of rows
    int tiles;       // U = ceil(out_dim / 32)
    int ksteps;      // multiple of 16
    int kind;        // SegmentKind
    int col_offset;  // column of W that feature 0 of this segment maps to
    int feat_lo;     // SEG_POINTS_PE: only encoding indices in [feat_lo, feat_hi) are wired (others zero)
    int feat_hi;
    int degree;      // PE degree for SEG_*_PE
    long long dst;   // offset (floats) into the packed stream
    int transposed;  // 0: forward operand (rows = out features).  1: dgrad operand W^T -- rows of the tile are IN
                     //    features (col_offset + 32u + i) and the k-steps run over OUT features in accumulator order
};

// Element of W for (transposed segment, tile u, row i, k-step, lane half): W[acc_feature(ks,h)][col_offset+32u+i].
__host__ __device__ inline long long transposed_index(const Segment& s, int u, int i, int ks, int h) {
    const int out = acc_feature(ks, h);
    const int in = 32 * u + i;
    if (out >= s.out_dim || in >= s.feat_hi) return -1;  // feat_hi = number of input features wired (e.g. 256)
    return (long long)out * s.ld + s.col_offset + in;
}

// Column of W for (segment, k-step, lane half), or -1 for a zero weight.
__host__ __device__ inline int segment_column(const Segment& s, int ks, int h) {
    if (s.kind == SEG_ACC) return s.col_offset + acc_feature(ks, h);
    if (s.kind == SEG_POINTS_PE) {
        const int e = pe_feature(ks, h, kPointsPairs, s.degree);
        if (e < s.feat_lo || e >= s.feat_hi) return -1;
        return s.col_offset + (e - s.feat_lo);
    }
    const int e = pe_feature(ks, h, kViewsPairs, s.degree);
    if (e < 0) return -1;
    return s.col_offset + e;
}

}  // namespace snerf
