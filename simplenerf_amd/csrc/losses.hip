// L1/L2: the training losses evaluated on the device.
//
// The reference evaluates nine loss classes per sub-batch as a few hundred small torch kernels (boolean-mask
// compaction, elementwise ops, means, and -- for the three patch-consistency losses -- 75 fancy-index gathers each).
// Here the whole set is three launches: the patch-decision kernel (one wave per ray), one reduction over every term,
// one gradient pass over every term.  Work is a few thousand rays, so everything is latency-bound; the design goal is
// launch count and determinism (fixed-order reductions, no floating-point atomics), not bandwidth.
#include "snerf_common.h"
#include "wave.h"

#include "../../include/simplenerf_train.h"

namespace {

constexpr int kMaxTerms = SNERF_LOSS_MAX_TERMS;
constexpr int kMaxGroups = SNERF_LOSS_MAX_GROUPS;
constexpr int kBlock = 256;
constexpr int kMaxBlocks = 64;

struct TermTable {
    snerf_loss_term term[kMaxTerms];
    int num_terms;
    int num_groups;
};

struct Workspace {
    float sums[kMaxBlocks][kMaxTerms];
    unsigned counts[kMaxBlocks][kMaxTerms];
    unsigned done;   // completion counter: zero between launches
};

__device__ __forceinline__ unsigned wave_sum_u(unsigned v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ float squared_error(const snerf_loss_term& t, long long ray) {
    if (t.channels == 1) {
        const float e = t.pred[ray] - t.target[ray];
        return e * e;
    }
    float s = 0.0f;
    for (int c = 0; c < t.channels; ++c) {
        const float e = t.pred[ray * t.channels + c] - t.target[ray * t.channels + c];
        s += e * e;
    }
    return s;
}

// Pass 1: per-term (sum of squared error on numerator rays, count of denominator rays).  Thread-sequential over a
// grid-stride slice, wave shuffle, four waves through LDS, blocks through the workspace; the last block to finish
// folds the block partials in block order and writes values / scales.
__global__ void __launch_bounds__(kBlock) loss_forward_kernel(TermTable table, long long num_rays, float* values,
                                                              float* scales, Workspace* ws) {
    __shared__ float lds_sum[kBlock / 64][kMaxTerms];
    __shared__ unsigned lds_cnt[kBlock / 64][kMaxTerms];
    __shared__ unsigned is_last;
    float sum[kMaxTerms];
    unsigned cnt[kMaxTerms];
#pragma unroll
    for (int t = 0; t < kMaxTerms; ++t) {
        sum[t] = 0.0f;
        cnt[t] = 0;
    }
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long ray = (long long)blockIdx.x * kBlock + threadIdx.x; ray < num_rays; ray += stride) {
#pragma unroll
        for (int t = 0; t < kMaxTerms; ++t) {
            if (t < table.num_terms) {
                const snerf_loss_term& term = table.term[t];
                if (!term.denominator_mask || term.denominator_mask[ray]) cnt[t] += 1;
                if (!term.numerator_mask || term.numerator_mask[ray]) sum[t] += squared_error(term, ray);
            }
        }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int t = 0; t < kMaxTerms; ++t) {
        const float s = snerf::wave_sum(sum[t]);
        const unsigned c = wave_sum_u(cnt[t]);
        if (lane == 0) {
            lds_sum[wave][t] = s;
            lds_cnt[wave][t] = c;
        }
    }
    __syncthreads();
    if (threadIdx.x < kMaxTerms) {
        float s = 0.0f;
        unsigned c = 0;
        for (int w = 0; w < kBlock / 64; ++w) {
            s += lds_sum[w][threadIdx.x];
            c += lds_cnt[w][threadIdx.x];
        }
        ws->sums[blockIdx.x][threadIdx.x] = s;
        ws->counts[blockIdx.x][threadIdx.x] = c;
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) is_last = (atomicAdd(&ws->done, 1u) == gridDim.x - 1) ? 1u : 0u;
    __syncthreads();
    if (!is_last) return;
    __threadfence();
    __shared__ float term_value[kMaxTerms];
    if (threadIdx.x < kMaxTerms) {
        const int t = threadIdx.x;
        float s = 0.0f;
        unsigned c = 0;
        for (unsigned b = 0; b < gridDim.x; ++b) {
            s += ws->sums[b][t];
            c += ws->counts[b][t];
        }
        float value = 0.0f, scale = 0.0f;
        if (t < table.num_terms && c > 0) {
            const float denom = (float)table.term[t].channels * (float)c;
            value = s / denom;
            scale = 2.0f / denom;
        }
        term_value[t] = value;
        if (t < table.num_terms) {
            values[t] = value;
            scales[t] = scale;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float total = 0.0f;
        for (int g = 0; g < table.num_groups; ++g) {
            float group_value = 0.0f;
            for (int t = 0; t < table.num_terms; ++t)
                if (table.term[t].group == g) group_value += term_value[t];
            values[table.num_terms + g] = group_value;
        }
        for (int t = 0; t < table.num_terms; ++t) total += table.term[t].weight * term_value[t];
        values[table.num_terms + table.num_groups] = total;
        ws->done = 0;
    }
}

// Pass 2: every gradient.  A thread owns ray r of every term, so two terms that share a gradient buffer (a depth read
// by several losses) are summed by the same thread in table order: race-free and deterministic.
__global__ void __launch_bounds__(kBlock) loss_backward_kernel(TermTable table, long long num_rays,
                                                               const float* __restrict__ scales,
                                                               const float* __restrict__ upstream) {
    const int T = table.num_terms, G = table.num_groups;
    const float up_total = upstream[T + G];
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long ray = (long long)blockIdx.x * kBlock + threadIdx.x; ray < num_rays; ray += stride) {
#pragma unroll 1
        for (int t = 0; t < T; ++t) {
            const snerf_loss_term& term = table.term[t];
            if (!term.d_pred) continue;
            const float factor = (upstream[t] + upstream[T + term.group] + up_total * term.weight) * scales[t];
            const bool on = !term.numerator_mask || term.numerator_mask[ray];
            for (int c = 0; c < term.channels; ++c) {
                const long long i = ray * term.channels + c;
                float g = on ? factor * (term.pred[i] - term.target[i]) : 0.0f;
                if (term.accumulate) g += term.d_pred[i];
                term.d_pred[i] = g;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
struct PatchArgs {
    const float* rays_o;
    const float* rays_d;
    const float* depth1;
    const float* depth2;
    const unsigned char* ray_mask;
    const int* pixel_id;
    long long num_rays;
    const float* poses;
    const float* intrinsic;
    const float* images;
    int num_views, height, width, patch_x, patch_y;
    float threshold;
    unsigned char* mask1;
    unsigned char* mask2;
    float* rmse1;
    float* rmse2;
};

struct Landing {
    long long x, y;   // clipped to the image
    bool valid;       // un-clipped position keeps the whole patch inside the image
};

// torch: (K @ flip @ R^T @ rel), evaluated left to right with plain (unfused) multiply-adds in index order -- verified
// bit-exact against the reference's outputs; then x/z, y/z, round-half-even, cast to int64.
__device__ __forceinline__ Landing land(const float m2[3][3], float px, float py, float pz, int height, int width, int hx,
                                        int hy) {
    float cam[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) cam[i] = (m2[i][0] * px + m2[i][1] * py) + m2[i][2] * pz;
    const float fx = rintf(cam[0] / cam[2]), fy = rintf(cam[1] / cam[2]);
    // float -> int64 of a NaN / out-of-range value is INT64_MIN on the reference's host; such a ray is simply invalid
    const bool finite = fabsf(fx) < 9.0e18f && fabsf(fy) < 9.0e18f;
    const long long lo = -0x7fffffffffffffffLL - 1;
    const long long x = fabsf(fx) < 9.0e18f ? (long long)fx : lo;
    const long long y = fabsf(fy) < 9.0e18f ? (long long)fy : lo;
    Landing out;
    out.valid = finite && x >= hx && x < width - hx && y >= hy && y < height - hy;
    out.x = x < 0 ? 0 : (x > width - 1 ? width - 1 : x);
    out.y = y < 0 ? 0 : (y > height - 1 ? height - 1 : y);
    return out;
}

// gt_images_padded[view, y, x]: the reference pads H by hx rows and W by hy columns at the far end only and lets
// negative indices wrap (python indexing) into that padding (:152-158).
__device__ __forceinline__ const float* padded_pixel(const PatchArgs& a, long long view, long long y, long long x, int hx, int hy) {
    if (y < 0) y += a.height + hx;
    if (x < 0) x += a.width + hy;
    if (y < 0 || y >= a.height || x < 0 || x >= a.width) return nullptr;
    return a.images + ((view * a.height + y) * a.width + x) * 3;
}

__global__ void __launch_bounds__(256) patch_masks_kernel(PatchArgs a) {
    const long long ray = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (ray >= a.num_rays) return;
    const long long view_id = a.pixel_id[3 * ray];
    // rows outside the pixel-ray mask, and rows whose view index is not a view (a loader's -1 fill), decide nothing
    if ((a.ray_mask && !a.ray_mask[ray]) || view_id < 0 || view_id >= a.num_views) {
        if (lane == 0) {
            a.mask1[ray] = 0;
            a.mask2[ray] = 0;
            if (a.rmse1) a.rmse1[ray] = 0.0f;
            if (a.rmse2) a.rmse2[ray] = 0.0f;
        }
        return;
    }
    const int hx = a.patch_x / 2, hy = a.patch_y / 2;
    const long long view_a = view_id, x_a = a.pixel_id[3 * ray + 1], y_a = a.pixel_id[3 * ray + 2];
    // nearest other view: second entry of a stable sort of the origin distances (kthvalue(.., 2), :128-132)
    const float* pa = a.poses + view_a * 16;
    const float oax = pa[3], oay = pa[7], oaz = pa[11];
    float d0 = INFINITY, d1 = INFINITY;
    int i0 = 0, i1 = 0;
    for (int j = 0; j < a.num_views; ++j) {
        const float* pj = a.poses + (long long)j * 16;
        const float ex = oax - pj[3], ey = oay - pj[7], ez = oaz - pj[11];
        const float d = sqrtf((ex * ex + ey * ey) + ez * ez);
        if (d < d0) {
            d1 = d0; i1 = i0;
            d0 = d; i0 = j;
        } else if (d < d1) {
            d1 = d; i1 = j;
        }
    }
    const long long view_b = i1;
    const float* pb = a.poses + view_b * 16;
    float m2[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float k0 = a.intrinsic[3 * i], k1 = -a.intrinsic[3 * i + 1], k2 = -a.intrinsic[3 * i + 2];
#pragma unroll
        for (int j = 0; j < 3; ++j) m2[i][j] = (k0 * pb[4 * j] + k1 * pb[4 * j + 1]) + k2 * pb[4 * j + 2];
    }
    const float ox = a.rays_o[3 * ray], oy = a.rays_o[3 * ray + 1], oz = a.rays_o[3 * ray + 2];
    const float dx = a.rays_d[3 * ray], dy = a.rays_d[3 * ray + 1], dz = a.rays_d[3 * ray + 2];
    const float t1 = a.depth1[ray], t2 = a.depth2[ray];
    const Landing l1 = land(m2, (ox + dx * t1) - pb[3], (oy + dy * t1) - pb[7], (oz + dz * t1) - pb[11], a.height, a.width, hx, hy);
    const Landing l2 = land(m2, (ox + dx * t2) - pb[3], (oy + dy * t2) - pb[7], (oz + dz * t2) - pb[11], a.height, a.width, hx, hy);
    const bool valid_a = x_a >= hx && x_a < a.width - hx && y_a >= hy && y_a < a.height - hy;

    float s1 = 0.0f, s2 = 0.0f;
    const int pixels = a.patch_x * a.patch_y;
    for (int p = lane; p < pixels; p += 64) {
        const int offy = p / a.patch_x - hy, offx = p % a.patch_x - hx;
        const float* qa = padded_pixel(a, view_a, y_a + offy, x_a + offx, hx, hy);
        const float* q1 = padded_pixel(a, view_b, l1.y + offy, l1.x + offx, hx, hy);
        const float* q2 = padded_pixel(a, view_b, l2.y + offy, l2.x + offx, hx, hy);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float va = qa ? qa[c] : 0.0f;
            const float e1 = va - (q1 ? q1[c] : 0.0f), e2 = va - (q2 ? q2[c] : 0.0f);
            s1 += e1 * e1;
            s2 += e2 * e2;
        }
    }
    s1 = snerf::wave_sum(s1);
    s2 = snerf::wave_sum(s2);
    if (lane == 0) {
        const float count = (float)(pixels * 3);
        const float r1 = sqrtf(s1 / count), r2 = sqrtf(s2 / count);
        a.mask1[ray] = ((r1 < r2) || !l2.valid) && (r1 < a.threshold) && l1.valid && valid_a;
        a.mask2[ray] = ((r2 < r1) || !l1.valid) && (r2 < a.threshold) && l2.valid && valid_a;
        if (a.rmse1) a.rmse1[ray] = r1;
        if (a.rmse2) a.rmse2[ray] = r2;
    }
}

int fill_table(const snerf_loss_term* terms, int num_terms, int num_groups, bool backward, TermTable* table) {
    SNERF_REQUIRE(terms, "loss: NULL term table");
    SNERF_REQUIRE(num_terms >= 1 && num_terms <= kMaxTerms, "loss: num_terms %d outside [1,%d]", num_terms, kMaxTerms);
    SNERF_REQUIRE(num_groups >= 1 && num_groups <= kMaxGroups, "loss: num_groups %d outside [1,%d]", num_groups, kMaxGroups);
    for (int t = 0; t < num_terms; ++t) {
        SNERF_REQUIRE(terms[t].pred && terms[t].target, "loss: term %d has a NULL pred/target", t);
        SNERF_REQUIRE(terms[t].channels >= 1 && terms[t].channels <= 4, "loss: term %d channels %d outside [1,4]", t, terms[t].channels);
        SNERF_REQUIRE(terms[t].group >= 0 && terms[t].group < num_groups, "loss: term %d group %d outside [0,%d)", t, terms[t].group, num_groups);
        if (backward && terms[t].accumulate) {
            bool earlier = false;
            for (int s = 0; s < t; ++s) earlier |= terms[s].d_pred == terms[t].d_pred && terms[s].channels == terms[t].channels;
            SNERF_REQUIRE(earlier, "loss: term %d accumulates into a buffer no earlier term wrote", t);
        }
        table->term[t] = terms[t];
    }
    table->num_terms = num_terms;
    table->num_groups = num_groups;
    return SNERF_OK;
}

}  // namespace

extern "C" long long snerf_loss_workspace_bytes(void) { return (long long)sizeof(Workspace); }

extern "C" int snerf_loss_forward(const snerf_loss_term* terms, int num_terms, int num_groups, long long num_rays,
                                  float* values, float* scales, void* workspace, snerf_stream_t stream) {
    TermTable table;
    if (int st = fill_table(terms, num_terms, num_groups, false, &table)) return st;
    SNERF_REQUIRE(values && scales && workspace, "loss_forward: NULL pointer");
    SNERF_REQUIRE(num_rays >= 0, "loss_forward: negative ray count");
    long long blocks = (num_rays + kBlock - 1) / kBlock;
    blocks = blocks < 1 ? 1 : (blocks > kMaxBlocks ? kMaxBlocks : blocks);
    hipLaunchKernelGGL(loss_forward_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, table, num_rays,
                       values, scales, (Workspace*)workspace);
    return snerf::check_launch("loss_forward");
}

extern "C" int snerf_loss_backward(const snerf_loss_term* terms, int num_terms, int num_groups, long long num_rays,
                                   const float* scales, const float* upstream, snerf_stream_t stream) {
    TermTable table;
    if (int st = fill_table(terms, num_terms, num_groups, true, &table)) return st;
    SNERF_REQUIRE(scales && upstream, "loss_backward: NULL pointer");
    SNERF_REQUIRE(num_rays >= 0, "loss_backward: negative ray count");
    if (num_rays == 0) return SNERF_OK;
    hipLaunchKernelGGL(loss_backward_kernel, dim3(snerf::stride_grid(num_rays, kBlock)), dim3(kBlock), 0,
                       (hipStream_t)stream, table, num_rays, scales, upstream);
    return snerf::check_launch("loss_backward");
}

extern "C" int snerf_patch_consistency_masks(const float* rays_o, const float* rays_d, const float* depth1,
                                             const float* depth2, const unsigned char* ray_mask, const int* pixel_id,
                                             long long num_rays, const float* poses, const float* intrinsic,
                                             const float* images, int num_views, int height, int width, int patch_x,
                                             int patch_y, float rmse_threshold, unsigned char* mask1,
                                             unsigned char* mask2, float* rmse1, float* rmse2, snerf_stream_t stream) {
    SNERF_REQUIRE(rays_o && rays_d && depth1 && depth2 && pixel_id && poses && intrinsic && images && mask1 && mask2,
                  "patch_consistency_masks: NULL pointer");
    SNERF_REQUIRE(num_rays >= 0, "patch_consistency_masks: negative ray count");
    SNERF_REQUIRE(num_views >= 2, "patch_consistency_masks: needs at least 2 views (the reference's kthvalue(.., 2) fails on one), got %d", num_views);
    SNERF_REQUIRE(height > 0 && width > 0, "patch_consistency_masks: bad resolution %dx%d", height, width);
    SNERF_REQUIRE(patch_x >= 1 && patch_y >= 1 && (patch_x & 1) && (patch_y & 1) && patch_x * patch_y <= 4096,
                  "patch_consistency_masks: patch size must be odd and at most 4096 pixels, got %dx%d", patch_x, patch_y);
    if (num_rays == 0) return SNERF_OK;
    PatchArgs a{rays_o, rays_d, depth1, depth2, ray_mask, pixel_id, num_rays, poses, intrinsic, images, num_views,
                height, width, patch_x, patch_y, rmse_threshold, mask1, mask2, rmse1, rmse2};
    hipLaunchKernelGGL(patch_masks_kernel, dim3((unsigned)((num_rays + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
    return snerf::check_launch("patch_consistency_masks");
}
