// B1-B3: training-batch assembly, shuffled index stream and random draws on the device.
//
// All three are HBM/latency-bound streaming kernels over a few thousand rows per step; what they buy is (a) no
// ~100 B/pixel ray cache in memory and no host->device copies per chunk, (b) draws that are a pure function of
// (seed, stream, global row, column), so results do not depend on the number of ranks.
#include "snerf_common.h"
#include "raygen_device.h"

#include "../../include/simplenerf_train.h"

namespace {

static_assert(sizeof(snerf::Camera) == SNERF_CAMERA_FLOATS * sizeof(float), "camera table row layout");

__global__ void camera_table_kernel(const float* __restrict__ intrinsics, const float* __restrict__ poses, int num_views,
                                    int height, int width, snerf::Camera* __restrict__ table) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= num_views) return;
    snerf::Camera c;
    if (!snerf::make_camera(intrinsics + 9 * v, poses + 16 * v, height, width, &c)) {
        float* raw = reinterpret_cast<float*>(&c);
        for (int i = 0; i < SNERF_CAMERA_FLOATS; ++i) raw[i] = __builtin_nanf("");
    }
    table[v] = c;
}

struct AssembleArgs {
    const long long* indices;
    long long num_rays, num_pixel_rays;
    const snerf::Camera* cameras;
    int num_views, height, width, ndc;
    const float* images;
    const float* sparse_depths;
    const float* sparse_errors;
    const float* sparse_depths_ndc;
    float near, far, near_ndc, far_ndc;
    long long first_pixel_row, first_sparse_row;
    snerf_batch out;
};

__device__ __forceinline__ void fill3(float* p, float v) {
    if (p) { p[0] = v; p[1] = v; p[2] = v; }
}

__global__ void __launch_bounds__(256) assemble_batch_kernel(AssembleArgs a) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long frame = (long long)a.height * a.width;
    const snerf_batch& o = a.out;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < a.num_rays; i += stride) {
        const long long idx = a.indices[i];
        const bool pixel_row = i < a.num_pixel_rays;
        const bool ok = idx >= 0 && idx < frame * a.num_views;
        o.mask_pixel_rays[i] = ok && pixel_row;
        if (o.mask_sparse_rays) o.mask_sparse_rays[i] = ok && !pixel_row;
        if (o.global_rows) o.global_rows[i] = pixel_row ? a.first_pixel_row + i : a.first_sparse_row + (i - a.num_pixel_rays);
        float* on = a.ndc ? o.rays_o_ndc + 3 * i : nullptr;
        float* dn = a.ndc ? o.rays_d_ndc + 3 * i : nullptr;
        float sd = -1.0f, se = -1.0f, sn = -1.0f, nr = -1.0f, fr = -1.0f, nn = -1.0f, fn = -1.0f;
        if (ok) {
            const int view = (int)(idx / frame);
            const int pix = (int)(idx - (long long)view * frame);
            const int y = pix / a.width, x = pix - y * a.width;
            snerf::pinhole_ray(a.cameras[view], (float)x, (float)y, a.near, a.ndc != 0, o.rays_o + 3 * i, o.rays_d + 3 * i,
                               o.view_dirs + 3 * i, on, dn);
            o.pixel_id[3 * i] = view; o.pixel_id[3 * i + 1] = x; o.pixel_id[3 * i + 2] = y;
            if (pixel_row) {
                const float* px = a.images + 3 * idx;
                o.target_rgb[3 * i] = px[0]; o.target_rgb[3 * i + 1] = px[1]; o.target_rgb[3 * i + 2] = px[2];
            } else {
                fill3(o.target_rgb + 3 * i, -1.0f);
                if (a.sparse_depths) sd = a.sparse_depths[idx];
                if (a.sparse_errors) se = a.sparse_errors[idx];
                if (a.sparse_depths_ndc) sn = a.sparse_depths_ndc[idx];
            }
            nr = a.near; fr = a.far; nn = a.near_ndc; fn = a.far_ndc;
        } else {
            fill3(o.rays_o + 3 * i, -1.0f); fill3(o.rays_d + 3 * i, -1.0f); fill3(o.view_dirs + 3 * i, -1.0f);
            fill3(on, -1.0f); fill3(dn, -1.0f); fill3(o.target_rgb + 3 * i, -1.0f);
            o.pixel_id[3 * i] = -1; o.pixel_id[3 * i + 1] = -1; o.pixel_id[3 * i + 2] = -1;
        }
        o.near[i] = nr; o.far[i] = fr;
        if (a.ndc) { o.near_ndc[i] = nn; o.far_ndc[i] = fn; }
        if (o.sparse_depth_values) o.sparse_depth_values[i] = sd;
        if (o.sparse_depth_errors) o.sparse_depth_errors[i] = se;
        if (o.sparse_depth_values_ndc) o.sparse_depth_values_ndc[i] = sn;
    }
}

// ---------------------------------------------------------------------------------------------------------------
constexpr int kFeistelRounds = 8;

struct ShuffleArgs {
    unsigned keys[kFeistelRounds];
    // `at` != NULL (snerf_shuffled_indices_at): epoch and first come from the device-resident iteration record -- the keys are
    // then derived in the kernel (the same splitmix64 the host evaluates) and `first` is the shard offset added to its position
    const snerf_iteration* at;
    unsigned long long seed;
    int sparse;
    long long first, count, domain;
    const long long* candidates;
    int half_bits;
    int height, width, crop_y0, crop_x0, crop_h, crop_w;
    long long* out;
};

__host__ __device__ inline unsigned long long splitmix64(unsigned long long z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__host__ __device__ inline unsigned mix32(unsigned h) {   // murmur3 finaliser
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}

__global__ void __launch_bounds__(256) shuffled_indices_kernel(ShuffleArgs a) {
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= a.count) return;
    const unsigned long long mask = (1ull << a.half_bits) - 1ull;
    unsigned keys[kFeistelRounds];
    long long first = a.first;
    if (a.at) {
        const unsigned long long epoch = (unsigned long long)(a.sparse ? a.at->sparse_epoch : a.at->pixel_epoch);
        first += a.sparse ? a.at->sparse_first : a.at->pixel_first;
#pragma unroll
        for (int r = 0; r < kFeistelRounds; ++r) keys[r] = (unsigned)splitmix64(a.seed ^ splitmix64(epoch * kFeistelRounds + r));
    } else {
#pragma unroll
        for (int r = 0; r < kFeistelRounds; ++r) keys[r] = a.keys[r];
    }
    unsigned long long x = (unsigned long long)(first + j);
    do {   // cycle walking: the network permutes [0, 2^(2 half_bits)); re-apply until the image falls inside the domain
        unsigned long long left = x >> a.half_bits, right = x & mask;
#pragma unroll
        for (int r = 0; r < kFeistelRounds; ++r) {
            const unsigned long long next = left ^ ((unsigned long long)mix32((unsigned)right + keys[r]) & mask);
            left = right;
            right = next;
        }
        x = (left << a.half_bits) | right;
    } while (x >= (unsigned long long)a.domain);
    long long index;
    if (a.candidates) {
        index = a.candidates[x];
    } else {
        const long long per_view = (long long)a.crop_h * a.crop_w;
        const long long view = (long long)x / per_view, rest = (long long)x - view * per_view;
        const long long y = a.crop_y0 + rest / a.crop_w, xx = a.crop_x0 + rest % a.crop_w;
        index = (view * a.height + y) * a.width + xx;
    }
    a.out[j] = index;
}


// ---------------------------------------------------------------------------------------------------------------
struct Philox { unsigned v[4]; };

__device__ __forceinline__ Philox philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return Philox{{c0, c1, c2, c3}};
}

// `at` != NULL (the _at entry points): stream_id = at->iter_num * num_kinds + kind, with (kind, num_kinds) passed in
// `stream_id` as kind | num_kinds << 16
template <bool NORMAL>
__global__ void __launch_bounds__(256) random_kernel(unsigned seed_lo, unsigned seed_hi, unsigned stream_id, long long first_row,
                                                     const long long* __restrict__ row_ids, long long num_rows, int row_width,
                                                     float scale, float* __restrict__ out, const snerf_iteration* __restrict__ at) {
    if (at) stream_id = (unsigned)((unsigned long long)at->iter_num * (stream_id >> 16) + (stream_id & 0xffffu));
    const int blocks_per_row = (row_width + 3) / 4;
    const long long total = num_rows * blocks_per_row;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const long long row = t / blocks_per_row;
        const int block = (int)(t - row * blocks_per_row);
        const unsigned long long grow = (unsigned long long)(row_ids ? row_ids[row] : first_row + row);
        const Philox p = philox4x32_10((unsigned)grow, (unsigned)(grow >> 32), (unsigned)block, stream_id, seed_lo, seed_hi);
        float v[4];
        if (NORMAL) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float u1 = (float)((p.v[2 * h] >> 8) + 1u) * 5.9604644775390625e-8f;   // (0,1]
                const float u2 = (float)(p.v[2 * h + 1] >> 8) * 5.9604644775390625e-8f;       // [0,1)
                const float radius = sqrtf(-2.0f * logf(u1));
                float s, c;
                sincospif(2.0f * u2, &s, &c);
                v[2 * h] = scale * (radius * c);
                v[2 * h + 1] = scale * (radius * s);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = (float)(p.v[k] >> 8) * 5.9604644775390625e-8f;   // 2^-24
        }
        float* dst = out + row * row_width + 4 * block;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (4 * block + k < row_width) dst[k] = v[k];
    }
}

int launch_random(bool normal, unsigned long long seed, unsigned stream_id, long long first_row, const long long* row_ids,
                  long long num_rows, int row_width, float scale, float* out, snerf_stream_t stream,
                  const snerf_iteration* at = nullptr) {
    SNERF_REQUIRE(out, "random draws: NULL output");
    SNERF_REQUIRE(num_rows >= 0 && row_width >= 1 && first_row >= 0, "random draws: bad shape (%lld rows from %lld, width %d)",
                  num_rows, first_row, row_width);
    if (num_rows == 0) return SNERF_OK;
    const long long work = num_rows * ((row_width + 3) / 4);
    const dim3 grid(snerf::stride_grid(work, 256)), block(256);
    if (normal)
        hipLaunchKernelGGL(random_kernel<true>, grid, block, 0, (hipStream_t)stream, (unsigned)seed, (unsigned)(seed >> 32),
                           stream_id, first_row, row_ids, num_rows, row_width, scale, out, at);
    else
        hipLaunchKernelGGL(random_kernel<false>, grid, block, 0, (hipStream_t)stream, (unsigned)seed, (unsigned)(seed >> 32),
                           stream_id, first_row, row_ids, num_rows, row_width, scale, out, at);
    return snerf::check_launch("random draws");
}

}  // namespace

extern "C" int snerf_camera_table(const float* intrinsics, const float* poses, int num_views, int height, int width,
                                  float* table, snerf_stream_t stream) {
    SNERF_REQUIRE(intrinsics && poses && table, "camera_table: NULL pointer");
    SNERF_REQUIRE(num_views >= 1 && height > 0 && width > 0, "camera_table: bad sizes (%d views, %dx%d)", num_views, height, width);
    hipLaunchKernelGGL(camera_table_kernel, dim3((num_views + 63) / 64), dim3(64), 0, (hipStream_t)stream, intrinsics, poses,
                       num_views, height, width, reinterpret_cast<snerf::Camera*>(table));
    return snerf::check_launch("camera_table");
}

extern "C" int snerf_assemble_batch(const long long* indices, long long num_rays, long long num_pixel_rays,
                                    const float* camera_table, int num_views, int height, int width, const float* images,
                                    const float* sparse_depths, const float* sparse_errors, const float* sparse_depths_ndc,
                                    int ndc, float near, float far, float near_ndc, float far_ndc, long long first_pixel_row,
                                    long long first_sparse_row, const snerf_batch* out, snerf_stream_t stream) {
    SNERF_REQUIRE(indices && camera_table && images && out, "assemble_batch: NULL pointer");
    SNERF_REQUIRE(num_rays >= 0 && num_pixel_rays >= 0 && num_pixel_rays <= num_rays,
                  "assemble_batch: %lld pixel rays of %lld rows", num_pixel_rays, num_rays);
    SNERF_REQUIRE(num_views >= 1 && height > 0 && width > 0, "assemble_batch: bad sizes (%d views, %dx%d)", num_views, height, width);
    SNERF_REQUIRE(out->rays_o && out->rays_d && out->view_dirs && out->pixel_id && out->target_rgb && out->near && out->far &&
                  out->mask_pixel_rays, "assemble_batch: a required output is NULL");
    SNERF_REQUIRE(!ndc || (out->rays_o_ndc && out->rays_d_ndc && out->near_ndc && out->far_ndc),
                  "assemble_batch: ndc requested but an NDC output is NULL");
    if (num_rays == 0) return SNERF_OK;
    AssembleArgs a{indices, num_rays, num_pixel_rays, reinterpret_cast<const snerf::Camera*>(camera_table), num_views, height,
                   width, ndc, images, sparse_depths, sparse_errors, sparse_depths_ndc, near, far, near_ndc, far_ndc,
                   first_pixel_row, first_sparse_row, *out};
    hipLaunchKernelGGL(assemble_batch_kernel, dim3(snerf::stride_grid(num_rays, 256)), dim3(256), 0, (hipStream_t)stream, a);
    return snerf::check_launch("assemble_batch");
}

static int shuffled_indices_impl(unsigned long long seed, unsigned long long epoch, long long first, long long count,
                                 long long domain, const long long* candidates, int num_views, int height, int width,
                                 int crop_y0, int crop_y1, int crop_x0, int crop_x1, long long* out, snerf_stream_t stream,
                                 const snerf_iteration* at, int sparse) {
    SNERF_REQUIRE(out, "shuffled_indices: NULL output");
    SNERF_REQUIRE(domain >= 1 && domain < (1LL << 62), "shuffled_indices: domain %lld outside [1, 2^62)", domain);
    SNERF_REQUIRE(first >= 0 && count >= 0 && first + count <= domain,
                  "shuffled_indices: positions [%lld, %lld) outside the epoch of %lld", first, first + count, domain);
    ShuffleArgs a;
    a.at = at; a.seed = seed; a.sparse = sparse;
    if (!candidates) {
        SNERF_REQUIRE(num_views >= 1 && crop_y0 >= 0 && crop_y0 < crop_y1 && crop_y1 <= height && crop_x0 >= 0 &&
                      crop_x0 < crop_x1 && crop_x1 <= width, "shuffled_indices: bad crop window [%d,%d)x[%d,%d) of %dx%d",
                      crop_y0, crop_y1, crop_x0, crop_x1, height, width);
        SNERF_REQUIRE(domain == (long long)num_views * (crop_y1 - crop_y0) * (crop_x1 - crop_x0),
                      "shuffled_indices: domain %lld is not views x crop window", domain);
    }
    if (count == 0) return SNERF_OK;
    int bits = 2;
    while ((1ull << bits) < (unsigned long long)domain) ++bits;
    bits += bits & 1;
    a.half_bits = bits / 2;
    for (int r = 0; r < kFeistelRounds; ++r)
        a.keys[r] = (unsigned)splitmix64(seed ^ splitmix64(epoch * kFeistelRounds + r));
    a.first = first; a.count = count; a.domain = domain; a.candidates = candidates;
    a.height = height; a.width = width; a.crop_y0 = crop_y0; a.crop_x0 = crop_x0;
    a.crop_h = crop_y1 - crop_y0; a.crop_w = crop_x1 - crop_x0;
    a.out = out;
    hipLaunchKernelGGL(shuffled_indices_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return snerf::check_launch("shuffled_indices");
}

extern "C" int snerf_shuffled_indices(unsigned long long seed, unsigned long long epoch, long long first, long long count,
                                      long long domain, const long long* candidates, int num_views, int height, int width,
                                      int crop_y0, int crop_y1, int crop_x0, int crop_x1, long long* out,
                                      snerf_stream_t stream) {
    return shuffled_indices_impl(seed, epoch, first, count, domain, candidates, num_views, height, width, crop_y0, crop_y1,
                                 crop_x0, crop_x1, out, stream, nullptr, 0);
}

extern "C" int snerf_shuffled_indices_at(unsigned long long seed, const snerf_iteration* current, int sparse, long long first_offset,
                                         long long count, long long domain, const long long* candidates, int num_views,
                                         int height, int width, int crop_y0, int crop_y1, int crop_x0, int crop_x1,
                                         long long* out, snerf_stream_t stream) {
    SNERF_REQUIRE(current, "shuffled_indices_at: NULL iteration record");
    return shuffled_indices_impl(seed, 0, first_offset, count, domain, candidates, num_views, height, width, crop_y0, crop_y1,
                                 crop_x0, crop_x1, out, stream, current, sparse ? 1 : 0);
}

// G1: the first node of a replayed training graph -- record (counter mod slots) of the host's pinned ring becomes the current one
namespace {
__global__ void iteration_advance_kernel(const snerf_iteration* __restrict__ ring, int slots, unsigned long long* counter,
                                         snerf_iteration* __restrict__ current) {
    const unsigned long long c = *counter;
    *current = ring[c % (unsigned long long)slots];
    *counter = c + 1;
}
}  // namespace

extern "C" int snerf_iteration_advance(const snerf_iteration* ring, int ring_slots, unsigned long long* counter,
                                       snerf_iteration* current, snerf_stream_t stream) {
    SNERF_REQUIRE(ring && counter && current, "iteration_advance: NULL pointer");
    SNERF_REQUIRE(ring_slots >= 2, "iteration_advance: the ring needs at least two slots, got %d", ring_slots);
    hipLaunchKernelGGL(iteration_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, ring, ring_slots, counter, current);
    return snerf::check_launch("iteration_advance");
}

extern "C" int snerf_random_uniform_at(unsigned long long seed, const snerf_iteration* current, int kind, int num_kinds,
                                       long long first_row, const long long* row_ids, long long num_rows, int row_width,
                                       float* out, snerf_stream_t stream) {
    SNERF_REQUIRE(current && kind >= 0 && num_kinds > kind && num_kinds < 65536, "random_uniform_at: bad record / kind %d of %d", kind, num_kinds);
    return launch_random(false, seed, (unsigned)kind | ((unsigned)num_kinds << 16), first_row, row_ids, num_rows, row_width, 1.0f, out,
                         stream, current);
}

extern "C" int snerf_random_normal_at(unsigned long long seed, const snerf_iteration* current, int kind, int num_kinds,
                                      long long first_row, const long long* row_ids, long long num_rows, int row_width, float scale,
                                      float* out, snerf_stream_t stream) {
    SNERF_REQUIRE(current && kind >= 0 && num_kinds > kind && num_kinds < 65536, "random_normal_at: bad record / kind %d of %d", kind, num_kinds);
    return launch_random(true, seed, (unsigned)kind | ((unsigned)num_kinds << 16), first_row, row_ids, num_rows, row_width, scale, out,
                         stream, current);
}

extern "C" int snerf_random_uniform(unsigned long long seed, unsigned int stream_id, long long first_row,
                                    const long long* row_ids, long long num_rows, int row_width, float* out,
                                    snerf_stream_t stream) {
    return launch_random(false, seed, stream_id, first_row, row_ids, num_rows, row_width, 1.0f, out, stream);
}

extern "C" int snerf_random_normal(unsigned long long seed, unsigned int stream_id, long long first_row,
                                   const long long* row_ids, long long num_rows, int row_width, float scale, float* out,
                                   snerf_stream_t stream) {
    return launch_random(true, seed, stream_id, first_row, row_ids, num_rows, row_width, scale, out, stream);
}
