/*
 * simplenerf_train.h -- C ABI of the training-step pieces either side of the ray-marching path (SURVEY 8f "next"
 * rows): loss evaluation (f1), batch assembly and random draws (f2), the optimiser update (f4).  Same conventions as
 * simplenerf_hip.h: device pointers to contiguous arrays in the reference's layouts, calls only enqueue on `stream`,
 * 0 / negative snerf_status return, no CPU fallback.
 */
#ifndef SIMPLENERF_TRAIN_H
#define SIMPLENERF_TRAIN_H

#include "simplenerf_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------------------------------------------------------
 * L1  fused loss evaluation.  One launch evaluates every masked mean-squared-error term of a training step and one
 * launch writes every gradient.  Replaces, per term:
 *   MSE.compute_mse                       (src/loss_functions/MSE01.py:57-67; MSE02/MSE03 identical)
 *   SparseDepthMSE.compute_depth_loss     (src/loss_functions/SparseDepthMSE01.py:58-71; 02/03 identical)
 *   ...DepthLoss.compute_depth_mse        (src/loss_functions/PointsAugmentationDepthLoss02.py:196-212)
 *   CoarseFineConsistencyLoss.compute_loss_sd (src/loss_functions/CoarseFineConsistencyLoss02.py:174-189)
 * and the weighted sum of LossComputer.compute_losses (src/loss_functions/LossComputer01.py:40-52).
 *
 *   term value  = sum over rays with numerator_mask of sum_c (pred - target)^2  /  (channels * #rays with
 *                 denominator_mask);  0 when the denominator count is 0 (the reference's `if numel() > 0 else 0`)
 *   numerator_mask == denominator_mask for the plain masked MSE; the patch-consistency depth terms average over all
 *   pixel rays but only count the rays the decision mask keeps (the reference zeroes the others, then takes the mean).
 *   `target` never receives a gradient (the reference detaches it or it is data).
 */
#define SNERF_LOSS_MAX_TERMS 16
#define SNERF_LOSS_MAX_GROUPS 16

typedef struct snerf_loss_term {
    const float* pred;                     /* device (num_rays, channels) */
    const float* target;                   /* device (num_rays, channels) */
    const unsigned char* numerator_mask;   /* device (num_rays) 0/1 bytes, or NULL = every ray */
    const unsigned char* denominator_mask; /* device (num_rays) 0/1 bytes, or NULL = every ray */
    float* d_pred;                         /* backward only: device (num_rays, channels) gradient buffer */
    int channels;                          /* 1 (depth) or 3 (colour) */
    int group;                             /* which loss (LossComputer entry) the term belongs to, 0 <= group < num_groups */
    int accumulate;                        /* backward only: 1 = add to d_pred (an earlier term of the table wrote it) */
    float weight;                          /* the loss weight of LossComputer.get_loss_weight for this iteration */
} snerf_loss_term;

/* Bytes of device scratch snerf_loss_forward needs (partial sums + a completion counter that must be zero on first
 * use; the kernel leaves it zero again). */
long long snerf_loss_workspace_bytes(void);

/* values  device (num_terms + num_groups + 1): [0,T) term values, [T,T+G) per-loss sums of their terms (what the
 *         reference reports as loss_value per loss name), [T+G] = sum_t weight_t * value_t (TotalLoss)
 * scales  device (num_terms): 2 / (channels * count), 0 for an empty term -- consumed by snerf_loss_backward */
int snerf_loss_forward(const snerf_loss_term* terms, int num_terms, int num_groups, long long num_rays, float* values,
                       float* scales, void* workspace, snerf_stream_t stream);

/* upstream  device (num_terms + num_groups + 1): gradient of the caller's scalar with respect to `values` (all zero
 *           except a 1 at [T+G] when the caller back-propagates TotalLoss).
 * Writes d_pred of every term:  (upstream[t] + upstream[T+group] + upstream[T+G]*weight) * scale * (pred - target)
 * on the numerator-masked rays, 0 elsewhere; terms with accumulate=1 add to what an earlier term wrote. */
int snerf_loss_backward(const snerf_loss_term* terms, int num_terms, int num_groups, long long num_rays,
                        const float* scales, const float* upstream, snerf_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * L2  patch-reprojection decision masks.  Replaces the decision part of compute_loss_nerf
 * (src/loss_functions/PointsAugmentationDepthLoss02.py:119-169; ViewsAugmentationDepthLoss02 and
 * CoarseFineConsistencyLoss02 share it) together with CommonUtils.reproject (src/utils/CommonUtils01.py:45-71):
 * the 3-D points of two depth estimates are projected into the nearest other training view, patch_x x patch_y
 * patches of the ground-truth images are compared with the patch around the source pixel, and
 *   mask1 = estimate 1 matches better (or estimate 2 lands outside), its patch RMSE < threshold, all patches inside
 *   mask2 = likewise for estimate 2
 *   rays_o, rays_d  device (n,3)     depth1, depth2  device (n)      ray_mask  device (n) bytes or NULL: rays outside
 *   pixel_id  device (n,3) int32 (view, x, y)                         it get mask1 = mask2 = 0
 *   poses  device (num_views,4,4) camera-to-world    intrinsic  device (3,3): the FIRST view's, used for every ray
 *   images device (num_views, height, width, 3) in [0,1]              (CommonUtils01.py:66)
 *   mask1, mask2  device (n) bytes;  rmse1, rmse2  device (n) or NULL (diagnostics)
 */
int snerf_patch_consistency_masks(const float* rays_o, const float* rays_d, const float* depth1, const float* depth2,
                                  const unsigned char* ray_mask, const int* pixel_id, long long num_rays,
                                  const float* poses, const float* intrinsic, const float* images, int num_views,
                                  int height, int width, int patch_x, int patch_y, float rmse_threshold,
                                  unsigned char* mask1, unsigned char* mask2, float* rmse1, float* rmse2,
                                  snerf_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * B1  training-batch assembly.  Replaces DataPreprocessor.load_cached_next_batch -> load_nerf_cached_batch /
 * load_sparse_depth_cached_batch (src/data_preprocessors/DataPreprocessor01.py:514-551, :586-636, :655-704) and the ray
 * cache they read (preprocess_nerf_data :284-349: ~100 B per pixel of every training view).  Instead of gathering
 * rows of a precomputed cache, each row's rays are recomputed from its view's camera (bit-identical to the cached
 * numpy values, same arithmetic as K1), so the only gathers left are the target colour and the sparse-depth tables.
 *
 * snerf_camera_table: per-view camera constants on the device.
 *   intrinsics device (num_views,3,3)   poses device (num_views,4,4) processed camera-to-world (what the reference
 *   passes to get_rays, :299)           table device (num_views, SNERF_CAMERA_FLOATS); a singular intrinsic yields NaNs
 */
#define SNERF_CAMERA_FLOATS 24
int snerf_camera_table(const float* intrinsics, const float* poses, int num_views, int height, int width, float* table,
                       snerf_stream_t stream);

typedef struct snerf_batch {
    float* rays_o;                  /* device (n,3) */
    float* rays_d;                  /* device (n,3) */
    float* view_dirs;               /* device (n,3) */
    float* rays_o_ndc;              /* device (n,3); NULL when ndc == 0 */
    float* rays_d_ndc;              /* device (n,3); NULL when ndc == 0 */
    int* pixel_id;                  /* device (n,3) int32 (view, x, y) */
    float* target_rgb;              /* device (n,3); sparse-depth rows hold -1 (the loader's fill value, :598) */
    float* near;                    /* device (n,1) */
    float* far;                     /* device (n,1) */
    float* near_ndc;                /* device (n,1); NULL when ndc == 0 */
    float* far_ndc;                 /* device (n,1); NULL when ndc == 0 */
    float* sparse_depth_values;     /* device (n,1) or NULL; pixel-ray rows hold -1 (:689-693) */
    float* sparse_depth_errors;     /* device (n,1) or NULL */
    float* sparse_depth_values_ndc; /* device (n,1) or NULL */
    unsigned char* mask_pixel_rays; /* device (n) bytes: indices_mask_nerf */
    unsigned char* mask_sparse_rays;/* device (n) bytes: indices_mask_sparse_depth, or NULL */
    long long* global_rows;         /* device (n) int64 or NULL: the row each ray has in the SINGLE-PROCESS batch (pixel rays
                                       first, then sparse-depth rays) -- first_pixel_row + i for the pixel rows,
                                       first_sparse_row + (i - num_pixel_rays) for the others.  A per-row tensor, so that it
                                       is cut along with the batch by a trainer that slices sub-batches
                                       (src/Trainer01.py:82-90); the renderer keys its training draws on it */
} snerf_batch;

/*   indices   device (n) int64 global pixel indices  view*height*width + y*width + x; the first num_pixel_rays rows
 *             are pixel rays, the rest sparse-depth rays (the concatenation order of select_batch_indices :553-584).
 *             An index outside [0, num_views*height*width) leaves its row at the loader's -1 fill with both masks 0.
 *   images    device (num_views,height,width,3)     sparse_*  device (num_views*height*width) dense tables or NULL
 *   near      also the near plane of the NDC warp (get_ndc_rays(..., near), :309) */
int snerf_assemble_batch(const long long* indices, long long num_rays, long long num_pixel_rays,
                         const float* camera_table, int num_views, int height, int width, const float* images,
                         const float* sparse_depths, const float* sparse_errors, const float* sparse_depths_ndc, int ndc,
                         float near, float far, float near_ndc, float far_ndc, long long first_pixel_row,
                         long long first_sparse_row, const snerf_batch* out, snerf_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * B2  shuffled index stream.  Replaces the host-side index list of generate_indices / select_batch_indices (:252-270,
 * :556-563: numpy.arange + numpy.random.shuffle per epoch, sliced per batch, copied to the device) by positions
 * [first, first+count) of a keyed pseudo-random PERMUTATION of the candidate set, computed on the fly (balanced
 * Feistel network with cycle walking; nothing of size "all rays" is ever stored or shuffled).  The numpy stream itself
 * cannot be reproduced on a device; callers that need the reference's exact order pass its indices to
 * snerf_assemble_batch instead.
 *   domain      number of candidates in an epoch; output position j holds candidate perm_{seed,epoch}(j)
 *   candidates  device (domain) int64 list to index (the sparse-depth pixels, :441), or NULL: the candidates are the
 *               pixels of the crop window rows [crop_y0,crop_y1) x columns [crop_x0,crop_x1) of every view
 *               (generate_indices' precrop :258-268; the full image is 0,height,0,width) and
 *               domain must equal num_views*(crop_y1-crop_y0)*(crop_x1-crop_x0)
 *   out         device (count) int64 global pixel indices
 */
int snerf_shuffled_indices(unsigned long long seed, unsigned long long epoch, long long first, long long count,
                           long long domain, const long long* candidates, int num_views, int height, int width,
                           int crop_y0, int crop_y1, int crop_x0, int crop_x1, long long* out, snerf_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * B3  training draws on the device.  Replaces the CPU-generator draws the reference makes per chunk and copies over
 * (src/models/SimpleNeRF01.py:299 stratified jitter, :341 inverse-CDF u, :670 density noise) by Philox4x32-10
 * (Salmon et al., SC'11; Random123 known-answer vectors hold) in counter mode: element (row, col) of a draw depends
 * only on (seed, stream_id, first_row + row, col), so a ray's draws do not depend on how rays are sharded over ranks.
 *   uniform: [0,1) with 24 random bits, like torch.rand;  normal: Box-Muller on pairs of 24-bit uniforms, times scale
 *   out  device (num_rows, row_width)
 *   row_ids  device (num_rows) int64 global row of each output row (snerf_batch.global_rows), or NULL: row r is global row
 *            first_row + r
 */
int snerf_random_uniform(unsigned long long seed, unsigned int stream_id, long long first_row, const long long* row_ids,
                         long long num_rows, int row_width, float* out, snerf_stream_t stream);
int snerf_random_normal(unsigned long long seed, unsigned int stream_id, long long first_row, const long long* row_ids,
                        long long num_rows, int row_width, float scale, float* out, snerf_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * O1  Adam update of every parameter tensor in one launch per 64 tensors.  Replaces the optimiser call of the
 * reference's trainer, torch.optim.Adam(params, lr, betas).step() (src/Trainer01.py:102, :516-517; PyTorch 2.x
 * single-tensor algorithm, weight_decay = 0, amsgrad = False), with the learning rate the trainer writes into
 * param_groups before each step (:293-295).  Element-wise arithmetic and its order are those of the CPU path of that
 * algorithm, bit for bit:
 *     m <- fma(1-beta1, g - m, m)            v <- fma((1-beta2) g, g, beta2 v)
 *     p <- p + (-(lr / (1-beta1^step)) m) / (sqrt(v) / sqrt(1-beta2^step) + eps)
 * with the scalar factors evaluated in double on the host and rounded to float once, as Python/ATen do.
 *   params, grads, exp_avg, exp_avg_sq   HOST arrays of num_tensors DEVICE pointers; grads[i] == NULL skips tensor i
 *   sizes   HOST array of element counts   step  1-based step count of this update (after the increment)
 */
int snerf_adam_step(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                    const long long* sizes, int num_tensors, long long step, double lr, double beta1, double beta2,
                    double eps, snerf_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * G1  per-iteration scalars in DEVICE memory (round 3), so that batch assembly, the training draws and the optimiser step
 * can be part of a captured HIP graph: a replayed graph cannot change kernel arguments, but the position in the epoch, the
 * iteration number that keys the draws and Adam's step-dependent factors change every iteration.  The host writes them,
 * ahead of time, into a ring of `snerf_iteration` records in PINNED host memory; the first node of the graph,
 * snerf_iteration_advance, copies record (counter mod ring_slots) into the device-resident `current` record and increments
 * the device-resident counter; the `_at` variants below read `current` instead of taking the scalars as arguments.  The
 * host must not run more than ring_slots - 1 replays ahead of the device (it fills slot i mod ring_slots for replay i).
 * Same arithmetic as the scalar-argument entry points, bit for bit.
 */
typedef struct snerf_iteration {
    long long iter_num;                      /* trainer iteration: draws use stream iter_num * num_kinds + kind */
    long long pixel_epoch, pixel_first;      /* this iteration's pixel rows: positions pixel_first.. of epoch pixel_epoch */
    long long sparse_epoch, sparse_first;    /* ... and its sparse-depth rows */
    float adam_neg_step_size;                /* -(lr / (1 - beta1^step)), evaluated in double by the host (as snerf_adam_step) */
    float adam_bias2_sqrt;                   /* sqrt(1 - beta2^step) */
    long long reserved[2];
} snerf_iteration;

/*   ring      PINNED host memory visible to the device (hipHostMalloc / torch pin_memory), ring_slots records
 *   counter   device, one unsigned 64-bit word (0 before the first replay)      current   device, one record */
int snerf_iteration_advance(const snerf_iteration* ring, int ring_slots, unsigned long long* counter,
                            snerf_iteration* current, snerf_stream_t stream);
/* snerf_shuffled_indices with (epoch, first) = current->{pixel,sparse}_{epoch,first} + first_offset (a rank's shard of the
 * slice); `sparse` selects the pair.  The caller guarantees that the slice lies inside the epoch (the host knows). */
int snerf_shuffled_indices_at(unsigned long long seed, const snerf_iteration* current, int sparse, long long first_offset,
                              long long count, long long domain, const long long* candidates, int num_views, int height,
                              int width, int crop_y0, int crop_y1, int crop_x0, int crop_x1, long long* out,
                              snerf_stream_t stream);
/* snerf_random_uniform / _normal with stream_id = current->iter_num * num_kinds + kind */
int snerf_random_uniform_at(unsigned long long seed, const snerf_iteration* current, int kind, int num_kinds, long long first_row,
                            const long long* row_ids, long long num_rows, int row_width, float* out, snerf_stream_t stream);
int snerf_random_normal_at(unsigned long long seed, const snerf_iteration* current, int kind, int num_kinds, long long first_row,
                           const long long* row_ids, long long num_rows, int row_width, float scale, float* out,
                           snerf_stream_t stream);
/* snerf_adam_step with the two step-dependent factors taken from `current` */
int snerf_adam_step_at(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                       const long long* sizes, int num_tensors, const snerf_iteration* current, double beta1, double beta2,
                       double eps, snerf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SIMPLENERF_TRAIN_H */
