/*
 * simplenerf_hip.h -- C ABI of the MI355X (gfx950) volumetric renderer for SimpleNeRF's ray-marching path.
 *
 * Drop-in boundary: every entry point replaces one function of the reference's Python path (file:line below are
 * relative to the reference checkout).  All pointers marked "device" are HIP device pointers to contiguous fp32
 * arrays in the reference's row-major layouts; `stream` is a hipStream_t passed as void* (NULL = default stream).
 * Calls only enqueue work on `stream`; they never synchronise, allocate or free (safe under hipGraph capture).
 *
 * Return value: 0 on success, negative snerf_status on failure; snerf_last_error() returns a thread-local
 * message for the last failure on the calling thread.  There is no CPU fallback: without a HIP device every
 * compute entry point fails with SNERF_E_HIP.
 */
#ifndef SIMPLENERF_HIP_H
#define SIMPLENERF_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* snerf_stream_t; /* hipStream_t */

enum snerf_status {
    SNERF_OK = 0,
    SNERF_E_INVALID = -1,     /* bad argument (null pointer, non-positive size, ...) */
    SNERF_E_UNSUPPORTED = -2, /* configuration outside what the kernels are built for */
    SNERF_E_HIP = -3,         /* HIP runtime error (message carries hipGetErrorString) */
    SNERF_E_RANGE = -4        /* an EARLIER fp16-mode launch on this device met a value outside the fp16 range (see
                                 snerf_range_status): its results are wrong; reported once by the next fp16-mode call */
};

/* ABI version of this header; bumped on any signature change (new enum values such as SNERF_PRECISION_F16 extend a
 * version without changing it: older callers never pass them). */
#define SNERF_ABI_VERSION 9
int snerf_abi_version(void);
const char* snerf_last_error(void);

/* ---------------------------------------------------------------------------------------------------------------
 * K1  ray generation.  Replaces DataPreprocessor.get_rays (src/data_preprocessors/DataPreprocessor01.py:351-368),
 * get_view_dirs (:392-394) and get_ndc_rays (:371-389) for the pixel range [first_ray, first_ray+num_rays) of an
 * height x width frame (row-major pixel index, x fastest), so that each rank of a sharded render generates only
 * its own rays.
 *   intrinsic   host, 3x3 row-major        pose  host, 4x4 row-major *processed* camera-to-world (the matrix
 *               the reference hands to get_rays; see create_test_data :816-831)
 *   pixel_offset  0 (reference default) or 0.5 (its 'mip_nerf' switch, :357-359)
 *   near        world near plane used by the NDC warp (model_configs['near'])
 *   outputs     device, (num_rays,3) each; rays_o_ndc / rays_d_ndc may be NULL when ndc == 0; view_dirs may be NULL
 */
int snerf_generate_rays(int height, int width, const float* intrinsic, const float* pose, float pixel_offset,
                        int ndc, float near, long long first_ray, long long num_rays, float* rays_o, float* rays_d,
                        float* view_dirs, float* rays_o_ndc, float* rays_d_ndc, snerf_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * K2  coarse depths.  Replaces SimpleNeRF.get_z_vals_coarse (src/models/SimpleNeRF01.py:272-302).
 *   near, far   device (num_rays)           t_rand  device (num_rays, num_samples) uniform [0,1) draws for the
 *               stratified jitter of :293-301, or NULL for the unperturbed (eval) depths
 *   depths      device (num_rays, num_samples)
 */
int snerf_coarse_depths(const float* near, const float* far, long long num_rays, int num_samples, int lindisp,
                        const float* t_rand, float* depths, snerf_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * K3  positional encoding + MLP.  Replaces run_network/batchify + MLP.forward (src/models/SimpleNeRF01.py:363-428,
 * :560-715) including the sample-position computation of render_rays (:139-142, :203-206).
 */
typedef struct snerf_mlp_desc {
    int points_net_depth;    /* number of trunk Linear layers (skip re-injection after layer index 4, :580) */
    int points_net_width;    /* 128 or 256 */
    int views_net_depth;     /* 1 (only value any shipped config uses) when view_dependent_rgb, else ignored */
    int views_net_width;     /* 64 or 128 */
    int points_pe_degree;    /* <= 10 */
    int views_pe_degree;     /* <= 4; ignored unless use_view_dirs */
    int sigma_pe_degree;     /* points_sigma_positional_encoding_degree (:576-578), or -1 when the key is absent */
    int use_view_dirs;
    int view_dependent_rgb;
    int predict_visibility;  /* :582-584, :599-602: views_output_linear gets a 4th row, the per-sample visibility of the point
                                from a view direction (sigmoid).  Built for view_dependent_rgb MLPs, fp32 arithmetic; off in
                                every shipped configuration */
} snerf_mlp_desc;

/* Number of entries of the `params` array below for this descriptor (0 if unsupported):
 *   [2i], [2i+1]  pts_linears.i.weight (out,in), .bias           i = 0 .. points_net_depth-1
 *   then          pts_output_linear.weight, .bias
 *   then, if view_dependent_rgb: feature_linear.weight, .bias; views_linears.0.weight, .bias;
 *                 views_output_linear.weight, .bias
 * i.e. the reference state_dict tensors (MLP.__init__ :586-608), each a device pointer. */
int snerf_mlp_num_params(const snerf_mlp_desc* desc);

/* Size in floats of the packed (MFMA-operand-ordered) weight stream; 0 if the descriptor is unsupported. */
size_t snerf_mlp_packed_floats(const snerf_mlp_desc* desc);

/* Re-order the reference-layout parameters into the packed stream (device -> device).  Call after every
 * parameter update; the stream is what snerf_mlp_forward consumes. */
int snerf_mlp_pack(const snerf_mlp_desc* desc, const float* const* params, int num_params, float* packed,
                   snerf_stream_t stream);

/* snerf_mlp_pack writes EVERY operand format of the weights (fp32 segments, the fp16 hi/lo streams for training and for
 * rendering, their bf16 counterparts): a trainer that re-packs after every optimiser step pays for seven of them and reads
 * two (the fp32 precision: the fp32 segments alone).  This variant writes what calls at ONE precision (enum snerf_precision, below) in ONE mode will read -- training != 0:
 * snerf_mlp_forward_train / snerf_mlp_backward / the render ops with saved activations; training == 0: additionally the
 * rendering layout -- and zeroes the rest: the caller re-packs when it changes precision or mode (the Python model keys its
 * packed streams by both).  No counterpart in the reference (its modules read their parameters directly).
 *
 * Round 5: the library remembers, per packed buffer (by address, host side), which formats its last pack call wrote, and every
 * entry point that reads a weight stream (snerf_mlp_forward[_train|_visibility], snerf_mlp_backward, the render calls) returns
 * SNERF_E_INVALID before enqueuing anything when the buffer does not hold what the call reads -- a stream packed for (f16,
 * training) used by an eval render used to multiply the zero-filled rendering layout and return bias-only output with SNERF_OK.
 * A buffer no pack call of this library instance has written (a device-to-device copy of a packed buffer) is not checked. */
int snerf_mlp_pack_for(const snerf_mlp_desc* desc, const float* const* params, int num_params, float* packed, int precision,
                       int training, snerf_stream_t stream);

enum snerf_precision {
    SNERF_PRECISION_FP32 = 0, /* fp32 MFMA (v_mfma_f32_32x32x2_f32), exact fp32 FMA chains */
    SNERF_PRECISION_F16X3 = 1, /* every operand split into two fp16 (hi + lo, ~22 significand bits), three fp16 MFMAs per
                                  product (hi.hi + hi.lo + lo.hi), fp32 accumulate: fp32-grade results at 3/16 of the
                                  fp32-MFMA time */
    SNERF_PRECISION_F16 = 2,   /* 16-bit mode: one fp16 MFMA per product (11 significand bits per operand), fp32 accumulate,
                                  fp32 master weights, biases, heads and outputs; training keeps activations as fp16 and
                                  layer gradients as bf16 (half the HBM traffic).  NOT within the fp32 parity bar: results
                                  agree with the fp32 path to ~1e-3 (tests/test_gpu_f16.py states the tolerances) */
    SNERF_PRECISION_F16S8 = 4, /* SNERF_PRECISION_F16 with the saved trunk activations h_1 .. h_D-1 kept as fp8 e4m3 (half their bytes
                                  on the way out and back in): they are read back only as the X operand of the weight-gradient
                                  contraction over ~10^6 samples, where a 2^-4 rounding error per element averages out; values above
                                  448 are clamped there.  Rendering, the training forward's arithmetic and the backward chain are
                                  SNERF_PRECISION_F16's; only the weight gradients differ (<= 1e-2 relative L2,
                                  tests/test_gpu_f16.py) */
    SNERF_PRECISION_BF16S8 = 5, /* SNERF_PRECISION_BF16 with the saved trunk activations as fp8 e4m3, as SNERF_PRECISION_F16S8 is to
                                   SNERF_PRECISION_F16: BASELINE config 5's literal dtype with 15 % fewer HBM bytes per iteration */
    SNERF_PRECISION_BF16 = 3   /* the same kernels on bf16 operands (v_mfma_f32_*_bf16, 8 significand bits, fp32's exponent
                                  range): BASELINE config 5's literal dtype.  No range limit -- nothing below under "Range"
                                  applies -- at 3 fewer significand bits than SNERF_PRECISION_F16; saved activations and layer
                                  gradients both bf16.  Own tolerances (tests/test_gpu_bf16.py) */
};
/* Range: both fp16 modes hold hidden activations and weights as fp16 numbers (pairs), so their magnitudes must stay
 * below 65504 -- far above what a NeRF MLP on encoded inputs produces (trained hidden units are O(1..100)).  Exceeding it
 * is DETECTED, not silent: an operand beyond the range converts to +-inf, which makes every pre-activation of the next layer
 * non-finite for that sample, so every fp16-mode forward kernel checks one pre-activation per sample and layer before its
 * ReLU could mask it (one instruction per layer; csrc/mlp_device_f16.h, RangeWatch), and snerf_mlp_pack checks the weights;
 * a launch that met a non-finite operand raises a per-device flag in pinned host memory when it finishes.  Launches are asynchronous, so the
 * flag is reported like an asynchronous HIP error: the NEXT fp16-mode call on that device (snerf_mlp_forward[_train],
 * snerf_mlp_backward, the render ops) fails with SNERF_E_RANGE before enqueuing anything and clears it; after a
 * synchronisation snerf_range_status tells at once.  Outputs of a flagged launch must be discarded (an overflowed unit can
 * be masked by a later ReLU, so they may look finite).  Gradients have no such limit (renormalised by powers of two).  The
 * fp32 mode has the full fp32 range and never raises the flag.
 *   snerf_range_status(clear)   bit 0: an activation / encoded input left the fp16 range; bit 1: a weight did (set by
 *                               snerf_mlp_pack, which cannot know the precision it packs for: only fp16-mode calls report it);
 *                               0 = clean; < 0 = error.  `clear` != 0 resets the flag of the current device.
 * The first fp16-mode call (or snerf_mlp_pack) of a process allocates the 256-byte flag table with hipHostMalloc: make it
 * outside of a graph capture. */
int snerf_range_status(int clear);

/*   origins, dirs   device (num_rays,3): the rays the depths are measured along (NDC rays when ndc)
 *   view_dirs       device (num_rays,3) or NULL when !use_view_dirs
 *   depths          device (num_rays, num_samples)
 *   sigma_noise     device (num_rays, num_samples) added to the raw density before the ReLU (:669-672), or NULL
 *   sigma           device (num_rays, num_samples)      = 'sigma' of MLP.forward, post-ReLU
 *   rgb             device (num_rays, num_samples, 3)   = 'rgb' of MLP.forward, post-sigmoid
 */
int snerf_mlp_forward(const snerf_mlp_desc* desc, const float* packed, const float* origins, const float* dirs,
                      const float* view_dirs, const float* depths, long long num_rays, int num_samples,
                      const float* sigma_noise, float* sigma, float* rgb, int precision, snerf_stream_t stream);

/* ---- training: forward that keeps the per-layer activations, and the parameter-gradient backward (K7) -----------
 * Together they replace what autograd records and replays for MLP.forward (src/models/SimpleNeRF01.py:626-715).
 *   saved_acts   device, snerf_mlp_saved_floats(desc, num_rays, num_samples) floats, written by forward_train and
 *                read by backward ([32-sample block][feature][32] tiles: encodings and every layer's input)
 *   d_sigma (n,S), d_rgb (n,S,3)   device: gradients w.r.t. the `sigma` / `rgb` outputs of the forward
 *   sigma, rgb                      device: those outputs themselves (ReLU / sigmoid derivatives)
 *   workspace    device, snerf_mlp_backward_workspace_floats(...) floats of scratch
 *   param_grads  num_params device pointers, same order and shapes as snerf_mlp_pack's `params`; each tensor is
 *                OVERWRITTEN with dL/dparam when accumulate == 0, or has dL/dparam ADDED to it when accumulate != 0
 *                (what autograd does with the gradients of the trainer's second sub-batch, src/Trainer01.py:82-96, without
 *                a separate add launch per tensor).  Sums over samples are taken in a fixed order: bit-reproducible
 *   precision    SNERF_PRECISION_FP32; SNERF_PRECISION_F16X3 for the fp16-split forward_train / dgrad chain / large
 *                weight-gradient products (the small head and encoding products stay on the fp32 matrix cores); or
 *                SNERF_PRECISION_F16 (16-bit mode).  The layout of saved_acts depends on the precision: pass to
 *                snerf_mlp_backward the precision that snerf_mlp_forward_train was called with (both buffer-size
 *                queries are valid for every precision)
 * Inputs (rays, depths, view directions) receive no gradient -- the reference detaches the sample depths (:312).
 */
size_t snerf_mlp_saved_floats(const snerf_mlp_desc* desc, long long num_rays, int num_samples);
int snerf_mlp_forward_train(const snerf_mlp_desc* desc, const float* packed, const float* origins, const float* dirs,
                            const float* view_dirs, const float* depths, long long num_rays, int num_samples,
                            const float* sigma_noise, float* sigma, float* rgb, float* saved_acts, int precision,
                            snerf_stream_t stream);
size_t snerf_mlp_backward_workspace_floats(const snerf_mlp_desc* desc, long long num_rays, int num_samples);
int snerf_mlp_backward(const snerf_mlp_desc* desc, const float* packed, const float* saved_acts, const float* sigma,
                       const float* rgb, const float* d_sigma, const float* d_rgb, long long num_rays, int num_samples,
                       float* workspace, float* const* param_grads, int num_params, int precision, int accumulate,
                       snerf_stream_t stream);

/* ---- predict_visibility (src/models/SimpleNeRF01.py:317-326, :646-649, :691-714, :479-482) -----------------------------
 * snerf_other_view_dirs: SimpleNeRF.compute_other_view_dirs -- unit directions from each of `num_other` secondary camera
 * centres to every sample point, (n, S, num_other, 3).  With ndc the sample depths are converted to world depths first
 * (near plane hard-coded to 1 as in the reference, :319-321).
 *   depths (n,S); rays_o, rays_d (n,3) WORLD rays; rays_o2 (n, num_other, 3)
 * snerf_mlp_forward_visibility: snerf_mlp_forward[_train] for a predict_visibility MLP; additionally
 *   visibility   (n,S)             'visibility' of MLP.forward for the primary view direction, or NULL
 *   view_dirs2   (n,S,num_other,3) per-sample secondary directions (snerf_other_view_dirs), or NULL with num_other = 0
 *   visibility2  (n,S,num_other)   'visibility2': the views head evaluated again per secondary direction (:646-649)
 *   saved_acts   NULL for inference; else as snerf_mlp_forward_train (the visibility outputs carry no gradient: no shipped
 *                loss reads them, and snerf_mlp_backward writes zeros for the 4th row of views_output_linear)
 * snerf_composite_visibility2: 'visibility2' of volume_rendering (:479-482): sum_s w[n,s] vis2[n,s,k] / (acc[n] + 1e-6).
 */
int snerf_other_view_dirs(const float* depths, const float* rays_o, const float* rays_d, const float* rays_o2,
                          long long num_rays, int num_samples, int num_other, int ndc, float* view_dirs2,
                          snerf_stream_t stream);
int snerf_mlp_forward_visibility(const snerf_mlp_desc* desc, const float* packed, const float* origins, const float* dirs,
                                 const float* view_dirs, const float* depths, long long num_rays, int num_samples,
                                 const float* sigma_noise, const float* view_dirs2, int num_other, float* sigma, float* rgb,
                                 float* visibility, float* visibility2, float* saved_acts, int precision,
                                 snerf_stream_t stream);
int snerf_composite_visibility2(const float* weights, const float* acc, const float* visibility2, long long num_rays,
                                int num_samples, int num_other, float* out, snerf_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * K4  alpha compositing.  Replaces SimpleNeRF.volume_rendering (src/models/SimpleNeRF01.py:430-483) and
 * convert_depth_from_ndc (:486-502).
 *   sigma (n,S), rgb (n,S,3), depths (n,S)      device
 *   march_dirs (n,3)   rays_d, or rays_d_ndc when ndc      rays_o, rays_d (n,3) world rays, used only when ndc
 *   out_rgb (n,3), out_acc (n), out_depth (n), out_depth_var (n)             device, required
 *   out_depth_ndc (n), out_depth_var_ndc (n)                                  device, required when ndc
 *   out_alpha, out_visibility, out_weights (n,S)                              device, each may be NULL
 */
int snerf_composite(const float* sigma, const float* rgb, const float* depths, const float* march_dirs,
                    const float* rays_o, const float* rays_d, long long num_rays, int num_samples, int ndc,
                    int white_bkgd, float* out_rgb, float* out_acc, float* out_alpha, float* out_visibility,
                    float* out_weights, float* out_depth, float* out_depth_var, float* out_depth_ndc,
                    float* out_depth_var_ndc, snerf_stream_t stream);

/* K6  backward of the compositing (autograd of volume_rendering :446-460): per-ray gradients of rgb / acc / depth /
 * depth_ndc (each (n,3) or (n), any may be NULL = zero) -> d_sigma (n,S), d_rgb (n,S,3).  `depth` is the world depth
 * for NDC scenes, as in the forward.  No gradient is produced for depth_var* (no reference loss reads them).
 */
int snerf_composite_backward(const float* sigma, const float* rgb, const float* depths, const float* march_dirs,
                             const float* rays_o, const float* rays_d, long long num_rays, int num_samples, int ndc,
                             int white_bkgd, const float* grad_rgb, const float* grad_acc, const float* grad_depth,
                             const float* grad_depth_ndc, float* d_sigma, float* d_rgb, snerf_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * K5  hierarchical resampling.  Replaces SimpleNeRF.get_z_vals_fine (src/models/SimpleNeRF01.py:304-315) and
 * sample_pdf (:329-361): inverse-CDF samples of the coarse weights merged with the coarse depths, ascending.
 *   depths_coarse, weights_coarse  device (n, num_coarse)
 *   u            device (n, num_fine) uniform draws (:341), or NULL for the deterministic linspace of :338
 *   depths_fine  device (n, num_coarse + num_fine)
 */
int snerf_resample_depths(const float* depths_coarse, const float* weights_coarse, long long num_rays,
                          int num_coarse, int num_fine, const float* u, float* depths_fine, snerf_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * The one-call render ops (SURVEY 8b "Ownership / errors": render_forward / render_backward).
 *
 * snerf_render_forward enqueues on ONE stream everything SimpleNeRF.render_rays does for a batch of rays
 * (src/models/SimpleNeRF01.py:108-270): coarse depths (K2) -> for the main coarse MLP and, when present, the points- and
 * views-augmentation coarse MLPs: fused encoding + MLP (K3) and compositing (K4) -> inverse-CDF resampling of the main
 * coarse weights + merge (K5) -> the same for the fine-level MLPs on the merged depths.  snerf_render_backward enqueues
 * what autograd replays for it: per level the compositing backward (K6) and the MLP parameter-gradient backward (K7).
 * Like every entry point they only enqueue: no allocation, no synchronisation; every buffer is the caller's.
 *
 * Levels (the six MLPs a SimpleNeRF model can hold, :17-41): a NULL `desc` means the level is absent.  Levels 1, 2, 4, 5
 * are the augmentation models, which the reference evaluates in training mode only -- the caller passes them only then.
 */
enum snerf_render_level {
    SNERF_LEVEL_MAIN_COARSE = 0, SNERF_LEVEL_POINTS_AUG_COARSE = 1, SNERF_LEVEL_VIEWS_AUG_COARSE = 2,
    SNERF_LEVEL_MAIN_FINE = 3, SNERF_LEVEL_POINTS_AUG_FINE = 4, SNERF_LEVEL_VIEWS_AUG_FINE = 5
};
#define SNERF_RENDER_LEVELS 6

typedef struct snerf_render_config {
    int ndc;              /* configs['data_loader']['ndc'] */
    int white_bkgd;       /* model.white_bkgd (:462-463) */
    int lindisp;          /* model.lindisp (:286-289) */
    int num_coarse;       /* coarse_mlp.num_samples */
    int num_fine;         /* fine_mlp.num_samples, or 0 when the model has no fine MLP */
    int precision;        /* enum snerf_precision, for every MLP of the call */
    int keep_activations; /* != 0: training forward (snerf_mlp_forward_train); level outputs need `saved_acts` */
    int fused;            /* != 0: an eval-mode render of a plain coarse + fine model (levels 0 and 3 only, fp32, 64+128 or
                             128+128 samples, no noise / visibility / fine-depth override) runs as ONE launch that keeps each
                             ray group's sample tile in LDS from the coarse depths to the fine colour (csrc/render_fused.hip);
                             bit-identical outputs; `sigma` / `raw_rgb` of the two levels may then be NULL (not produced).
                             Any other call takes the stage-by-stage path as if the flag were 0 */
} snerf_render_config;

typedef struct snerf_render_mlp {
    const snerf_mlp_desc* desc; /* NULL = level absent */
    const float* packed;        /* device, snerf_mlp_pack's stream for that descriptor */
} snerf_render_mlp;

typedef struct snerf_render_rays {          /* device pointers, layouts of the reference's input_batch (:113-137) */
    const float *rays_o, *rays_d;           /* (n,3) world rays */
    const float* view_dirs;                 /* (n,3), or NULL when no MLP uses view directions */
    const float *rays_o_ndc, *rays_d_ndc;   /* (n,3), required when ndc */
    const float *near, *far;                /* (n): the columns the depths are spaced between (near_ndc/far_ndc when ndc) */
    const float* t_rand;                    /* (n, num_coarse) stratified-jitter draws (:299) or NULL */
    const float* u;                         /* (n, num_fine) inverse-CDF draws (:341) or NULL (deterministic linspace) */
    const float* sigma_noise[SNERF_RENDER_LEVELS]; /* per level (n, S) density noise, already scaled (:669-672), or NULL */
    const float* depths_fine;               /* (n, num_coarse+num_fine): use these fine depths instead of resampling
                                               (parity-test hook: sample_pdf has a rounding-dependent discontinuity), or NULL */
    const float* rays_o2;                   /* (n, num_other, 3) secondary camera centres ('rays_o2', :120-133) for the
                                               predict_visibility MLPs when sec_views_vis, or NULL */
    int num_other;                          /* num_frames - 1; 0 = no secondary views */
} snerf_render_rays;

typedef struct snerf_render_level_out {     /* S = num_coarse for levels 0-2, num_coarse + num_fine for levels 3-5 */
    float *rgb, *acc, *depth, *depth_var;   /* (n,3), (n), (n), (n): required for a present level */
    float *depth_ndc, *depth_var_ndc;       /* (n): required when ndc */
    float *alpha, *visibility, *weights;    /* (n,S): each may be NULL */
    float *sigma, *raw_rgb;                 /* (n,S), (n,S,3): the MLP outputs ('raw_sigma', 'raw_rgb'), required */
    float* saved_acts;                      /* snerf_mlp_saved_floats(desc, n, S) floats, required when keep_activations */
    float* raw_visibility;                  /* (n,S): predict_visibility levels, may be NULL */
    float* raw_visibility2;                 /* (n,S,num_other): required for such a level when rays.num_other > 0 */
    float* visibility2;                     /* (n,num_other): composited, required with raw_visibility2 */
    float* view_dirs2;                      /* (n,S,num_other,3) scratch, required with raw_visibility2 */
} snerf_render_level_out;

typedef struct snerf_render_outputs {
    float* depths_coarse;                   /* (n, num_coarse) 'z_vals_coarse', required */
    float* depths_fine;                     /* (n, num_coarse+num_fine) 'z_vals_fine'; required when num_fine > 0 and
                                               rays.depths_fine is NULL */
    snerf_render_level_out level[SNERF_RENDER_LEVELS];
} snerf_render_outputs;

/* Scratch of snerf_render_forward in floats (the main coarse weights, when level[0].weights is NULL).
 *
 * Round 5: below 65 536 coarse samples per call (num_rays x num_coarse) and with more than two MLP levels, snerf_render_forward and
 * snerf_render_backward run the levels SIDE BY SIDE on streams of the library's own -- forward {main coarse -> fine} | points-aug
 * | views-aug, backward every level -- forked from `stream` and joined back to it with events before the call returns: the call
 * is still enqueue-only, work enqueued on `stream` afterwards is ordered after all of it, and a graph capture of `stream`
 * records the side streams as parallel branches.  Results are bit-identical to the levels in order.  The side streams and
 * events are created on first use per (device, stream) and kept for the life of the process (at most 64 sets; further caller
 * streams get the levels in order). */
size_t snerf_render_workspace_floats(const snerf_render_config* cfg, long long num_rays);
int snerf_render_forward(const snerf_render_config* cfg, const snerf_render_mlp* mlps, const snerf_render_rays* rays,
                         long long num_rays, const snerf_render_outputs* out, float* workspace, snerf_stream_t stream);

typedef struct snerf_render_level_grads {   /* dL/d(output) of one level, each (shape as the output) or NULL = zero */
    const float *rgb, *acc, *depth, *depth_ndc, *sigma, *raw_rgb;
    float* const* param_grads;              /* the level's parameter gradients, as snerf_mlp_backward; NULL = skip level */
    int num_params;
    int accumulate;                         /* as snerf_mlp_backward */
} snerf_render_level_grads;

/* Scratch of snerf_render_backward in floats: d sigma + d rgb of the largest level + the largest snerf_mlp_backward
 * workspace among the present levels (levels run one after another on the stream and share it); for a call whose levels run
 * side by side (above) the SUM over the levels -- each level its own region.  Always ask this function. */
size_t snerf_render_backward_workspace_floats(const snerf_render_config* cfg, const snerf_render_mlp* mlps, long long num_rays);
/* `out` = the buffers snerf_render_forward (keep_activations != 0) filled for the same cfg / mlps / rays.  A level whose
 * param_grads is NULL, or whose six gradient pointers are all NULL, is skipped (its param_grads are not touched). */
int snerf_render_backward(const snerf_render_config* cfg, const snerf_render_mlp* mlps, const snerf_render_rays* rays,
                          long long num_rays, const snerf_render_outputs* out, const snerf_render_level_grads* grads,
                          float* workspace, snerf_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Output post-processing (next-row f3).  Replaces post_process_image / post_process_depth
 * (src/data_preprocessors/DataPreprocessor01.py:1106-1114) on the device, so a rendered frame crosses PCIe as uint8.
 *   rgb (n,3) device; depth (n) device or NULL;  image (n,3) uint8 device;  depth_out (n) device or NULL
 *   image (and rgb) may be NULL to convert a depth column alone (retrieve_inference_outputs :906-918 converts four).
 * Pinned edge cases (tests/golden/display.npz, made by the reference's functions): colour NaN -> 0, +inf -> 255,
 * exact .5 ties round to even; depth NaN stays NaN, -0.0 stays -0.0, -inf -> 0.
 */
int snerf_to_display(const float* rgb, const float* depth, long long num_rays, unsigned char* image, float* depth_out,
                     snerf_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Opt-in event timing of the dominant kernels (the measurement row, SURVEY 8d: "achieved" of the roofline is measured
 * live with HIP events on the stream the kernel is launched on).  While enabled, every snerf_mlp_forward[_train] launch
 * (kind SNERF_PROFILE_MLP_FORWARD) and every snerf_mlp_backward call (SNERF_PROFILE_MLP_BACKWARD) -- also those issued
 * from inside snerf_render_forward / _backward -- is bracketed by an event pair owned by the library.
 *   snerf_profile_enable(capacity)   create `capacity` event pairs and start recording; capacity <= 0 stops and frees
 *   snerf_profile_collect(kind, ms, samples, capacity)   waits for the recorded events of that kind and writes each
 *        launch's duration in milliseconds and its number of samples (rays x samples); returns how many, or < 0
 *   snerf_profile_reset()            forget the recorded launches (and the dropped count), keep recording
 *   snerf_profile_dropped()          launches since enable / reset that were NOT timed because all `capacity` pairs were in
 *                                    use -- a measurement whose count is not zero under-counts kernel time and FLOPs
 * Process-wide, off by default (one relaxed atomic load on the launch path).  Launches enqueued on a stream that is being
 * captured into a graph are never timed (hipStreamIsCapturing), so the hooks may stay enabled across a capture. */
enum snerf_profile_kind { SNERF_PROFILE_MLP_FORWARD = 0, SNERF_PROFILE_MLP_BACKWARD = 1 };
int snerf_profile_enable(int capacity);
int snerf_profile_collect(int kind, float* milliseconds, long long* samples, int capacity);
int snerf_profile_reset(void);
long long snerf_profile_dropped(void);

#ifdef __cplusplus
}
#endif
#endif /* SIMPLENERF_HIP_H */
