/*
 * topo4d_raster.h — C ABI of the MI355X-native differentiable Gaussian-splatting rasterizer.
 *
 * This is the drop-in boundary for the ONE hot path of Topo4D: the rasterizer behind
 *     im, radius, depth, alpha = Renderer(raster_settings=cam)(**rendervar)
 * (/root/reference train.py:307, :388, :463, :484; imports train.py:19, helpers.py:18-19).
 * Upstream binds that path through a pybind11 module `diff_gaussian_rasterization._C` with the entry points
 * `rasterize_gaussians`, `rasterize_gaussians_backward` and `mark_visible` (not vendored in the reference —
 * README.md:22-24).  The functions below are what a maintainer binds instead (ctypes stub in
 * INTEGRATION.md; the shipped Python host side is topo4d_amd/rasterizer.py):
 *
 *   t4d_rasterize_forward   <->  _C.rasterize_gaussians            (called from train.py:307/388/463/484)
 *   t4d_rasterize_backward  <->  _C.rasterize_gaussians_backward   (reached from loss.backward(), train.py:667/738)
 *   t4d_mark_visible        <->  _C.mark_visible                   (GaussianRasterizer.markVisible; unused by Topo4D)
 *   t4d_state_bytes / t4d_backward_scratch_bytes  <->  upstream's resize-callback buffers (geom/binning/img)
 *
 * Conventions
 *   - plain C: raw DEVICE pointers (fp32 unless noted), sizes, and a hipStream_t passed as void*.
 *     No torch types.  The library never allocates or frees device memory and never synchronises the
 *     stream unless T4D_FLAG_CHECKED / T4D_FLAG_DEBUG_SYNC asks for it.
 *   - one call renders n_views views of the SAME P Gaussians (the 24 cameras of a Topo4D frame, or one rank's
 *     shard of them); n_views = 1 is the reference's call shape.  All views share H and W.
 *     (T4DProblem.views_per_param_set lets one call carry the views of SEVERAL frames, each frame with its own P Gaussians.)
 *   - per-view camera record = T4D_VIEW_FLOATS floats on the device, built from the fields of
 *     GaussianRasterizationSettings (helpers.py:73-86):
 *        [0..15]  viewmatrix  — the 16 floats of the [1,4,4] tensor helpers.py:67 builds (transposed w2c ⇒
 *                 element (row r, col c) of the mathematical matrix sits at [c*4+r])
 *        [16..31] projmatrix  — same layout (helpers.py:71-72)
 *        [32..34] campos      [35..37] bg      [38] tanfovx      [39] tanfovy
 *   - outputs are planar, one image after another: color [V,3,H,W], depth [V,1,H,W], alpha [V,1,H,W],
 *     radii int32 [V,P] — for V = 1 exactly the four tensors train.py:307 unpacks.
 *   - gradients are per view: dL_dX has a leading V dimension; the caller sums over views if it wants the
 *     gradient of a multi-view loss (or passes T4D_FLAG_... none: summation is not done here).
 *   - every function returns T4D_OK (0) or an error code; nothing throws or exits across the ABI.
 */
#ifndef TOPO4D_RASTER_H
#define TOPO4D_RASTER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define T4D_ABI_VERSION 4
#define T4D_VIEW_FLOATS 40
#define T4D_GRAD_PAIR_FLOATS 10   /* per (Gaussian,tile) partial-gradient record in the backward scratch */

enum {
    T4D_OK = 0,
    T4D_ERR_ARG = 1,            /* bad argument combination (mirrors the upstream Python-level checks) */
    T4D_ERR_HIP = 2,            /* a HIP runtime call failed; see t4d_last_error() */
    T4D_ERR_PAIR_OVERFLOW = 3,  /* pair_capacity too small (CHECKED mode): status->max_pairs_per_view says how much */
    T4D_ERR_STATE_SIZE = 4      /* state/scratch buffer smaller than t4d_*_bytes() */
};

enum {
    T4D_FLAG_CHECKED = 1u,     /* forward: sync once after binning sizes are known and fail with PAIR_OVERFLOW
                                  instead of rendering truncated tile lists (what upstream's num_rendered D2H does) */
    T4D_FLAG_DEBUG_SYNC = 2u,  /* `debug=True` of the settings tuple: synchronise + check after every kernel */
    T4D_FLAG_PREFILTERED = 4u, /* accepted for API parity (helpers.py:84 passes False); no effect */
    T4D_FLAG_ASYNC_STATUS = 8u,/* forward, without CHECKED: `status` must point at PINNED (device-mapped) host memory; the first
                                  16 bytes receive { uint32 overflow; uint32 max_pairs_per_view; uint64 total_pairs } without
                                  any host synchronisation: written by the binning kernel itself (one view of at most 1,024
                                  tiles; the two 8-byte words may land one after the other) or by an asynchronous copy
                                  enqueued behind the binning kernels */
    T4D_FLAG_NO_LONG_BINS = 16u,/* forward: the caller knows (T4DStatus.max_tile_pairs of an earlier call on this scene) that no
                                  tile list exceeds 2048 pairs: the launch of the long-bin sort kernel is skipped, and a one-view
                                  launch of more than 8,192 tiles keeps no snapshots for a depth-segmented backward of its long
                                  tiles.  Only a speed hint — longer bins that show up anyway are still sorted and replayed
                                  correctly, just slowly */
    T4D_FLAG_SHORT_BINS = 32u, /* forward: ... and that none exceeds 512 pairs (the one-pass ranking sort): a small launch then
                                  sorts every bin inside the render workgroup of its tile instead of launching a sort kernel.
                                  A speed hint like the one above */
    T4D_FLAG_LONG_LISTS = 64u, /* forward: the caller knows that some tile list exceeds 1,024 pairs: launches of up to 24 x CUs
                                  tiles (instead of 12 x) run the latency build of the forward, which such a list bounds.
                                  A speed hint */
    T4D_FLAG_RAW_PARAMS = 128u /* forward AND backward of a call: `rotations`, `opacities`, `scales` are Topo4D's optimiser
                                  parameters - un-normalised quaternions, logit opacities, log scales (helpers.py:95-97) - the
                                  library applies F.normalize / sigmoid / exp itself (the arithmetic of t4d_activate_forward), and
                                  the backward returns dL/d(unnorm_rotations), dL/d(logit_opacities), dL/d(log_scales) in
                                  dL_drotations, dL_dopacities, dL_dscales (per view; the arithmetic of t4d_activate_backward):
                                  params2rendervar and its autograd without a launch of their own */
};

typedef struct T4DProblem {
    int32_t abi_version;     /* = T4D_ABI_VERSION */
    int32_t n_views;         /* V */
    int32_t P;               /* Gaussians */
    int32_t H, W;            /* image_height, image_width */
    int32_t sh_degree;       /* active SH degree (settings.sh_degree); used only when shs != NULL */
    int32_t sh_coeffs;       /* M: coefficients stored per Gaussian in shs [P,M,3] */
    float   scale_modifier;
    int64_t pair_capacity;   /* capacity, PER VIEW, of the (Gaussian,tile) pair arena */
    uint32_t flags;
    uint32_t views_per_param_set; /* 0 (or n_views): every view renders the same P Gaussians - one frame per call.
                                     k > 0: the call carries n_views / k PARAMETER SETS (frames) of P Gaussians each, view v renders
                                     set v / k; every per-Gaussian INPUT array (means3D, opacities, scales, rotations,
                                     cov3D_precomp, colors_precomp, shs) then holds the sets one after the other, [n_views / k][P, .].
                                     n_views must be a multiple of k.  Outputs and gradients keep their per-view layout ([V, ...]),
                                     so nothing else changes: a rank of a view-sharded job (BASELINE config 3: 3 of 24 cameras per
                                     rank) renders its cameras of several independent frames in ONE launch set, at the efficiency
                                     of a 24-view launch.  Results per view are those of the same view in a one-frame call of the
                                     same launch shape */
} T4DProblem;

typedef struct T4DStatus {
    int64_t max_pairs_per_view;  /* largest per-view number of (Gaussian,tile) pairs this call needed */
    int64_t total_pairs;         /* sum over views (upstream's num_rendered, summed) */
    int32_t overflow;            /* 1 if any view exceeded pair_capacity (its tile lists were truncated) */
    int32_t max_tile_pairs;      /* longest per-tile list of the call (dense passes reach > 10^4; see T4D_FLAG_NO_LONG_BINS) */
} T4DStatus;

typedef struct T4DForwardIO {
    const float *views;           /* [V][T4D_VIEW_FLOATS] */
    const float *means3D;         /* [P,3] */
    const float *opacities;       /* [P] (the [P,1] tensor of helpers.py:96) */
    const float *scales;          /* [P,3] or NULL when cov3D_precomp */
    const float *rotations;       /* [P,4] (r,x,y,z), caller-normalised (helpers.py:95), or NULL */
    const float *cov3D_precomp;   /* [P,6] or NULL */
    const float *colors_precomp;  /* [P,3] or NULL when shs */
    const float *shs;             /* [P,M,3] or NULL */
    float   *out_color;           /* [V,3,H,W] */
    float   *out_depth;           /* [V,1,H,W] */
    float   *out_alpha;           /* [V,1,H,W] */
    int32_t *out_radii;           /* [V,P] */
    void    *state;               /* opaque, t4d_state_bytes(); must stay alive and unmodified until backward */
    size_t   state_bytes;
} T4DForwardIO;

typedef struct T4DBackwardIO {
    const float *views, *means3D, *opacities, *scales, *rotations, *cov3D_precomp, *colors_precomp, *shs;
    const int32_t *radii;         /* the [V,P] forward output */
    const void  *state;           /* the forward's state buffer */
    size_t       state_bytes;
    const float *dL_dcolor;       /* [V,3,H,W] */
    const float *dL_ddepth;       /* [V,1,H,W] or NULL (= zeros; Topo4D discards depth, train.py:307) */
    const float *dL_dalpha;       /* [V,1,H,W] or NULL */
    float *dL_dmeans3D;           /* [V,P,3] */
    float *dL_dmeans2D;           /* [V,P,3] (z = 0): the screen-space gradient kept alive by train.py:304 */
    float *dL_dcolors;            /* [V,P,3]   or NULL (required when colors_precomp) */
    float *dL_dshs;               /* [V,P,M,3] or NULL (required when shs) */
    float *dL_dopacities;         /* [V,P] */
    float *dL_dscales;            /* [V,P,3]   or NULL (required unless cov3D_precomp) */
    float *dL_drotations;         /* [V,P,4]   or NULL (required unless cov3D_precomp) */
    float *dL_dcov3D;             /* [V,P,6]   or NULL (required when cov3D_precomp) */
    void  *scratch;               /* t4d_backward_scratch_bytes() */
    size_t scratch_bytes;
    float *cotangent_dot;         /* [V] or NULL.  out[v] = <color, dL_dcolor> + <depth, dL_ddepth> + <alpha, dL_dalpha> of view v:
                                   * the replay's suffix sum IS this inner product when it reaches the eye, so the backward
                                   * emits it for one reduction per tile instead of a second pass over both images
                                   * (deterministic; agrees with the sum over the forward's outputs to fp32 rounding) */
} T4DBackwardIO;

uint32_t    t4d_abi_version(void);
const char *t4d_last_error(void);

size_t t4d_state_bytes(const T4DProblem *prob);
size_t t4d_backward_scratch_bytes(const T4DProblem *prob);

/* Forward: preprocess -> per-tile binning -> per-tile depth sort -> alpha blend.  `status` may be NULL; it is
 * filled only in CHECKED / DEBUG_SYNC mode (otherwise use t4d_fetch_status after the stream has drained). */
int t4d_rasterize_forward(const T4DProblem *prob, const T4DForwardIO *io, T4DStatus *status, void *hip_stream);

/* Backward: per-tile back-to-front replay (atomic-free, deterministic) -> per-Gaussian gather + chain rule.
 * If the forward that produced `state` overflowed its pair arena (only possible without T4D_FLAG_CHECKED), every
 * gradient of this call is written as ZERO: a truncated pass never yields garbage. */
int t4d_rasterize_backward(const T4DProblem *prob, const T4DBackwardIO *io, void *hip_stream);

/* Copies the status block of a forward's state buffer to the host (synchronises the stream). */
int t4d_fetch_status(const T4DProblem *prob, const void *state, T4DStatus *out, void *hip_stream);

/* present[i] = 1 iff Gaussian i passes the near-plane test of view record `view` (upstream markVisible). */
int t4d_mark_visible(int32_t P, const float *means3D, const float *view, uint8_t *present, void *hip_stream);

/* out[v] = sum_i a[v][i] * b[v][i] for v < n_views (one fused pass, deterministic): per-view scalars for a multi-GPU loss
 * gather.  (For sum(colour * dL/dcolour) itself, T4DBackwardIO.cotangent_dot is free; bench.py uses that.) */
size_t t4d_view_dot_scratch_bytes(int32_t n_views);
int t4d_view_dot(int32_t n_views, int64_t n_per_view, const float *a, const float *b, float *out, void *scratch,
                 void *hip_stream);

/* The gradient of a loss that ADDS per-view terms (rasterize_views: one launch set per frame, one optimiser step per frame):
 * dst[k][i] = sum over v < n_views of src[k][v * n_per_view[k] + i], v ascending (deterministic), for up to T4D_SUM_MAX_TENSORS
 * per-view gradient tensors of t4d_rasterize_backward in ONE launch (six torch reductions otherwise).  NULL entries are skipped. */
#define T4D_SUM_MAX_TENSORS 8
int t4d_sum_views(int32_t n_views, int32_t n_tensors, const float *const *src, float *const *dst, const int64_t *n_per_view,
                  void *hip_stream);

/* Fused photometric loss of Topo4D's render loop, forward AND gradient in one call (train.py:310,315;
 * helpers.py:115-116 l1_loss_v1; external.py:73-116 calc_ssim):
 *     im' = exp(cam_m[v,c]) * im + cam_c[v,c];   loss[v] = 0.8*mean|im'-gt| + 0.2*(1 - mean SSIM_11x11(im', gt))
 * im, gt, dL_dim: [V,3,H,W]; cam_m, cam_c, dL_dcam_m, dL_dcam_c: [V,3] (both NULL = no affine / no gradient wanted);
 * view_weight [V] = dL/dloss[v] (NULL = 1).  dL_dim is exactly the dL_dcolor input of t4d_rasterize_backward. */
size_t t4d_photometric_scratch_bytes(int32_t n_views, int32_t H, int32_t W);
int t4d_photometric_loss(int32_t n_views, int32_t H, int32_t W, const float *im, const float *gt, const float *cam_m,
                         const float *cam_c, const float *view_weight, float *loss, float *dL_dim, float *dL_dcam_m,
                         float *dL_dcam_c, void *scratch, size_t scratch_bytes, void *hip_stream);

/* Masked L1 of the dense (texture) pass, train.py:394-405 (get_loss_dense with use_mask=True): no camera affine, no SSIM,
 *     loss[v] = sum over { mask == 1 } of |im - gt|  /  count{ mask == 1 },     dL_dim = view_weight[v] * sign(im - gt) / count there, 0 elsewhere.
 * im, gt, mask, dL_dim: [V,3,H,W]; mask is the float image of zeros and ones helpers.get_mask (helpers.py:811-823) returns
 * (the three channels carry the same plane; they are all counted, like `masked_index.sum()`).  An empty mask gives NaN, as
 * in the reference. */
size_t t4d_masked_l1_scratch_bytes(int32_t n_views);
int t4d_masked_l1_loss(int32_t n_views, int32_t H, int32_t W, const float *im, const float *gt, const float *mask,
                       const float *view_weight, float *loss, float *dL_dim, void *scratch, size_t scratch_bytes, void *hip_stream);

/* helpers.get_mask (helpers.py:811-823) and the masked target of get_loss's later-frame branch (train.py:320-326; train.py:631,647
 * hard-code use_mask = True, so this is what every frame after the first optimises against), for the n_views cameras of a
 * frame in ONE launch - the reference recomputes both in every iteration although they depend on (frame, camera) only.
 *     hit(pixel)    = OR over the n_labels selected labels of  AND over channels c of  | mask_image[c]*255 - label_colors[k][c] | < 1
 *     filtered_mask = 1 where hit else 0, on all three channels           (get_mask's return value; NULL = not wanted)
 *     target        = gt * scale where hit else gt                         (masked_gt, scale = 0.1 at train.py:326; NULL = not wanted)
 * mask_image, gt, filtered_mask, target: [V,3,H,W] device floats; mask_image is the label image as get_dataset loads it
 * (train.py:84-92: 8-bit colours / 255).  label_colors: HOST array [n_labels,3] of the selected labels' colours as floats, in the
 * channel order of the mask image (helpers.py:806 `cmap`: the pascal colormap of 14 labels with its columns swapped to BGR),
 * n_labels <= T4D_MAX_MASK_LABELS.  Arithmetic is the reference's, rounding for rounding (float32 product, then float32
 * difference): filtered_mask and target are bit-identical to torch's. */
#define T4D_MAX_MASK_LABELS 16
int t4d_label_mask_target(int32_t n_views, int32_t H, int32_t W, const float *mask_image, const float *label_colors /* host */,
                          int32_t n_labels, const float *gt, float scale, float *filtered_mask, float *target, void *hip_stream);

/* The 'soft_color' term of get_loss_dense (train.py:407; weight 0.02, train.py:541-543): helpers.l1_loss_v2 (helpers.py:119-120)
 *     *loss = mean over rows of ( sum over width of |x - y| )                                   (UNWEIGHTED, device scalar)
 *     grad[i] (+)= (weight / rows) * sign(x[i] - y[i])     (sign(0) = 0 as torch's abs; accumulate != 0: added to what grad holds -
 *                                                           e.g. t4d_rasterize_backward's dL_dcolors; grad NULL = no gradient)
 * x = dense_rgb_colors, y = dense_init_colors: [rows,width] device floats.  Deterministic (fixed partial-sum order). */
size_t t4d_soft_color_scratch_bytes(void);
int t4d_soft_color_loss(int64_t rows, int32_t width, const float *x, const float *y, float weight, float *loss, float *grad,
                        int32_t accumulate, void *scratch, size_t scratch_bytes, void *hip_stream);

/* Fused optimiser step of Topo4D's loop: torch.optim.Adam (one group per tensor, train.py:272-297) for up to
 * T4D_ADAM_MAX_TENSORS tensors in ONE launch, followed by the per-iteration region freezes of train.py:676-700
 * (`params[name][mask] = values`) expressed as a per-row pin mask + pinned values.  grad == NULL: the tensor only gets its
 * pins (torch skips parameters without a gradient). */
#define T4D_ADAM_MAX_TENSORS 12
typedef struct T4DAdamTensor {
    float *param;               /* [rows, width] updated in place */
    const float *grad;          /* same shape or NULL */
    float *exp_avg, *exp_avg_sq;/* Adam state, same shape (required when grad != NULL) */
    const uint8_t *pin_mask;    /* [rows] or NULL */
    const float *pin_values;    /* [rows, width] (read where pin_mask != 0) or NULL */
    int64_t rows;
    int32_t width;
    float lr;
    int32_t step;               /* 1-based count of the gradient steps THIS tensor has taken, this one included (torch keeps
                                   the step per parameter; a tensor skipped for lack of a gradient does not advance) */
    int32_t flags;              /* T4D_ADAM_CLEAR_GRAD: the step leaves zeros in `grad` (a persistent gradient buffer that the next
                                   iteration fills only in part - the per-camera rows of cam_m / cam_c, train.py:310 - needs no
                                   separate fill launch); 0 otherwise */
} T4DAdamTensor;
#define T4D_ADAM_CLEAR_GRAD 1
int t4d_adam_pin_step(const T4DAdamTensor *tensors /* host array */, int32_t n_tensors, float beta1, float beta2, float eps,
                      void *hip_stream);
/* The same step with its hyper-parameters in DEVICE memory, so that the launch can be recorded in a HIP graph and replayed - ONE
 * launch: lr_dev [n_tensors] float learning rates; step_dev [t4d_adam_step_counters(tensors, n)] int32 step counts, one PER WORKGROUP
 * of the launch (workgroups of 256 elements, tensor after tensor in descriptor order: tensor k owns ceil(rows_k * width_k / 256)
 * consecutive counters, all equal - read any of them).  A workgroup of a tensor that has a gradient advances its own counter, then
 * uses it for the bias corrections (a tensor skipped for lack of a gradient does not advance).  The `step` and `lr` fields of the
 * descriptors are ignored.  Bias corrections are evaluated in double precision on the device: results equal t4d_adam_pin_step's.
 * The descriptors' shapes must be the same in every call that shares a step_dev array: n_step_counters - the length of step_dev -
 * is checked against t4d_adam_step_counters(tensors, n_tensors) (T4D_ERR_ARG otherwise; nothing is launched). */
int64_t t4d_adam_step_counters(const T4DAdamTensor *tensors /* host array */, int32_t n_tensors);
int t4d_adam_pin_step_graph(const T4DAdamTensor *tensors /* host array */, int32_t n_tensors, float beta1, float beta2, float eps,
                            int32_t *step_dev, int64_t n_step_counters, const float *lr_dev, void *hip_stream);

/* Dense-attribute interpolation: helpers.py:237-253 `compute_vertex_attribute_by_weight_2` on the device (Topo4D runs it in
 * numpy after a device->host copy every frame, train.py:504-506).  out [n_coarse+n_dense, width]: the first n_coarse rows
 * copy `attribute` [n_coarse, width]; dense row d = sum_k attribute[quad_faces[vertex_father[d]][k]] * weight[d][k], k < 4,
 * accumulated in float64 like numpy and rounded to float32 once (bit-identical to the reference followed by `.float()`). */
int t4d_dense_interpolate(const float *attribute, const int32_t *quad_faces /* [n_quads,4] */,
                          const int32_t *vertex_father /* [n_dense] */, const double *weight /* [n_dense,4] */,
                          int64_t n_coarse, int64_t n_dense, int32_t width, float *out, void *hip_stream);

/* Parameter activations of params2rendervar (helpers.py:91-100: rotations = F.normalize(unnorm_rotations), opacities =
 * sigmoid(logit_opacities), scales = exp(log_scales)) in one launch, and their vector-Jacobian products in one launch
 * (torch runs three kernels forward and about a dozen through autograd backward, every iteration: SURVEY.md row a2).
 * All pointers are device pointers; rotations [P,4], opacities [P,1], scales [P,3].  In the backward a NULL cotangent
 * counts as zeros and a NULL output is skipped; `opacities` / `scales` are the forward OUTPUTS. */
int t4d_activate_forward(int64_t P, const float *unnorm_rotations, const float *logit_opacities, const float *log_scales,
                         float *rotations, float *opacities, float *scales, void *hip_stream);
int t4d_activate_backward(int64_t P, const float *unnorm_rotations, const float *opacities, const float *scales,
                          const float *dL_drotations, const float *dL_dopacities, const float *dL_dscales,
                          float *dL_dunnorm_rotations, float *dL_dlogit_opacities, float *dL_dlog_scales, void *hip_stream);

/* UV-space texture bake (BASELINE config 5): drop-in for the reference's CPU rasterizer
 *     void _render_colors_core(float* image, float* vertices, int* triangles, float* colors, float* depth_buffer,
 *                              int nver, int ntri, int h, int w, int c)        face3d/mesh/cython/mesh_core.h:63-69
 * reached from helpers.py:953-960 (write_texture) via face3d/mesh/render.py:52-86 (render_colors).  Same argument
 * meaning (all pointers are DEVICE pointers here): vertices [nver,3] in pixel space (x, y, depth), triangles [ntri,3]
 * int32, colors [nver,c], image [h,w,c] in/out (zeros or a background), depth_buffer [h,w] in/out (the caller fills it
 * with -999999 like render.py:72).  Results are bit-identical to the reference, including its border-ring dilation
 * (mesh_core.cpp:211) and "first triangle wins on equal depth".  rows [row_begin,row_end) select a horizontal band of
 * the image (multi-GPU: one band per rank); pass 0,h for everything.  pair_capacity bounds the (triangle, 32x32 tile)
 * pairs; on T4D_ERR_PAIR_OVERFLOW *pairs_needed (host) holds the size to retry with.  Synchronises the stream once. */
size_t t4d_texture_bake_scratch_bytes(int32_t h, int32_t w, int64_t pair_capacity);
int t4d_texture_bake(const float *vertices, const int32_t *triangles, const float *colors, int32_t nver, int32_t ntri,
                     int32_t h, int32_t w, int32_t c, int32_t row_begin, int32_t row_end, float *image, float *depth_buffer,
                     void *scratch, size_t scratch_bytes, int64_t pair_capacity, int64_t *pairs_needed, void *hip_stream);

/* The reference's `render_colors(vertices, triangles, colors, h, w, c, BG)` as ONE call (face3d/mesh/render.py:52-86: image =
 * BG or zeros, depth_buffer = -999999, then _render_colors_core): `image` and `depth_buffer` are pure OUTPUTS here - every
 * texel of rows [row_begin,row_end) is written, winner or background (`background` [h,w,c] or NULL = zeros) - so the caller
 * fills nothing and the kernel reads no depth buffer (8192^2: 1.07 GB of fills and 0.27 GB of reads less than
 * t4d_texture_bake).  Same results bit for bit, same scratch (t4d_texture_bake_scratch_bytes) and overflow protocol. */
int t4d_texture_render_colors(const float *vertices, const int32_t *triangles, const float *colors, const float *background,
                              int32_t nver, int32_t ntri, int32_t h, int32_t w, int32_t c, int32_t row_begin, int32_t row_end,
                              float *image, float *depth_buffer, void *scratch, size_t scratch_bytes, int64_t pair_capacity,
                              int64_t *pairs_needed, void *hip_stream);

/* Optional per-kernel timing with HIP events recorded on the stream the kernels are launched on.  Between
 * t4d_profile_begin() and t4d_profile_end() every kernel launch of this library is bracketed by two events;
 * t4d_profile_end() synchronises them and returns, per kernel, the summed elapsed time and the launch count.
 * bench.py uses this for the `roofline` object (duration of the dominant kernel).  Not thread-safe. */
typedef struct T4DKernelTime {
    const char *name;
    double total_ms;
    int64_t launches;
} T4DKernelTime;
int t4d_profile_begin(void);
int t4d_profile_end(T4DKernelTime *out, int max_entries, int *n_entries);

/* Test/debug only: byte offsets of the arrays inside a state buffer, in this order:
 *   status, view_total, view_cursor, tile_count, bucket_fill, tile_off, xy, depth, conic_opacity, rgb, clamped,
 *   pair_off, keys, final_T, n_contrib, total_bytes.
 * The layout is NOT part of the stable ABI; tests use it to check the integer state (tile bins, sort order,
 * n_contrib) bit-for-bit against the oracle. */
#define T4D_DEBUG_LAYOUT_FIELDS 16
int t4d_debug_state_layout(const T4DProblem *prob, int has_sh, uint64_t *offsets, int n);

#ifdef __cplusplus
}
#endif
#endif /* TOPO4D_RASTER_H */
