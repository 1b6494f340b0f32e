/*
 * include/gsvc_hip.h — C-ABI of libgsvc_hip.so, the MI355X (gfx950) hot path of GSVC.
 *
 * Drop-in boundary: these entry points are what a GSVC maintainer binds in place of the two native
 * extensions the renderer path imports (INTEGRATION.md shows the ctypes stubs):
 *
 *   diff_gaussian_rasterization.cuda_ortho_gaussian_rasterizer   (external pip dependency, reference README.md:52)
 *       GaussianRasterizer.visible_filter   call site reference ortho_gaussian_renderer/preprocess.py:99-104
 *       GaussianRasterizer.forward          call site reference ortho_gaussian_renderer/renderer.py:90-98
 *       GaussianRasterizer.backward         implicit via autograd, reference pipeline/train.py:462
 *   _gridencoder                                                  (reference submodules/gridencoder.zip)
 *       grid_encode_forward / grid_encode_backward   gridencoder.zip!gridencoder/src/gridencoder.h:12-36,
 *                                                    call sites reference utils/encodings.py:529-553,582-610
 *   utils/entropy_models.py EntropyGaussian (+Low_bound)           reference utils/entropy_models.py:32-68,159-175
 *
 * TWO KINDS OF ENTRY POINTS.  A maintainer who swaps GSVC's native extensions for this library binds the FIFTEEN marked
 * [BOUNDARY] — they have the shape of the calls GSVC makes today — and can ignore the rest:
 *
 *   rasterizer   gsvc_raster_sizes_query, gsvc_raster_visible_filter, gsvc_raster_forward, gsvc_raster_backward_scratch_bytes,
 *                gsvc_raster_backward   (+ gsvc_raster_forward_pair: the decoder's two-view frame of report_utils.py:297-319 in one pass)
 *   hash grid    gsvc_grid_forward, gsvc_grid_backward                      (= grid_encode_forward / grid_encode_backward)
 *   rate         gsvc_rate_forward, gsvc_rate_backward                      (= EntropyGaussian.forward + Low_bound's backward)
 *   codec        gsvc_ans_segments, gsvc_ans_scratch_bytes, gsvc_ans_encode, gsvc_ans_decode_scratch_bytes, gsvc_ans_decode
 *                                                                             (= gsvc_cuda_ans.ANSCoder behind encoder_gaussian / decoder_gaussian)
 *   init         gsvc_knn3_mean_dist2                                         (= simple_knn._C.distCUDA2)
 *   always       gsvc_last_error, gsvc_version
 *
 * Everything else is [INTERNAL]: fusion entries that gsvc_amd's own host code (gsvc_amd/ *.py) calls to run the fitting step and
 * the decoder loop in ~170 launches instead of ~2 000 — batched / un-compacted forms of the same arithmetic (several views, whole
 * MLPs, the step's loss terms, the optimizer, the step plan), each with a parity test against the tensor expression or reference
 * function it replaces (cited at the declaration).  They are exported because the host side is Python over ctypes; their
 * signatures follow gsvc_amd's needs, not GSVC's API, and may change between rounds.  INTEGRATION.md section 1 has the same map
 * with the reference call site of every [BOUNDARY] entry.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory unless its name ends in _host;
 *   - the caller owns every buffer (outputs, scratch blobs); nothing is allocated or freed inside, nothing
 *     synchronises the device, so every call may be captured into a hipGraph;
 *   - every call is enqueued on `stream` (a hipStream_t passed as void*);
 *   - return value: 0 = ok, negative = GSVC_E_*; gsvc_last_error() gives the message of the calling
 *     thread's last failure;
 *   - float = IEEE binary32, row-major, contiguous.
 */
#ifndef GSVC_HIP_H
#define GSVC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSVC_OK 0
#define GSVC_E_INVALID (-1)   /* bad argument (null pointer, negative size, unsupported shape) */
#define GSVC_E_LAUNCH (-2)    /* HIP reported a launch error */
#define GSVC_E_UNSUPPORTED (-3)

const char *gsvc_last_error(void);
/* "gsvc_hip <version> gfx950" */
const char *gsvc_version(void);

/* Per-kernel timing for bench.py (not a drop-in entry point).  While enabled, every kernel launch made by
 * this library is bracketed by hipEvents on its own stream.  gsvc_profile_collect synchronises those events
 * and returns, per kernel name, the launch count and total milliseconds since the last collect/enable.
 * names_out: `max_kernels` slots of 64 bytes.  Returns the number of kernels written (<0 on error).
 * `on` is a bit set: 1 = the per-kernel timing; 2 = the compositing backward runs its diagnostic instantiation and adds, per
 * launch, three uint64 to bytes 64..87 of the render's counters block (the start of the binning blob): (entry, quadrant) replays,
 * lanes of those replays that held a contributing pixel, list entries replayed — bench.py's roofline_valu reads them. */
int gsvc_profile_enable(int on);
int gsvc_profile_collect(char *names_out_host, int32_t *launches_out_host, float *total_ms_out_host, int max_kernels);

/* ------------------------------------------------------------------------------------------------------
 * Orthographic sliding-window rasterizer   [BOUNDARY; gsvc_raster_visible_masks and the two *_layout queries are INTERNAL]
 * ---------------------------------------------------------------------------------------------------- */

/* Fields of GaussianRasterizationSettings the extension reads (reference renderer.py:63-83).
 * sh_degree / campos / prefiltered / debug carry no information for the colours_precomp path and are
 * not part of the ABI.  viewmatrix is the LOGICAL 4x4 the reference passes (`frame.view_matrix.permute(1,0)`),
 * row-major: p_view = M[:3,:3] p + M[:3,3]. */
typedef struct gsvc_raster_settings {
    int32_t image_height;
    int32_t image_width;
    float x_min;
    float y_min;
    float scale;
    float threshold;
    float scale_modifier;
    float bg[3];
    float viewmatrix[16];
    uint32_t flags;   /* GSVC_RASTER_* convention switches below; 0 = the conventions of DESIGN.md "Raster spec" */
    float low_pass;   /* added to the diagonal of the 2-D covariance, pixels^2; 0 = the default 0.3 (reference
                         arguments/__init__.py:55); GSVC_RASTER_NO_LOW_PASS for none */
} gsvc_raster_settings;

/* Convention switches.  The external CUDA rasterizer's source is not part of the reference tree (README.md:52), so the
 * conventions its call sites do not pin are kept switchable; INTEGRATION.md "Calibrating the rasterizer conventions" shows
 * how to find the combination that reproduces the real extension on one small scene.  Every combination is covered by the
 * oracle <-> dense-float64 cross-check and by HIP <-> oracle parity (tests/test_oracle_raster.py, tests/test_raster_gpu.py). */
#define GSVC_RASTER_SLAB_ONE_SIDED 1u        /* keep -threshold <= z_view <= 0 instead of |z_view| <= threshold (preprocess.py:108-112) */
#define GSVC_RASTER_PIXEL_CORNER 2u          /* u = (x_view - x_min) * scale, without the -0.5 that puts pixel centres on integers */
#define GSVC_RASTER_DEPTH_DESCENDING 4u      /* composite from the largest view-space z to the smallest (ties still by index) */
#define GSVC_RASTER_MEANS2D_PIXEL_UNITS 8u   /* dL_dmeans2D = (dL/du, dL/dv) in pixels instead of NDC units (x W/2, x H/2) */
#define GSVC_RASTER_CLAMP_STOPS_GRADIENT 16u /* no gradient to conic / mean / opacity where the alpha <= 0.99 clamp is active */
#define GSVC_RASTER_NO_LOW_PASS 32u          /* low_pass = 0 exactly */
#define GSVC_RASTER_TIGHT_BINNING 64u       /* a Gaussian is LISTED only in the tiles of its 3-sigma rectangle that its alpha >= 1/255 box
                                               touches (the other instances can contribute nothing: the compositing kernels skip them
                                               anyway): same image, radii, num_rendered (still the 3-sigma count) and gradients (up to
                                               the order large rectangles' rows are added in), shorter lists — late in a fit 40 % of the
                                               3-sigma instances are such.  The pair forward clamps both views' rectangles by the (shared) box */

/* Byte sizes of the three opaque state blobs of one forward call (the 3DGS-lineage "geomBuffer /
 * binningBuffer / imgBuffer" the reference extension hands back to autograd). */
typedef struct gsvc_raster_sizes {
    uint64_t geom_bytes;
    uint64_t binning_bytes;
    uint64_t image_bytes;
} gsvc_raster_sizes;

/* counters written by forward into the first 32 bytes of the binning blob */
typedef struct gsvc_raster_counters {
    int32_t num_rendered; /* sum of the 3-sigma rectangles' tiles (true count even when it overflowed) = the lists' length, except under
                             GSVC_RASTER_TIGHT_BINNING, where the lists are shorter: their length is tile_offsets[T] (gsvc_raster_binning_layout) */
    int32_t overflow;     /* 1 when the lists' length (reserved[2]; = num_rendered except under GSVC_RASTER_TIGHT_BINNING) > max_instances:
                             image/state are NOT valid, retry with max_instances >= reserved[2] */
    int32_t num_visible;  /* Gaussians with radius > 0 */
    int32_t max_tile_len; /* longest per-tile list */
    int32_t num_big_tiles; /* tiles whose list is sorted by the workgroup kernel (internal) */
    int32_t reserved[3];  /* [0], [1]: internal cursors; [2]: after the forward, the tile lists' length = tile_offsets[T] (true length even
                             when it overflowed): point_list / inst_bbox / the backward's scratch rows are sized by THIS, not by num_rendered */
} gsvc_raster_counters;

int gsvc_raster_sizes_query(const gsvc_raster_settings *settings, int64_t P, int64_t max_instances,
                            gsvc_raster_sizes *sizes_host);

/* radii[P] (int32): >0 iff the Gaussian's 3-sigma ellipse intersects the z-slab and the screen. */
int gsvc_raster_visible_filter(const gsvc_raster_settings *settings, int64_t P, const float *means3D,
                               const float *scales, const float *rotations, int32_t *radii, void *stream);
/* The same test for 1..8 views over the same P Gaussians in one launch: masks[r P + i] = (radius under settings[r] > 0).
 * scales: rows scale_stride floats apart, the first three read; scale_exp != 0: they are logarithms (reference
 * scene/gaussian_model.py scaling_activation = exp); rot_normalise != 0: rotations are un-normalised quaternions (rotation_activation
 * = normalize, eps 1e-12) — the per-anchor activations of reference ortho_gaussian_renderer/preprocess.py:88-104 folded in. */
int gsvc_raster_visible_masks(const gsvc_raster_settings *const *settings, int32_t views, int64_t P, const float *means3D,
                              const float *scales, int32_t scale_stride, int32_t scale_exp, const float *rotations,
                              int32_t rot_normalise, uint8_t *masks, void *stream);

/* Forward: means3D[P,3] colors[P,3] opacities[P] scales[P,3] rotations[P,4]  ->  image[3,H,W], radii[P].
 * State blobs sized by gsvc_raster_sizes_query(settings, P, max_instances).  The counters struct sits at
 * byte 0 of `binning`; copy it to the host when num_rendered is needed (the reference API returns it as a
 * python int, reference renderer.py:90). */
int gsvc_raster_forward(const gsvc_raster_settings *settings, int64_t P, int64_t max_instances,
                        const float *means3D, const float *colors, const float *opacities, const float *scales,
                        const float *rotations, float *image, int32_t *radii, void *geom, void *binning,
                        void *image_state, void *stream);

/* Two-view frame in one pass (no backward): image_pair[3,H,W] = (render(view) + flip_W(render(opposite view))) / 2,
 * the frame GSVC's evaluation / decoder output (reference utils/report_utils.py:297-319): the opposite view
 * (frame.view_matrix_s) composites the same Gaussians at the mirrored pixels in reverse depth order, so one binning
 * and one walk of the tile lists produce both composites.  Same arguments as gsvc_raster_forward with `settings`
 * holding the forward view; final_T / n_contrib are not produced.  Agrees with two gsvc_raster_forward calls to
 * < 2e-4 per pixel (the opposite view's own early exit is not replayed). */
int gsvc_raster_forward_pair(const gsvc_raster_settings *settings, int64_t P, int64_t max_instances,
                             const float *means3D, const float *colors, const float *opacities, const float *scales,
                             const float *rotations, float *image_pair, int32_t *radii, void *geom, void *binning,
                             void *image_state, void *stream);

/* Backward: dL_dimage[3,H,W] + the forward's inputs and state  ->  gradients (all overwritten):
 * dL_dmeans3D[P,3], dL_dmeans2D[P,3] (screen-space gradient in NDC units, the tensor GSVC's densification
 * reads, reference scene/gaussian_model.py:1311), dL_dcolors[P,3], dL_dopacities[P], dL_dscales[P,3],
 * dL_drotations[P,4].  `scratch`: gsvc_raster_backward_scratch_bytes(P, max_instances) bytes (one 64-byte row of
 * partial sums per (tile, Gaussian) instance; every row that is read is written first, nothing needs zeroing).
 * No float atomics: two calls on the same inputs return bit-identical gradients. */
int64_t gsvc_raster_backward_scratch_bytes(int64_t P, int64_t max_instances);
int gsvc_raster_backward(const gsvc_raster_settings *settings, int64_t P, int64_t max_instances,
                         const float *means3D, const float *colors, const float *opacities, const float *scales,
                         const float *rotations, const int32_t *radii, const void *geom, const void *binning,
                         const void *image_state, const float *dL_dimage, float *dL_dmeans3D, float *dL_dmeans2D,
                         float *dL_dcolors, float *dL_dopacities, float *dL_dscales, float *dL_drotations,
                         void *scratch, void *stream);

/* Test/inspection helpers: locate the sorted per-tile lists inside the binning blob.
 * tile_offsets_host_out: byte offset of int32 tile_offsets[T+1]; point_list: byte offset of int32 ids. */
int gsvc_raster_binning_layout(const gsvc_raster_settings *settings, int64_t P, int64_t max_instances,
                               uint64_t *tile_offsets_byte_off_host, uint64_t *point_list_byte_off_host);
int gsvc_raster_image_layout(const gsvc_raster_settings *settings, uint64_t *final_T_byte_off_host,
                             uint64_t *n_contrib_byte_off_host);

/* ------------------------------------------------------------------------------------------------------
 * Multi-resolution hash grid (replaces _gridencoder.grid_encode_forward / grid_encode_backward)   [BOUNDARY; the _ex / _packed
 * forms and gsvc_pack_sign_bits are INTERNAL]
 * ---------------------------------------------------------------------------------------------------- */

/* inputs[N,D] in [0,1]; embeddings[rows,C]; offsets[L+1], resolutions[L] (int32, device, already sliced to
 * the levels to compute); outputs[L,N,C]; dy_dx[N, L*D*C] or NULL.
 * D in {1,2,3}, C in {1,2,4,8,16,32} (else GSVC_E_UNSUPPORTED with the reference's message). */
int gsvc_grid_forward(const float *inputs, const float *embeddings, const int32_t *offsets,
                      const int32_t *resolutions, float *outputs, uint32_t N, uint32_t D, uint32_t C, uint32_t L,
                      float *dy_dx, void *stream);

/* grad[L,N,C]; grad_embeddings[rows,C] is ACCUMULATED into (caller zero-fills, reference encodings.py:574);
 * grad_inputs[N,D] is overwritten when dy_dx != NULL. */
int gsvc_grid_backward(const float *grad, const float *inputs, const float *embeddings, const int32_t *offsets,
                       const int32_t *resolutions, float *grad_embeddings, uint32_t N, uint32_t D, uint32_t C,
                       uint32_t L, const float *dy_dx, float *grad_inputs, void *stream);

/* The same kernels with the caller's layout (no input-gradient path): the points' coordinates are columns in_col[0..D) of a
 * row-major [N, in_stride] matrix, the features of (level l, point b) are the C floats at l * feat_level_stride +
 * b * feat_point_stride of `outputs` / `grad` (strides in floats, multiples of min(C, 4)).  Several grids over the same
 * positions — Mix3d2dEncoding, reference scene/gaussian_model.py:81-147 — write their column blocks of ONE [N, sum L C] matrix
 * (feat_level_stride = C, feat_point_stride = its row length) and read the gradient the same way: no slices, permutes or cat. */
typedef struct gsvc_grid_io {
    int64_t feat_level_stride, feat_point_stride;
    int32_t in_stride, in_col[3];
} gsvc_grid_io;
int gsvc_grid_forward_ex(const float *inputs, const float *embeddings, const int32_t *offsets, const int32_t *resolutions,
                         float *outputs, uint32_t N, uint32_t D, uint32_t C, uint32_t L, const gsvc_grid_io *layout, void *stream);
int gsvc_grid_backward_ex(const float *grad, const float *inputs, const int32_t *offsets, const int32_t *resolutions,
                          float *grad_embeddings, uint32_t N, uint32_t D, uint32_t C, uint32_t L, const gsvc_grid_io *layout,
                          void *stream);

/* Several grids over the SAME points in one launch each way (Mix3d2dEncoding: one 3-D + three 2-D grids, reference
 * scene/gaussian_model.py:81-147 calls the four encoders one after the other): `features` is the grid's column block of the shared
 * [N, sum L C] matrix (forward: written; backward: the gradient, read), `layout` as for gsvc_grid_forward_ex.  Same arithmetic per
 * (grid, level, point) as the single-grid entries — the tables' gradients are added into grad_embeddings (zero them first); 1..4
 * grids of 2 or 3 dimensions, C in {2, 4, 8} features per level for all of them. */
typedef struct gsvc_grid_many_job {
    const float *embeddings;        /* forward: the table; backward: unused (may be NULL) */
    float *features;                /* forward: output block; backward: gradient block (read only) */
    float *grad_embeddings;         /* backward: the table's gradient (accumulated) */
    const int32_t *offsets, *resolutions;
    uint32_t D, L;
    gsvc_grid_io layout;
} gsvc_grid_many_job;
int gsvc_grid_forward_many(const float *inputs, const gsvc_grid_many_job *jobs, int32_t n_jobs, uint32_t N, uint32_t C, void *stream);
int gsvc_grid_backward_many(const float *inputs, const gsvc_grid_many_job *jobs, int32_t n_jobs, uint32_t N, uint32_t C, void *stream);

/* Lookup in BIT-PACKED binarised tables (8 features per row: one byte per row, bit f set = +1, clear = -1 — the {-1, +1} tables of
 * reference utils/encodings.py:375-392 STE_binary as the bitstream carries them: 1/32 of the float tables).  Same interpolation,
 * same results bit for bit as gsvc_grid_forward_ex on the float tables; layout may be NULL (default layout).  Inference only.
 * gsvc_pack_sign_bits: bits[row] from a float table x [rows, 8] (x >= 0 -> 1). */
int gsvc_pack_sign_bits(const float *x, int64_t rows, uint8_t *bits, void *stream);
int gsvc_grid_forward_packed(const float *inputs, const uint8_t *table_bits, const int32_t *offsets, const int32_t *resolutions,
                             float *outputs, uint32_t N, uint32_t D, uint32_t L, const gsvc_grid_io *layout, void *stream);

/* ------------------------------------------------------------------------------------------------------
 * Entropy-rate estimator (replaces utils/entropy_models.py EntropyGaussian.forward + Low_bound backward)   [BOUNDARY: gsvc_rate_forward /
 * _backward; the sampled-rate entries below them are INTERNAL]
 * ---------------------------------------------------------------------------------------------------- */

/* bits[n,c] = -log2(max(Phi((x+Q/2-mu)/sigma) - Phi((x-Q/2-mu)/sigma), 2^-16)) with x clamped to
 * [x_lo, x_hi] first (the caller computes x_mean -/+ 15000*mean(Q), reference entropy_models.py:40-47;
 * NULL disables; device scalars, or one pair per row [n] when bounds_per_row != 0 — used when several renders
 * are batched into one call).  x, mean, scale: [n,c];  Q: [n] (per row) or NULL with Q_scalar.
 * row_weight: [n,c] multiplier folded into the sum (the offsets mask) or NULL.
 * Outputs: bits[n,c] (may be NULL) and bits_sum[1] (double precision accumulate, float result, ADDED to). */
int gsvc_rate_forward(const float *x, const float *mean, const float *scale, const float *Q, float Q_scalar,
                      const float *weight, const float *x_lo, const float *x_hi, int32_t bounds_per_row, int64_t n,
                      int64_t c, float *bits, float *bits_sum, void *stream);

/* d(sum(weight*bits)*gscale)/d{x, mean, scale, Q[n], weight}; any output pointer may be NULL.
 * Low_bound rule: gradient passes only where the likelihood >= 2^-16 (reference entropy_models.py:166-175,
 * net effect). gscale_dev: device scalar multiplying every gradient (dL/d bits_sum). dQ[n] is accumulated. */
int gsvc_rate_backward(const float *x, const float *mean, const float *scale, const float *Q, float Q_scalar,
                       const float *weight, const float *x_lo, const float *x_hi, int32_t bounds_per_row, int64_t n,
                       int64_t c, const float *gscale_dev, float *dx, float *dmean, float *dscale, float *dQ, float *dweight,
                       void *stream);

/* Sampled rate of the R (<= 16) renders of a fitting step in one pass (reference ortho_gaussian_renderer/guassian.py:73-132,
 * evaluated per render there): the n_sel selected rows (sel: row of x / Q / mask; sel_ctx: row of mean / scale, NULL = sel)
 * are gathered, clamped to x_mean[g] (a device array) -+ 15000 * (mean of Q[g] over the selected rows of the row's render), priced as in
 * gsvc_rate_forward and summed per render and group: S[r][g], g = 0 feature, 1 scaling, 2 offsets (weighted by
 * mask[row][col / 3] when mask != NULL).  row_bounds[renders + 1] (host): row offsets of the renders.
 * scratch: gsvc_rate_sample_scratch_floats(n_sel) floats; the forward leaves the per-render step statistics there and the
 * backward reads them.  Backward: gS[r][g] = dL/dS; dx[g] / dQ[g] rows are STORED (selected rows are distinct), dmean[g] /
 * dscale[g] / dmask are ADDED (rows of the context may repeat): all of them dense, zero-filled by the caller. */
typedef struct gsvc_rate_sample {
    const float *x[3], *mean[3], *scale[3], *Q[3];
    const float *mask;
    const int64_t *sel, *sel_ctx, *row_bounds;
    const float *x_mean;            /* device, 3 floats */
    int32_t C[3];
    int32_t K, renders;
    int64_t n_sel;
} gsvc_rate_sample;
int64_t gsvc_rate_sample_scratch_floats(int64_t n_sel);
int gsvc_rate_sample_forward(const gsvc_rate_sample *desc, float *scratch, float *S, void *stream);
int gsvc_rate_sample_backward(const gsvc_rate_sample *desc, const float *scratch, const float *gS, float *const *dx,
                              float *const *dmean, float *const *dscale, float *const *dQ, float *dmask, void *stream);

/* Normalisation of the sampled rate (reference gaussian_renderer/guassian.py:110-132 per render): out[r] = {all, features,
 * scalings, offsets} bits per coded parameter = S[r][g] / (n_sel_r dims[g]) x keep rate_r, keep rate = the render's rows whose K
 * offset masks (offset_masks [rows, K]) sum to more than 0 / its rows; n_sel_r = entries of the sorted row list sel inside
 * [row_bounds[r], row_bounds[r+1]).  coef [R, 4] (saved for the backward) = d out[r][i] / d S; sum[0] = sum_r out[r][0].
 * backward: gS [R, 3] from g_out [R, 4] and / or g_sum [1] (either may be NULL). */
int64_t gsvc_rate_normalise_scratch_bytes(void);
int gsvc_rate_normalise_forward(const float *S, const float *offset_masks, int32_t K, const int64_t *sel, int64_t n_sel,
                                const int64_t *row_bounds_host, int32_t R, const float *dims3_host, void *scratch, float *out,
                                float *coef, float *sum, void *stream);
int gsvc_rate_normalise_backward(const float *coef, const float *g_out, const float *g_sum, int32_t R, float *gS, void *stream);

/* Densification statistics of un-compacted renders (reference scene/gaussian_model.py:1281-1314 training_statis): row i of
 * the concatenated renders is anchor vis[i] with K Gaussians; opacity_accum[a] += sum_k max(opacity, 0), anchor_denom[a] += 1,
 * and where seen[i K + k]: grad_accum[a K + k] += |grad[i K + k][0:2]|, denom[a K + k] += 1 (grad rows grad_stride floats apart). */
int gsvc_training_statis(const int64_t *vis, const float *opacity, const uint8_t *seen, const float *grad, int32_t grad_stride,
                         int64_t rows, int32_t K, float *opacity_accum, float *anchor_denom, float *grad_accum, float *denom,
                         void *stream);

/* ------------------------------------------------------------------------------------------------------
 * Image distortion of the fitting step (replaces utils/loss_utils.py l1_loss_func + ssim_func and their autograd)   [INTERNAL]
 * ---------------------------------------------------------------------------------------------------- */

/* img1, img2: [C,H,W].  workspace: 2048 floats of scratch.  sums[2] (device): sums[0] = sum of the SSIM map (11x11 Gaussian window,
 * sigma 1.5, zero padding, C1=.01^2, C2=.03^2, reference loss_utils.py:52-72), sums[1] = sum |img1-img2|
 * (:20-21); divide by C*H*W for the reference's means.  dm_*: [C,H,W] partial maps kept for the backward, or
 * all three NULL when no gradient is needed. */
int gsvc_ssim_l1_forward(const float *img1, const float *img2, int32_t C, int32_t H, int32_t W, float *sums,
                         float *workspace, float *dm_dmu1, float *dm_de11, float *dm_de12, void *stream);

/* dL_dimg1[C,H,W] = grads[0] * d(mean SSIM)/d img1 + grads[1] * d(mean |img1-img2|)/d img1; grads: device [2]. */
int gsvc_ssim_l1_backward(const float *img1, const float *img2, int32_t C, int32_t H, int32_t W, const float *grads,
                          const float *dm_dmu1, const float *dm_de11, const float *dm_de12, float *dL_dimg1,
                          void *stream);

/* The same with the first image formed on load as the two-view frame 0.5 * (img_f + flip_W(img_b)) (reference
 * pipeline/train.py:368-375); avg_out (may be NULL) receives that frame; the backward writes both views' gradients. */
int gsvc_ssim_l1_pair_forward(const float *img_f, const float *img_b, const float *img2, int32_t C, int32_t H, int32_t W,
                              float *sums, float *workspace, float *dm_dmu1, float *dm_de11, float *dm_de12, float *avg_out,
                              void *stream);
int gsvc_ssim_l1_pair_backward(const float *img_f, const float *img_b, const float *img2, int32_t C, int32_t H, int32_t W,
                               const float *grads, const float *dm_dmu1, const float *dm_de11, const float *dm_de12,
                               float *dL_dimg_f, float *dL_dimg_b, void *stream);

/* STE_binary forward (reference utils/encodings.py:375-392: y = x >= 0 ? +1 : -1) and count[0] = number of +1 entries
 * (the hash-bit term of the loss, pipeline/train.py:456, needs it); n < 2^24. */
int gsvc_ste_binary_count(const float *x, int64_t n, float *y, float *count, void *stream);
/* The same for up to 8 tables in one launch (x[k], y[k]: n[k] floats; counts[k]); its straight-through backward
 * grad_x[k][i] = |x[k][i]| <= 1 ? grad_y[k][i] + 0.5 count_grads[k count_grad_stride] : 0 (grad_y[k] or count_grads may be NULL = zero: a table
 * entry's share of its count is 1/2); and the Bernoulli code length of the tables from their counts (reference
 * utils/encodings.py:34-51 get_binary_vxl_size over all tables, pipeline/train.py:456): out[0] = bits, out[1] = d bits / d count. */
int gsvc_ste_binary_count_many(const float *const *x, float *const *y, const int64_t *n, int32_t tables, float *counts, void *stream);
int gsvc_ste_binary_backward_many(const float *const *x, const float *const *grad_y, const int64_t *n, int32_t tables,
                                  const float *count_grads, int32_t count_grad_stride, float *const *grad_x, void *stream);
int gsvc_table_bits(const float *counts, int32_t tables, int64_t total, float *out, void *stream);

/* ------------------------------------------------------------------------------------------------------
 * [INTERNAL] Per-Gaussian loss terms of the fitting step over UN-COMPACTED renders (every visible anchor contributes its K
 * Gaussians; mask[i] = opacity_i > 0)
 * ---------------------------------------------------------------------------------------------------- */

/* Optical-flow consistency of one adjacent-frame pair (reference utils/loss_utils.py:76-135,
 * calc_optical_loss_one_frame): over the Gaussians alive in both renders (same anchor, same offset slot) whose
 * frame-t pixel round((xy1 - (x_min, y_min)) * scale) lies inside [0, x_pix_max) x [0, y_pix_max):
 *   sums[0] = sum |(xy2 - xy1) - flow[:, py, px] / scale|_1,  sums[1] = number of such Gaussians;  loss = sums[0] / (2 sums[1]).
 * world{1,2}: [n,3] positions before the bound clamp; vis{1,2}: int64 anchor index of each row of K Gaussians;
 * flow: [2, flow_h, flow_w].  Scratch (caller-allocated): table int32[anchors*K], partner int32[n1] (kept for backward),
 * partial float[2*ceil(n1/256)] (per-workgroup sums; a second small kernel adds them — no atomics, deterministic). */
int gsvc_optical_forward(const float *world1, const uint8_t *mask1, const int64_t *vis1, int64_t n1, const float *world2,
                         const uint8_t *mask2, const int64_t *vis2, int64_t n2, int32_t K, int64_t anchors, const float *flow,
                         int32_t flow_h, int32_t flow_w, float x_min, float y_min, float scale, int32_t x_pix_max,
                         int32_t y_pix_max, int32_t *table, int32_t *partner, float *sums, float *partial, void *stream);
/* grad_world{1,2}[n,3] = d(loss)/d(world) * grad_out[0] (device scalar); both are overwritten. */
int gsvc_optical_backward(const int32_t *partner, int64_t n1, int64_t n2, const float *sums, const float *grad_out,
                          float *grad_world1, float *grad_world2, void *stream);

/* The same loss summed over P disjoint pairs of renders that are row ranges of ONE set of tensors (a fitting step's renders
 * f1, b1, f2, b2 concatenated, pairs f1 -> f2 and b1 -> b2: reference utils/loss_utils.py:138-153 calc_optical_loss).  world [N, 3],
 * mask [N], vis [N / K] cover all R renders, gaussian_offsets_host [R + 1]; pair p takes its frame-t Gaussians from render
 * pair_src_host[p] and their partners from pair_dst_host[p].  loss[0] = sum_p sums[2p] / (2 sums[2p + 1]).  Scratch: table
 * int32 [R, anchors] (anchor -> row + 1, kept for the backward), partner int32 [N], sums float [2 P], partial float
 * [gsvc_optical_many_partial_floats].  backward: grad_world [N, 3] is written completely (zeros outside the pairs). */
int64_t gsvc_optical_many_partial_floats(const int64_t *gaussian_offsets_host, int32_t R, const int32_t *pair_src_host,
                                         const int32_t *pair_dst_host, int32_t P, int32_t K);
int gsvc_optical_many_forward(const float *world, const uint8_t *mask, const int64_t *vis, const int64_t *gaussian_offsets_host, int32_t R,
                              const int32_t *pair_src_host, const int32_t *pair_dst_host, int32_t P, int32_t K, int64_t anchors,
                              const float *flow, int32_t flow_h, int32_t flow_w, float x_min, float y_min, float scale,
                              int32_t x_pix_max, int32_t y_pix_max, int32_t *table, int32_t *partner, float *sums, float *partial,
                              float *loss, void *stream);
int gsvc_optical_many_backward(const int32_t *partner, const int64_t *vis, const int64_t *gaussian_offsets_host, int32_t R,
                               const int32_t *pair_src_host, const int32_t *pair_dst_host, int32_t P, int32_t K, int64_t anchors,
                               const int32_t *table, const float *sums, const float *grad_out, float *grad_world, void *stream);

/* Regularisers of the R renders of one step (reference pipeline/train.py:417-424) over their concatenated Gaussians
 * (render r = [seg_offsets[r], seg_offsets[r+1]), host array of R+1 entries, R <= 8):
 *   out[0] = sum_r mean_{i in r, mask_i}(scaling_i0 scaling_i1 scaling_i2),  out[1] = sum_r mean_{i in r}(1 - neural_opacity_i).
 * sums: float[3R] kept for backward; partial: float[gsvc_regs_partial_floats()] scratch (per-workgroup sums). */
int64_t gsvc_regs_partial_floats(const int64_t *seg_offsets_host, int32_t R);
int gsvc_regs_forward(const float *scaling, const float *neural_opacity, const uint8_t *mask, const int64_t *seg_offsets_host,
                      int32_t R, float *sums, float *partial, float *out, void *stream);
int gsvc_regs_backward(const float *scaling, const uint8_t *mask, const int64_t *seg_offsets_host, int32_t R, const float *sums,
                       const float *grad_out, float *grad_scaling, float *grad_opacity, void *stream);

/* Training-time noise quantisation (reference utils/encodings.py:395-409 UniformQuantizer) of x[rows, C] for R renders
 * at once (render r = rows [row_offsets[r], row_offsets[r+1]), host array, R <= 8):
 *   y = clamp(x / Q, c_r - 15000, c_r + 15000) * Q + noise * Q,   c_r = mean(x over render r) / mean(Q over render r),
 * Q = q_rows[row] (per-row step, device) or q_scalar when q_rows is NULL; noise[rows, C] ~ U(-1/2, 1/2) is drawn by the
 * caller.  centre: float[R] (kept for backward); scratch: float[gsvc_noise_quant_scratch_floats()]. */
int64_t gsvc_noise_quant_scratch_floats(const int64_t *row_offsets_host, int32_t R);
int gsvc_noise_quant_forward(const float *x, const float *q_rows, float q_scalar, const float *noise,
                             const int64_t *row_offsets_host, int32_t R, int32_t C, float *scratch, float *centre, float *y,
                             void *stream);
/* grad_x[rows, C] and grad_q_rows[rows] (may be NULL) are overwritten; the centre carries no gradient (C <= 256). */
int gsvc_noise_quant_backward(const float *grad_y, const float *x, const float *q_rows, float q_scalar, const float *noise,
                              const float *centre, const int64_t *row_offsets_host, int32_t R, int32_t C, float *grad_x,
                              float *grad_q_rows, void *stream);

/* STE_multistep over the rows of R renders (reference utils/encodings.py:395-420 with input_mean given, as ortho_gaussian_renderer/
 * guassian.py:205-207 calls it): y = x1 + (round(x1 / Q) Q - x1), x1 = clamp(x / Q, trunc(c_r - 15000), trunc(c_r + 15000)) Q,
 * c_r = x_mean / (mean of Q over render r's rows); x_mean: one float on the device (the whole parameter's mean).  No backward: the
 * caller detaches the result, as the reference does.  scratch: gsvc_noise_quant_scratch_floats floats; bounds: 2 R floats (out). */
int gsvc_ste_quant_forward(const float *x, const float *q_rows, float q_scalar, const float *x_mean, const int64_t *row_offsets_host,
                           int32_t R, int32_t C, float *scratch, float *bounds, float *y, void *stream);

/* Tail of the anchor -> neural-Gaussian generation for un-compacted renders (reference
 * ortho_gaussian_renderer/guassian.py:262-296: opacity mask, sigmoid scaling, normalised rotation, world position, bound
 * clamp), n = rows*K Gaussians, Gaussian i belongs to anchor row i / K.  Inputs: opacity_raw[n], offset_mask[n],
 * grid_offsets[n,3], neural_offset[n,3], scale_rot[n,7], grid_scaling[rows,6], anchor[rows,3]; bounds: 3 host floats each.
 * Outputs: neural_opacity[n], mask[n] (opacity > 0), scaling[n,3], rot[n,4], world[n,3], xyz[n,3]. */
int gsvc_gen_tail_forward(const float *opacity_raw, const float *offset_mask, const float *grid_offsets,
                          const float *neural_offset, const float *scale_rot, const float *grid_scaling, const float *anchor,
                          const float *bound_min3_host, const float *bound_max3_host, int64_t rows, int32_t K,
                          float *neural_opacity, uint8_t *mask, float *scaling, float *rot, float *world, float *xyz, void *stream);
/* Incoming gradients g_* may be NULL (output unused).  d_offsets is the gradient of both grid_offsets and neural_offset;
 * d_anchor may be NULL.  All d_* are overwritten. */
int gsvc_gen_tail_backward(const float *opacity_raw, const float *offset_mask, const float *grid_offsets,
                           const float *neural_offset, const float *scale_rot, const float *grid_scaling, const float *world,
                           const float *bound_min3_host, const float *bound_max3_host, int64_t rows, int32_t K,
                           const float *g_neural_opacity, const float *g_scaling, const float *g_rot, const float *g_world,
                           const float *g_xyz, float *d_opacity_raw, float *d_offset_mask, float *d_offsets, float *d_scale_rot,
                           float *d_grid_scaling, float *d_anchor, void *stream);

/* ------------------------------------------------------------------------------------------------------
 * Optimizer (reference scene/gaussian_model.py:1034-1058: torch.optim.Adam(eps=1e-15), 15 parameter groups)   [INTERNAL]
 * ---------------------------------------------------------------------------------------------------- */
typedef struct gsvc_adam_tensor {
    float *param;             /* updated in place */
    const float *grad;
    float *exp_avg;           /* first / second moment, updated in place */
    float *exp_avg_sq;
    int64_t n;                /* elements */
    float lr;
    float bias_correction1;   /* 1 - beta1^t, t = number of updates of this tensor including this one */
    float bias_correction2;   /* 1 - beta2^t */
} gsvc_adam_tensor;

/* One Adam update (no weight decay, no amsgrad) of every tensor in the host array, in one launch per 64 tensors. */
int gsvc_adam_step(int32_t n_tensors, const gsvc_adam_tensor *tensors_host, double beta1, double beta2, double eps, void *stream);
/* The same update unless one of the (<= 4) device words guards_host[k] points to is non-zero when the launch runs: then nothing is
 * written.  A fitting step that updates some tensors before it has read the rasterizer's overflow words back (so that the next
 * step's visibility test can be queued early) passes those words: a step that has to be repeated leaves its parameters alone. */
int gsvc_adam_step_guarded(int32_t n_tensors, const gsvc_adam_tensor *tensors_host, double beta1, double beta2, double eps,
                           const int32_t *const *guards_host, int32_t n_guards, void *stream);

/* ------------------------------------------------------------------------------------------------------
 * [BOUNDARY] Model creation: mean squared distance to the 3 nearest neighbours (replaces
 * simple_knn._C.distCUDA2, reference simple-knn.zip simple_knn.cu:185-220, call sites scene/gaussian_model.py:762,784).
 * The caller bins the points into a uniform grid: points_by_cell[n,3] sorted by cell index ((z * gy + y) * gx + x, cell of
 * a point = floor((p - origin) / cell_edge) clamped to the grid), cell_start[gx*gy*gz + 1].  out[n] in the sorted order.
 * ---------------------------------------------------------------------------------------------------- */
int gsvc_knn3_mean_dist2(const float *points_by_cell, const int32_t *cell_start, const float *origin3_host, float cell_edge,
                         int32_t gx, int32_t gy, int32_t gz, int64_t n, float *out, void *stream);

/* ------------------------------------------------------------------------------------------------------
 * [BOUNDARY; gsvc_ans_decode_many and gsvc_ans_table_checksum are INTERNAL] Entropy coding of quantised attributes with the learned
 * Gaussian model (replaces the external
 * gsvc_cuda_ans.ANSCoder the reference calls through utils/encodings.py:102-245 encoder_gaussian / decoder_gaussian:
 * integer symbols in [min_symbol, max_symbol], one Normal(mu, sigma) per symbol in symbol units).
 * rANS, 20-bit frequencies from the model itself (no tables), independent segments of seg_len symbols; the byte layout
 * is this library's own (the reference's coder is not in its tree): segment k = bytes [seg_offsets[k], seg_offsets[k+1]).
 * ---------------------------------------------------------------------------------------------------- */

/* number of segments / bytes of encode scratch for n symbols */
int64_t gsvc_ans_segments(int64_t n, int32_t seg_len);
/* checksum of the fixed-point Phi table encoder and decoder build their frequencies from: a stream records it, a decoder
 * whose own table differs (another platform's erfc rounded an entry the other way) must refuse the stream */
uint32_t gsvc_ans_table_checksum(void);
int64_t gsvc_ans_scratch_bytes(int64_t n, int32_t seg_len);

/* symbols[n] (device, int32) -> out (device; capacity gsvc_ans_scratch_bytes is always enough), seg_bytes[segments]
 * (uint32), seg_offsets[segments + 1] (uint64, seg_offsets[segments] = total bytes).  error_flag (device int32, zeroed by
 * the caller) becomes non-zero if a symbol lies outside [min_symbol, max_symbol]. */
int gsvc_ans_encode(const int32_t *symbols, const float *mu, const float *sigma, int64_t n, int32_t min_symbol,
                    int32_t max_symbol, int32_t seg_len, void *scratch, uint32_t *seg_bytes, uint64_t *seg_offsets, uint8_t *out,
                    int32_t *error_flag, void *stream);

/* inverse: bytes + seg_offsets[segments + 1] + the same mu, sigma -> symbols[n].  error_flag non-zero: corrupt stream.
 * `bytes` must be readable up to 16 bytes past seg_offsets[segments] (the kernel reads the stream in aligned 16-byte pieces).
 * `scratch`: gsvc_ans_decode_scratch_bytes(n, seg_len) bytes, 16-byte aligned: the part of the model that does not depend on
 * the coder state (the cumulative frequencies of every symbol's mode) is computed by a chip-wide parallel kernel first and laid
 * out so that the serial per-segment lanes of the decode kernel read it as one coalesced stream. */
int64_t gsvc_ans_decode_scratch_bytes(int64_t n, int32_t seg_len);
int gsvc_ans_decode(const uint8_t *bytes, const uint64_t *seg_offsets, const float *mu, const float *sigma, int64_t n,
                    int32_t min_symbol, int32_t max_symbol, int32_t seg_len, int32_t *symbols, int32_t *error_flag, void *scratch,
                    void *stream);

/* Per-anchor parameters of the batch's rows (reference ortho_gaussian_renderer/guassian.py:160-176 + the getters of
 * scene/gaussian_model.py:641-700), v = vis[row]: feat = anchor_feat[v] [rows,F], offsets = offset[v] [rows,3K], scaling =
 * exp(scaling_p[v]) [rows,S], mask = straight-through (sigmoid(mask_p[v]) > 0.01) [rows,K]; decoded != 0: the stored values as
 * they are.  F, K or S may be 0 (that group is left out: its pointers are not read).  Backward ADDS into the dense, caller-zeroed
 * d_* (an anchor is visible in several renders); g_* may be NULL. */
int gsvc_gather_rows_forward(const float *feat_p, const float *offset_p, const float *scaling_p, const float *mask_p,
                             const int64_t *vis, int64_t rows, int32_t F, int32_t K, int32_t S, int32_t decoded, float *feat,
                             float *offsets, float *scaling, float *mask, void *stream);
int gsvc_gather_rows_backward(const float *scaling_p, const float *mask_p, const int64_t *vis, int64_t rows, int32_t F, int32_t K,
                              int32_t S, int32_t decoded, const float *g_feat, const float *g_offsets, const float *g_scaling,
                              const float *g_mask, float *d_feat, float *d_offset, float *d_scaling, float *d_mask, void *stream);
/* The same gradients without atomics (and without a zero fill: every element of the d_* tensors is written), for rows that
 * are the concatenation of R ascending index lists over A anchors: seen [R*A] (1 = view r holds anchor a), rank [R*A] =
 * inclusive scan of seen (rank - 1 = the row), R <= 8.  Sums in view order: deterministic. */
int gsvc_gather_rows_backward_ranked(const float *scaling_p, const float *mask_p, const uint8_t *seen, const int64_t *rank, int32_t R,
                                     int64_t A, int32_t F, int32_t K, int32_t S, int32_t decoded, const float *g_feat,
                                     const float *g_offsets, const float *g_scaling, const float *g_mask, float *d_feat,
                                     float *d_offset, float *d_scaling, float *d_mask, void *stream);

/* Masks of a fitting step's plan in one pass over the A anchors: visible_host[r] = view r's visibility mask (bytes, 0 / 1),
 * M [R*A] = the masks side by side, present [A] = some view holds the anchor, chosen [R*A] (may be NULL: then mask_raw / u are
 * not read) = visible & live & (u <= rate), live = an offset mask of the anchor is on (mask_raw [A,K]: sigmoid(.) > 0.01, or the
 * stored 0 / 1 values when decoded; reference scene/gaussian_model.py get_mask_anchor), u [R*A] uniform draws of the caller. */
int gsvc_plan_masks(const uint8_t *const *visible_host, int32_t R, int64_t A, const float *mask_raw, int32_t K, int32_t decoded,
                    const float *u, float rate, uint8_t *M, uint8_t *present, uint8_t *chosen, void *stream);
/* The three quant_step networks of the EntropyParamsNets (reference scene/gaussian_model.py:198-232: Linear(in -> hidden), GELU,
 * Linear(hidden -> 1), evaluated at :1569-1597) on the same input rows X [M, in] in ONE launch each way (instantiated for
 * 192 -> 50 -> 1; other widths: GSVC_E_UNSUPPORTED, the caller keeps the layer kernels).  forward: z[i] [M, hidden] = the first
 * layer's pre-activation, a[i] = GELU(z[i]) (the operands of the backward and of the weight gradients), q[i] [M] = the raw output.
 * backward: dz[i] [M, hidden] = dq[i] W2_i GELU'(z[i]) (dq[i] NULL = zeros) and dX [M, in] = sum_i dz[i] W1_i; the weight
 * gradients are the products dz[i]^T X, dq[i]^T a[i] (gsvc_linear_wgrad_*). */
typedef struct gsvc_quant_step_net {
    const float *W1, *b1; /* [hidden][in], [hidden] */
    const float *W2, *b2; /* [1][hidden], [1] */
} gsvc_quant_step_net;
int gsvc_quant_step_nets_forward(const gsvc_quant_step_net *nets3, const float *X, int64_t M, int32_t in_dim, int32_t hidden,
                                 float *const *z3, float *const *a3, float *const *q3, void *stream);
int gsvc_quant_step_nets_backward(const gsvc_quant_step_net *nets3, const float *const *z3, const float *const *dq3, int64_t M,
                                  int32_t in_dim, int32_t hidden, float *const *dz3, float *dX, void *stream);

/* Per-row quantisation steps: out3 [3, rows], out3[g][i] = q_g adj_g[ctx_row ? ctx_row[i] : i] for the features, scalings and
 * offsets (reference gaussian_renderer/guassian.py:250-262, Q * Q_adj of the entropy context; ctx_row maps a row to its row of a
 * context evaluated once per distinct anchor).  raw != 0: adj_g holds the quant_step networks' raw outputs q and the adjustment
 * exp(clamp(q, -10, 10)) (reference scene/gaussian_model.py:1586-1596) is applied here.
 * backward: grad_adj3 [3, D] = scatter-add of q_g g_g[i] (any g_g may be NULL), times adj_g [|q| <= 10] when raw (the gradient
 * w.r.t. the raw outputs; adj_* are read only then). */
int gsvc_q_rows_forward(const float *adj_feat, const float *adj_scaling, const float *adj_offsets, const int64_t *ctx_row,
                        float q_feat, float q_scaling, float q_offsets, int64_t rows, int32_t raw, float *out3, void *stream);
int gsvc_q_rows_backward(const float *g_feat, const float *g_scaling, const float *g_offsets, const int64_t *ctx_row, float q_feat,
                         float q_scaling, float q_offsets, int64_t rows, int64_t D, const float *adj_feat, const float *adj_scaling,
                         const float *adj_offsets, int32_t raw, float *grad_adj3, void *stream);

/* Scans and compactions of a step plan in three launches.  view_masks [R, A] (bytes, 0 / 1), chosen [R, A] (a subset of
 * view_masks: the rate sample, or NULL), present [A] (the union of the views).  Outputs: scan [R A] = inclusive scan of the
 * flattened view masks (row of (view, anchor) in the concatenated rows + 1); flat = the anchors (position mod A) at the set
 * positions of view_masks in order — the views' visible-anchor lists one after the other (capacity R A); sel_rows = scan - 1 at the chosen positions (capacity R A); pos [A] = inclusive scan of present - 1; distinct =
 * the set positions of present (capacity A); counts [R + 2]: the scan at each view's end, the chosen pairs, the distinct anchors. */
int64_t gsvc_plan_scans_scratch_bytes(int32_t R, int64_t A);
int gsvc_plan_scans(const uint8_t *view_masks, const uint8_t *chosen, const uint8_t *present, int32_t R, int64_t A, void *scratch,
                    int64_t *scan, int64_t *flat, int64_t *sel_rows, int64_t *pos, int64_t *distinct, int64_t *counts, void *stream);

/* Row maps for FiLM rows shared by the opposite views of a frame (views 2 f and 2 f + 1 of R have the same camera z, hence the
 * same condition per anchor: reference frame_cube/frame.py:18-43, gaussian_renderer/guassian.py:225-230).  vis [rows]: anchor of
 * every chain row (views concatenated, row_bounds_host [R + 1]); pos [A]: anchor -> index in the list distinct [D] of the anchors
 * some view sees; view_masks [R A] / scan [R A]: the flattened view masks and their inclusive scan.  Outputs: row_of [rows] =
 * (view / 2) D + pos[vis]; src_a / src_b [R / 2, D] = chain row of the anchor in view 2 f / 2 f + 1, or -1. */
int gsvc_film_row_maps(const int64_t *vis, const int64_t *row_bounds_host, int32_t R, const int64_t *pos, int64_t D, int64_t A,
                   const uint8_t *view_masks, const int64_t *scan, const int64_t *distinct, int32_t *row_of, int32_t *src_a,
                   int32_t *src_b, void *stream);

/* out [rows_u, C] = g[src_a[j], :] + g[src_b[j], :] per row j, a source of -1 counting as zeros: the gradient of the generators'
 * outputs computed once per (frame, anchor) from the gradients of the frame's two opposite views (FULL_PRECISION / STE_ENTROPY
 * steps: both views see the same Gaussians, reference gaussian_renderer/guassian.py:225-273 evaluates the networks per view);
 * src_a / src_b as gsvc_film_row_maps writes them.  A gather per output row: no atomics, a fixed order of the two terms. */
int gsvc_pair_rows_sum(const float *g, const int32_t *src_a, const int32_t *src_b, int64_t rows_u, int32_t C, float *out, void *stream);

/* Deterministic mode (GSVC_DETERMINISTIC=1 through gsvc_amd.switches; SURVEY section 5 "deterministic-mode switch for bwd atomics").
 * gsvc_set_deterministic(1): the launches whose float sums depend on the order workgroups retire in take a fixed-order form —
 * the hash-grid backward gives every (level, table slice) to ONE workgroup (its 64-bit fixed-point LDS sums are exact, the single
 * float add per table word is the only rounding), and the host routes every scatter-add of rows through gsvc_segment_rows_sum:
 * dst[sorted_idx[p]][c] (= or +=, `accumulate`) the sum over the run of equal targets starting at p of src[order[j]][c], j in list
 * order (order = a STABLE argsort of the rows' targets, sorted_idx the targets in that order).  Replaces the float atomics of
 * reference-shaped scatter-adds (utils/entropy_models.py:159-175's gradient, the index_add of guassian.py:160-176's gathers). */
int gsvc_set_deterministic(int on);

/* The weight gradients of gsvc_generators_backward / gsvc_generator_backward / gsvc_deform_backward on ANOTHER stream: with a non-NULL
 * stream set here those entries queue their chain kernels on the caller's stream as before, record an event there, and queue the
 * dW = G^T X products and their reduces (reference: autograd's weight gradients of scene/gaussian_model.py:150-196, 468-489) on
 * `stream` behind that event — so the caller's stream can go on with the feature gradient while they run.  The caller then owns
 * the hazards: the products read feat, cond, the saved activations, the scratch (which the two entries must therefore NOT share) and
 * gy of the deformation network, and write the gradient tensors, until `stream` has passed; the optimizer's stream must wait for
 * it.  NULL (the default): everything on the caller's stream.  Process-wide; not for concurrent callers. */
int gsvc_set_wgrad_stream(void *stream);
/* gsvc_wgrad_hold(1): while a stream is set, those entries keep their weight-gradient launches back (their operands by value);
 * gsvc_wgrad_flush(stream) records the event on `stream` NOW and queues every held product behind it — for a caller that runs
 * several chain backward passes in a row and wants all their chain kernels queued before the first product starts beside them
 * (a chain workgroup needs its whole CU).  Held launches that are never flushed are dropped by the next flush's owner: flush before
 * gsvc_wgrad_hold(0). */
int gsvc_wgrad_hold(int32_t on);
int gsvc_wgrad_flush(void *stream);
int gsvc_segment_rows_sum(const float *src, const int64_t *order, const int64_t *sorted_idx, int64_t n, int32_t C, float *dst,
                          int32_t accumulate, void *stream);

/* out[scan[i] - 1] = (value ? value[i] + value_bias : i) for every i with mask[i] != 0, `scan` = inclusive scan of the mask
 * (int64): the index lists of the step plan in one elementwise pass each; entries of `out` past the count are not written. */
int gsvc_compact_by_scan(const uint8_t *mask, const int64_t *scan, const int64_t *value, int64_t value_bias, int64_t n, int64_t *out,
                         void *stream);

/* Means of three whole parameter tensors in one pass (the x_mean terms of the rate's clamp bounds: reference
 * utils/entropy_models.py EntropyGaussian.forward, x_mean = mean of the full attribute tensor): out3 = (mean(feat),
 * mean(scaling_exp ? exp(scaling) : scaling), mean(offset)).  scratch: gsvc_param_means_scratch_floats() floats.  Fixed order. */
int64_t gsvc_param_means_scratch_floats(void);
int gsvc_param_means(const float *feat, int64_t n_feat, const float *scaling, int64_t n_scaling, int32_t scaling_exp,
                     const float *offset, int64_t n_offset, float *scratch, float *out3, void *stream);

/* Tail of an EntropyParamsNet (reference scene/gaussian_model.py:1586-1596): params [n,2C] = [mean | scale], q [n] ->
 * mean [n,C], scale = max(scale, 1e-9) [n,C], adj = exp(clamp(q, -10, 10)) [n]; backward -> dparams [n,2C], dq [n]. */
int gsvc_ctx_post_forward(const float *params, const float *q, int64_t n, int32_t C, float *mean, float *scale, float *adj,
                          void *stream);
int gsvc_ctx_post_backward(const float *params, const float *q, const float *adj, int64_t n, int32_t C, const float *g_mean,
                           const float *g_scale, const float *g_adj, float *dparams, float *dq, void *stream);

/* FiLM of the generator networks (reference scene/gaussian_model.py:150-166) over n = rows * features elements (a multiple
 * of 4, 16-byte aligned tensors): y = gamma * h + beta; backward of the product: dgamma = g * h, dh = g * gamma. */
int gsvc_film_forward(const float *gamma, const float *h, const float *beta, float *y, int64_t n, void *stream);
int gsvc_film_backward(const float *g, const float *h, const float *gamma, float *dgamma, float *dh, int64_t n, void *stream);

/* Conditioning input of the generator / deformation MLPs for the concatenated rows of `renders` (<= 16) views (reference
 * ortho_gaussian_renderer/guassian.py:225-230, utils/time_util.py:7-55): pe[row] = [embed(cam_z[r]) | embed(anchor[row].z -
 * cam_z[r])] for the rows row_bounds[r] <= row < row_bounds[r + 1]; embed(x) = [x, sin(2^k x), cos(2^k x)]_{k < freqs}.
 * anchor: [rows, 3]; pe: [rows, 2 * (2 * freqs + 1)].  row_bounds / cam_z are host arrays. */
int gsvc_embed_pe(const float *anchor, const int64_t *row_bounds, const float *cam_z, int32_t renders, int32_t freqs, float *pe,
                  void *stream);

/* Several streams in one launch (each job as gsvc_ans_decode): a coded model is dozens of streams of a few waves each; one
 * launch lasts as long as a lane's seg_len serial symbols whatever the number of streams it carries. */
typedef struct gsvc_ans_decode_job {
    const uint8_t *bytes;
    const uint64_t *seg_offsets;
    const float *mu, *sigma;
    int64_t n;
    int32_t min_symbol, max_symbol, seg_len;
    int32_t *symbols, *error_flag;
    void *scratch;
} gsvc_ans_decode_job;
int gsvc_ans_decode_many(const gsvc_ans_decode_job *jobs, int32_t n_jobs, void *stream);

/* ------------------------------------------------------------------------------------------------------
 * [INTERNAL] Linear layers of the generator / deformation / entropy-parameter MLPs (reference scene/gaussian_model.py:
 * 150-232: every nn.Linear applied to the [anchors, features] matrix)
 * ---------------------------------------------------------------------------------------------------- */

/* Y[M,N] = X[M,K] W[N,K]^T + bias[N] (bias may be NULL; torch.nn.Linear layout), fp32 MFMA, weight matrix resident
 * in LDS and X streamed through registers: K, N <= 192 (GSVC_E_UNSUPPORTED beyond).
 * w_in_out != 0: W is [K][N] (input-major), so Y = X W — the backward's dX = G W without a transposed copy.
 * relu != 0: Y = max(Y, 0) fused into the store (the Linear -> ReLU pairs of the reference's nn.Sequential MLPs). */
int gsvc_linear_forward(const float *X, const float *W, const float *bias, float *Y, int64_t M, int32_t K, int32_t N,
                        int32_t w_in_out, int32_t relu, void *stream);

/* The same product with an epilogue program (the activations of the MLPs ride on the GEMM that produces their operand;
 * gsvc_amd/mlp.py runs every network of scene/gaussian_model.py:150-232 as one chain of these calls).  aux1 / aux2 / Y2 / Y3
 * have Y's shape [M,N]; v = X W^T + bias:
 *   GSVC_LIN_NONE / _RELU       as gsvc_linear_forward
 *   GSVC_LIN_GELU_DUAL          Y = v (the pre-activation the backward needs), Y2 = GELU(v) (exact, erf form)
 *   GSVC_LIN_TANH / _SIGMOID    Y = tanh(v) / sigmoid(v)
 *   GSVC_LIN_MUL_GELU_GRAD      Y = v * GELU'(aux1)            (dX of a layer behind a GELU: aux1 = its pre-activation)
 *   GSVC_LIN_MUL_RELU_MASK      Y = aux1 > 0 ? v : 0           (dX of a layer behind a ReLU: aux1 = its output)
 *   GSVC_LIN_FILM               Y = v (gamma), Y2 = v * aux1 + aux2          (FiLM: aux1 = h, aux2 = beta)
 *   GSVC_LIN_FILM_GRAD          Y = v (d beta), Y2 = v * aux1 (d gamma, aux1 = h), Y3 = v * aux2 (d h, aux2 = gamma)
 *   GSVC_LIN_ADD                Y = v + aux1 (aux1 may be Y itself: the input gradients of several networks that read the same
 *                               matrix — the six entropy sub-networks all read the hash-grid feature — accumulate in place) */
enum {
    GSVC_LIN_NONE = 0, GSVC_LIN_RELU = 1, GSVC_LIN_GELU_DUAL = 2, GSVC_LIN_TANH = 3, GSVC_LIN_SIGMOID = 4,
    GSVC_LIN_MUL_GELU_GRAD = 5, GSVC_LIN_MUL_RELU_MASK = 6, GSVC_LIN_FILM = 7, GSVC_LIN_FILM_GRAD = 8, GSVC_LIN_ADD = 9
};
int gsvc_linear_forward_ex(const float *X, const float *W, const float *bias, float *Y, int64_t M, int32_t K, int32_t N,
                           int32_t w_in_out, int32_t epilogue, const float *aux1, const float *aux2, float *Y2, float *Y3,
                           void *stream);

/* Y[M,N] = sum_p X_p[M,K_p] W_p with W_p given [K_p][N] (input-major: the dX = G W products of a backward pass whose input
 * gradients meet in one matrix — the six sub-networks of the three EntropyParamsNets all read the hash-grid feature, reference
 * scene/gaussian_model.py:198-232, 1569-1597) in ONE launch: a wave keeps its rows' accumulators in registers across the
 * products and stores the sum once.  At most 8 products, N, K_p <= 192, M <= 65536 (GSVC_E_UNSUPPORTED beyond: use
 * gsvc_linear_forward_ex with GSVC_LIN_ADD product by product). */
typedef struct {
    const float *X;      /* [M][K] */
    const float *W;      /* [K][N] */
    int32_t K;
    int32_t pad;
} gsvc_accum_job;
int gsvc_linear_accumulate_many(const gsvc_accum_job *jobs, int32_t n_jobs, float *Y, int64_t M, int32_t N, void *stream);

/* The forward counterpart: several layers that read the SAME input, Y_p[M,N_p] = X[M,K] W_p^T + b_p with W_p [N_p][K] as nn.Linear
 * stores it; Y2_p != NULL: Y_p keeps the pre-activation and Y2_p = GELU(Y_p) (the GSVC_LIN_GELU_DUAL pair).  One launch: a wave
 * reads its rows' fragments once and keeps them across the products.  At most 8 products, N_p <= 160, K <= 192 and a multiple
 * of 4, M <= 65536 (GSVC_E_UNSUPPORTED beyond: gsvc_linear_forward_ex layer by layer). */
typedef struct {
    const float *W;      /* [N][K] */
    const float *bias;   /* [N] or NULL */
    float *Y, *Y2;       /* [M][N]; Y2 may be NULL */
    int32_t N;
    int32_t pad;
} gsvc_shared_input_job;
int gsvc_linear_forward_shared_input(const float *X, int64_t M, int32_t K, const gsvc_shared_input_job *jobs, int32_t n_jobs,
                                     void *stream);

/* dW[N,K] = G[M,N]^T X[M,K] and (db != NULL) db[N] = column sums of G: the weight / bias gradients of the same
 * layers.  Rows are split over the chip instead of the tiny output; every workgroup writes its partial sums to its
 * slot of `workspace` ((N*K + N) floats per slot, gsvc_linear_wgrad_workspace() = 256 slots) and a second small
 * kernel adds the slots, so the result is deterministic (no atomics).  K, N <= 192. */
int64_t gsvc_linear_wgrad_workspace(int32_t N, int32_t K);
int gsvc_linear_wgrad(const float *G, const float *X, float *dW, float *db, int64_t M, int32_t N, int32_t K,
                      float *workspace, int64_t workspace_floats, void *stream);

/* The two halves of gsvc_linear_wgrad apart, so that one small launch adds up the partial sums of every layer of a network's
 * backward pass: _partial runs the row-split kernel only (workspace as above, one region per pending layer; *slots_used = the
 * slots it filled), _reduce_many adds the slots of n_jobs layers (db may be NULL when the partials were taken without it;
 * `partial` must then have been produced with want_db = 0). */
typedef struct gsvc_wgrad_reduce_job {
    const float *partial;
    float *dW, *db;
    int32_t slots, N, K;
} gsvc_wgrad_reduce_job;
int gsvc_linear_wgrad_partial(const float *G, const float *X, int32_t want_db, int64_t M, int32_t N, int32_t K,
                              float *workspace, int64_t workspace_floats, int32_t *slots_used, void *stream);
int gsvc_linear_wgrad_reduce_many(const gsvc_wgrad_reduce_job *jobs, int32_t n_jobs, void *stream);
/* The partial sums of n_jobs products at once: products of the usual shapes share launches (up to 8 per launch, the workgroups
 * dealt to them by their MFMA work: the weight gradients of a whole network over the same rows), anything else is launched alone.
 * Every job names its own workspace region; slots_used is written per job (the `slots` of its reduce job). */
typedef struct gsvc_wgrad_partial_job {
    const float *G, *X;
    float *workspace;
    int64_t M, workspace_floats;
    int32_t want_db, N, K;
    int32_t slots_used;      /* out */
} gsvc_wgrad_partial_job;
int gsvc_linear_wgrad_partial_many(gsvc_wgrad_partial_job *jobs, int32_t n_jobs, void *stream);

/* ------------------------------------------------------------------------------------------------------
 * [INTERNAL] Whole-network chain kernels of the generator and deformation MLPs (csrc/mlp_chain.hip): a 16-row block's activations stay
 * in registers from the network's input to its output; weights resident in LDS; fp32 MFMA.
 * Replaces, per anchor row: GeneratorNet.forward = out_act(out_linear(film(linear2(GELU(linear1(feature))), condition)))
 * with FiLM = gamma(c) * x + beta(c), gamma / beta = fc_*1(ReLU(fc_*0(c))) (reference scene/gaussian_model.py:150-196, used at
 * ortho_gaussian_renderer/guassian.py:251-262), and mlp_deform = 5 x Linear with GELU between (scene/gaussian_model.py:468-489,
 * guassian.py:264-273) on cat([feature, condition]).  All matrices row-major fp32, weights in torch.nn.Linear layout [out, in],
 * every pointer 16-byte aligned.  Instantiated widths: feature 50, condition 66, hidden 100, outputs 10 / 30 / 70 (generators)
 * and 30 (deform); anything else returns GSVC_E_UNSUPPORTED (callers keep the gsvc_linear_* layer path).
 *   _forward : y[M, out]; `saved` (gsvc_*_saved_floats floats, caller-owned) receives what the backward reads.
 *   _backward: from gy[M, out] forms d feature (gfeat[M, feat]; accumulate_gfeat != 0 adds to what is there — the feature
 *              matrix feeds four networks), and every weight / bias gradient (dW = G^T X by the row-split kernels of
 *              gsvc_linear_wgrad_partial + one batched slot reduce: deterministic, no atomics) into the pointers of `grads`
 *              (a NULL weight pointer skips that layer).  The condition receives no gradient (it is the positional embedding of
 *              detached anchor positions / the frame time).  `scratch`: gsvc_*_scratch_floats floats.
 * out_act: 0 identity, 1 tanh, 2 sigmoid. */
typedef struct gsvc_generator_net {
    const float *W1, *b1, *W2, *b2, *W3, *b3;                             /* linear1, linear2, out_linear */
    const float *Wg0, *bg0, *Wg1, *bg1, *Wb0, *bb0, *Wb1, *bb1;           /* film.fc_gamma0/1, film.fc_beta0/1 */
    int32_t feat_dim, cond_dim, hidden_dim, out_dim, out_act;
} gsvc_generator_net;
typedef struct gsvc_generator_grads {
    float *W1, *b1, *W2, *b2, *W3, *b3, *Wg0, *bg0, *Wg1, *bg1, *Wb0, *bb0, *Wb1, *bb1;
} gsvc_generator_grads;
 /* gsvc_generators_*: n_nets (1 .. 3) networks on the same (feat, cond) rows in ONE pair of launches each way (workgroup b serves
 * network b % n_nets: one prologue and one partial last round for all of them).  saved / y / gy / gfeat are arrays of n_nets
 * pointers; `scratch` holds the networks' regions back to back (sum of gsvc_generator_scratch_floats, each rounded up to a
 * multiple of 4 floats); gfeat[i] receives network i's feature gradient (distinct buffers, written, not accumulated: pass them
 * to gsvc_deform_backward as addends, or add them up). */
/* Shared FiLM rows.  gamma / beta depend on the condition only, and the two opposite views of a frame (the training step renders
 * both: reference pipeline/train.py:353-387) have the same condition for the same anchor — same camera z, same anchor z — while
 * their features differ (independent quantisation noise per render).  With a gsvc_film_rows the FiLM networks (4 of a
 * generator's 7 layers, forward, backward and weight gradients) run once per (frame, anchor) instead of once per (view, anchor):
 *   rows      number of FiLM rows;  cond [rows, cond_dim] their condition
 *   row_of    [M] int32: chain row (view, anchor) -> its FiLM row
 *   src_a/_b  [rows] int32: the (up to) two chain rows of a FiLM row, -1 = that view does not see the anchor
 * NULL (or rows = 0): one FiLM row per chain row, conditions = cond.  The sizes below take the FiLM row count (0 = M). */
typedef struct gsvc_film_rows {
    int64_t rows;
    const float *cond;
    const int32_t *row_of, *src_a, *src_b;
} gsvc_film_rows;
int64_t gsvc_generator_saved_floats(const gsvc_generator_net *net, int64_t M, int64_t film_rows);
int64_t gsvc_generator_scratch_floats(const gsvc_generator_net *net, int64_t M, int64_t film_rows);
int gsvc_generator_forward(const gsvc_generator_net *net, const float *feat, const float *cond, int64_t M, float *saved, float *y,
                           void *stream);
int gsvc_generators_forward(const gsvc_generator_net *nets, int32_t n_nets, const float *feat, const float *cond, int64_t M,
                            const gsvc_film_rows *film, float *const *saved, float *const *y, void *stream);
/* Forward only (the decoder's render loop, evaluation): the same kernels with every store the backward alone would read left out.
 * scratch[i]: gsvc_generator_inference_floats(net, M, film_rows) floats (gamma / beta of the FiLM networks); deform: HID M floats. */
int64_t gsvc_generator_inference_floats(const gsvc_generator_net *net, int64_t M, int64_t film_rows);
int gsvc_generators_forward_inference(const gsvc_generator_net *nets, int32_t n_nets, const float *feat, const float *cond, int64_t M,
                                      const gsvc_film_rows *film, float *const *scratch, float *const *y, void *stream);
int gsvc_generators_backward(const gsvc_generator_net *nets, int32_t n_nets, const float *feat, const float *cond, int64_t M,
                             const gsvc_film_rows *film, const float *const *saved, const float *const *y, const float *const *gy,
                             float *scratch, float *const *gfeat, const gsvc_generator_grads *grads, void *stream);
int gsvc_generator_backward(const gsvc_generator_net *net, const float *feat, const float *cond, int64_t M, const float *saved,
                            const float *y, const float *gy, float *scratch, float *gfeat, int32_t accumulate_gfeat,
                            const gsvc_generator_grads *grads, void *stream);

typedef struct gsvc_deform_net {
    const float *W[5], *b[5];                                             /* mlp_deform's five Linear layers, input = [feat | cond] */
    int32_t feat_dim, cond_dim, hidden_dim, out_dim;
} gsvc_deform_net;
typedef struct gsvc_deform_grads {
    float *W[5], *b[5];
} gsvc_deform_grads;
int64_t gsvc_deform_saved_floats(const gsvc_deform_net *net, int64_t M);
int64_t gsvc_deform_scratch_floats(const gsvc_deform_net *net, int64_t M);
int gsvc_deform_forward(const gsvc_deform_net *net, const float *feat, const float *cond, int64_t M, float *saved, float *y, void *stream);
/* gfeat_addends: n_addends (0 .. 3) further [M, feat] gradients summed into gfeat in the same pass (the generators' feature gradients) */
int gsvc_deform_forward_inference(const gsvc_deform_net *net, const float *feat, const float *cond, int64_t M, float *scratch, float *y,
                                  void *stream);
int gsvc_deform_backward(const gsvc_deform_net *net, const float *feat, const float *cond, int64_t M, const float *saved, const float *gy,
                         float *scratch, float *gfeat, int32_t accumulate_gfeat, const float *const *gfeat_addends, int32_t n_addends,
                         const gsvc_deform_grads *grads, void *stream);

/* ------------------------------------------------------------------------------------------------------
 * [INTERNAL] Anchor geometry decode (the G-PCC tmc3 step of reference utils/encodings.py:780-826 decode_anchor; coder and bitstream are this
 * library's own: gsvc_amd/anchor_codec.py).  An occupancy octree over the integer anchor lattice, one 8-bit mask per node,
 * per level a static 12-bit model and an interleaved rANS stream (32-bit states, 16-bit words, symbol i in lane i % lanes).
 * gsvc_anchor_rans_decode: all levels' symbols in one launch (one workgroup per level); error: OR of 1 bad table, 2 truncated,
 * 4 the coder did not return to its initial state.  gsvc_octree_popcount / _expand: a level's masks -> the next level's nodes
 * (Morton keys; counts_inclusive_scan = inclusive scan of the popcounts; error |= 8 on an empty mask or more than `capacity`
 * children).  gsvc_morton_decode: keys (x at bit 3 b + 2, y at 3 b + 1, z at 3 b) -> xyz [n, 3].
 * ---------------------------------------------------------------------------------------------------- */
typedef struct gsvc_anchor_level {
    const uint32_t *states;     /* [lanes] */
    const uint16_t *words;      /* [n_words] */
    const uint16_t *freq;       /* [256], sum 4096 */
    uint8_t *out;               /* [n] decoded masks */
    int64_t n, n_words;
    int32_t lanes;
} gsvc_anchor_level;
int gsvc_anchor_rans_decode(const gsvc_anchor_level *levels_host, int32_t n_levels, int32_t *error, void *stream);
int gsvc_octree_popcount(const uint8_t *occ, int64_t n, int64_t *counts, void *stream);
int gsvc_octree_expand(const int64_t *nodes, const uint8_t *occ, const int64_t *counts_inclusive_scan, int64_t n, int64_t capacity,
                       int64_t *nodes_out, int32_t *error, void *stream);
int gsvc_morton_decode(const int64_t *keys, int64_t n, int64_t *xyz, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GSVC_HIP_H */
