/*
 * gsraster.h -- C ABI of libgsraster.so, the MI355X (gfx950) differentiable Gaussian rasterizer.
 *
 * This is the drop-in boundary for the ONE hot path of W-Ted/GScream: the native module
 * `diff_gaussian_rasterization._C` (reference: submodules/diff-gaussian-rasterization, "DGR").
 * The reference binds five pybind11 functions (DGR/ext.cpp:15-21) that take torch::Tensor; this
 * library exposes the same operations as plain `extern "C"` entry points taking raw DEVICE
 * pointers, sizes and a hipStream_t -- no torch types, no C++ types, no exceptions.
 * gscream_amd/_native.py (ctypes) is the only caller; INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - every pointer is a device pointer unless its name ends in _host;
 *   - all float arrays are contiguous row-major fp32, images are CHW (DGR forward.cu:563);
 *   - matrices are the transposed / row-vector form GScream passes (scene/cameras.py:64-67):
 *     p_view = [x y z 1] * viewmatrix, indexed m[0]x + m[4]y + m[8]z + m[12] (auxiliary.h:58-77);
 *   - a NULL optional pointer means "not provided", like the reference's empty tensors
 *     (DGR forward.cu:209,245; backward.cu:400,404);
 *   - every function returns 0 on success and a negative gsr_status on failure;
 *     gsr_last_error() returns a thread-local message for the last failure on this thread;
 *   - the library owns no device memory: workspaces are sized by the gsr_*_bytes queries, allocated by the
 *     caller (PyTorch's caching allocator) and passed in.  All workspace pointers must be 256-byte aligned.
 *     State it does keep, none of it observable through results: per host thread and device one 64-byte pinned,
 *     device-mapped word block (the read-back of num_rendered; a flag "the last backward met a group of very large
 *     splats", which only selects between two launch plans that produce the same bits) and one event, created on
 *     first use and released when the thread exits; the last-error string (thread-local); the optional profiler's event list
 *     (process-wide, mutex-guarded, only between gsr_profile_begin / _end).  Entry points are re-entrant.
 *   - `stream` is a hipStream_t (NULL = the legacy default stream, which is what the reference
 *     launches on).  debug != 0 synchronises and checks for errors after every stage, the
 *     semantics of the reference's CHECK_CUDA (auxiliary.h:166-173).
 */
#ifndef GSRASTER_H_INCLUDED
#define GSRASTER_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum gsr_status {
    GSR_OK = 0,
    GSR_ERR_INVALID_ARGUMENT = -1,
    GSR_ERR_HIP = -2,          /* a HIP runtime call or kernel launch failed */
    GSR_ERR_UNSUPPORTED = -3,  /* e.g. more tiles than the binning kernels support */
    GSR_ERR_NO_DEVICE = -4,
    GSR_NEED_CAPACITY = 1      /* gsr_forward only: the binning workspace was too small, see there */
} gsr_status;

/* Result of forward stage 1, written to host memory (pinned memory avoids a staging copy). */
typedef struct gsr_stage1_result {
    int32_t num_rendered;  /* R: number of binned (Gaussian, tile) instances.  With tile culling disabled this
                              is the reference's return value of Rasterizer::forward (rasterizer_impl.cu:346) */
    int32_t max_tile_count; /* longest per-tile list; selects the per-tile sort variant */
    int32_t num_slots;      /* number of gradient slots the backward scratch must hold: one per BINNED instance, i.e.
                               equal to num_rendered (tiles the cull or the occlusion cut-off dropped own no slot) */
    int32_t num_occluded;   /* instances the conservative occlusion cut-off removed before binning (gsr_tuning.occlusion_cut;
                               0 when it is off): lets the caller judge whether the pass pays on this kind of frame */
} gsr_stage1_result;

/* Tunables; zero-initialise for defaults.  Pure performance knobs: images, radii and gradients do not
 * depend on them. */
typedef struct gsr_tuning {
    int32_t disable_tile_cull; /* 1 = bin every tile of the rectangle like the reference (the internal per-tile
                                  lists and num_rendered become bit-identical to the reference's; slower).
                                  Default 0: skip tiles the Gaussian cannot change (gsr_math.h) */
    int32_t disable_speculation; /* host-side hint (the library ignores it): 1 = the binding should always use the
                                  two-stage forward instead of gsr_forward */
    int32_t disable_partial_sort; /* 1 = sort every per-tile list completely, like the reference.  Default 0: lists longer
                                  than 2048 entries are put in depth order only for their nearest <= 2048 instances first
                                  (the blend stops where the tile's pixels saturate); a tile whose pixels are still
                                  blending at the end of that prefix is sorted completely and blended again */
    int32_t inference; /* 1 = no backward will follow this forward (render under no_grad, the reference's eval loops
                                  train.py:756-763,861-878): the forward writes no depth checkpoints, contributor counts,
                                  per-tile traversal depths or gradient-slot offsets and clears no slot flags.  Images and
                                  radii are bit-identical to the training forward; gsr_backward on that state is undefined */
    int32_t scatter_bands; /* 0 = automatic.  n > 0 forces the scatter launch to n bands of tile rows per chunk (binning.hip: large images /
                                  instance counts stage their keys per band so that they leave in runs; a test / tuning knob) */
    int32_t occlusion_cut; /* 1 = conservative per-tile occlusion cut-off in front of the binning (frames of large splats: lists of
                                  thousands of instances of which the blend walks a tenth).  Every (Gaussian, tile) instance that covers
                                  its whole tile with alpha >= 1/255 adds -log2(1 - its smallest alpha in the tile) to a fixed-point sum per
                                  (tile, depth bucket; 16 buckets per octave of view depth); behind the bucket at which a tile's sum says
                                  "every pixel's transmittance is below 1e-4" nothing can blend (DGR forward.cu:537), and those instances
                                  are never binned.  Images, radii, gradients unchanged bit for bit; num_rendered and the lists shrink */
    int32_t heavy_groups; /* per-Gaussian backward, groups of 64 Gaussians with more than 1024 gradient slots (large splats).  0 = automatic:
                                  a second, cooperative kernel takes them when the caller's previous backwards met enough of them to pay for
                                  its launch; 1 = always launch it, 2 = never (the one-wave kernel does them).  Same bits in every mode */
    int32_t walk_depths_valid; /* see walk_depths: 1 = the array holds a previous visit's depths (order by them), 0 = only record */
    uint64_t walk_depths; /* device address of 4 * ceil(W/16) * ceil(H/16) uint32 the caller keeps PER VIEW (camera), or 0.  The forward's
                                  launch lasts as long as its deepest walks, and the ones that start late are what it waits for; how deep a
                                  quadrant walks hardly changes between two visits of one view, and training revisits its views every
                                  epoch.  With this array the forward records the list depth every 8x8 quadrant walked, and -- when
                                  walk_depths_valid says the array holds a previous visit's -- first dispatches its tasks deepest first
                                  (forward blend -7 % config 2, -11 % config 3, -16 % on GScream's iteration-0 frame).  Any content is safe:
                                  images and gradients do not depend on it.  The array must not be written by other work while the
                                  forward runs. */
} gsr_tuning;

/* Pipeline stages, for the optional per-stage timing below. */
enum { GSR_STAGE_PREPROCESS = 0, GSR_STAGE_COUNT_SCAN, GSR_STAGE_SCATTER, GSR_STAGE_TILE_SORT, GSR_STAGE_BLEND_FWD,
       GSR_STAGE_BLEND_BWD, GSR_STAGE_GAUSS_BWD, GSR_NUM_STAGES };

typedef struct gsr_profile {
    double total_ms[GSR_NUM_STAGES];   /* sum of HIP-event elapsed times per stage */
    int64_t launches[GSR_NUM_STAGES];  /* number of timed stage invocations */
} gsr_profile;

const char* gsr_version(void);
/* Integer version of this header's binary interface: bumped whenever an entry point's argument list, a struct layout or a
 * workspace size formula changes incompatibly.  Bindings compare it with GSR_ABI_VERSION at load time. */
#define GSR_ABI_VERSION 8
int gsr_abi_version(void);
const char* gsr_last_error(void);
/* Forget what the calling thread's previous calls taught the launch heuristics that learn from feedback (ABI 7): today the partial
 * sort's bet -- after a forward in which a quadrant ran off its tile's partially sorted prefix, the next 64 forwards sort lists of up to
 * 4096 entries completely at once instead of paying a second forward-blend launch (api.hip gsr_partial_bet; no counterpart in the
 * reference, which always sorts everything: rasterizer_impl.cu:300-320).  Results never depend on these heuristics; tests and
 * benchmarks that compare launch patterns call this between cases (gscream_amd.set_tuning does). */
void gsr_adaptive_reset(void);
/* Number of visible HIP devices, or a negative gsr_status. */
int gsr_device_count(void);

/* ---- workspace sizes (replace the obtain()/required<T>() carving of rasterizer_impl.h:21-74) ---- */
size_t gsr_geom_bytes(int P);                 /* per-Gaussian state kept from forward to backward   */
size_t gsr_image_bytes(int P, int W, int H);  /* per-pixel / per-tile state kept forward -> backward (16 B + 192 B of
                                                 depth checkpoints per pixel, of which only the reached ones are touched;
                                                 + 5 KiB per tile, up to 8192 tiles, for the occlusion cut-off's mass tables,
                                                 touched only when gsr_tuning.occlusion_cut is set) */
size_t gsr_binning_bytes(int R);              /* per-instance state: sorted point list (+ sort keys)  */
size_t gsr_backward_scratch_bytes(int P, int num_slots); /* per-instance gradient slots + the heavy-group list of the per-Gaussian backward */

/*
 * Forward, stage 1 of 2: per-Gaussian preprocess + tile histogram + scans.
 * Replaces FORWARD::preprocess, cub::DeviceScan::InclusiveSum and the 4-byte D2H read of
 * num_rendered in CudaRasterizer::Rasterizer::forward (DGR rasterizer_impl.cu:252-287).
 *   means3D[P,3] opacities[P] features[P] (GScream's per-Gaussian `uncertainty`)
 *   scales[P,3]+rotations[P,4]  XOR  cov3D_precomp[P,6]
 *   colors_precomp[P,3]         XOR  shs[P,M,3] with degree D (campos[3] needed for SH)
 * Outputs: radii[P] (int32; 0 = culled), *result_host.  The call synchronises `stream` once,
 * like the reference's blocking cudaMemcpy (rasterizer_impl.cu:287).
 * prefiltered: the caller's promise that no point fails the near-plane test.  The reference traps the kernel on a
 * violation (auxiliary.h:154-162); here, with prefiltered != 0 AND debug != 0, the entry points that take the flag
 * (stage 1, gsr_forward, gsr_filter) check the promise first and fail with GSR_ERR_INVALID_ARGUMENT and the
 * reference's message; without debug the flag is ignored (culled points are culled).
 */
int gsr_forward_stage1(int P, int D, int M, int W, int H,
                       const float* means3D, const float* scales, float scale_modifier, const float* rotations,
                       const float* opacities, const float* features, const float* shs,
                       const float* cov3D_precomp, const float* colors_precomp,
                       const float* viewmatrix, const float* projmatrix, const float* campos,
                       float tan_fovx, float tan_fovy, int prefiltered,
                       void* geom, void* image, int32_t* radii, gsr_stage1_result* result_host,
                       const gsr_tuning* tuning, int debug, void* stream);

/*
 * Forward, stage 2 of 2: instance scatter into per-tile segments, per-tile depth sort, blend.
 * Replaces duplicateWithKeys, cub::DeviceRadixSort::SortPairs, identifyTileRanges and
 * FORWARD::render (DGR rasterizer_impl.cu:295-344).  `binning` must hold gsr_binning_bytes(R).
 * Outputs (every pixel is written): out_color[3,H,W], out_depth[1,H,W], out_feature[1,H,W].
 */
int gsr_forward_stage2(int P, int W, int H, int R, int max_tile_count, const float* background,
                       void* geom, void* image, void* binning,
                       float* out_color, float* out_depth, float* out_feature,
                       const gsr_tuning* tuning, int debug, void* stream);

/*
 * Forward in ONE call, without the GPU-idle window of the two-stage form.  The caller passes a binning workspace
 * sized for `binning_capacity` instances (gsr_binning_bytes(binning_capacity); e.g. 1.25 x the previous frame's
 * num_rendered).  Stage 2 is enqueued before the host waits for num_rendered, so the device never waits for the
 * host.  `max_tile_count_hint` (> 0) bounds the longest per-tile list the caller expects: only the sort variants
 * needed for it are launched; <= 0 launches all of them.  Returns GSR_OK when num_rendered <= binning_capacity and
 * max_tile_count <= hint (outputs valid; keep `binning_capacity` for gsr_backward).  Returns GSR_NEED_CAPACITY (> 0)
 * otherwise: nothing was written out of bounds, but the images are invalid -- allocate
 * gsr_binning_bytes(result_host->num_rendered) and call gsr_forward_stage2 to redo stage 2.
 * (Performance only: the instance scatter sizes its staging -- and decides whether to split its launch into bands of tile
 * rows -- for 0.8 x binning_capacity instances, i.e. it assumes the provision carries the 25 % of slack suggested above.)
 */
int gsr_forward(int P, int D, int M, int W, int H,
                const float* means3D, const float* scales, float scale_modifier, const float* rotations,
                const float* opacities, const float* features, const float* shs,
                const float* cov3D_precomp, const float* colors_precomp,
                const float* viewmatrix, const float* projmatrix, const float* campos,
                float tan_fovx, float tan_fovy, int prefiltered, const float* background,
                void* geom, void* image, void* binning, int binning_capacity, int max_tile_count_hint, int32_t* radii,
                float* out_color, float* out_depth, float* out_feature, gsr_stage1_result* result_host,
                const gsr_tuning* tuning, int debug, void* stream);

/*
 * Backward.  Replaces CudaRasterizer::Rasterizer::backward (DGR rasterizer_impl.cu:536-643):
 * BACKWARD::render + computeCov2DCUDA + preprocessCUDA(bwd).  geom/image/binning are the buffers
 * the forward filled.  Every output row is written (culled Gaussians get exact zeros), so the
 * caller need not zero-fill.  Optional outputs may be NULL: dL_dcov3D[P,6], dL_dsh[P,M,3].
 *   dL_dmeans2D[P,3] (z = 0)  dL_dcolors[P,3]  dL_dopacity[P]  dL_dfeatures[P]
 *   dL_dmeans3D[P,3]  dL_dscales[P,3]  dL_drotations[P,4]
 * dL_dout_depth and dL_dout_feature may each be NULL (= no gradient flows into that map; with both NULL a
 * cheaper kernel variant runs).  No floating-point atomics on global memory are used.
 * The image workspace (backward task list) and the binning workspace (one "slot written" byte per instance: cleared
 * by the forward, set by the blend -- to the same set on every backward of one forward state) are used as scratch
 * during the call, so the backward may run again on the same forward state; two backward calls on ONE forward state
 * must not overlap on different streams.
 */
int gsr_backward(int P, int D, int M, int W, int H, int R, int binning_capacity /* what the forward's binning
                 workspace was sized for; = R after the two-stage forward */,
                 int max_tile_count /* gsr_stage1_result.max_tile_count of THAT forward (its longest per-tile list), or <= 0 if the
                 caller did not keep it: sizes the grid of depth-segment tasks -- a frame none of whose lists reaches the second
                 segment tier needs half the workgroups; unknown = the full grid, same results */,
                 const float* background,
                 const float* means3D, const int32_t* radii, const float* colors_precomp, const float* shs,
                 const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                 const float* viewmatrix, const float* projmatrix, const float* campos,
                 float tan_fovx, float tan_fovy,
                 const float* dL_dout_color, const float* dL_dout_depth, const float* dL_dout_feature,
                 const void* geom, const void* image, const void* binning, void* scratch,
                 float* dL_dmeans2D, float* dL_dcolors, float* dL_dopacity, float* dL_dfeatures,
                 float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscales, float* dL_drotations,
                 const gsr_tuning* tuning, int debug, void* stream);

/*
 * Visibility filters on a point cloud (GScream calls them on the anchors every iteration,
 * train.py:433).  Replaces Rasterizer::visible_filter (px = py = NULL) and
 * Rasterizer::position2D_filter (DGR rasterizer_impl.cu:350-406, :470-530).  No workspace.
 * radii[P]; px[P], py[P] = pixel-space centre, 0 when culled (forward.cu:378-379).
 */
int gsr_filter(int P, int W, int H, const float* means3D, const float* scales, float scale_modifier,
               const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
               const float* projmatrix, float tan_fovx, float tan_fovy, int prefiltered,
               int32_t* radii, float* px, float* py, int debug, void* stream);

/* Replaces Rasterizer::markVisible (DGR rasterizer_impl.cu:141-153): present[i] = view-space z > 0.2. */
int gsr_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                     uint8_t* present, void* stream);

/*
 * ---- fused neural-Gaussian decode + compaction, the step right before the rasterizer ----------
 * Replaces the body of gaussian_renderer/__init__.py:18-102 generate_neural_gaussians after the visible-anchor
 * gather: view vector/distance, the four MLPs 36->32->{K, K, 3K, 7K} (scene/gaussian_model.py:118-144: opacity+Tanh,
 * uncertainty+Sigmoid, color+Sigmoid, cov linear), the opacity>0 mask, the boolean-mask compaction and the
 * post-processing.  N anchors, K = n_offsets <= 10, feat_dim = 32.  All pointers are device pointers, fp32
 * contiguous: feat[N,32], anchor[N,3], offsets[N,K,3], grid_scaling[N,6] (= exp(_scaling)), campos[3].
 * weights[16] = { w1[4], b1[4], w2[4], b2[4] } for the MLPs {opacity, uncertainty, color, cov} (torch Linear layout).
 * visible (optional): int32[N] rows of the model-sized tensors to decode -- the visible-anchor gather (:25-28) folded
 * in; NULL = rows 0..N-1.  With `visible`, feat/anchor/offsets/grid_scaling and the four d_* outputs are MODEL-sized
 * (the backward writes the visible rows only: pre-zero them, or let
 * gsr_decode_zero_hidden_rows fill the others).
 * visible_count (optional, with `visible`): device word holding how many entries of `visible` are valid, read by the kernels
 * instead of being passed by the host (N is then the upper bound everything was sized for); with gsr_decode_visible_rows, which
 * compacts a boolean mask into that row list + count on the device, a training iteration needs no host round trip for the
 * visible-anchor gather (the reference's x[visible_mask] synchronises; so does torch.nonzero).
 *   gsr_decode_visible_rows : rows[<= N] = ascending indices of the set bytes of visible_mask[N], count[1]; block_scratch: ceil(N/256) u32
 *   gsr_decode_count : neural_opacity[N*K], mask[N*K] (u8), count[N] (u8), first[N] (u32, exclusive scan), total[1] (u32);
 *                      block_scratch: ceil(N/256) u32 of scratch
 *   gsr_decode_emit  : the total[0] surviving rows, in boolean-mask order: xyz[M,3], color[M,3], opacity[M], uncertainty[M],
 *                      scaling[M,3], rot[M,4]
 *   gsr_decode_backward : upstream gradients of those rows (each of the six may be NULL = no gradient flows into that
 *                      output: read as zeros) -> d_feat[N,32], d_anchor[N,3], d_offsets[N,K,3],
 *                      d_grid_scaling[N,6] AND the 16 weight / bias gradients grads16 = { gw1[4] [32,36], gb1[4] [32],
 *                      gw2[4] [out,32], gb2[4] [out] } in the order of `weights` (every element written; bit-reproducible),
 *                      in one pass on the f32 matrix cores; workspace of gsr_decode_weight_grad_workspace_bytes() bytes
 *                      (workgroup partials, nothing to initialise).  Replaces the autograd backward of
 *                      generate_neural_gaussians including the reference's nn.Linear layers (scene/gaussian_model.py:118-144).
 */
int gsr_decode_visible_rows(int N, const uint8_t* visible_mask, int32_t* rows, uint32_t* count, uint32_t* block_scratch, void* stream);
int gsr_decode_count(int N, int K, const float* const* weights, const int32_t* visible, const uint32_t* visible_count, const float* feat, const float* anchor, const float* campos,
                     float* neural_opacity, uint8_t* mask, uint8_t* count, uint32_t* first, uint32_t* total,
                     uint32_t* block_scratch, void* stream);
int gsr_decode_emit(int N, int K, const float* const* weights, const int32_t* visible, const uint32_t* visible_count, const float* feat, const float* anchor, const float* offsets,
                    const float* grid_scaling, const float* campos, const float* neural_opacity /* from gsr_decode_count */,
                    const uint8_t* mask, const uint32_t* first, float* xyz,
                    float* color, float* opacity, float* uncertainty, float* scaling, float* rot, void* stream);
int gsr_decode_backward(int N, int K, const float* const* weights, const int32_t* visible, const float* feat, const float* anchor,
                        const float* offsets, const float* grid_scaling, const float* campos, const uint8_t* mask,
                        const uint32_t* first, const float* g_xyz, const float* g_color, const float* g_opacity,
                        const float* g_uncertainty, const float* g_scaling, const float* g_rot, float* d_feat, float* d_anchor,
                        float* d_offsets, float* d_grid_scaling, void* workspace, float* const* grads16, void* stream);
size_t gsr_decode_weight_grad_workspace_bytes(void);
/* With `visible`, gsr_decode_backward writes the visible rows of the model-sized d_* tensors only.  This fills the OTHER rows
 * (visible_mask[r] == 0, one byte per model row: the boolean mask the row list was made from) with zeros, so that the caller
 * can hand in uninitialised tensors instead of zero-filling all N rows.  N = model rows here. */
int gsr_decode_zero_hidden_rows(int N, int K, const uint8_t* visible_mask, float* d_feat, float* d_anchor, float* d_offsets,
                                float* d_grid_scaling, void* stream);

/*
 * Densification statistics of one training iteration: replaces the body of
 * GaussianModel.training_statis (scene/gaussian_model.py:730-757).  Nv visible anchors, K offsets each.
 * visible[Nv] (int32 rows of the model, NULL = identity), neural_opacity[Nv*K], selection[Nv*K] (u8, the decode's
 * mask), first[Nv] (u32, first output row of each anchor: gsr_decode_count), update_filter[M] (u8: radii > 0 of the
 * decoded Gaussians), viewspace_grad[M,3] (gradient of the screen-space means).  Accumulates IN PLACE into the
 * model-sized opacity_accum[N], anchor_demon[N], offset_gradient_accum[N*K], offset_denom[N*K] (fp32).
 */
int gsr_training_stats(int Nv, int K, int M, const int32_t* visible, const float* neural_opacity, const uint8_t* selection,
                       const uint32_t* first, const uint8_t* update_filter, const float* viewspace_grad,
                       float* opacity_accum, float* anchor_demon, float* offset_gradient_accum, float* offset_denom,
                       void* stream);

/*
 * ---- the image-space RGB loss that follows the rasterizer ----------------------------------
 * Fused weighted L1 + weighted SSIM (11x11 Gaussian window, sigma 1.5, zero padding), value and gradient:
 *     L = a_l1 * mean(|img - gt| * m) + a_ssim * mean(ssim_map(img, gt) * m),   m = weight[H,W] (1 when NULL),
 * means over C*H*W.  Replaces GScream's utils/loss_utils.py:26-30 (l1_loss, l1_loss_masked) and :131-190 (ssim,
 * ssim_masked: five depthwise conv2d + ~10 elementwise passes each) as composed in train.py:538-545
 * (a_l1 = w (1 - lambda), a_ssim = -w lambda, the constant w lambda is added by the caller).
 * img, gt: [C,H,W] fp32 contiguous.  out3 (device): {L, mean(|d| m), mean(ssim m)}.  keep_state = 1 also stores what
 * the backward needs in the workspace (gsr_loss_workspace_bytes).  upstream: device pointer to dL/dL (NULL = 1).
 */
size_t gsr_loss_workspace_bytes(int C, int H, int W);
int gsr_rgb_loss_forward(int C, int H, int W, const float* img, const float* gt, const float* weight, float a_l1,
                         float a_ssim, void* workspace, float* out3, int keep_state, void* stream);
int gsr_rgb_loss_backward(int C, int H, int W, const float* img, const float* gt, const float* weight, float a_l1,
                          float a_ssim, const void* workspace, const float* upstream, float* dL_dimg, void* stream);
/* The same with ssim's `window_size` argument (utils/loss_utils.py:131,165; create_window :117-121): odd, 1..11 (11 = the two
 * entry points above = what GScream's trainer uses everywhere); other values -> GSR_ERR_UNSUPPORTED. */
int gsr_rgb_loss_forward_window(int C, int H, int W, const float* img, const float* gt, const float* weight, float a_l1,
                                float a_ssim, int window_size, void* workspace, float* out3, int keep_state, void* stream);
int gsr_rgb_loss_backward_window(int C, int H, int W, const float* img, const float* gt, const float* weight, float a_l1,
                                 float a_ssim, int window_size, const void* workspace, const float* upstream, float* dL_dimg,
                                 void* stream);

/*
 * The depth terms of the same loss (train.py:548-573): least-squares scale/shift alignment of the rendered depth to the
 * target over lsq_mask (utils/loss_utils.py:77-104), scale = |scale|, then
 *     L = lambda_l1 * mean(|a - y| * l1_weight) + sum_{k=0..3} 0.5 * lambda_smooth * gradient_loss(a[::2^k], y[::2^k], grad_mask[::2^k])
 * (utils/loss_utils.py:26-30, 40-49, 58-74), a = scale * depth + shift.  The gradient includes the path through the
 * fit.  depth, target, masks: [H,W] fp32 (masks may be NULL = ones).  out5 (device): {L, mean(|a-y| w), smooth part,
 * scale, shift}.  The forward keeps what the backward needs in the workspace.
 */
size_t gsr_depth_loss_workspace_bytes(int H, int W);
int gsr_depth_loss_forward(int H, int W, const float* depth, const float* target, const float* lsq_mask,
                           const float* l1_weight, const float* grad_mask, float lambda_l1, float lambda_smooth,
                           void* workspace, float* out5, void* stream);
int gsr_depth_loss_backward(int H, int W, const float* depth, const float* target, const float* lsq_mask,
                            const void* workspace, const float* upstream, float* dL_ddepth, void* stream);

/*
 * ---- SURVEY 8(f) rank 4 (init-only): simple_knn ------------------------------------------------------------------
 * mean_dist2[i] = mean of the three smallest squared fp32 distances from point i to the OTHER points.  Replaces
 * simple_knn._C.distCUDA2 (submodules/simple-knn/simple_knn.cu:185-220, spatial.cu), called once by
 * scene/gaussian_model.py at initialisation.  points: [P,3] fp32 contiguous; workspace: gsr_knn_workspace_bytes(P).
 */
size_t gsr_knn_workspace_bytes(int P);
int gsr_knn_mean_dist2(int P, const float* points, float* mean_dist2, void* workspace, void* stream);

/*
 * Optional per-stage timing (bench.py's roofline numbers; the reference has no counterpart -- its only
 * timing is two CUDA events around a whole training iteration, train.py:343-344,406,578).
 * Between gsr_profile_begin() and gsr_profile_end() every stage of every call is bracketed by a pair
 * of HIP events recorded on the stream the stage is launched on.  gsr_profile_end() waits for them and
 * returns the per-stage totals.  This is the only process-wide state in the library; off by default.
 */
int gsr_profile_begin(unsigned stage_mask /* bit i = time stage i; 0 = all */);
/* The same, timing only every `every`-th invocation of each selected stage (the first one included): an event pair costs the
 * stream a bubble on either side of the stage, so a benchmark that wants a stage's duration from inside its timed region samples. */
int gsr_profile_begin_sampled(unsigned stage_mask, unsigned every);
int gsr_profile_end(gsr_profile* out_host);
const char* gsr_stage_name(int stage);

#ifdef __cplusplus
}
#endif
#endif /* GSRASTER_H_INCLUDED */
