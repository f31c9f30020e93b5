/*
 * qed_splat.h -- C ABI of libqed_splat.so, the MI355X (gfx950) native replacement for the
 * render hot path behind qed_splatter/model.py get_outputs() / get_loss_dict().
 *
 * The reference has NO native boundary of its own: its hot path is the single Python call
 *     render, alpha, info = gsplat.rendering.rasterization(...)
 * at /root/reference/qed_splatter/model.py:267-288 (gsplat is a third-party CUDA package, not
 * vendored).  The entry points below are what a native FFI for that call decomposes into
 * (SURVEY.md section 8b); each cites the piece of the reference call path it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless it is named h_*; all float data is fp32
 *   - the caller (PyTorch) owns every buffer: the library never allocates or frees device
 *     memory, never synchronises the stream, reads no environment variable and keeps no global
 *     mutable state besides a thread-local error string -> re-entrant per stream, safe to capture
 *     into a hipGraph
 *   - `stream` is a hipStream_t passed as void*
 *   - return value: 0 = ok, <0 = error (see QED_E_*); qed_last_error() describes it
 *   - C = cameras, N = Gaussians, M = tile/Gaussian intersections, T = tile_w * tile_h
 *   - quaternions are wxyz; pixel centres are at (x + 0.5, y + 0.5); tile size is 16
 */
#ifndef QED_SPLAT_H
#define QED_SPLAT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QED_OK 0
#define QED_E_INVALID_ARG (-1)
#define QED_E_WORKSPACE (-2)
#define QED_E_LAUNCH (-3)
#define QED_E_UNSUPPORTED (-4)

#define QED_TILE 16             /* BLOCK_WIDTH = 16, model.py:243 */
#define QED_SPLAT_FLOATS 12     /* packed per-(camera,Gaussian) record, see qed_project_fwd */
#define QED_METRICS_WS_DOUBLES (10 * 1024)  /* workspace of qed_image_metrics */
#define QED_STEP_METRICS_WS_DOUBLES (16 * 1024)  /* workspace of qed_step_metrics */
#define QED_LOSS_SUMS_FLOATS (8 + 4 * 1024) /* sums workspace of qed_loss_reduce / qed_loss_grad */
#define QED_VSPLAT_FLOATS 16    /* packed per-(camera,Gaussian) gradient row, see qed_composite_bwd */
#define QED_SH_JAC_FLOATS 10    /* per-(camera,Gaussian) floats of qed_project_fwd's sh_jac hand-over */

/* flags of qed_project_fwd / qed_project_bwd */
#define QED_F_ANTIALIASED 1u    /* rasterize_mode == "antialiased": opacity *= compensation */
#define QED_F_LOG_SCALES 2u     /* `scales` holds log-scales: fuse torch.exp        (model.py:270) */
#define QED_F_LOGIT_OPAC 4u     /* `opacities` holds logits:  fuse torch.sigmoid    (model.py:271) */
#define QED_F_DEPTH_CHANNEL 8u  /* render_mode "RGB+D": depth is colour channel 3   (model.py:256-259) */
#define QED_F_SIGMOID_COLORS 16u /* sh_degree None path: fuse torch.sigmoid(colors) (model.py:264) */
#define QED_F_SH_GRAD_COMPACT 64u /* qed_project_bwd (one camera): v_sh0 receives the clamp-masked colour gradient
                                    (3 floats) instead of b_0 v, v_shN is not written; qed_sh_grad_from_views
                                    rebuilds all coefficient gradients from the views' colour gradients */
#define QED_F_TIGHT_TILES 32u   /* list only the tiles of the 3-sigma square that can reach alpha >= 1/255 (the
                                   rectangle of that ellipse; with qed_project_fwd's tile_masks exactly the tiles in
                                   which some pixel can): tiles_per_gauss / the sorted list become subsets of gsplat's,
                                   images and gradients are unchanged (pass `splats` -- and the masks -- to
                                   qed_bin_tiles) */
#define QED_F_CAMERA_C2W 128u   /* qed_project_fwd: `viewmats` holds camera-to-world matrices c2w[C,3,4] (OpenGL) and `Ks`
                                   the intrinsics [C,4] = (fx, fy, cx, cy); the kernel derives view matrix and K itself
                                   (get_viewmat, model.py:22-38: what qed_camera_setup does in a launch of its own) and
                                   writes them to viewmats_out[C,4,4] / Ks_out[C,3,3] for the kernels that follow */

/* Version of this C ABI: bumped whenever an entry point's argument list or a buffer layout changes incompatibly.
 *   1  rounds 1-3.
 *   2  round 4: qed_composite_fwd / qed_composite_bwd gained `t_final` in the middle of their argument lists; the row of
 *      the compact data-parallel message grew from 3 N + 16 to 3 N + 20 floats (still reported as 1 by that round's
 *      library).  A caller built against another version must not call further: the pointers would be shifted.
 *   3  round 5: qed_composite_fwd gained `tile_order` behind `tile_cost`. */
#define QED_ABI_VERSION 3
int qed_version(void);   /* == QED_ABI_VERSION of the header the library was built from */
const char* qed_last_error(void);

/* Device-side address of a pinned, mapped host allocation (hipHostGetDevicePointer): what qed_bin_tiles' host_words must
 * be given.  Host code, no launch. */
int qed_host_device_pointer(void* host, void** device);
/* size in bytes of the error/overflow status word block a caller passes as `status` (int32[4]):
 * [0] != 0 -> intersection buffer capacity exceeded (value = required M),
 * [1] != 0 -> radix-sort look-back watchdog fired. */
#define QED_STATUS_WORDS 4

/* ---- a1/a3: camera setup in one launch --------------------------------------------------------
 * get_viewmat (model.py:22-38): c2w[C,3,4] (OpenGL) -> viewmats[C,4,4] = rigid inverse after flipping
 * the y/z columns of R; intrinsics[C,4] = (fx, fy, cx, cy) -> Ks[C,3,3] (model.py:247). */
int qed_camera_setup(int32_t C, const float* c2w, const float* intrinsics, float* viewmats, float* Ks,
                     void* stream);

/* ---- K1+K2: projection + SH colour, fused ------------------------------------------------
 * Replaces, inside rasterization() (model.py:267-288): world->camera, 3D covariance from
 * quat+scale, EWA 2D covariance + eps2d blur, conic, radius, near/far/frustum culling
 * (near_plane=0.01, far_plane=1e10: model.py:279-280), SH evaluation to degree sh_degree
 * (model.py:261-262, 282) with +0.5 and clamp, and the per-Gaussian tile count.
 *
 * means[N,3] quats[N,4] scales[N,3] opacities[N] viewmats[C,4,4] Ks[C,3,3] (row major).
 * sh0 / shN: degree-0 coefficient [.,3] and higher coefficients [.,K-1,3] with row strides
 * (in floats) sh0_stride / shN_stride -- `colors[N,K,3]` is (colors, 3K, colors+3, 3K);
 * features_dc / features_rest (model.py:241) can be passed without the torch.cat.
 * sh_degree < 0: sh0 holds ready colours [N,3] (model.py:263-265).
 *
 * outputs: radii[C,N] i32, means2d[C,N,2], depths[C,N], conics[C,N,3], opac_out[C,N],
 *   colors_out[C,N,3], tiles_per_gauss[C,N] i32 and the packed record splats[C*N][12] =
 *   {x, y, conic_a, conic_b | conic_c, opacity, r, g | b, depth, ln(255 opacity), 0} read by the
 *   compositing kernels; block_sums[ceil(C*N/256)] i32 = tile counts summed per 256
 *   consecutive (camera,Gaussian) slots (input of qed_isect_scan).
 * Culled Gaussians get radius 0 and zeros everywhere.
 * tile_masks (may be NULL; with QED_F_TIGHT_TILES only): u64[C*N][2], 16-byte aligned.  Exact tile lists: word 1 of a slot
 *   is a mask whose bit l says whether tile l (row-major) of the rectangle in record slot 11 can be reached -- whether ANY
 *   pixel centre of the tile can have alpha >= 1/255, the rectangle test the compositing kernels apply to a tile's
 *   quadrants, applied to the tile; tiles_per_gauss then counts the set bits.  A rectangle of more than 64 tiles keeps
 *   every tile (mask ~0).  Word 0 repeats the count (low half) and the packed rectangle (high half), so that the emit
 *   pass reads ONE 16-byte descriptor per slot.  Hand the array to qed_bin_tiles together with `splats`.  Images and
 *   gradients do not change; the list loses ~18 % at config B.
 * sh_jac (may be NULL; used with sh_degree >= 0): QED_SH_JAC_FLOATS planes of C*N floats that the backward pass takes
 *   instead of the coefficients -- planes 0..8 = d colour_ch / d unit direction_axis at [3 axis + ch] (sum over k of
 *   d b_k / d axis * c_k,ch, before the clamp), plane 9 = the clamp mask as an integer (bit ch: colour_ch + 0.5 >= 0).
 *   Written for visible (camera, Gaussian) slots only.  With it qed_project_bwd reads 40 B per slot where it would
 *   re-read all 3 K coefficients (192 B at degree 3) and rebuild the basis derivatives beside the projection state. */
int qed_project_fwd(int32_t N, int32_t C, const float* means, const float* quats, const float* scales,
                    const float* opacities, const float* sh0, int32_t sh0_stride, const float* shN,
                    int32_t shN_stride, int32_t sh_degree, const float* viewmats, const float* Ks,
                    int32_t width, int32_t height, int32_t tile_w, int32_t tile_h, float eps2d,
                    float near_plane, float far_plane, float radius_clip, uint32_t flags,
                    int32_t* radii, float* means2d, float* depths, float* conics, float* opac_out,
                    float* colors_out, float* splats, int32_t* tiles_per_gauss, uint64_t* tile_masks,
                    int32_t* block_sums, float* viewmats_out, float* Ks_out, float* sh_jac, void* stream);

/* Backward of qed_project_fwd (autograd backward of the projection + SH part of model.py:267-288).
 * vsplat[C*N][16] = packed gradient row written by qed_composite_bwd:
 *   {v_x, v_y, |v_x|, |v_y|, v_conic_a, v_conic_b, v_conic_c, v_opacity, v_r, v_g, v_b, v_depth, 0,0,0,0}
 * Outputs (overwritten, summed over cameras): v_means[N,3] v_quats[N,4] v_scales[N,3]
 * v_opacities[N] v_sh0 (stride v_sh0_stride) v_shN (stride v_shN_stride); with
 * QED_F_LOG_SCALES / QED_F_LOGIT_OPAC / QED_F_SIGMOID_COLORS the exp / sigmoid Jacobians are
 * applied so the gradients are w.r.t. the raw parameters.  v_viewmats[C,4,4] (nullable) is
 * accumulated with atomics and must be zeroed by the caller (camera optimiser, model.py:212).
 * sh_jac (may be NULL): the planes qed_project_fwd wrote for the SAME inputs; when given (and sh_degree >= 0) sh0 / shN
 * are not read and may be NULL.  Same gradients either way (the sums run in the same order). */
int qed_project_bwd(int32_t N, int32_t C, const float* means, const float* quats, const float* scales,
                    const float* opacities, const float* sh0, int32_t sh0_stride, const float* shN,
                    int32_t shN_stride, int32_t sh_degree, const float* viewmats, const float* Ks,
                    int32_t width, int32_t height, float eps2d, uint32_t flags, const int32_t* radii,
                    const float* vsplat, float* v_means, float* v_quats, float* v_scales,
                    float* v_opacities, float* v_sh0, int32_t v_sh0_stride, float* v_shN,
                    int32_t v_shN_stride, float* v_viewmats, const float* sh_jac, void* stream);

/* ---- data-parallel exchange of the SH gradients -----------------------------------------
 * d L / d sh_k = sum over views of b_k(dir_view) * v_view (b_k the real SH basis, dir = mean - camera
 * position, v_view the clamp-masked colour gradient written by qed_project_bwd with
 * QED_F_SH_GRAD_COMPACT).  View c's colour gradients are v_views + c * view_stride ([N,3]; a view that
 * does not see a Gaussian holds zeros) and its 4x4 view matrix viewmats + c * viewmat_stride (strides in
 * floats, so that one all-gathered buffer can carry both); v_sh0 / v_shN receive scale * the sums
 * (scale = 1/n_views averages). */
/* The same clamp-masked colour gradient BEFORE qed_project_bwd has run: out[total,3] = vsplat row slots 8..10 (what
 * qed_composite_bwd accumulated) under the clamp mask qed_project_fwd kept in plane 9 of sh_jac (total = C * N slots, one
 * camera per rank in the exchange).  Lets the all-gather of the colour gradients start ahead of the projection backward. */
int qed_pack_color_grad(int32_t total, const float* vsplat, const float* sh_jac, float* out, void* stream);

int qed_sh_grad_from_views(int32_t N, int32_t n_views, const float* means, const float* viewmats,
                           int64_t viewmat_stride, const float* v_views, int64_t view_stride,
                           int32_t sh_degree, float scale, float* v_sh0, int32_t v_sh0_stride,
                           float* v_shN, int32_t v_shN_stride, void* stream);

/* ---- K3: tile intersection ------------------------------------------------------------------
 * qed_isect_scan: exclusive scan of block_sums -> block_offsets[n_blocks] and the total
 * M -> n_isect[0] (device int32).  If M > capacity, status[0] = M (emit/sort then do nothing).
 * qed_isect_emit: for every (camera,Gaussian) with radius > 0 writes one (key,value) per touched
 * tile: key = (cam << tile_bits | tile_id) << 32 | float_bits(depth), value = cam*N + n, in
 * Gaussian-index then row-major tile order (gsplat isect_tiles, behind model.py:267-288). */
int qed_isect_scan(const int32_t* block_sums, int32_t n_blocks, int32_t* block_offsets,
                   int32_t* n_isect, int64_t capacity, int32_t* status, void* stream);
int qed_isect_emit(int32_t N, int32_t C, const float* means2d, const int32_t* radii, const float* depths,
                   const int32_t* tiles_per_gauss, const int32_t* block_offsets, int32_t tile_w,
                   int32_t tile_h, int32_t tile_bits, const int32_t* n_isect, int64_t capacity,
                   uint64_t* keys, int32_t* vals, void* stream);

/* ---- K4: device radix sort of (u64 key, i32 value) pairs, LSD, 8-bit digits, stable ------------
 * Sorts the first *n_dev (device int32, <= capacity) pairs on key bits [0, end_bit).  Ping-pongs
 * between (keys,vals) and (keys_alt,vals_alt); returns 0 if the sorted result is in (keys,vals),
 * 1 if it is in (keys_alt,vals_alt), <0 on error.  `workspace` needs
 * qed_sort_workspace_bytes(capacity) bytes. */
int64_t qed_sort_workspace_bytes(int64_t capacity);
int qed_sort_pairs(uint64_t* keys, int32_t* vals, uint64_t* keys_alt, int32_t* vals_alt,
                   const int32_t* n_dev, int64_t capacity, int32_t end_bit, void* workspace,
                   int64_t workspace_bytes, int32_t* status, void* stream);

/* ---- K3+K4+K5 in one call: two-stage tile binning (what rasterization() uses) ---------------------
 * Produces exactly the sorted list of qed_isect_emit + qed_sort_pairs + qed_tile_offsets (same
 * order, ties included) with ~3x less sort traffic: (A) the C*N (camera,Gaussian) slots are radix
 * sorted by their 32 depth bits, (B) intersections are emitted in that order with the 32-bit key
 * cam|tile and STABLY sorted on the tile bits only (2 passes at 1080p instead of 6 on 64-bit keys).
 * Outputs: flatten_ids[capacity] (first M valid), offsets[C*T+1] (offsets[C*T] = M), n_isect[1],
 * and, if isect_ids != NULL, the 64-bit keys (cam|tile) << 32 | depth_bits of the sorted list.
 * Overflow (M > capacity) sets status[0] = M and leaves M = 0.  workspace: qed_bin_workspace_bytes.
 * splats (may be NULL): the records of qed_project_fwd; when given, each Gaussian's tile rectangle is
 * taken from record slot 11 (the rectangle project_fwd counted) instead of being recomputed from
 * means2d / radii -- REQUIRED when project_fwd ran with QED_F_TIGHT_TILES.
 * tile_masks (may be NULL; needs splats): qed_project_fwd's per-slot masks -- REQUIRED when project_fwd wrote them
 * (tiles_per_gauss then counts the set bits, not the rectangle).
 * mode: QED_BIN_TWO_STAGE is the pipeline above.  QED_BIN_TILE_SORT gives the same list another way: entries
 * are emitted in slot order and stably sorted on the tile bits, then one workgroup per tile sorts its run by
 * the 32 depth bits (stable, in LDS for runs of <= 2048 entries, through global scratch beyond) -- no global
 * sort of the C*N slots, 10 launches instead of 23, faster while the runs are short.  QED_BIN_AUTO picks by
 * capacity per tile.  block_sums (may be NULL): qed_project_fwd's per-256-slot sums of tiles_per_gauss, which
 * save the tile-sort pipeline one counting launch.
 * host_words (may be NULL): int32[4], 16-byte aligned, in HOST-MAPPED memory (hipHostMalloc / a pinned torch tensor);
 * the call's last list kernel stores {M, status[0], status[1], 0} there in ONE 16-byte store.  A caller that wants the
 * count without blocking sets word 0 to -1 before the call and looks at it later (M >= 0 once the store has landed):
 * neither a device-to-host copy nor an event enters the stream. */
#define QED_BIN_AUTO 0
#define QED_BIN_TWO_STAGE 1
#define QED_BIN_TILE_SORT 2
int64_t qed_bin_workspace_bytes(int64_t n_slots, int64_t capacity);
int qed_bin_tiles(int32_t N, int32_t C, const float* means2d, const int32_t* radii, const float* depths,
                  const int32_t* tiles_per_gauss, const float* splats, const uint64_t* tile_masks,
                  const int32_t* block_sums, int32_t tile_w, int32_t tile_h, int64_t capacity, int32_t mode,
                  int32_t* flatten_ids,
                  int32_t* offsets, int32_t* n_isect, uint64_t* isect_ids, void* workspace,
                  int64_t workspace_bytes, int32_t* status, int32_t* host_words, void* stream);

/* ---- K5: tile offsets --------------------------------------------------------------------------
 * offsets[C*T + 1]: offsets[t] = first sorted index whose (cam,tile) >= t; offsets[C*T] = M. */
int qed_tile_offsets(const uint64_t* sorted_keys, const int32_t* n_dev, int64_t capacity, int32_t C,
                     int32_t n_tiles, int32_t tile_bits, int32_t* offsets, void* stream);

/* ---- K6: alpha compositing forward -------------------------------------------------------------
 * One workgroup per 16x16 tile; front-to-back over the tile's run of the sorted list:
 * sigma = .5(a dx^2 + c dy^2) + b dx dy, alpha = min(.999, o e^-sigma), skip if sigma < 0 or
 * alpha < 1/255, stop (Gaussian not applied) when T(1-alpha) <= 1e-4.
 * channels = 3 (RGB) or 4 (RGB+D).  backgrounds[C,channels] may be NULL (model.py:267-288 passes
 * none).  Outputs render[C,H,W,channels], alpha[C,H,W], last_ids[C,H,W] i32.
 * t_final (may be NULL; [C,H,W]): receives every pixel's final transmittance T itself.  alpha = 1 - T loses up to
 * 3e-8 / T of it (a saturated pixel's alpha sits just below 1), and every gradient term of a pixel scales with its T:
 * qed_composite_bwd, given this image, reconstructs the transmittances from T instead of from 1 - alpha.
 * launch_flags: 0 in production.  One wave composites a whole tile, or one 8x8 quadrant of it for the
 * last tiles of a launch (finer work items fill the end of the launch); QED_CL_TILE_WAVES / _QUADRANT_WAVES
 * / _HALF_AND_HALF force one shape and QED_CL_NO_CULL turns the per-quadrant culling off -- results must
 * not change (parity tests).
 * tile_cost (may be NULL; [C*tiles][4] i32): receives, per tile and quadrant wave, the number of (Gaussian, quadrant)
 * visits + a staging term per batch -- the work predictor qed_composite_bwd orders its launch by.
 * tile_order (may be NULL; [C * tiles + 1]): a launch order for THIS kernel -- the order_ws an ordering job (qed_ssim_fwd_step,
 * or qed_composite_bwd's own) produced for an EARLIER frame of the same camera and tile grid.  The forward kernel cannot
 * know its tiles' costs in advance (the list length does not predict them), an earlier frame's counts do: heaviest tiles
 * first, 114-117 us against 122-126 at config B.  Must be a permutation of the tiles as those jobs write it (any valid
 * order gives the same image; a poor predictor only costs time).  NULL: raster order with a tail of quadrant waves. */
/* get_outputs' post-processing (model.py:295-297, 304-306) folded into the compositing kernels -- the reference-shaped route
 * then has no pass of its own over the image between the rasterizer and the loss, in either direction:
 *   qed_post_t (forward, may be NULL): rgb[C,H,W,3] = clamp(render[..., :3] + (1 - alpha) background[3], 0, 1) is written by
 *     the compositing kernel beside render; with a depth channel, depth[C,H,W] = alpha > 0 ? render[..., 3] : max render[..., 3]
 *     (one extra pass over alpha that patches the empty pixels; tile_dmax: C * tiles * 4 floats of scratch);
 *   qed_post_grad_t (backward, may be NULL): v_render / v_alpha are derived per pixel from v_rgb[C,H,W,3] and v_depth[C,H,W]
 *     (either may be NULL = no gradient) in the tile prologue: pass NULL for v_render and v_alpha; render = the forward
 *     pass's output.  Same arithmetic as qed_post_process_fwd / _bwd. */
typedef struct {
    const float* background;
    float* rgb;
    float* depth;
    float* tile_dmax;
} qed_post_t;
typedef struct {
    const float* background;
    const float* render;
    const float* v_rgb;
    const float* v_depth;
} qed_post_grad_t;
#define QED_CL_TILE_WAVES 1
#define QED_CL_QUADRANT_WAVES 2
#define QED_CL_HALF_AND_HALF 3
#define QED_CL_NO_CULL 4
#define QED_CL_ORDER_READY 8    /* qed_composite_bwd: order_ws already holds the launch order of this tile_cost
                                   (qed_ssim_fwd_step wrote it): no ordering launch in front of the kernel */
int qed_composite_fwd(int32_t C, int32_t N, const float* splats, const int32_t* flatten_ids,
                      const int32_t* offsets, int32_t width, int32_t height, int32_t tile_w,
                      int32_t tile_h, int32_t channels, const float* backgrounds, float* render,
                      float* alpha, float* t_final, int32_t* last_ids, int32_t* tile_cost,
                      const int32_t* tile_order, const qed_post_t* post, int32_t launch_flags, void* stream);

/* ---- K7: alpha compositing backward ------------------------------------------------------------
 * Back-to-front replay from last_ids; per-pixel gradients are reduced across each 64-wide wave
 * (one permlane level, the rest through LDS) and added once per (tile,Gaussian) to the
 * 64-byte row vsplat[C*N][16] (layout at qed_project_bwd; includes absgrad, model.py:284).
 * vsplat must be zeroed by the caller.
 * t_final (may be NULL; [C,H,W]): qed_composite_fwd's image of the final transmittances; with it render_alpha is not read
 * (1 - t_final is that alpha bit for bit).  NULL: T_final = 1 - render_alpha, as gsplat's backward pass forms it.
 * tile_cost (may be NULL; [C*tiles][4] i32, 16-byte aligned): what qed_composite_fwd wrote for the same list -- the
 * tiles are then handed out costliest first (greedy longest-processing-time scheduling of the launch; tiles heavier than
 * the average wave slot's whole share are dealt as four quadrant waves), ordered by one extra one-workgroup launch into
 * order_ws (C*tiles + 1 ints of scratch).  Same gradients up to the order of the float atomics. */
int qed_composite_bwd(int32_t C, int32_t N, const float* splats, const int32_t* flatten_ids,
                      const int32_t* offsets, int32_t width, int32_t height, int32_t tile_w,
                      int32_t tile_h, int32_t channels, const float* backgrounds,
                      const float* render_alpha, const float* t_final, const int32_t* last_ids,
                      const float* v_render, const float* v_alpha, float* vsplat, const int32_t* tile_cost, int32_t* order_ws,
                      const qed_post_grad_t* post, int32_t launch_flags, void* stream);

/* ---- K8: fused image-space loss + gradient ------------------------------------------------------
 * Collapses model.py:295-297 (background composite + clamp), :304-306 (depth fix-up), :87-116
 * (masked depth-L1, depth_lambda) and the L1 part of the parent's RGB loss into one pass.
 * Pass 1 (qed_loss_reduce) fills per-workgroup partials in sums[QED_LOSS_SUMS_FLOATS] (only {n_valid,
 * max depth} are needed before gradients; the two loss sums are accumulated by pass 2, which leaves
 * sums[0..3] = {sum|rgb-gt|, sum|d-dgt|, n_valid, max depth}); pass 2 (qed_loss_grad) writes v_render[H,W,channels] and v_alpha[H,W] for
 *   loss = rgb_weight * mean|rgb - gt| + depth_lambda * sum|d - dgt| / n_valid
 * and the scalar losses -> losses[0..2] = {rgb term, depth term, their sum}.  mask[H,W] may be
 * NULL; when given it multiplies the rendered AND the ground-truth image before the L1 (and the SSIM)
 * term, as the parent's get_loss_dict does behind model.py:83-85, and both depths (model.py:93-97).
 * An additional term on the same clamped colour (the SSIM part of the
 * parent's loss, qed_ssim_* below) enters through v_rgb_extra[H,W,3] = its gradient w.r.t. rgb and
 * extra_sum[extra_n] (qed_ssim_fwd's per-workgroup partials): losses[0] += extra_offset + extra_scale * sum(extra_sum);
 * both pointers may be NULL. */
int qed_loss_reduce(int32_t n_pix, int32_t channels, const float* render, const float* alpha,
                    const float* background, const float* gt_rgb, const float* gt_depth,
                    const float* mask, float* sums, void* stream);
int qed_loss_grad(int32_t n_pix, int32_t channels, const float* render, const float* alpha,
                  const float* background, const float* gt_rgb, const float* gt_depth,
                  const float* mask, const float* sums, float rgb_weight, float depth_lambda,
                  float* v_render, float* v_alpha, float* losses, const float* v_rgb_extra,
                  const float* extra_sum, int32_t extra_n, float extra_scale, float extra_offset, void* stream);

/* ---- SSIM term of the parent's RGB loss (reached from model.py:83-85) ----------
 * pytorch_msssim semantics: data_range 1, 11-tap Gaussian window (sigma 1.5) applied separably with
 * no padding, K = (0.01, 0.03), mean over the (H-10) x (W-10) map and 3 channels.
 * pred is either a plain [H,W,3] image (alpha == NULL) or the compositor's render[H,W,channels]
 * together with alpha[H,W] and background[3], in which case the colour clamp(render + (1-alpha) bg)
 * of model.py:296-297 is formed on the fly.  qed_ssim_fwd writes one partial sum of the SSIM map per workgroup into
 * ssim_sum[qed_ssim_sum_floats(H, W)] (SSIM = sum of them / (3 (H-10)(W-10)); nothing to zero, no same-address
 * atomics) and, unless maps == NULL (value only: the rgb_ssim metric), the
 * coefficient maps (qed_ssim_maps_floats floats) that
 * qed_ssim_bwd turns into v_pred[H,W,3] = scale * (scale_dev ? scale_dev[0] : 1) * d ssim_sum / d colour
 * (scale_dev: an upstream gradient that lives in device memory).  mask[H,W] (may be NULL) multiplies
 * both images before the SSIM, as the parent's loss does; v_pred is the gradient w.r.t. the colour
 * BEFORE that multiply. */
int64_t qed_ssim_maps_floats(int32_t height, int32_t width);
int64_t qed_ssim_sum_floats(int32_t height, int32_t width);
int qed_ssim_fwd(int32_t height, int32_t width, int32_t channels, const float* pred, const float* alpha,
                 const float* background, const float* gt_rgb, const float* mask, float* maps,
                 float* ssim_sum, void* stream);
int qed_ssim_bwd(int32_t height, int32_t width, int32_t channels, const float* pred, const float* alpha,
                 const float* background, const float* gt_rgb, const float* mask, const float* maps,
                 float scale, const float* scale_dev, float* v_pred, void* stream);
/* qed_ssim_fwd of the fused training step, with two optional PASSENGERS riding in the same launch instead of in launches of
 * their own on the step's critical chain:
 *  - tile_cost != NULL: one extra workgroup sorts the tiles of the compositing backward that follows by the cost
 *    qed_composite_fwd counted (tile_cost[n_tiles][4], n_tiles = C * tile_w * tile_h) and leaves the launch order in
 *    order_ws[n_tiles + 1] -- pass that order_ws to qed_composite_bwd with QED_CL_ORDER_READY;
 *  - sums != NULL: pass 1 of the image loss, exactly qed_loss_reduce(height * width, channels, render, ..., gt_depth, mask,
 *    sums), as extra workgroups (no separate qed_loss_reduce call then).
 * render / alpha / background as in qed_ssim_fwd's composite mode (alpha must be given). */
int qed_ssim_fwd_step(int32_t height, int32_t width, int32_t channels, const float* render, const float* alpha,
                      const float* background, const float* gt_rgb, const float* mask, float* maps,
                      float* ssim_sum, const int32_t* tile_cost, int64_t n_tiles, int32_t* order_ws,
                      const float* gt_depth, float* sums, void* stream);

/* qed_ssim_bwd and qed_loss_grad in ONE launch (the fused training step), after qed_ssim_fwd and qed_loss_reduce on
 * the same buffers: every thread of the SSIM backward pass finishes its pixels on the spot -- SSIM gradient (scale
 * ssim_scale = -ssim_lambda / (3 (H-10)(W-10))) + L1 gradient through the clamp and the background composite
 * (model.py:296-297) -> v_render[H,W,channels] / v_alpha[H,W], depth-L1 gradient (model.py:87-116, :304-306) into channel
 * 3 -- instead of handing v_pred to a second pass that re-reads render, alpha and the ground truth (116 MB at 1080p).
 * losses[0..2] as qed_loss_grad with extra_sum = ssim_sum, extra_scale = ssim_scale, extra_offset = ssim_offset (=
 * ssim_lambda); sums is qed_loss_reduce's workspace (which also zeroes the slots this pass adds its loss sums to).
 * zero_buf (may be NULL; 16-byte aligned, zero_floats a multiple of 4): a buffer the same launch zeroes -- the vsplat
 * accumulator qed_composite_bwd needs zeroed, which saves the fill launch in front of that kernel.
 * tick (may be NULL): the optimiser's device-resident step state is advanced by the one-workgroup fold launch of this call
 * (what qed_adam_step_dev / qed_adam_step_sh otherwise do in a one-thread launch of their own: this call sits between
 * the previous step's Adam launches and this step's); follow with qed_adam_step_sh(parts | QED_ADAM_PART_TICKED). */
typedef struct {
    float* dev_state;        /* as qed_adam_step_dev */
    float beta1, beta2;
    float* dev_lr_slot;      /* NULL, or &dev_lr[scheduled group]: receives the exponential decay of qed_lr_exp_decay_dev */
    float lr_init, lr_final;
    int32_t max_steps;
    const int32_t* skip_flag; /* NULL, or as qed_adam_step's: non-zero -> the state is not advanced */
} qed_adam_tick_t;
int qed_loss_grad_ssim(int32_t height, int32_t width, int32_t channels, const float* render, const float* alpha,
                       const float* background, const float* gt_rgb, const float* gt_depth, const float* mask,
                       const float* maps, float* sums, float rgb_weight, float depth_lambda, float ssim_scale,
                       float* v_render, float* v_alpha, float* losses, const float* ssim_sum,
                       int32_t ssim_sum_n, float ssim_offset, float* zero_buf, int64_t zero_floats,
                       const qed_adam_tick_t* tick, void* stream);

/* qed_ssim_bwd and qed_image_losses_bwd in ONE launch (get_loss_dict's backward): v_rgb[H,W,3] = g_main[0] * d main_loss /
 * d rgb (the L1 term joins the SSIM term inside the SSIM backward pass, which has the pixel's colours at hand) and
 * v_depth[H,W] (may be NULL) = g_depth[0] * d depth_loss / d depth.  sums: qed_image_losses_fwd's; maps: qed_ssim_fwd's;
 * ssim_scale = -ssim_lambda / (3 (H-10)(W-10)); g_depth may be NULL (no upstream gradient: zeros).  zero_buf (may be
 * NULL): as qed_loss_grad_ssim's -- the accumulator the compositing backward that follows adds into. */
int qed_image_losses_ssim_bwd(int32_t height, int32_t width, const float* rgb, const float* depth,
                              const float* gt_rgb, const float* gt_depth, const float* mask, const float* maps,
                              const float* sums, float rgb_weight, float depth_lambda, float ssim_scale,
                              const float* g_main, const float* g_depth, float* v_rgb, float* v_depth,
                              float* zero_buf, int64_t zero_floats, void* stream);

/* ---- the same arithmetic behind the reference's OWN call sequence --------------------------------
 * get_outputs() returns images and get_loss_dict() turns them into a dict of scalar losses that the
 * trainer sums and differentiates; each half is one autograd node on the host side.
 *
 * qed_post_process_fwd/bwd = model.py:295-297 + 304-306:
 *   rgb[H,W,3] = clamp(render[..., :3] + (1 - alpha) background, 0, 1)
 *   depth[H,W] = alpha > 0 ? render[..., 3] : max(render[..., 3])      (channels == 4; max is detached)
 * workspace: QED_LOSS_SUMS_FLOATS floats.  bwd: v_rgb / v_depth (either may be NULL) -> v_render, v_alpha.
 *
 * qed_image_losses_fwd/bwd = the parent's main loss (behind model.py:83-85) + the depth term (:87-116):
 *   losses[0] = rgb_weight * mean|m rgb - m gt| + extra_offset + extra_scale * sum(extra_sum[0 .. extra_n))
 *   losses[1] = depth_lambda * sum|m d - m dgt| / n_valid   (finite & dgt > 0; 0 when nothing is valid)
 * with extra_* the SSIM term (qed_ssim_fwd on the same images).  depth / gt_depth / mask may be NULL.
 * sums: QED_LOSS_SUMS_FLOATS floats, kept for the backward.  bwd: v_rgb = g_main[0] * d losses[0] / d rgb
 * (accumulate != 0: added to what qed_ssim_bwd left there), v_depth = g_depth[0] * d losses[1] / d depth;
 * g_main / g_depth are DEVICE scalars (NULL = 0): the trainer may weight or scale each term. */
int qed_post_process_fwd(int32_t n_pix, int32_t channels, const float* render, const float* alpha,
                         const float* background, float* rgb, float* depth, float* workspace, void* stream);
int qed_post_process_bwd(int32_t n_pix, int32_t channels, const float* render, const float* alpha,
                         const float* background, const float* v_rgb, const float* v_depth,
                         float* v_render, float* v_alpha, void* stream);
int qed_image_losses_fwd(int32_t n_pix, const float* rgb, const float* depth, const float* gt_rgb,
                         const float* gt_depth, const float* mask, float rgb_weight, float depth_lambda,
                         const float* extra_sum, int32_t extra_n, float extra_scale, float extra_offset,
                         float* sums, float* losses, void* stream);
int qed_image_losses_bwd(int32_t n_pix, const float* rgb, const float* depth, const float* gt_rgb,
                         const float* gt_depth, const float* mask, const float* sums, float rgb_weight,
                         float depth_lambda, const float* g_main, const float* g_depth, int32_t accumulate,
                         float* v_rgb, float* v_depth, void* stream);

/* ---- per-step evaluation metrics (model.py:120-197, metrics.py:84-156) ----------
 * One streaming pass, results left in DEVICE memory (the reference synchronises ~12 times per step
 * with .item()).  out[10] = {rgb_mse, rgb_psnr, depth_abs_rel, depth_sq_rel, depth_rmse,
 * depth_rmse_log, depth_a1, depth_a2, depth_a3, n_valid_depth}; the depth entries are NaN when no
 * pixel is valid (metrics.py:134-143) or when pred_depth/gt_depth are NULL; the rgb entries are NaN
 * when pred_rgb/gt_rgb are NULL.  pred_rgb, gt_rgb: [n_pix,3]; depths: [n_pix]; valid depth =
 * finite(pred) & finite(gt) & gt > tolerance (0.1 in the reference).  workspace: QED_METRICS_WS_DOUBLES
 * doubles (per-workgroup partial sums; no zeroing needed).
 * rgb_ssim is qed_ssim_fwd's value; LPIPS (pretrained network) is not provided. */
int qed_image_metrics(int32_t n_pix, const float* pred_rgb, const float* gt_rgb, const float* pred_depth,
                      const float* gt_depth, float tolerance, double* workspace, float* out, void* stream);
/* Everything get_metrics_dict (model.py:120-197) needs from one training step's images in ONE streaming pass + ONE fold,
 * with what get_loss_dict (model.py:73-118) needs from the same images riding along -- two launches instead of the eight
 * short ones of qed_image_metrics + the SSIM map's sum + qed_nanmean_exp + qed_image_losses_fwd:
 *   out[0..9]  as qed_image_metrics;
 *   out[10]    rgb_ssim = ssim_norm * sum(ssim_sum[0 .. ssim_n))  (qed_ssim_fwd's partials; ssim_norm = 1 / (3 (H-10)(W-10));
 *              NaN when ssim_sum is NULL);
 *   out[11]    avg_min_scale = nanmean_i exp(scales[i * scale_stride]), i < n_scales (model.py:192-194; NaN when NULL);
 *   losses[3], loss_sums[QED_LOSS_SUMS_FLOATS] (both may be NULL): exactly what qed_image_losses_fwd(rgb_weight,
 *              depth_lambda, extra = the SSIM term with weight ssim_lambda, mask = loss_mask) would write for the same
 *              images -- get_loss_dict then launches nothing in its forward pass.
 * workspace: QED_STEP_METRICS_WS_DOUBLES doubles. */
int qed_step_metrics(int32_t n_pix, const float* pred_rgb, const float* gt_rgb, const float* pred_depth,
                     const float* gt_depth, float tolerance, const float* ssim_sum, int32_t ssim_n, float ssim_norm,
                     const float* scales, int32_t n_scales, int32_t scale_stride, const float* loss_mask,
                     float rgb_weight, float depth_lambda, float ssim_lambda, float* loss_sums, float* losses,
                     double* workspace, float* out, void* stream);
/* out[0] = nanmean_i exp(x[i * stride]), i < n (NaN when nothing is left) -- the "avg_min_scale" entry of
 * get_metrics_dict (model.py:192-194).  workspace: QED_METRICS_WS_DOUBLES doubles. */
int qed_nanmean_exp(int32_t n, const float* x, int32_t stride, double* workspace, float* out, void* stream);

/* ---- densification / culling ---------------------------------------------------
 * GPU side of the parent class's callbacks that consume model.py:249,289-292 (self.xys.absgrad,
 * self.radii, self.last_size): SplatfactoModel.after_train / refinement_after / split_gaussians /
 * dup_gaussians / cull_gaussians / dup_in_all_optim / remove_from_all_optim, restated in
 * oracle/densify_oracle.py.  All buffers are the flat [means | scales | quats | opacities |
 * features_dc | features_rest] layout of this package.
 *
 * qed_densify_accumulate (every step): for radii > 0: vis_counts += 1, xys_grad_norm +=
 *   |absgrad| (absgrad = [N] rows of 2 floats, stride_floats apart), max_2Dsize = max(., radii *
 *   inv_max_dim) with inv_max_dim = 1 / max(H, W).
 * qed_densify_classify: flags[N] (bit0 split, bit1 duplicate, bit2 keep old, bit3 keep children,
 *   bit4 keep duplicate), pos[4][N] exclusive scans {split rank, kept-old slot, kept-children slot,
 *   kept-duplicate slot} followed by 4 * ceil(N/256) ints of scratch (qed_densify_pos_ints), totals[4] = {n_split, K_old, K_child, K_dup} (device).  densify = 0 is the
 *   cull-only pass after stop_split_at.  Negative split_screen_size / cull_scale_thresh /
 *   cull_screen_size switch the corresponding test off (step-dependent in the reference).
 * qed_densify_emit: writes the N' = K_old + n_samples * K_child + K_dup rows of the new parameter and
 *   Adam-moment buffers in the reference's order [kept old | children (sample-major) | duplicates];
 *   children: mean + R(q/|q|) (exp(scale) * samples[s * n_split + split_rank]), scale = log(exp(s)/1.6),
 *   zero moments.  h_totals = totals copied to the host (the caller needs N' to allocate).
 * qed_densify_reset_opacity: opacities = min(opacities, max_logit), opacity moments = 0. */
int qed_densify_accumulate(int32_t N, const float* absgrad, int32_t stride_floats, const int32_t* radii,
                           float inv_max_dim, float* xys_grad_norm, float* vis_counts, float* max_2Dsize,
                           void* stream);
int64_t qed_densify_pos_ints(int32_t N);
int qed_densify_classify(int32_t N, const float* scales, const float* opacities, const float* xys_grad_norm,
                         const float* vis_counts, const float* max_2Dsize, int32_t densify,
                         float half_max_dim, float densify_grad_thresh, float densify_size_thresh,
                         float split_screen_size, float cull_alpha_thresh, float cull_scale_thresh,
                         float cull_screen_size, uint8_t* flags, int32_t* pos, int32_t* totals, void* stream);
int qed_densify_emit(int32_t N, int32_t n_samples, const uint8_t* flags, const int32_t* pos,
                     const int32_t* h_totals, const float* samples, const float* old_params,
                     const float* old_exp_avg, const float* old_exp_avg_sq, const int64_t* h_old_begin,
                     float* new_params, float* new_exp_avg, float* new_exp_avg_sq,
                     const int64_t* h_new_begin, void* stream);
int qed_densify_reset_opacity(int32_t N, float* opacities, float* exp_avg, float* exp_avg_sq,
                              float max_logit, void* stream);

/* ---- depth back-projection for the initial point cloud --------------------------
 * Geometry of qed-init-pc's backproject_frame (create_init_pointcloud.py:148-196): every stride-th pixel
 * (u, v) with a finite depth 0 < d < depth_max becomes the world point c2w * ((u-cx) d/fx, (v-cy) d/fy, d)
 * (Open3D create_from_depth_image with depth_scale 1).  h_c2w_opengl: HOST pointer to the OpenGL
 * camera-to-world pose, row-major with row stride 4 (a 3x4 or 4x4 matrix); the OpenCV flip of
 * _opengl_c2w_to_opencv_w2c (:61-70) is applied inside.  points[capacity,3] receives the points in
 * row-major pixel order, n_points[1] their number; more than `capacity` sets status[0] to the number
 * needed and n_points to 0.  workspace: qed_backproject_workspace_ints ints. */
int64_t qed_backproject_workspace_ints(int32_t height, int32_t width, int32_t stride);
int qed_backproject_depth(int32_t height, int32_t width, const float* depth, float fx, float fy, float cx,
                          float cy, const float* h_c2w_opengl, float depth_max, int32_t stride,
                          int64_t capacity, float* points, int32_t* n_points, int32_t* workspace,
                          int32_t* status, void* stream);

/* ---- fused multi-tensor Adam over one flat parameter buffer (config.py:44-68) --
 * n_groups contiguous segments; segment g covers elements [h_group_begin[g], h_group_begin[g+1])
 * and uses learning rate h_lr[g].  bias corrections use `step` (1-based).  The betas are doubles: 1 - beta is
 * formed in double and rounded once, as torch.optim.Adam's Python scalars are (exp_avg_sq then agrees to 1e-6).
 * skip_flag (may be NULL; all three Adam entry points and qed_adam_tick_t): a DEVICE word; when it is non-zero
 * the launch updates nothing.  Pass qed_bin_tiles' status[0]: a frame whose intersection list overflowed its
 * buffer renders empty, and a step enqueued behind it without a host round trip must not train on it. */
int qed_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                  int32_t n_groups, const int64_t* h_group_begin, const float* h_lr, double beta1,
                  double beta2, float eps, int32_t step, const int32_t* skip_flag, void* stream);

/* Same update with the step state and learning rates in DEVICE memory, so that a captured hipGraph
 * can be replayed: dev_state[3] = {step, 1/(1-beta1^step), 1/sqrt(1-beta2^step)} is advanced by one
 * (zero it before the first step), dev_lr[8] holds the per-group learning rates. */
int qed_adam_step_dev(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                      int32_t n_groups, const int64_t* h_group_begin, const float* dev_lr, double beta1,
                      double beta2, float eps, float* dev_state, const int32_t* skip_flag, void* stream);

/* Adam step that never materialises the SH-coefficient gradients.  Every group is [N, width]; the LAST
 * two must be features_dc [N,3] and features_rest [N,KR,3].  Their gradient is the rank-1 product
 *   g[n][k][c] = scale * sum_{view} b_k(direction of Gaussian n from that view's camera) * v_views[view][n][c]
 * (what qed_sh_grad_from_views would write), evaluated on the fly from the 3 N clamp-masked colour
 * gradients qed_project_bwd leaves with QED_F_SH_GRAD_COMPACT; `grads` is read for the other groups only
 * (v_views may point into it).  `means` are the positions the views were rendered with (normally the
 * means group of `params`: the SH launch, which reads them, precedes the launch that updates them).
 * State: either device (dev_state + dev_lr as qed_adam_step_dev; h_lr and step ignored; with
 * sched_group >= 0 the same tick launch first sets dev_lr[sched_group] to the exponential decay of
 * qed_lr_exp_decay_dev for the step about to be taken) or host (h_lr + 1-based step; dev_* NULL,
 * sched_group < 0).  Saves, for one camera per step at 500 k Gaussians, 96 MB written + 96 MB read; for
 * data-parallel steps also the rebuild pass.
 * `parts`: QED_ADAM_PART_SH (the tick + the two SH groups: needs the gathered views only), QED_ADAM_PART_LEADING
 * (the other groups: needs `grads`), or both (3).  A data-parallel step issues the SH part while the all-reduce of
 * the leading groups' gradients is still on the wire, then the leading part -- in that order (see `means`), with the
 * same `step`; with device state only the SH part advances it. */
#define QED_ADAM_PART_SH 1
#define QED_ADAM_PART_LEADING 2
#define QED_ADAM_PART_TICKED 4   /* device state already advanced for this step (qed_loss_grad_ssim's tick) */
int qed_adam_step_sh(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                     int32_t n_groups, const int64_t* h_group_begin, const float* h_lr, float* dev_lr,
                     double beta1, double beta2, float eps, int32_t step, float* dev_state,
                     int32_t sched_group, float sched_lr_init, float sched_lr_final,
                     int32_t sched_max_steps, int32_t N, int32_t sh_degree, const float* means,
                     int32_t n_views, const float* viewmats, int64_t viewmat_stride,
                     const float* v_views, int64_t view_stride, float scale, int32_t parts,
                     const int32_t* skip_flag, void* stream);

/* ExponentialDecayScheduler of one group (the reference schedules "means": 1.6e-4 -> 1.6e-6 over
 * 30000 steps, config.py:46-51) from the device step counter, for graph replay: dev_lr_slot[0] =
 * exp((1-t) log lr_init + t log lr_final), t = clip(dev_state[0] / max_steps, 0, 1).  Call it before
 * qed_adam_step_dev (dev_state[0] is then the 0-based index of the step about to be taken). */
int qed_lr_exp_decay_dev(float* dev_lr_slot, const float* dev_state, float lr_init, float lr_final,
                         int32_t max_steps, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* QED_SPLAT_H */
