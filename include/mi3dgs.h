/* mi3dgs C-ABI: the MI355X (gfx950) 3D Gaussian Splatting hot path.
 *
 * This is the drop-in boundary for the one path of krishan44/pipeline-pointcloud that this
 * repository accelerates: the `Train-Stage1` component (reference
 * source/container/src/main.py:1270-1316 single GPU `ns-train splatfacto`, and
 * main.py:1318-1347 multi GPU `gsplat/examples/simple_trainer.py`).  The reference has no
 * in-process FFI for that path -- it shells out (pipeline/pipeline.py:217-222) to
 * third-party CUDA code (gsplat) whose operator surface is what each entry point below
 * replaces.  The "replaces" notes name that upstream operator and the reference call site
 * that reaches it.
 *
 * Conventions
 *   - plain C, no torch types; every pointer is a DEVICE pointer unless it says "host";
 *   - all tensors are dense, row-major, float32 unless stated; C cameras, N Gaussians;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); nothing
 *     synchronises the host, nothing allocates: scratch comes in as `workspace`;
 *   - return 0 on success, non-zero on failure with mi3dgs_last_error() describing it
 *     (thread-local).  Inputs are borrowed, outputs are written in place.
 *
 * Packed records (one 64-byte line each, so the rasteriser's gather and its gradient
 * atomics cost ONE memory request per (tile, Gaussian)):
 *   splat record  [C*N][16]: 0 x, 1 y (pixels) | 2,3,4 conic A,B,C | 5 opacity (after
 *       activation / compensation) | 6,7,8 r,g,b (after SH + 0.5, clamp >= 0) | 9 depth |
 *       10 compensation | 11..15 zero.           == gsplat means2d/conics/opacities/colors/depths
 *   grad record   [C*N][16]: 0,1 d/dxy | 2,3,4 d/dconic | 5 d/dopacity | 6,7,8 d/drgb |
 *       9,10 sum |d/dxy| (absgrad) | 11 d/ddepth (input to project_bwd) | 12..15 unused.
 */
#ifndef MI3DGS_H
#define MI3DGS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI3DGS_SPLAT_STRIDE 16
#define MI3DGS_GRAD_STRIDE 16

/* `flags` bits of project_fwd / project_bwd */
#define MI3DGS_FLAG_LOG_SCALES 1   /* scales are log-space parameters; exp() fused */
#define MI3DGS_FLAG_LOGIT_OPAC 2   /* opacities are logits; sigmoid() fused */
#define MI3DGS_FLAG_ANTIALIASED 4  /* rasterize_mode "antialiased": opacity *= compensation */
#define MI3DGS_FLAG_CLEAR_VSPLATS 8 /* project_bwd / project_bwd_adam, one camera with SH colours only: the v_splats row of every
                                      visible Gaussian is zeroed once it has been read (rasterize_bwd touches no other rows), so
                                      that the next step's rasterize_bwd finds the buffer clear without a fill pass of its own.
                                      The buffer is written through the `const` pointer: it must be writable. */

/* mi3dgs_project_bwd_adam only: the call handles only the aligned 64-Gaussian groups NONE of whose members is visible (their
 * update needs no gradient: a pure Adam stream over their parameters, which depends on mi3dgs_project_fwd's radii alone), or
 * only the groups with a visible member.  Two calls, one with each bit, update every Gaussian exactly once: the first can run on a
 * second stream under the rasterisers (which leave HBM idle) while the second waits for mi3dgs_rasterize_bwd. */
#define MI3DGS_FLAG_ONLY_CULLED_GROUPS 16
#define MI3DGS_FLAG_ONLY_VISIBLE_GROUPS 32

/* The update of the fully culled groups as a kernel of its own (a pure stream: no gradient, ~30 registers, no LDS -- it fits
 * beside the rasterisers' and the loss kernels' waves on a second stream): the same bits as mi3dgs_project_bwd_adam gives those
 * Gaussians.  params / exp_avg / exp_avg_sq: HOST arrays of 6 device pointers (group order of mi3dgs_project_bwd_adam, 16-byte
 * aligned; opacities may be NULL in all three); radii[N][2]: this step's mi3dgs_project_fwd output, one camera.  Follow it with
 * mi3dgs_project_bwd_adam(..., flags | MI3DGS_FLAG_ONLY_VISIBLE_GROUPS, ...) with the SAME scale_reg_weight / _max_ratio: in a step
 * that applies splatfacto's scale regulariser (reference main.py:1288) that is the one gradient a culled Gaussian has, and this
 * kernel forms it for the scales group (weight 0 = no regulariser in this step). */
int mi3dgs_adam_culled_groups(int N, float* const* params, float* const* exp_avg, float* const* exp_avg_sq,
                              const int32_t* radii, const float* lrs, int step, float beta1, float beta2, float eps,
                              float scale_reg_weight, float scale_reg_max_ratio, void* stream);

/* colour modes of project_fwd / project_bwd */
#define MI3DGS_COLOR_SH 0          /* sh0[N,1,3] + shN[N,15,3], degree `sh_degree` */
#define MI3DGS_COLOR_PER_GAUSSIAN 1 /* colors[N,3] */
#define MI3DGS_COLOR_PER_CAMERA 2  /* colors[C,N,3] */

/* Environment variables the PRODUCT library reads, all of them size thresholds between code paths that give the same
 * results (tests/test_cabi_cpu.py checks that the binary holds no other MI3DGS_ name):
 *   MI3DGS_OS_SMALL_KEYS      sorts up to this many keys use 2 048-key onesweep tiles (default 512 K)
 *   MI3DGS_OS_MAX_KEYS        sorts above this many keys use the classic radix passes (default 4 M)
 *   MI3DGS_EMIT_SMALL_SPLATS  up to this many Gaussians the tile emit gives a wave 16 splats instead of 64 (default 256 K)
 *   MI3DGS_KEYS16=0           never sort 16-bit tile keys
 * Everything that can return wrong results (timing experiments) and every rejected variant is compiled only into
 * libmi3dgs_exp.so (make -C csrc: -DMI3DGS_EXPERIMENTS), which the product never loads. */
const char* mi3dgs_last_error(void);
int mi3dgs_abi_version(void);      /* 7 */
int mi3dgs_splat_stride(void);
int mi3dgs_grad_stride(void);

/* Optional per-kernel profiler: while enabled, every kernel launch made by this library is
 * bracketed by HIP events recorded on the launch stream.  mi3dgs_profile_read waits for
 * them, writes one text line per kernel tag ("<tag> <launches> <total_ms>\n") into the HOST
 * buffer `out`, clears the table and returns the bytes written (0 if `cap` is too small). */
int mi3dgs_profile_enable(int on);
size_t mi3dgs_profile_read(char* out, size_t cap);

/* ---- per-Gaussian stage --------------------------------------------------------------
 * Replaces gsplat fully_fused_projection_fwd + quat_scale_to_covar_preci +
 * spherical_harmonics_fwd + the exp/sigmoid/clamp glue of gsplat.rasterization()
 * (reached via main.py:1312 `ns-train`, main.py:1343 `simple_trainer.py`).
 * viewmats[C,4,4] world->camera, Ks[C,3,3]; radii[C*N,2] int32 (0,0 = culled). */
int mi3dgs_project_fwd(int C, int N, const float* means, const float* quats, const float* scales,
                       const float* opacities /* nullable */, const float* sh0, const float* shN,
                       const float* colors, int color_mode, int sh_degree, const float* viewmats,
                       const float* Ks, int width, int height, float eps2d, float near_plane,
                       float far_plane, float radius_clip, int flags, int32_t* radii, float* splats,
                       uint32_t* depth_keys_opt, void* stream);

/* Replaces fully_fused_projection_bwd + spherical_harmonics_bwd + the autograd of the
 * glue.  v_splats is the packed gradient record written by rasterize_bwd.  Every output is
 * written once (sum over cameras inside the kernel, no atomics).  stat_* (nullable,
 * accumulated in place) are gsplat DefaultStrategy._update_state's running statistics:
 * screen-space gradient norm (from |d/dxy| when stat_use_abs), visibility count, max
 * normalised screen radius. */
int mi3dgs_project_bwd(int C, int N, const float* means, const float* quats, const float* scales,
                       const float* opacities, const float* sh0, const float* shN, int color_mode,
                       int sh_degree, const float* viewmats, const float* Ks, int width, int height,
                       float eps2d, int flags, const int32_t* radii, const float* splats,
                       const float* v_splats, float* v_means, float* v_quats, float* v_scales,
                       float* v_opacities /* nullable */, float* v_sh0, float* v_shN,
                       float* v_colors /* nullable */, float* stat_grad2d, float* stat_count,
                       float* stat_radii, int stat_use_abs, void* stream);

/* The same backward for the training path (ONE camera, SH colours) with the Adam step -- and,
 * when scale_reg_weight > 0, the splatfacto scale regulariser (reference main.py:1288) -- fused
 * in: the six parameter arrays are updated IN PLACE and no gradient array is written, which
 * removes the 944-byte-per-Gaussian gradient round trip of a separate mi3dgs_adam_step.
 * exp_avg / exp_avg_sq / lrs are HOST arrays of 6 in the group order means[3], quats[4],
 * scales[3], opacities[1], sh0[3], shN[45]; `step` is 1-based.  quats, shN and the shN
 * moments must be 16-byte aligned.  Not for the data-parallel mode (gradients must be
 * all-reduced first): use mi3dgs_project_bwd + mi3dgs_adam_step there. */
int mi3dgs_project_bwd_adam(int N, float* means, float* quats, float* scales, float* opacities,
                            float* sh0, float* shN, int sh_degree, const float* viewmats,
                            const float* Ks, int width, int height, float eps2d, int flags,
                            const int32_t* radii, const float* splats, const float* v_splats,
                            float* const* exp_avg, float* const* exp_avg_sq, const float* lrs,
                            int step, float beta1, float beta2, float eps, float scale_reg_weight,
                            float scale_reg_max_ratio, float* stat_grad2d, float* stat_count,
                            float* stat_radii, int stat_use_abs, void* stream);

/* The same for gsplat's MCMC strategy (simple_trainer mcmc / splatfacto-mcmc): its two regularisers -- opacity_reg * mean(sigmoid(o))
 * + scale_reg * mean(exp(s)), over EVERY Gaussian, every step -- are folded into the fused kernel, so that the step is one launch
 * instead of backward + mi3dgs_mcmc_regularise + mi3dgs_adam_step.  (No densify statistics: MCMC does not use them.) */
int mi3dgs_project_bwd_adam_mcmc(int N, float* means, float* quats, float* scales, float* opacities,
                                 float* sh0, float* shN, int sh_degree, const float* viewmats,
                                 const float* Ks, int width, int height, float eps2d, int flags,
                                 const int32_t* radii, const float* splats, const float* v_splats,
                                 float* const* exp_avg, float* const* exp_avg_sq, const float* lrs,
                                 int step, float beta1, float beta2, float eps, float mcmc_opacity_reg,
                                 float mcmc_scale_reg, void* stream);

/* ---- tile binning --------------------------------------------------------------------
 * Replaces gsplat isect_tiles (count + emit + cub radix sort) and isect_offset_encode.
 * Two phases so that the caller may read the intersection count back between them (exact
 * allocation) or skip the read-back and size flatten_ids / tile_keys by capacity
 * (`max_isect`): every kernel takes its live count from n_isect_dev[0] on the device. */
size_t mi3dgs_bin_workspace_bytes(int C, int N, long long max_isect);
/* `tight` = 0: bin by the splat's bounding box, exactly gsplat's tile lists (flatten_ids,
 * isect_ids, tiles_per_gauss identical to upstream).  `tight` = 1: per tile row keep only the
 * span the ellipse sigma <= ln(255 opacity) reaches -- renders and gradients are bit-identical
 * (the dropped pairs fail the rasteriser's alpha >= 1/255 test on every pixel), the lists are
 * shorter.  `height` is the image height in pixels.  Both phases must use the same value.
 * `tight` is a bit field: bit 0 as above; bit 1 (value 2) = the splat records carry their integer radii in
 * slots 11 and 12 (mi3dgs_project_fwd writes them there), so the emit pass gathers ONE line per splat instead of
 * a record plus an 8-byte radii entry from a second random sector.  Leave it clear for hand-made records.
 * Bit 2 (value 4, mi3dgs_bin_tiles only) = the caller does not read `tile_keys` back: the buffer (still max_isect words)
 * is scratch and its contents are unspecified on return.  The library then sorts 16-bit tile keys whenever every tile id
 * fits and isect_ids_opt is null; flatten_ids and isect_offsets are the same bit for bit. */
int mi3dgs_bin_count(int C, int N, const int32_t* radii, const float* splats, int tile_size,
                     int tile_width, int tile_height, int height, int tight,
                     int32_t* tiles_per_gauss /* [C*N], nullable */, int32_t* n_isect_dev /* [1] */,
                     void* workspace, size_t workspace_bytes, long long max_isect, void* stream);
/* flatten_ids[max_isect] (index into the C*N records, sorted by (camera, tile, depth)),
 * tile_keys[max_isect] (camera*tiles + tile), isect_offsets[C*tile_height*tile_width],
 * isect_ids_opt[max_isect] int64 (nullable): gsplat's (tile << 32 | depth bits) keys. */
int mi3dgs_bin_emit(int C, int N, const int32_t* radii, const float* splats, int tile_size,
                    int tile_width, int tile_height, int height, int tight,
                    int32_t* n_isect_dev /* in/out: clamped to max_isect */, long long max_isect,
                    int32_t* flatten_ids, int32_t* tile_keys, int32_t* isect_offsets,
                    int64_t* isect_ids_opt, void* workspace, size_t workspace_bytes, void* stream);

/* Both phases in one call for callers that bring a capacity (no count to read back): depth sort,
 * then counting and emission fused in one chained pass over the depth-sorted splats (no separate
 * tile_count / scan, the row tables are built once).  Outputs as above; n_isect_dev receives the
 * number of intersections CLAMPED to max_isect (so that no later kernel walks past the buffers);
 * when the true count was larger, bit 2 of the sticky word of mi3dgs_async_errors() is set and
 * the lists are truncated.  tiles_per_gauss_opt[C*N] is nullable.  depth_keys_opt[C*N]
 * (nullable): the sort keys mi3dgs_project_fwd wrote (depth bits, 0xFFFFFFFF for culled splats);
 * CONSUMED (the sort ping-pongs through the buffer).  Without it the keys are gathered from the
 * splat records. */
int mi3dgs_bin_tiles(int C, int N, const int32_t* radii, const float* splats, int tile_size,
                     int tile_width, int tile_height, int height, int tight,
                     int32_t* n_isect_dev, long long max_isect,
                     int32_t* flatten_ids, int32_t* tile_keys, int32_t* isect_offsets,
                     int64_t* isect_ids_opt, int32_t* tiles_per_gauss_opt,
                     uint32_t* depth_keys_opt,
                     void* workspace, size_t workspace_bytes, void* stream);

/* building blocks of the above, exported for reuse and for tests */
size_t mi3dgs_sort_workspace_bytes(long long n);
int mi3dgs_sort_pairs_u32(uint32_t* keys, uint32_t* vals, long long n, int nbits, void* workspace,
                          size_t workspace_bytes, void* stream); /* stable, ascending, in place */
/* A/B switch: 0 = classic radix passes (histogram + 3-kernel scan + scatter), 1 = onesweep
 * (one histogram kernel for all passes, one chained-look-back kernel per pass -- or, for small
 * sorts of at most 96 tiles (196 k keys; 786 k above the small-tile limit), ONE kernel for all passes with device-wide
 * barriers in between), 2 = onesweep up to 4 M keys, classic above (the default; see binning.hip
 * for the measurements), 3 = as 1 but always one launch per pass. */
int mi3dgs_debug_set_sort_mode(int mode);
/* A/B switch for the fused exact emit of mi3dgs_bin_tiles: 1 (default) = wave-granular (one wave = 64 depth-sorted
 * splats, no barriers), 0 = block-cooperative (round 1).  Same output bit for bit. */
int mi3dgs_debug_set_emit_mode(int mode);
/* The chained kernels (onesweep radix pass, device-wide scan, fused tile emit) wait on one another
 * with BOUNDED spins; a wait that runs out sets a bit in one device word instead of hanging the
 * GPU, and the results of that call are then wrong.  This reads (and optionally clears) the word;
 * it synchronises, so call it where the host waits anyway.  0 = every chain resolved.
 * Bits: 1 = onesweep radix pass, 2 = chained scan / fused tile emit, 4 = more tile intersections
 * than `max_isect` (lists truncated; size the capacity up). */
int mi3dgs_async_errors(uint32_t* out, int reset);
size_t mi3dgs_scan_workspace_bytes(long long n);
int mi3dgs_scan_exclusive_u32(const uint32_t* in, uint32_t* out, long long n,
                              uint32_t* total_dev /* nullable */, void* workspace,
                              size_t workspace_bytes, void* stream);

/* ---- rasteriser ----------------------------------------------------------------------
 * Replaces gsplat rasterize_to_pixels_fwd / _bwd.  tile_size must be 16.
 * render[C,H,W,3], alphas[C,H,W,1], last_ids[C,H,W] int32; backgrounds[C,3] nullable. */
int mi3dgs_rasterize_fwd(int C, int width, int height, int tile_size, int tile_width,
                         int tile_height, const float* splats, const int32_t* isect_offsets,
                         const int32_t* flatten_ids, const int32_t* n_isect_dev,
                         const float* backgrounds, float* render, float* alphas, int32_t* last_ids,
                         void* seg_workspace /* nullable */, size_t seg_workspace_bytes,
                         void* stream);
/* Segment workspace (optional, training only).  One block walks a tile's list serially in the backward, so a launch is as
 * long as its heaviest tile.  Given this workspace the forward leaves its per-pixel state (T, r, g, b) at every 256-entry
 * boundary a tile's block walks past, and the backward -- given the SAME workspace and the forward's `render` -- processes the
 * entries in front of each boundary as work items of their own (512 extra blocks looping over the list, resident from the
 * start of the launch) while the tile's block keeps the rest.  Same gradients up to f32 rounding of (final colour -
 * checkpoint colour), ~1e-5 relative.  Nothing is left for tiles that stop before their first boundary.  Where the capacity
 * exceeds 1 024 entries per tile on a grid of at least 4 096 tiles (lists long everywhere and blocks enough to fill the device:
 * nothing to balance, and every item has a fixed cost) both calls ignore the workspace.  Size: mi3dgs_raster_seg_workspace_bytes(C * tile_width * tile_height, max_isect) -- 4 KB per
 * possible boundary (max_isect / 256 of them), touched only where boundaries exist. */
size_t mi3dgs_raster_seg_workspace_bytes(int n_tiles, long long max_isect);
/* Clears the workspace's control words.  Call it once after allocating the workspace.  Every mi3dgs_rasterize_bwd given the
 * workspace leaves them clear again (its workers reset the counter they consumed), so a training loop -- forward, backward,
 * forward, backward -- never clears anything; call it again before a forward only if the forward before it was NOT followed by
 * its backward (the counter would still hold that forward's work items). */
int mi3dgs_raster_seg_workspace_init(void* seg_workspace, size_t seg_workspace_bytes, void* stream);
/* v_splats[C*N][16] must be zeroed by the caller; gradients are ACCUMULATED into it.
 * Numerics: log2 alpha of a (pixel, splat) pair is evaluated by the forward's own instruction sequence (three-term bf16
 * coefficients against an exact bf16 basis, f32 accumulation), so forward and backward take the same alpha >= 1/255 decision;
 * the per-splat sums over a tile's pixels are carried to their f32 accumulators as two bf16 terms per value (relative error
 * <= 2^-16 per term, unbiased; bench.py says so in `dtype_note`).  The all-f32 reduction of the same sums exists in the
 * experiments library only (mi3dgs_debug_set_raster_mode(3) there), as the yardstick of tests/test_gpu_configs.py. */
int mi3dgs_rasterize_bwd(int C, int width, int height, int tile_size, int tile_width,
                         int tile_height, const float* splats, const int32_t* isect_offsets,
                         const int32_t* flatten_ids, const int32_t* n_isect_dev,
                         const float* backgrounds, const float* alphas, const int32_t* last_ids,
                         const float* v_render, const float* v_alphas, int absgrad, float* v_splats,
                         long long n_gaussians /* rows of `splats` per camera: picks the kernel shape (0 = unknown) */,
                         const float* render /* the forward's output; needed with seg_workspace only */,
                         void* seg_workspace /* nullable: the one the forward of this step was given */,
                         size_t seg_workspace_bytes, void* stream);

/* Product library: accepts 1 (the MFMA rasterisers, the only ones it holds) and fails for anything else.  Experiments
 * library: 0 = round-1 all-VALU kernels, 3 = f32 reduce-scatter backward, 4 = three-term bf16 backward, 14 = wave-flush
 * backward, 21 / 22 = the product backward forced to its DEEP / WIDE shape, 11..13 = timing experiments with wrong results
 * (csrc/rasterize.hip). */
int mi3dgs_debug_set_raster_mode(int mode);

/* Process-wide.  0 (default): mi3dgs_rasterize_fwd walks every tile's list serially (and, given a segment workspace, leaves the
 * backward's checkpoints).  1: given a segment workspace it walks the lists of more than 256 entries as segments of 256 side by
 * side (four launches: plan; segments + short tiles; combine; the segments pixels stop in, once more).  Results agree to float
 * rounding (T_in * prod(1 - alpha) is associated differently; a pixel whose transmittance comes within an ulp of the 1e-4 stop may
 * end one segment early: bounded by 1e-4 in colour).  Halves the forward where lists are walked to their ends; slower where the
 * pixels saturate early, which is most of training (docs/FINDINGS_r03.md 4.2).  The backward is the same either way. */
int mi3dgs_debug_set_raster_fwd_segments(int on);

/* ---- loss ----------------------------------------------------------------------------
 * Replaces the L1 + SSIM(11x11, sigma 1.5) loss of splatfacto / simple_trainer.
 * sums[2] (zeroed by the caller; nullable when the loss VALUE of the step is not wanted) receives
 * {sum |r-t|, sum SSIM map}; the three dm_* maps [C,H,W,3] are scratch handed from fwd to bwd. */
int mi3dgs_loss_fwd(int C, int height, int width, const float* render, const float* target,
                    float* dm_dmu1, float* dm_dsigma1, float* dm_dsigma12, float* sums, void* stream);
int mi3dgs_loss_bwd(int C, int height, int width, const float* render, const float* target,
                    const float* dm_dmu1, const float* dm_dsigma1, const float* dm_dsigma12,
                    float ssim_lambda, float loss_scale, float* v_render, void* stream);
/* The same with the target straight from a uint8 image cache [C,H,W,3]: the kernels form value * scale (scale = 1 / 255)
 * themselves -- the same product mi3dgs_image_u8_to_f32 forms, hence the same results -- and the per-step conversion launch
 * of the nerfstudio-style cache (cache_images_type uint8) disappears. */
int mi3dgs_loss_fwd_u8(int C, int height, int width, const float* render, const uint8_t* target_u8, float scale,
                       float* dm_dmu1, float* dm_dsigma1, float* dm_dsigma12, float* sums, void* stream);
int mi3dgs_loss_bwd_u8(int C, int height, int width, const float* render, const uint8_t* target_u8, float scale,
                       const float* dm_dmu1, const float* dm_dsigma1, const float* dm_dsigma12,
                       float ssim_lambda, float loss_scale, float* v_render, void* stream);
/* splatfacto use_scale_regularization (reference main.py:1288): loss_sum[0] (nullable) +=
 * weight*mean(max(ratio,max_ratio)-max_ratio); v_scales (nullable, log-space) accumulated. */
int mi3dgs_scale_reg(int N, const float* scales_log, float weight, float max_ratio, float* v_scales,
                     float* loss_sum, void* stream);

/* ---- optimiser -----------------------------------------------------------------------
 * Replaces torch.optim.Adam over the Gaussian parameter groups: one launch for up to 8
 * flat segments.  The pointer arrays are HOST arrays of device pointers. */
/* grad_scale: every gradient is multiplied by it on the way in (1 for a single GPU; 1 / world behind a sum
 * reduce-scatter, reference main.py:1318-1347). */
int mi3dgs_adam_step(int nseg, float* const* params, const float* const* grads, float* const* exp_avg,
                     float* const* exp_avg_sq, const long long* numel, const float* lrs, int step,
                     float beta1, float beta2, float eps, float grad_scale, void* stream);
/* HBM yardstick for the bench line (roofline.copy_GBps_this_box): a 16-byte-per-lane stream that reads n_read arrays and
 * writes their sum to n_write arrays; buf holds (n_read + n_write) arrays of floats_per_array floats, reads first.
 * (n_read, n_write) in {(1,1) copy, (2,1), (4,3), (5,4) the mix of mi3dgs_project_bwd_adam, (1,0), (0,1)}. */
int mi3dgs_debug_hbm_stream(float* buf, long long floats_per_array, int n_read, int n_write, void* stream);

/* ---- adaptive density control --------------------------------------------------------
 * Replaces gsplat DefaultStrategy._grow_gs/_prune_gs and strategy.ops duplicate / split /
 * remove / reset_opa.  decide -> mi3dgs_scan_exclusive_u32(out_count) -> scatter.
 * Group order everywhere: means[3], quats[4], scales[3], opacities[1], sh0[3], shN[45]. */
/* stat_radii (nullable): the running maximum of radius / max(W, H) that mi3dgs_project_bwd* accumulates.  Given it, the
 * screen-size rules of gsplat's DefaultStrategy apply (grow_scale2d / prune_scale2d; nerfstudio splatfacto's split_screen_size
 * 0.05 / cull_screen_size 0.15 until stop_screen_size_at 4000 -- `ns-train splatfacto`, reference main.py:1270-1306): split
 * also where it exceeds grow_scale2d, prune (when check_too_big) also where it exceeds prune_scale2d.  The caller passes null
 * from the stop iteration on.  flags: bit 0 duplicate, bit 1 split (both: the copy and the two samples, 3 outputs), bit 2 prune. */
int mi3dgs_densify_decide(int N, const float* scales_log, const float* opacities_logit,
                          const float* stat_grad2d, const float* stat_count, const float* stat_radii,
                          float grow_grad2d, float grow_scale3d_abs, float grow_scale2d, float prune_opa,
                          float prune_scale3d_abs, float prune_scale2d, int do_grow, int check_too_big,
                          uint8_t* flags, uint32_t* out_count, void* stream);
/* n_out = the scan's total (the caller has read it back to size / swap its buffers anyway);
 * map_workspace[capacity] u32: for every output row, where it comes from (a row-gather per array follows). */
int mi3dgs_densify_scatter(int N, long long n_out, const float* const* params_in,
                           const float* const* exp_avg_in, const float* const* exp_avg_sq_in,
                           float* const* params_out, float* const* exp_avg_out,
                           float* const* exp_avg_sq_out, const uint8_t* flags, const uint32_t* offsets,
                           long long capacity, uint32_t seed, uint32_t* map_workspace, void* stream);
int mi3dgs_reset_opacity(int N, float* opacities_logit, float max_logit, float* exp_avg,
                         float* exp_avg_sq, void* stream);

/* ---- MCMC strategy ----------------------------------------------------------------------
 * Replaces gsplat MCMCStrategy's compute_relocation kernel, inject_noise_to_position and the
 * opacity / scale regularisers of `simple_trainer.py mcmc` / `splatfacto-mcmc` (reference
 * main.py:1285-1291, 1324-1327).  relocation: ACTIVATED opacities[n] / scales[n,3], ratios[n]
 * copies each, binoms[51*51] = C(n,k) table.  inject_noise: means += Sigma (randn * gate(o) *
 * scaler), gate = sigmoid(-100 (o - 0.005)).  regularise: accumulates d/d(logit, log-scale) of
 * opacity_reg*mean(sigmoid(o)) + scale_reg*mean(exp(s)). */
int mi3dgs_mcmc_relocation(int n, const float* opacities, const float* scales, const int32_t* ratios,
                           const float* binoms, float* new_opacities, float* new_scales, void* stream);
int mi3dgs_mcmc_inject_noise(int N, float* means, const float* quats, const float* scales_log,
                             const float* opacities_logit, float scaler, uint32_t seed, void* stream);
int mi3dgs_mcmc_regularise(int N, const float* opacities_logit, const float* scales_log,
                           float opacity_reg, float scale_reg, float* v_opacities, float* v_scales,
                           void* stream);

/* ---- input side ----------------------------------------------------
 * knn: exact k nearest neighbours (k <= 4, the point itself excluded by index) of points[n,3];
 * out_d2[n,k] ascending squared distances (+inf where fewer than k other points exist),
 * out_idx_opt[n,k] (nullable) their indices (-1 where missing).  Replaces the sklearn
 * NearestNeighbors call of splatfacto's / simple_trainer's scale initialisation, reached through
 * reference main.py:1271 and :1328.  No host synchronisation.
 * image_downscale_area: cv2.resize(..., INTER_AREA) of an [H,W,channels] u8 image to
 * [out_height,out_width,channels], u8 (round to nearest even) or f32 in [0,1]
 * (reference main.py:419-481 ensure_downscaled_images).  image_u8_to_f32: dst = src * scale. */
size_t mi3dgs_knn_workspace_bytes(long long n);
int mi3dgs_knn(long long n, const float* points, int k, float* out_d2, int32_t* out_idx_opt,
               void* workspace, size_t workspace_bytes, void* stream);
int mi3dgs_image_downscale_area(const uint8_t* src, int height, int width, int channels, void* dst,
                                int out_height, int out_width, int dst_is_f32, void* stream);
int mi3dgs_image_u8_to_f32(const uint8_t* src, long long n, float* dst, float scale, void* stream);
/* Lens undistortion to a pinhole image (cv2.undistort as nerfstudio's datamanager applies it before
 * training, reached through reference main.py:1303-1306): dst(u,v) = bilinear sample of src at
 * K_src * distort(K_dst^-1 (u,v)), zero outside.  k_src / k_dst / dist are HOST arrays: (fx, fy,
 * cx, cy) each; model 0 = OpenCV (k1 k2 p1 p2 k3 k4 k5 k6, missing ones zero), 1 = OpenCV fisheye
 * (k1..k4).  dst u8 (round to nearest) or f32 in [0,1]. */
int mi3dgs_image_undistort(const uint8_t* src, int height, int width, int channels, void* dst,
                           int out_height, int out_width, const float* k_src, const float* k_dst,
                           int model, const float* dist, int n_dist, int dst_is_f32, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MI3DGS_H */
