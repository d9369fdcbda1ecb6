/*
 * jt_render.h -- C ABI of the MI355X (gfx950) TensoRF-VM joint pose + radiance-field renderer.
 *
 * The reference (Nemo1999/Joint-TensoRF) is pure Python/PyTorch and has no FFI of its own
 * (SURVEY.md §8(b)); its seam is the duck-typed scene class reached through
 * `getattr(tensorf_repr, opt.arch.tensorf.model)` (model/tensorf.py:375-397) and its
 * `forward(...)` (model/tensorf_repr/batBase.py:44-165).  This header is the boundary a
 * maintainer binds instead of the stock torch ops on that path; every entry point names the
 * reference code it replaces (paths relative to the reference repo root).
 *
 * Conventions
 *   - plain C: raw DEVICE pointers, sizes, a hipStream_t passed as void*; no torch types.
 *   - every function returns 0 on success, JT_ERR_* (>0) on bad arguments, or the negated
 *     hipError_t of a failed launch.  No exceptions, no hidden allocation, no internal
 *     threads, no host synchronisation: re-entrant per stream and hipGraph-capturable.
 *   - VM factors are CHANNEL-LAST: plane i is [H_i][W_i][C] floats (logical torch tensor
 *     [1,C,H,W] = [1,C,g[m1],g[m0]], tensoRF.py:165), line i is [L_i][C] (logical [1,C,g[v],1]).
 *     matMode = {{0,1},{0,2},{1,2}}, vecMode = {2,1,0} (tensorBase.py:405-406).
 *   - rays are [R][3] fp32 (origins, un-normalised directions), outputs [R][3] / [R].
 *   - gradient buffers are ACCUMULATED into (atomics); the caller zeroes them.
 */
#ifndef JT_RENDER_H
#define JT_RENDER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever a struct layout, an argument list or what an argument must point to changes (round 4 added
 * JtScene.near_plane_dev and grew the regulariser scratch to 640 floats without bumping it: a caller built against the old
 * header would have handed over a shorter struct).  jt_version() returns the value the LIBRARY was built with; the Python
 * binding refuses to load a library whose version differs from the one it was written against (joint_tensorf_amd/_lib.py:
 * JT_ABI_VERSION).  include/jt_render.abi holds a hash of every prototype, struct and constant of this header next to the
 * version it was taken at: tests/test_abi.py fails when the hash changes without this number changing (round 5 added
 * jt_reg_losses_fused, removed jt_pose_fused* and redefined matrix-mode bit 2 at version 1100; 1200 = round 6: those changes,
 * jt_shade_lean_tape / jt_shade_set_lean_tape, the workspace no longer carries the tile lists unless that variant is selected;
 * 1201: + jt_chip_geometry; 1202: + jt_shade_workspace_layout; 1203: + jt_march_forward_pose / jt_march_backward_pose; 1204: +
 * jt_lattice_indices.  Additions bump the last two digits, anything a caller built against the old header would get wrong bumps the
 * hundreds). */
#define JT_VERSION 1204

#define JT_OK 0
#define JT_ERR_ARG 1         /* null pointer / bad size */
#define JT_ERR_UNSUPPORTED 2 /* shape outside what the kernels are instantiated for */

#define JT_ACT_SOFTPLUS 0 /* softplus(x + shift)  tensorBase.py:696-698 */
#define JT_ACT_RELU 1     /* relu(x + shift)      tensorBase.py:699-700 */

#define JT_MLP_FEA 0      /* MLPRender_Fea          tensorBase.py:101-126 */
#define JT_MLP_WEAKVIEW 1 /* MLPRender_Fea_WeakView tensorBase.py:180-214 */

#define JT_SHADE_POSE_ONLY 2  /* jt_shade_forward flags: the backward of this forward will be called without
                                 g_factors and g_mlp (test-time pose optimisation); only the records that
                                 backward reads are written */
#define JT_SHADE_SKIP_WGRAD 1 /* jt_shade_backward flags: leave out the MLP/basis weight-gradient pass
                                 (kernel timing probes only; g_mlp is then not written) */

/* Scene / call description: 4-byte fields only (mirrored 1:1 by ctypes in joint_tensorf_amd/_lib.py).
 * Replaces the TensorBase attributes set in tensorBase.py:430-488 plus the per-call keyword
 * arguments of BatBase.forward (batBase.py:44). */
typedef struct JtScene {
  float aabb_lo[3];
  float aabb_hi[3];
  int32_t plane_h[3];   /* rows  of plane i as stored (g[m1]; swapped when the reference's    */
  int32_t plane_w[3];   /* cols  non-cubic blur quirk is reproduced)         */
  int32_t line_len[3];  /* g[v]                                                               */
  int32_t n_comp_density; /* channels of every density plane/line (16 in both BAT yamls)      */
  int32_t n_comp_app;     /* channels of every appearance plane/line (48 Blender, 20 LLFF)    */
  float step_size;      /* tensorBase.py:484                                                  */
  float near_plane;     /* near_far[0]                                                        */
  float far_plane;      /* near_far[1]                                                        */
  float distance_scale; /* batBase.py:122                                                     */
  float density_shift;
  int32_t density_act;  /* JT_ACT_*                                                           */
  float weight_thres;   /* rayMarch_weight_thres, batBase.py:127                              */
  int32_t n_samples;    /* N_samples                                                          */
  int32_t ndc;          /* 1: sample_ray_ndc semantics (tensorBase.py:554-571, batBase.py:61-66) */
  int32_t white_bg;     /* resolved `white_bg or (is_train and coin<0.5)` batBase.py:154      */
  int32_t app_dim;      /* basis_mat rows (27 / 20)                                           */
  int32_t mlp_kind;     /* JT_MLP_*                                                           */
  int32_t mlp_hidden;   /* featureC (64 / 32)                                                 */
  int32_t view_pe;
  int32_t fea_pe;
  float view_pe_progress;
  float fea_pe_progress;
  /* alpha-mask volume (AlphaGridMask, tensorBase.py:80-98), used when JtFactors.alpha_volume != NULL:
   * samples whose trilinear mask value is not > 0 are dropped (batBase.py:76-82). */
  int32_t mask_dims[3]; /* the volume's gridSize (x, y, z); the tensor is [z][y][x]                      */
  float mask_lo[3];     /* its own box: aabb[0]                                                         */
  float mask_inv[3];    /* 1.0 / (aabb[1] - aabb[0]) * 2, rounded like AlphaGridMask.invgridSize        */
  /* optional (NULL = use near_plane): one float in DEVICE memory holding near_far[0] for the depth map's
   * "- near_far[0] + 0.05" (batBase.py:147-150).  A launch captured into a hipGraph keeps reading the current
   * value while tensorf_near_plane_schedule (model/tensorf.py:230-232) moves the near plane between replays. */
  const float* near_plane_dev;
} JtScene;

/* the 12 VM factor tensors (or their gradients) */
typedef struct JtFactors {
  float* density_plane[3];
  float* density_line[3];
  float* app_plane[3];
  float* app_line[3];
  const float* alpha_volume; /* optional alpha-mask volume [z][y][x] (0 / 1 floats); NULL = no mask; ignored in
                                gradient structs */
} JtFactors;

/* basis_mat.weight [app_dim][3*n_comp_app] and the render MLP (torch Linear layout [out][in]) */
typedef struct JtMlp {
  float* basis;
  float* w1;
  float* b1;
  float* w2;
  float* b2;
  float* w3;
  float* b3;
} JtMlp;

int jt_version(void);
/* {compute units, XCDs} of the current device as the library sees them (hipDeviceGetAttribute, once per device).  The
 * persistent kernels -- shade forward / backward chain / scatter, the density walk -- run one workgroup per CU or a fixed
 * fraction of the CUs: their grids are derived from these two numbers (256 and 8 on a full MI355X; 32 and 1 on a CPX
 * partition), not from constants. */
int jt_chip_geometry(int32_t* out2);

/* ---------------------------------------------------------------------------------------------
 * Ray generation for the SAMPLED pixels only.
 * Replaces camera.get_center_and_ray (camera.py:231-261) + `[:,ray_idx]` (model/tensorf.py:159-161)
 * and, when `ndc` != 0, camera.convert_NDC (camera.py:303-340).
 *   pose [B][12] (3x4 row-major, world->camera), intr_inv [B][9], intr [B][9] (NDC only, may be
 *   NULL otherwise), ray_idx [r] int64 flat pixel index y*W+x (same lattice for every view).
 *   out: rays_o, rays_d [B*r][3].
 * backward: g_rays_o/g_rays_d [B*r][3] -> g_pose [B][12] (overwritten, deterministic per view). */
int jt_raygen_forward(const float* pose, const float* intr_inv, const float* intr, const int64_t* ray_idx,
                      int n_views, int rays_per_view, int image_w, int ndc, float ndc_near,
                      float* rays_o, float* rays_d, void* stream);
int jt_raygen_backward(const float* pose, const float* intr_inv, const float* intr, const int64_t* ray_idx,
                       int n_views, int rays_per_view, int image_w, int ndc, float ndc_near,
                       const float* g_rays_o, const float* g_rays_d, float* g_pose, void* stream);
/* Flat pixel indices of the `all_view_rand_grid` lattice (model/nerf.py:660-667: x = ox + i * step, y = oy + j * step, index
 * y * image_w + x, row-major over (j, i)) with the two offsets read from DEVICE memory (offsets: int32[2]) -- the form a replayed
 * hipGraph needs: the host draws ox, oy per iteration and pokes them in front of the replay.  ray_idx: [ny * nx] int64. */
int jt_lattice_indices(const int32_t* offsets, int step, int nx, int ny, int image_w, int64_t* ray_idx, void* stream);
/* The same for a RAGGED batch of views (batched test-time pose optimisation of model/bat.py:265-292: every held-out view on its own
 * pixel lattice): ray_idx [n_rays] is the concatenation of the views' pixel lists, view b owns rays view_offset[b] ..
 * view_offset[b + 1] - 1 (view_offset [n_views + 1], int32, device memory).  Per ray / per view the arithmetic and the order of
 * the sums are those of a single-view call: a view's rays and pose gradient do not depend on what else is in the batch. */
int jt_raygen_forward_ragged(const float* pose, const float* intr_inv, const float* intr, const int64_t* ray_idx,
                             const int32_t* view_offset, int n_views, int n_rays, int image_w, int ndc, float ndc_near,
                             float* rays_o, float* rays_d, void* stream);
int jt_raygen_backward_ragged(const float* pose, const float* intr_inv, const float* intr, const int64_t* ray_idx,
                              const int32_t* view_offset, int n_views, int n_rays, int image_w, int ndc, float ndc_near,
                              const float* g_rays_o, const float* g_rays_d, float* g_pose, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Learnable pose: pose = exp(se3) o noise o gt.
 * Replaces bat.Graph.get_pose train branch (model/bat.py:341-353), Lie.se3_to_SE3 with the
 * nth=8 Taylor series (camera.py:81-99,122-145) and Pose.compose_pair (camera.py:50-57).
 *   se3 [B][6]; noise [B][12] or NULL; gt [B][12] (gt_stride 12) or one [12] shared (gt_stride 0).
 * backward: g_pose [B][12] -> g_se3 [B][6] (overwritten). */
int jt_pose_forward(const float* se3, const float* noise, const float* gt, int gt_stride, int n_views,
                    float* pose, void* stream);
int jt_pose_backward(const float* se3, const float* noise, const float* gt, int gt_stride, int n_views,
                     const float* g_pose, float* g_se3, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Separable 1-D blur of a channel-last factor, replicate padding, cross-correlation.
 * Replaces BAT_VMSplit.convolute_plane / convolute_line (bateRF.py:8-39); taps come from
 * kernels.get_gaussian_kernel (kernels.py:16-22), n_taps odd (65 for c2f_kernel_size 64).
 *   in/out [H][W][C]; a line is H = L, W = 1 (only the H pass runs).  tmp: H*W*C floats scratch.
 * backward is the exact adjoint (g_out -> g_in, overwritten). */
int jt_blur_forward(const float* in, float* out, float* tmp, int H, int W, int C, const float* taps, int n_taps,
                    void* stream);
int jt_blur_backward(const float* g_out, float* g_in, float* tmp, int H, int W, int C, const float* taps,
                     int n_taps, void* stream);

/* The same for up to 12 factors at once (all VM factors of a scene: BAT_VMSplit.forward blurs every plane and
 * line before rendering, bateRF.py:41-130): two launches per direction instead of three per factor.  For the
 * backward `in` is g_out and `out` is g_in. */
typedef struct JtBlurItem {
  const float* in;
  float* out;
  float* tmp;        /* H*W*C floats scratch for planes, may be NULL for lines */
  const float* taps;
  int32_t H, W, C, n_taps;
} JtBlurItem;
int jt_blur_batch_forward(const JtBlurItem* items, int n_items, void* stream);
int jt_blur_batch_backward(const JtBlurItem* items, int n_items, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Staged renderer (forward).  Replaces BatBase.forward (batBase.py:44-165) and everything it
 * calls: sample_ray / sample_ray_ndc (tensorBase.py:554-612), compute_densityfeature
 * (bateRF.py:41-94), feature2density (tensorBase.py:696-700), raw2alpha (tensorBase.py:57-65),
 * compute_appfeature (bateRF.py:97-130), basis_mat (tensoRF.py:156), MLPRender_Fea[_WeakView]
 * (tensorBase.py:101-126,180-214) and the compositing tail (batBase.py:142-165).
 *
 * jt_march_forward: sampling + density + transmittance scan.
 *   jitter: [R] one uniform draw per ray (tensorBase.py:592-596) or NULL; NDC: zvals [S] holds the
 *   (jittered) linspace row shared by all rays (tensorBase.py:557-559), jitter ignored.
 *   out: sigma_feat [R][S], weight [R][S], tmin [R], shade_count [R], shade_offset [R+1]
 *   (exclusive scan; [R] = number of shaded samples), shade_idx [R][S] uint16 (sample indices with
 *   weight > thres, in order), opacity [R], depth [R] (batBase.py:147-151).
 * jt_shade_list: entry -> (ray, sample) map + the positions the appearance path needs.
 *   out: entry_ray [n], entry_smp [n] (int32), viewdirs [n][3] (normalised when ndc).
 * jt_shade_forward (fused, MFMA): prod -> basis -> MLP -> sigmoid, rgb_s [n][3].
 * jt_composite_forward: rgb [R][3] = sum_k w_k c_k (+ white bg) clamped, clamp_mask [R] bit ch
 *   set when the un-clamped value lies in [0,1]. */
/* Opacity 1 - exp(-sigma * length) of one step at n arbitrary world points xyz [n][3] (masked points: 0).
 * Replaces BatBase.compute_alpha (batBase.py:27-41) under TensorBase.getDenseAlpha (tensorBase.py:618-634). */
int jt_dense_alpha(const JtScene* scene, const JtFactors* factors, const float* xyz, long n, float length,
                   float* alpha, void* stream);
int jt_march_forward(const JtScene* scene, const JtFactors* factors, const float* rays_o, const float* rays_d,
                     const float* jitter, const float* zvals, int n_rays, float* sigma_feat, float* weight,
                     float* tmin, int32_t* shade_count, int32_t* shade_offset, uint16_t* shade_idx,
                     float* opacity, float* depth, void* stream);
int jt_shade_list(const JtScene* scene, const float* rays_d, int n_rays, const int32_t* shade_offset,
                  const uint16_t* shade_idx, int32_t* entry_ray, int32_t* entry_smp, float* viewdirs,
                  int n_entries_max, void* stream);
int jt_composite_forward(const JtScene* scene, int n_rays, const int32_t* shade_offset,
                         const uint16_t* shade_idx, const float* weight, const float* rgb_s,
                         const float* opacity, float* rgb, int32_t* clamp_mask, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Staged renderer (backward) -- replaces autograd of all of the above (loss.all.backward(),
 * model/base.py:162; the scatter-adds of grid_sampler_2d_backward).
 * jt_composite_backward: g_rgb [R][3] -> g_rgb_s [n][3] (= weight * g_rgb masked by clamp).
 * jt_march_backward: density + transmittance backward:
 *   += g_factors.density_* (g_factors may be NULL: gradients w.r.t. the rays only); g_rays_o, g_rays_d
 *   [R][3] overwritten (include the app path's
 *   coordinate gradients read from g_xyz_app).  g_opacity [R] may be NULL.  workspace: caller-provided,
 *   jt_march_backward_workspace_bytes(scene, n_rays) bytes (per-sample density gradients + run lists + the rays'
 *   gradient sums in 2^48 fixed point: the runs of a ray meet in integer atomics, so g_rays_o / g_rays_d do not depend
 *   on the order in which they arrive -- the density part of a pose gradient is bit-reproducible in every mode). */
int jt_composite_backward(const JtScene* scene, int n_rays, const int32_t* shade_offset,
                          const int32_t* entry_ray, const int32_t* entry_smp, const float* weight,
                          const int32_t* clamp_mask, const float* g_rgb, float* g_rgb_s, int n_entries_max,
                          void* stream);
int jt_march_backward(const JtScene* scene, const JtFactors* factors, const float* rays_o, const float* rays_d,
                      const float* jitter, const float* zvals, int n_rays, const float* sigma_feat,
                      const float* weight, const float* tmin, const int32_t* shade_offset,
                      const uint16_t* shade_idx, const float* rgb_s, const int32_t* clamp_mask,
                      const float* g_rgb, const float* g_opacity, const float* g_xyz_app,
                      const JtFactors* g_factors, float* g_rays_o, float* g_rays_d, void* workspace,
                      size_t workspace_bytes, void* stream);
size_t jt_march_backward_workspace_bytes(const JtScene* scene, int n_rays);
/* The pair for a render of which only the RAYS want a gradient (test-time pose optimisation, model/bat.py:265-292: the scene is
 * frozen).  jt_march_forward_pose is jt_march_forward that also leaves d(density feature) / d(normalised coordinates) of every
 * in-box sample in dfeat_dn -- three planes of [n_rays * n_samples] floats (x, y, z), written while the taps of
 * compute_densityfeature (bateRF.py:41-94) are in registers; jt_march_backward_pose is jt_march_backward with g_factors = NULL
 * that reads them back instead of gathering the density factors a second time (same sums, same order per ray). */
int jt_march_forward_pose(const JtScene* scene, const JtFactors* factors, const float* rays_o, const float* rays_d,
                          const float* jitter, const float* zvals, int n_rays, float* sigma_feat, float* weight,
                          float* tmin, int32_t* shade_count, int32_t* shade_offset, uint16_t* shade_idx,
                          float* opacity, float* depth, float* dfeat_dn, void* stream);
int jt_march_backward_pose(const JtScene* scene, const JtFactors* factors, const float* rays_o, const float* rays_d,
                           const float* jitter, const float* zvals, int n_rays, const float* sigma_feat,
                           const float* weight, const float* tmin, const int32_t* shade_offset,
                           const uint16_t* shade_idx, const float* rgb_s, const int32_t* clamp_mask,
                           const float* g_rgb, const float* g_opacity, const float* g_xyz_app,
                           const float* dfeat_dn, float* g_rays_o, float* g_rays_d, void* workspace,
                           size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused appearance path on the matrix cores (fp32 accuracy: fp32 MFMA, or bf16 MFMA on three-piece operands, see
 * jt_shade_matrix_mode):
 * gather(app planes/lines) -> outer product -> basis_mat -> PE -> MLP -> sigmoid, per shaded
 * sample, without materialising the [n][3*Ca] product matrix.
 * Replaces compute_appfeature + basis_mat + MLPRender_Fea[_WeakView].forward and their autograd.
 *   forward : rgb_s [n][3]; with workspace != NULL (training) it also leaves the per-sample layer inputs
 *             (products, basis output, hidden activations, ReLU masks, sample coordinates) in the workspace
 *   backward: g_rgb_s [n][3] -> += g_factors.app_*, += g_mlp.*, g_xyz_app [n][3] (overwritten); g_factors and /
 *             or g_mlp may be NULL when that group of gradients is not wanted (test-time pose optimisation,
 *             model/bat.py:265-292, only needs g_xyz_app); consumes the
 *             records the forward of the SAME samples left in `workspace` (the autograd tape of the chain:
 *             nothing is gathered or evaluated twice) and the forward's rgb_s
 * workspace: jt_shade_workspace_bytes(scene, n_entries_max) bytes, caller-provided; forward with
 *            workspace == NULL is the inference path (no records).
 * aux_stream / ev_fork / ev_join (hipStream_t / hipEvent_t, all three or none): when given, the MLP / basis
 * weight-gradient GEMMs are enqueued on aux_stream behind ev_fork (recorded on `stream` after the per-sample
 * kernels) and ev_join is recorded on aux_stream at their end; the caller makes whoever consumes g_mlp wait
 * for ev_join.  This lets the MFMA-bound GEMMs overlap the atomics-bound density backward.  ev_fork alone (no
 * aux_stream): recorded on `stream` behind the per-sample backward kernels, in front of the weight-gradient GEMMs
 * (a timing mark). */
size_t jt_shade_workspace_bytes(const JtScene* scene, int n_entries_max);
/* Where the pieces of that workspace sit under the library's current modes (chunk size, split mode, lean tape): byte offsets
 * from the workspace base and byte sizes, out23[0] = the total jt_shade_workspace_bytes reports, [1] the record array (offset
 * 0), [2] / [3] / [4] offset of the per-chunk weight-gradient slabs, bytes per chunk, chunks, [5] rows of a tile's record block,
 * [6] / [7] offset inside a chunk's slabs and maximal size of what the scatter's workgroups write when dBasis is formed there,
 * [8] whether the tile-owned scatter's lists are part of the workspace and, if so, (offset, bytes) of its seven pieces in
 * [9..22].  A diagnostic for tests (every piece must lie inside the total for any capacity); JT_ERR_ARG for n_entries_max < 1. */
int jt_shade_workspace_layout(const JtScene* scene, int n_entries_max, int64_t* out23);
/* Shaded samples per backward launch ("chunk": batBase.py has no counterpart, it is how the build bounds the
 * per-launch record block).  jt_shade_chunk_entries() = the current value (2^22 unless JT_SHADE_CHUNK_LOG2 says
 * otherwise); jt_shade_set_chunk_log2(l) sets 2^l for l in 16..22 and returns the previous log2 (any other l only
 * queries).  jt_shade_workspace_bytes depends on it: re-query after a change, never change it between a forward
 * and its backward. */
/* Where the training forward leaves the ReLU sign words of the two hidden layers inside `workspace` (test / debug
 * readers): out[0] = record rows per 32-sample tile, out[1] = first of the four sign rows (2 * layer + lane half),
 * out[2] = hidden width, out[3] = samples per tile.  Bit mt * 16 + r of the word of half h stands for hidden unit
 * mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h (tensorBase.py:101-126: the ReLUs of MLPRender_Fea). */
int jt_shade_record_layout(const JtScene* scene, int32_t* out);
/* JT_DETERMINISTIC mode (also the environment variable of that name, read once): bit-reproducible gradients.  While
 * it is on, the g_factors pointers handed to jt_shade_backward / jt_march_backward must be INT64 shadow buffers with
 * the element count and indexing of the float gradient buffers (zero-initialised; sums arrive as 2^56 fixed point: value
 * = word / 2^56), the cross-block sums of jt_render_loss_forward / jt_reg_losses_forward / the weight gradients run in
 * a fixed order, and the ray gradients of jt_march_backward are combined in fixed point inside its workspace.  Returns
 * the previous setting; any argument other than 0 / 1 only queries.  A debugging aid (race detection), slower. */
int jt_set_deterministic(int on);
/* Device-side failure reporting of the gradient scatters (the reference's NaN checks, model/tensorf.py:43-44,147-151).  The
 * ray (pose) gradients of jt_march_backward, and in JT_DETERMINISTIC mode the factor gradients, are summed in 2^48 fixed
 * point; an addend that is NaN, infinite or >= 8 192 in magnitude is dropped and raises a sticky flag inside the library.
 * While the flag is up jt_march_backward writes NaN into g_rays_o / g_rays_d (as the float sums it replaces would have
 * carried it) and ORs JT_STATUS_FINITE_GRAD into the int32 status word bound with jt_status_bind (device memory, may be
 * NULL: nothing is reported then).  jt_status_clear zeroes the flag and the bound word on `stream`. */
#define JT_STATUS_FINITE_GRAD 8
int jt_status_bind(int32_t* status_word);
int jt_status_clear(void* stream);
int jt_shade_chunk_entries(void);
/* Which matrix stages of the appearance path run on the bf16 matrix cores with every fp32 operand split into three bf16
 * pieces and six products accumulated per K step (fp32-level accuracy; the reference's fp32 torch.nn.Linear chain,
 * tensorBase.py:43-131, is what both variants are held to): bit 0 the forward chain (basis product, layers 1 and 2),
 * bit 1 the weight-gradient GEMMs, bit 2 (round 5) layers 2 and 1 of the backward chain where the backward runs SPLIT
 * (chain kernel + scatter: the 20-channel scene, the tile-owned variant, the pose-only backward; the fused VM-48 kernel's LDS
 * has no room for the pre-split transposed weights).  0 = everything on the fp32 matrix cores; build default 7.  The
 * environment variable JT_BF16X3 (read once) overrides the build default; jt_shade_set_matrix_mode(mode) sets it for the
 * launches that follow and returns the previous value (a mode outside 0..7 only queries).  A forward and its backward may run under different modes: the
 * records they exchange are the same fp32 values in the same layout. */
int jt_shade_matrix_mode(void);
int jt_shade_set_matrix_mode(int mode);
int jt_shade_set_chunk_log2(int log2_entries);
/* How jt_shade_backward runs the per-sample part of the appearance backward (the autograd of bateRF.py:97-130 +
 * tensorBase.py:116-126): 0 = one kernel (MLP backward chain and the factor-gradient scatter of a 32-sample tile in the same
 * wave); 8 / 16 = two launches, the chain (which leaves the feature gradients in the record rows) and a scatter kernel whose
 * 16-lane groups walk runs of that many consecutive samples and which sums the LINE gradients in a workgroup-private LDS copy;
 * 1 (round 5) = the chain and the TILE-OWNED scatter (pairs counting-sorted by 4 x 4-texel tile, one wave per tile, the gradient
 * slice accumulated in matrix-core registers; needs factor gradients, float accumulation and at most 131 072 tiles per plane,
 * otherwise the default takes over);
 * -1 (the default) = chosen per scene kind: split 16 whenever the chain runs on the bf16 matrix cores (matrix-mode bit 2: the
 * build default) or the scene has fewer than 48 appearance channels, one kernel otherwise.  Same gradients
 * up to the order of the float sums.  The environment variable JT_BWD_SPLIT (read once) overrides the default; the setter
 * returns the previous value (any other argument only queries). */
int jt_shade_bwd_split(void);
int jt_shade_set_bwd_split(int run);
/* "Lean tape" (round 6; default 1, environment variable JT_LEAN_TAPE read once): whenever the backward is the split form with
 * the walker scatter (jt_shade_bwd_split() 8, 16, or -1 resolving to 16), the training forward does NOT record the 3 Ca
 * plane x line products of a shaded sample (bateRF.py:112-116) -- 576 of the 1 920 record bytes for VM-48, 240 of 1 104 for the
 * 20-channel scene -- and dBasis (the gradient of tensoRF.py:156's basis_mat) is formed inside the scatter kernel from the
 * products its walkers hold, instead of by a GEMM over recorded rows.  Same gradients up to the order of the float sums.
 * jt_shade_workspace_bytes / jt_shade_record_layout follow the mode; like the chunk size and the split mode it must not change
 * between a jt_shade_forward and its jt_shade_backward (the backward returns JT_ERR_ARG when it can tell).  The setter returns
 * the previous value; any argument other than 0 / 1 only queries. */
int jt_shade_lean_tape(void);
int jt_shade_set_lean_tape(int on);
int jt_shade_forward(const JtScene* scene, const JtFactors* factors, const JtMlp* mlp, const float* rays_o,
                     const float* rays_d, const float* jitter, const float* zvals, const float* tmin,
                     const int32_t* shade_offset, int n_rays, const int32_t* entry_ray,
                     const int32_t* entry_smp, const float* viewdirs, float* rgb_s, int n_entries_max,
                     void* workspace, size_t workspace_bytes, int flags, void* stream);
int jt_shade_backward(const JtScene* scene, const JtFactors* factors, const JtMlp* mlp, const float* rays_o,
                      const float* rays_d, const float* jitter, const float* zvals, const float* tmin,
                      const int32_t* shade_offset, int n_rays, const int32_t* entry_ray,
                      const int32_t* entry_smp, const float* viewdirs, const float* rgb_s, const float* g_rgb_s,
                      const JtFactors* g_factors, const JtMlp* g_mlp, float* g_xyz_app, int n_entries_max,
                      void* workspace, size_t workspace_bytes, int flags, void* stream, void* aux_stream,
                      void* ev_fork, void* ev_join);

/* (The single-launch test-time kernel jt_pose_fused lives in an OPTIONAL module of its own since round 5: include/jt_fused.h,
 * libjt_fused.so -- it is slower than the staged kernels it would replace and no default path uses it.) */

/* ---------------------------------------------------------------------------------------------
 * Regularisers over one channel-last factor [H][W][C] (a line is W = 1) in a single pass.
 * Replaces TensorVMSplit.density_L1 (tensoRF.py:212-216) and TVLoss.forward (tensorBase.py:16-41) and
 * their autograd:
 *   forward : out3[0] += sum |x|, out3[1] += sum_{y} (x[y+1,x]-x[y,x])^2, out3[2] += sum_{x} (x[y,x+1]-x[y,x])^2
 *             (out3 zeroed by the caller; the means / weights are applied by the caller)
 *   backward: g (+)= coef3[0]*sign(x) + coef3[1]*d(out3[1])/dx + coef3[2]*d(out3[2])/dx, coef3 on the device;
 *             accumulate != 0 adds to g, 0 overwrites it. */
int jt_factor_reg_forward(const float* x, int H, int W, int C, float* out3, void* stream);
int jt_factor_reg_backward(const float* x, int H, int W, int C, const float* coef3, float* g, int accumulate,
                           void* stream);

/* ---------------------------------------------------------------------------------------------
 * Photometric loss.  Replaces the render term of tensorf.Graph.compute_loss (model/tensorf.py:96-124) with
 * Graph.MSE_loss = nanmean of the squared error (model/base.py:259-261), including the `image[:, ray_idx]`
 * gather: rgb [B][r][3], image [B][3][n_pixels], ray_idx [r] int64; edge_mask [B][n_pixels] u8 or NULL.
 *   edge_mask == NULL: loss = nanmean((rgb - img)^2)
 *   else             : loss = edge_factor * nanmean((rgb*m - img*m)^2) + non_edge_factor * nanmean((rgb*(1-m) - img*(1-m))^2)
 * acc4: 4 floats of device scratch shared by forward and backward; g_loss: dL/dloss on the device. */
int jt_render_loss_forward(const float* rgb, const float* image, const int64_t* ray_idx, const uint8_t* edge_mask,
                           int n_views, int rays_per_view, int n_pixels, float edge_factor, float non_edge_factor,
                           float* acc4, float* loss, void* stream);
int jt_render_loss_backward(const float* rgb, const float* image, const int64_t* ray_idx, const uint8_t* edge_mask,
                            int n_views, int rays_per_view, int n_pixels, float edge_factor, float non_edge_factor,
                            const float* acc4, const float* g_loss, float* g_rgb, void* stream);
/* One photometric nanmean PER VIEW over a ragged batch (layout as jt_raygen_forward_ragged; image [n_views][3][n_pixels], no edge
 * mask): loss [n_views], acc2 [n_views][2] = (sum of squares, count) kept for the backward, g_rgb[i] = g_loss[view of i] * 2 (rgb -
 * gt) / count -- the single-view jt_render_loss_forward / backward bit for bit. */
int jt_render_loss_views_forward(const float* rgb, const float* image, const int64_t* ray_idx, const int32_t* view_offset,
                                 int n_views, int n_pixels, float* acc2, float* loss, void* stream);
int jt_render_loss_views_backward(const float* rgb, const float* image, const int64_t* ray_idx, const int32_t* view_offset,
                                  int n_views, int n_rays, int n_pixels, const float* acc2, const float* g_loss, float* g_rgb,
                                  void* stream);
/* The same with the supervising buffers named by DEVICE memory: slots[0] = address of the image buffer, slots[1] = address
 * of the edge-mask buffer (read only when with_edge_mask != 0).  The reference picks one of five blur scales of the
 * ground-truth images per iteration (model/nerf.py:209-227); a captured hipGraph stays the same for all of them when the
 * caller rewrites `slots` with jt_poke in front of a replay. */
int jt_render_loss_forward_ind(const float* rgb, const uint64_t* slots, const int64_t* ray_idx, int with_edge_mask,
                               int n_views, int rays_per_view, int n_pixels, float edge_factor, float non_edge_factor,
                               float* acc4, float* loss, void* stream);
int jt_render_loss_backward_ind(const float* rgb, const uint64_t* slots, const int64_t* ray_idx, int with_edge_mask,
                                int n_views, int rays_per_view, int n_pixels, float edge_factor, float non_edge_factor,
                                const float* acc4, const float* g_loss, float* g_rgb, void* stream);

/* Non-finite guard without a host sync: ORs items[k].bit into *status_word (device memory, caller-owned, cleared by
 * the caller) for every item whose `data[0..n)` holds a NaN or an infinity; one launch for up to JT_FINITE_MAX
 * tensors.  Stands in for the per-iteration `pose.isnan().any()` / `assert not torch.isnan(loss[key])` host reads of
 * model/tensorf.py:43-44,147-151; the host reads the word when it wants to. */
#define JT_FINITE_MAX 8
typedef struct JtFiniteItem {
  const float* data;
  int64_t n;
  int32_t bit;
  int32_t pad_;
} JtFiniteItem;
int jt_finite_check(const JtFiniteItem* items, int n_items, int32_t* status_word, void* stream);

/* TV of the rendered depth over the pixel lattice (model/tensorf.py:126-135, `TV_depth`), VALUE only: out[0] = sum over views of
 * sum_h (d[h+1][w] - d[h][w])^2 / grid_h + sum_w (d[h][w+1] - d[h][w])^2 / grid_w for depth [n_views][grid_h][grid_w].  The BAT
 * yamls weight the term 0.0 and only log it; a non-zero weight goes through the host mirror's differentiable ops. */
int jt_tv_depth_forward(const float* depth, int n_views, int grid_h, int grid_w, float* out, void* stream);

/* The weighted sum of Model.summarize_loss (model/tensorf.py:31-47) over the photometric term and the three
 * regularisers, one launch each way:  total = w_render render[0] + w_l1 reg3[0] + w_tv_density reg3[1] +
 * w_tv_color reg3[2]  (reg3 = the out3 of jt_reg_losses_forward; a term with weight 0 is left out, as the reference
 * leaves it out);  backward: g_render[0] = g_total[0] w_render, g_reg3[k] = g_total[0] w_k. */
int jt_loss_sum_forward(const float* render, const float* reg3, float w_render, float w_l1, float w_tv_density,
                        float w_tv_color, float* total, void* stream);
int jt_loss_sum_backward(const float* g_total, float w_render, float w_l1, float w_tv_density, float w_tv_color,
                         float* g_render, float* g_reg3, void* stream);
/* The same with w4 = {w_render, w_l1, w_tv_density, w_tv_color} in DEVICE memory: a captured hipGraph keeps working while
 * the host schedule changes the weights every iteration (LLFF: the TV weights decay per iteration, model/tensorf.py:441-447);
 * the caller rewrites w4 with jt_poke in front of a replay. */
int jt_loss_sum_forward_dyn(const float* render, const float* reg3, const float* w4, float* total, void* stream);
/* jt_loss_sum_forward (w4_host: the four weights as host floats) or _dyn (w4_dev, used when w4_host is NULL) AND
 * jt_finite_check of `items` in ONE launch; the total itself is checked as well and reports `loss_bit`. */
int jt_loss_sum_check_forward(const float* render, const float* reg3, const float* w4_host, const float* w4_dev,
                              float* total, const JtFiniteItem* items, int n_items, int32_t loss_bit,
                              int32_t* status_word, void* stream);
int jt_loss_sum_backward_dyn(const float* g_total, const float* w4, float* g_render, float* g_reg3, void* stream);

/* All regularisers of one scene in one call (replaces the loop bodies of model/tensorf.py:127-130):
 *   out3 = { density_L1(), TV_loss_density(TVLoss()), TV_loss_app(TVLoss()) }   (tensoRF.py:212-228).
 * plane_hw_line[9] = {H_i, W_i, L_i} for i = 0..2; scratch640 (forward): 640 floats of device scratch that must be ZERO when the
 * first call sees them and are left zero by every call (the one launch sums into them and its last workgroup combines and resets
 * them: no zero fill and no combine launch per iteration); not to be shared by calls that run concurrently.
 * with_tv_density / with_tv_app == 0: that TV term has weight zero in the run; it is not evaluated and out3
 * carries 0 for it (the L1 term is always evaluated).
 * backward: g3 = dL/d out3 on the device; writes (accumulate == 0) or adds (accumulate != 0) the gradients of
 * the density planes + lines, and of the appearance planes when with_tv_app != 0, into g_factors. */
int jt_reg_losses_forward(const JtFactors* factors, const int32_t* plane_hw_line, int n_comp_density,
                          int n_comp_app, int with_tv_density, int with_tv_app, float* scratch640, float* out3,
                          void* stream);
int jt_reg_losses_backward(const JtFactors* factors, const int32_t* plane_hw_line, int n_comp_density,
                           int n_comp_app, const float* g3, int with_tv_density, int with_tv_app,
                           const JtFactors* g_factors, int accumulate, float* scratch640, void* stream);
/* Value AND gradient in ONE launch (round 5): out3 as jt_reg_losses_forward, and the gradients that jt_reg_losses_backward
 * (accumulate == 0) would WRITE for the upstream gradients w3 = dL/d out3 -- in a training step the loss weights of
 * Model.summarize_loss (model/tensorf.py:31-47), known before the forward: three host floats (w3_host) or, when w3_host is
 * NULL, three floats in device memory (w3_dev: a replayed hipGraph reads this iteration's weights from there).  Every factor
 * is read once instead of twice.  scratch640 as for the forward.  JT_ERR_UNSUPPORTED in deterministic mode. */
int jt_reg_losses_fused(const JtFactors* factors, const int32_t* plane_hw_line, int n_comp_density, int n_comp_app,
                        int with_tv_density, int with_tv_app, const float* w3_host, const float* w3_dev,
                        const JtFactors* g_factors, float* scratch640, float* out3, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Dense Adam step over all tensors of an optimizer in one launch.  Replaces torch.optim.Adam.step of the
 * optimizer built by tensorf.NeRF._get_optimizer (model/tensorf.py:463-478, betas (0.9, 0.99)):
 *   m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= lr / bias_correction1 * m / (sqrt(v) / sqrt(bias_correction2) + eps)
 * p, g, m, v: n floats each in the same (dense) layout, 16-byte aligned; updated in place. */
typedef struct JtAdamItem {
  float* p;
  const float* g;
  float* m;
  float* v;
  int64_t n;
  float lr;
  float bias_correction1; /* 1 - beta1^step */
  float bias_correction2; /* 1 - beta2^step */
  int32_t pad_;
} JtAdamItem;
int jt_adam_step(const JtAdamItem* items, int n_items, float beta1, float beta2, float eps, void* stream);

/* The same step with the two per-tensor coefficients that change every iteration read from DEVICE memory:
 * dyn[2 i] = lr_i / (1 - beta1^step_i), dyn[2 i + 1] = 1 / sqrt(1 - beta2^step_i) for item i (the lr /
 * bias_correction fields of the items are ignored).  A hipGraph that captured this launch (the whole steady-state
 * iteration is captured, joint_tensorf_amd/graphed.py) is replayed with new coefficients written by jt_poke. */
int jt_adam_step_dyn(const JtAdamItem* items, int n_items, float beta1, float beta2, float eps, const float* dyn,
                     void* stream);
/* The same step with the SAME two coefficients per item given as floats in HOST memory (coefs_host[2 i], [2 i + 1]; the
 * lr / bias_correction fields are ignored): they travel as launch arguments.  An eager iteration steps with exactly the
 * float values a replayed one reads from `dyn`, without the jt_poke launch in front. */
int jt_adam_step_coefs(const JtAdamItem* items, int n_items, float beta1, float beta2, float eps,
                       const float* coefs_host, void* stream);

/* Write n_words (1..256) 32-bit words from HOST memory `words` to device memory `dst` on `stream`: the values
 * travel as launch arguments (no host staging buffer, no synchronisation, the host array may be reused at once).
 * The per-iteration host scalars of the reference's loop -- lattice offsets (model/nerf.py:663), Adam coefficients
 * (model/tensorf.py:441-447 decays the lr every iteration) -- reach a replayed hipGraph this way. */
int jt_poke(void* dst, const uint32_t* words, int n_words, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* JT_RENDER_H */
