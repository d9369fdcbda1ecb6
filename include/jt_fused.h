/* jt_fused.h -- OPTIONAL module libjt_fused.so (joint_tensorf_amd/csrc/jt_fused.hip): the single-launch render + loss +
 * backward kernel for test-time pose optimisation (BASELINE.json north_star: "forward and backward fused ... a single launch").
 *
 * Built, parity-green, measured -- and slower than the staged kernels it would replace (2.27 ms against 1.76 ms per iteration on
 * the dense 400^3 scene since the walker-free pose-only backward of round 5, 0.56 against 0.52 on the blob scene; its matrix phases
 * spill at the two waves per SIMD they need).  It is therefore NOT part of libjt_render.so any more: a
 * separate shared library that depends on libjt_render.so, loaded only when a caller asks for it
 * (opt.optim.test_fused / joint_tensorf_amd._lib.fused_lib()).  Same conventions as jt_render.h. */
#ifndef JT_FUSED_H
#define JT_FUSED_H

#include "jt_render.h"

#ifdef __cplusplus
extern "C" {
#endif

/*  * Single-launch render + photometric loss + backward to the rays (csrc/jt_fused.hip) -- test-time pose optimisation,
 * model/bat.py:265-292: per iteration Graph.forward(mode "test-optim") -> compute_loss -> loss.all.backward() with only a
 * 6-vector trained (scene frozen).  Replaces, for that mode, tensorf.Graph.render_rays + BatBase.forward
 * (model/tensorf.py:169-267, batBase.py:44-165), the render term of compute_loss (model/tensorf.py:96-124 with
 * base.py:259-261's mean over the 3 R colour values) and autograd's walk back to (center, ray_dir): one wave owns one ray
 * from its first sample to its gradient, nothing is recorded between forward and backward.
 *   rays_o / rays_d [R][3]; zvals [S] for NDC scenes (no jitter at test time); image [views][3][image_pixels] and
 *   ray_idx [rays_per_view] (int64): ray r looks at pixel ray_idx[r % rays_per_view] of view r / rays_per_view.
 *   loss_scale = 1 / (3 R) times whatever weight the caller wants folded in.  Values are assumed finite (the mean is
 *   over all 3 R values; the reference's nanmean would drop NaNs -- the caller's non-finite guard reports those).
 *   out: rgb [R][3], depth [R], opacity [R], sqerr [R] (per-ray sum of squared colour differences), loss [1] =
 *   loss_scale * sum(sqerr), g_rays_o / g_rays_d [R][3] = d loss / d rays (plain stores: bit-reproducible).
 *   workspace: jt_pose_fused_workspace_bytes(scene) bytes, ZERO before the first launch that uses it (its head holds the
 *   loss accumulator and an arrival counter, which every launch leaves zeroed again); contents otherwise scratch.
 * No blur (the caller blurs factors itself if a schedule asks for it and hands the blurred ones), alpha mask honoured. */
size_t jt_pose_fused_workspace_bytes(const JtScene* scene);
int jt_pose_fused(const JtScene* scene, const JtFactors* factors, const JtMlp* mlp, const float* rays_o,
                  const float* rays_d, const float* zvals, int n_rays, const float* image, const int64_t* ray_idx,
                  int rays_per_view, int image_pixels, float loss_scale, float* rgb, float* depth, float* opacity,
                  float* sqerr, float* loss, float* g_rays_o, float* g_rays_d, void* workspace, size_t workspace_bytes,
                  void* stream);


#ifdef __cplusplus
}
#endif

#endif /* JT_FUSED_H */
