/* TEST INFRASTRUCTURE, not part of the product library: the two entry points of the STAGED appearance path
 * (tests/staged_path.py), built into tests/lib/libjt_test_staged.so by joint_tensorf_amd/build.py.  The staged path
 * materialises the [n][3*Ca] plane x line products and runs basis / encoding / MLP in stock torch ops: a second,
 * independent implementation of the appearance chain that tests/test_gpu_parity.py runs next to the fused MFMA kernels
 * against the same golden vectors.
 * jt_app_gather_forward: prod [n][3*Ca] = plane_i^c(p) * line_i^c(p)  (bateRF.py:124-128).
 * jt_app_gather_backward: g_prod [n][3*Ca] -> += g_factors.app_*, g_xyz_app [n][3] (overwritten). */
#ifndef JT_TEST_STAGED_H
#define JT_TEST_STAGED_H
#include "jt_render.h"
#ifdef __cplusplus
extern "C" {
#endif
int jt_app_gather_forward(const JtScene* scene, const JtFactors* factors, const float* rays_o,
                          const float* rays_d, const float* jitter, const float* zvals, const float* tmin,
                          const int32_t* shade_offset, int n_rays, const int32_t* entry_ray,
                          const int32_t* entry_smp, float* prod, int n_entries_max, void* stream);
int jt_app_gather_backward(const JtScene* scene, const JtFactors* factors, const float* rays_o,
                           const float* rays_d, const float* jitter, const float* zvals, const float* tmin,
                           const int32_t* shade_offset, int n_rays, const int32_t* entry_ray,
                           const int32_t* entry_smp, const float* g_prod, const JtFactors* g_factors,
                           float* g_xyz_app, int n_entries_max, void* stream);
#ifdef __cplusplus
}
#endif
#endif
