// Fused appearance path on the matrix cores -- placeholder until the MFMA kernels land.
#include "jt_common.h"

extern "C" size_t jt_shade_workspace_bytes(const JtScene* scene) { return 0; }

extern "C" int jt_shade_forward(const JtScene* scene, const JtFactors* factors, const JtMlp* mlp, const float* rays_o,
                                const float* rays_d, const float* jitter, const float* zvals, const float* tmin,
                                const int32_t* shade_offset, int n_rays, const int32_t* entry_ray,
                                const int32_t* entry_smp, const float* viewdirs, float* rgb_s, int n_entries_max,
                                void* workspace, size_t workspace_bytes, void* stream) {
  return JT_ERR_UNSUPPORTED;
}

extern "C" int jt_shade_backward(const JtScene* scene, const JtFactors* factors, const JtMlp* mlp,
                                 const float* rays_o, const float* rays_d, const float* jitter, const float* zvals,
                                 const float* tmin, const int32_t* shade_offset, int n_rays,
                                 const int32_t* entry_ray, const int32_t* entry_smp, const float* viewdirs,
                                 const float* g_rgb_s, const JtFactors* g_factors, const JtMlp* g_mlp,
                                 float* g_xyz_app, int n_entries_max, void* workspace, size_t workspace_bytes,
                                 void* stream) {
  return JT_ERR_UNSUPPORTED;
}
