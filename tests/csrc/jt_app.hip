// Appearance-factor gather / scatter (staged form): prod[e][i*Ca + c] = plane_i^c(p_e) * line_i^c(p_e)
// for every shaded sample e (bateRF.py:97-128), and its backward (plane/line scatter-add + the
// gradient w.r.t. the sample position, which is what carries the pose gradient; bateRF.py:100-102
// keeps the grid coordinates attached).  Used by the staged pipeline and as the cross-check for the
// fused MFMA shade kernels in jt_shade.hip.
#include "jt_common.h"
#include "jt_test_staged.h"

namespace jt {

__device__ inline void entry_point(const Dev& D, const float* rays_o, const float* rays_d, const float* jitter,
                                   const float* zvals, const float* tmin, int ray, int smp, float n[3], float* z) {
  Ray r;
  load_ray(D, rays_o, rays_d, jitter, tmin, ray, r);
  *z = sample_z(D, r, zvals, smp);
  float p[3];
  sample_point(D, r, *z, p);
  normalize(D, p, n);
}

// one 16-lane group per (entry, plane): lane c4 handles channel quads c4, c4+16, ...  -> 64-byte
// contiguous loads per tap and 256-byte contiguous stores of the product row.
__global__ __launch_bounds__(256) void k_app_gather_fwd(Dev D, const float* __restrict__ rays_o,
                                                        const float* __restrict__ rays_d,
                                                        const float* __restrict__ jitter,
                                                        const float* __restrict__ zvals,
                                                        const float* __restrict__ tmin,
                                                        const int* __restrict__ offset, int R,
                                                        const int* __restrict__ eray, const int* __restrict__ esmp,
                                                        float* __restrict__ prod, int cap) {
  const int total = min(offset[R], cap);
  const int C = D.Ca;
  const int sub = threadIdx.x >> 4, q0 = (threadIdx.x & 15) * 4;
  const int groups = gridDim.x * 16;
  for (int item = blockIdx.x * 16 + sub; item < total * 3; item += groups) {
    const int e = item / 3, pl = item - e * 3;
    float n[3], z;
    entry_point(D, rays_o, rays_d, jitter, zvals, tmin, eray[e], esmp[e], n, &z);
    PlaneTaps t = plane_taps(n[kM0(pl)], n[kM1(pl)], D.ph[pl], D.pw[pl], C);
    Axis l = axis_taps(n[kV(pl)], D.ll[pl]);
    const float* P = D.aP[pl];
    const float* L = D.aL[pl];
    float* out = prod + (size_t)e * (3 * C) + pl * C;
    for (int q = q0; q < C; q += 64) {
      float4 a = ld4(P + t.o00 + q), b = ld4(P + t.o10 + q), c = ld4(P + t.o01 + q), d = ld4(P + t.o11 + q);
      float4 u = ld4(L + l.c0 * C + q), v = ld4(L + l.c1 * C + q);
      float4 o;
      o.x = (t.w00 * a.x + t.w10 * b.x + t.w01 * c.x + t.w11 * d.x) * (l.w0 * u.x + l.w1 * v.x);
      o.y = (t.w00 * a.y + t.w10 * b.y + t.w01 * c.y + t.w11 * d.y) * (l.w0 * u.y + l.w1 * v.y);
      o.z = (t.w00 * a.z + t.w10 * b.z + t.w01 * c.z + t.w11 * d.z) * (l.w0 * u.z + l.w1 * v.z);
      o.w = (t.w00 * a.w + t.w10 * b.w + t.w01 * c.w + t.w11 * d.w) * (l.w0 * u.w + l.w1 * v.w);
      *reinterpret_cast<float4*>(out + q) = o;
    }
  }
}

// backward: 16 lanes per (entry, plane), lane = channel (mod 16): dword atomics, 64 B contiguous per tap
__global__ __launch_bounds__(256) void k_app_gather_bwd(Dev D, JtFactors G, const float* __restrict__ rays_o,
                                                        const float* __restrict__ rays_d,
                                                        const float* __restrict__ jitter,
                                                        const float* __restrict__ zvals,
                                                        const float* __restrict__ tmin,
                                                        const int* __restrict__ offset, int R,
                                                        const int* __restrict__ eray, const int* __restrict__ esmp,
                                                        const float* __restrict__ g_prod,
                                                        float* __restrict__ g_xyz, int cap) {
  const int total = min(offset[R], cap);
  const int C = D.Ca;
  const int sub = threadIdx.x >> 4, ch = threadIdx.x & 15;
  const int groups = gridDim.x * 16;
  // all 16 lanes of a group iterate together (the shuffles below need the whole group)
  const int items = total * 3;
  const int rounds = (items + groups - 1) / groups;
  for (int rd = 0; rd < rounds; ++rd) {
    const int item = rd * groups + blockIdx.x * 16 + sub;
    const bool on = item < items;
    const int e = on ? item / 3 : 0, pl = on ? item - e * 3 : 0;
    float n[3], z;
    entry_point(D, rays_o, rays_d, jitter, zvals, tmin, eray[e], esmp[e], n, &z);
    PlaneTaps t = plane_taps(n[kM0(pl)], n[kM1(pl)], D.ph[pl], D.pw[pl], C);
    Axis l = axis_taps(n[kV(pl)], D.ll[pl]);
    const float* P = D.aP[pl];
    const float* L = D.aL[pl];
    float* gP = G.app_plane[pl];
    float* gL = G.app_line[pl];
    const float* gp = g_prod + (size_t)e * (3 * C) + pl * C;
    float aix = 0.f, aiy = 0.f, ail = 0.f;
    for (int cq = ch; cq < C; cq += 16) {
      float a = P[t.o00 + cq], b = P[t.o10 + cq], c = P[t.o01 + cq], d = P[t.o11 + cq];
      float u = L[l.c0 * C + cq], v = L[l.c1 * C + cq];
      float pv = t.w00 * a + t.w10 * b + t.w01 * c + t.w11 * d;
      float lv = l.w0 * u + l.w1 * v;
      float g = on ? gp[cq] : 0.f;
      float gpv = g * lv, glv = g * pv;
      if (on) {
        if (t.w00 != 0.f) atomicAdd(gP + t.o00 + cq, t.w00 * gpv);
        if (t.w10 != 0.f) atomicAdd(gP + t.o10 + cq, t.w10 * gpv);
        if (t.w01 != 0.f) atomicAdd(gP + t.o01 + cq, t.w01 * gpv);
        if (t.w11 != 0.f) atomicAdd(gP + t.o11 + cq, t.w11 * gpv);
        if (l.w0 != 0.f) atomicAdd(gL + l.c0 * C + cq, l.w0 * glv);
        if (l.w1 != 0.f) atomicAdd(gL + l.c1 * C + cq, l.w1 * glv);
      }
      float a_ = a * t.ax.m0 * t.ay.m0, b_ = b * t.ax.m1 * t.ay.m0, c_ = c * t.ax.m0 * t.ay.m1,
            d_ = d * t.ax.m1 * t.ay.m1;
      aix += gpv * ((b_ - a_) * (1.f - t.ay.f) + (d_ - c_) * t.ay.f);
      aiy += gpv * ((c_ - a_) * (1.f - t.ax.f) + (d_ - b_) * t.ax.f);
      ail += glv * (v * l.m1 - u * l.m0);
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      aix += __shfl_xor(aix, o);
      aiy += __shfl_xor(aiy, o);
      ail += __shfl_xor(ail, o);
    }
    if (on && ch == 0) {
      const int m0 = kM0(pl), m1 = kM1(pl), v = kV(pl);
      atomicAdd(g_xyz + (size_t)e * 3 + m0, aix * t.ax.scale * D.inv[m0]);
      atomicAdd(g_xyz + (size_t)e * 3 + m1, aiy * t.ay.scale * D.inv[m1]);
      atomicAdd(g_xyz + (size_t)e * 3 + v, ail * l.scale * D.inv[v]);
    }
  }
}

__global__ void k_zero(float* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0.f;
}

}  // namespace jt

using namespace jt;

extern "C" int jt_app_gather_forward(const JtScene* scene, const JtFactors* factors, const float* rays_o,
                                     const float* rays_d, const float* jitter, const float* zvals,
                                     const float* tmin, const int32_t* shade_offset, int n_rays,
                                     const int32_t* entry_ray, const int32_t* entry_smp, float* prod,
                                     int n_entries_max, void* stream) {
  Dev D;
  int rc = make_dev(scene, factors, &D);
  if (rc) return rc;
  if (!factors || !rays_o || !rays_d || !tmin || !shade_offset || !entry_ray || !entry_smp || !prod)
    return JT_ERR_ARG;
  if (D.ndc && !zvals) return JT_ERR_ARG;
  if (D.Ca < 4 || (D.Ca % 4) != 0) return JT_ERR_UNSUPPORTED;
  if (n_entries_max < 1) return JT_OK;
  long items = (long)n_entries_max * 3;
  int blocks = (int)min((items + 15) / 16, 8192L);
  hipLaunchKernelGGL(k_app_gather_fwd, dim3(blocks), dim3(256), 0, (hipStream_t)stream, D, rays_o, rays_d, jitter,
                     zvals, tmin, shade_offset, n_rays, entry_ray, entry_smp, prod, n_entries_max);
  JT_LAUNCH_CHECK();
  return JT_OK;
}

extern "C" int jt_app_gather_backward(const JtScene* scene, const JtFactors* factors, const float* rays_o,
                                      const float* rays_d, const float* jitter, const float* zvals,
                                      const float* tmin, const int32_t* shade_offset, int n_rays,
                                      const int32_t* entry_ray, const int32_t* entry_smp, const float* g_prod,
                                      const JtFactors* g_factors, float* g_xyz_app, int n_entries_max,
                                      void* stream) {
  Dev D;
  int rc = make_dev(scene, factors, &D);
  if (rc) return rc;
  if (!factors || !g_factors || !rays_o || !rays_d || !tmin || !shade_offset || !entry_ray || !entry_smp ||
      !g_prod || !g_xyz_app)
    return JT_ERR_ARG;
  for (int a = 0; a < 3; ++a)
    if (!g_factors->app_plane[a] || !g_factors->app_line[a]) return JT_ERR_ARG;
  if (D.ndc && !zvals) return JT_ERR_ARG;
  if (D.Ca < 4 || (D.Ca % 4) != 0) return JT_ERR_UNSUPPORTED;
  if (n_entries_max < 1) return JT_OK;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_zero, dim3(256), dim3(256), 0, st, g_xyz_app, (size_t)n_entries_max * 3);
  JT_LAUNCH_CHECK();
  long items = (long)n_entries_max * 3;
  int blocks = (int)min((items + 15) / 16, 8192L);
  hipLaunchKernelGGL(k_app_gather_bwd, dim3(blocks), dim3(256), 0, st, D, *g_factors, rays_o, rays_d, jitter,
                     zvals, tmin, shade_offset, n_rays, entry_ray, entry_smp, g_prod, g_xyz_app, n_entries_max);
  JT_LAUNCH_CHECK();
  return JT_OK;
}
