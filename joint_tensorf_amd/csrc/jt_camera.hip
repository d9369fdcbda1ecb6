// Camera-side kernels: learnable pose composition and ray generation for the sampled pixels,
// forward and analytic backward.
//   pose   = exp(se3_refine) o pose_noise o pose_GT          (model/bat.py:341-353)
//   exp    = Lie.se3_to_SE3 with the reference's nth=8 Taylor series for sin(t)/t, (1-cos t)/t^2,
//            (t - sin t)/t^3                                  (camera.py:81-99,122-145)
//   rays   : dirs = ([x+.5, y+.5, 1] K^-T) R, center = -(t^T R)   (camera.py:231-261), optional NDC
//            re-parametrisation (camera.py:303-340)
// Only the pixels of the iteration's lattice are generated (the reference builds all H*W rays of all
// views and indexes afterwards, model/tensorf.py:154-161).
#include "jt_common.h"

namespace jt {

struct M3 {
  float m[3][3];
};

__device__ inline M3 mm(const M3& a, const M3& b) {
  M3 r;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
  return r;
}
__device__ inline M3 tr(const M3& a) {
  M3 r;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[j][i];
  return r;
}
__device__ inline float dot9(const M3& a, const M3& b) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) s += a.m[i][j] * b.m[i][j];
  return s;
}
__device__ inline M3 skew3(const float w[3]) {
  M3 r;
  r.m[0][0] = 0.f;   r.m[0][1] = -w[2]; r.m[0][2] = w[1];
  r.m[1][0] = w[2];  r.m[1][1] = 0.f;   r.m[1][2] = -w[0];
  r.m[2][0] = -w[1]; r.m[2][1] = w[0];  r.m[2][2] = 0.f;
  return r;
}

// Taylor coefficients of A, B, C in powers of theta^2 (9 terms, camera.py:122-145 with nth=8);
// val = sum a_i x^i, dval = d val / d(theta^2), x = theta^2
__device__ inline void taylor(float x, int kind, float* val, float* dval) {
  float denom = 1.f, v = 0.f, dv = 0.f, xp = 1.f, xpm = 0.f;  // xp = x^i, xpm = x^(i-1)
  for (int i = 0; i <= 8; ++i) {
    if (kind == 0) {
      if (i > 0) denom *= (float)((2 * i) * (2 * i + 1));
    } else if (kind == 1) {
      denom *= (float)((2 * i + 1) * (2 * i + 2));
    } else {
      denom *= (float)((2 * i + 2) * (2 * i + 3));
    }
    float sgn = (i & 1) ? -1.f : 1.f;
    v += sgn * (xp / denom);
    if (i > 0) dv += sgn * ((float)i * xpm / denom);
    xpm = xp;
    xp *= x;
  }
  *val = v;
  *dval = dv;
}

struct Se3 {
  M3 R, V, wx, wx2;
  float t[3];
  float A, B, C, dA, dB, dC;
};

__device__ inline Se3 se3_exp(const float* wu) {
  Se3 s;
  float w[3] = {wu[0], wu[1], wu[2]};
  float x = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  taylor(x, 0, &s.A, &s.dA);
  taylor(x, 1, &s.B, &s.dB);
  taylor(x, 2, &s.C, &s.dC);
  s.wx = skew3(w);
  s.wx2 = mm(s.wx, s.wx);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      float I = (i == j) ? 1.f : 0.f;
      s.R.m[i][j] = I + s.A * s.wx.m[i][j] + s.B * s.wx2.m[i][j];
      s.V.m[i][j] = I + s.B * s.wx.m[i][j] + s.C * s.wx2.m[i][j];
    }
#pragma unroll
  for (int i = 0; i < 3; ++i) s.t[i] = s.V.m[i][0] * wu[3] + s.V.m[i][1] * wu[4] + s.V.m[i][2] * wu[5];
  return s;
}

// base = gt o noise  (compose_pair(noise, gt): R = R_gt R_noise, t = R_gt t_noise + t_gt)
__device__ inline void base_pose(const float* noise, const float* gt, M3& Rb, float tb[3]) {
  M3 Rg;
  float tg[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) Rg.m[i][j] = gt[i * 4 + j];
    tg[i] = gt[i * 4 + 3];
  }
  if (!noise) {
    Rb = Rg;
#pragma unroll
    for (int i = 0; i < 3; ++i) tb[i] = tg[i];
    return;
  }
  M3 Rn;
  float tn[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) Rn.m[i][j] = noise[i * 4 + j];
    tn[i] = noise[i * 4 + 3];
  }
  Rb = mm(Rg, Rn);
#pragma unroll
  for (int i = 0; i < 3; ++i) tb[i] = Rg.m[i][0] * tn[0] + Rg.m[i][1] * tn[1] + Rg.m[i][2] * tn[2] + tg[i];
}

__global__ void k_pose_fwd(const float* __restrict__ se3, const float* __restrict__ noise,
                           const float* __restrict__ gt, int gt_stride, int B, float* __restrict__ pose) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  Se3 s = se3_exp(se3 + b * 6);
  M3 Rb;
  float tb[3];
  base_pose(noise ? noise + b * 12 : nullptr, gt + (size_t)b * gt_stride, Rb, tb);
  // pose = compose_pair(refine, base): R = R_base R_refine ; t = R_base t_refine + t_base
  M3 R = mm(Rb, s.R);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) pose[b * 12 + i * 4 + j] = R.m[i][j];
    pose[b * 12 + i * 4 + 3] = Rb.m[i][0] * s.t[0] + Rb.m[i][1] * s.t[1] + Rb.m[i][2] * s.t[2] + tb[i];
  }
}

__global__ void k_pose_bwd(const float* __restrict__ se3, const float* __restrict__ noise,
                           const float* __restrict__ gt, int gt_stride, int B, const float* __restrict__ g_pose,
                           float* __restrict__ g_se3) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float* wu = se3 + b * 6;
  Se3 s = se3_exp(wu);
  M3 Rb;
  float tb[3];
  base_pose(noise ? noise + b * 12 : nullptr, gt + (size_t)b * gt_stride, Rb, tb);
  M3 GR;
  float gt_[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) GR.m[i][j] = g_pose[b * 12 + i * 4 + j];
    gt_[i] = g_pose[b * 12 + i * 4 + 3];
  }
  M3 RbT = tr(Rb);
  M3 gR = mm(RbT, GR);  // dL/dR_refine
  float gtr[3];         // dL/dt_refine
#pragma unroll
  for (int i = 0; i < 3; ++i) gtr[i] = RbT.m[i][0] * gt_[0] + RbT.m[i][1] * gt_[1] + RbT.m[i][2] * gt_[2];
  // t = V u
  float gu[3];
  M3 gV;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    gu[i] = s.V.m[0][i] * gtr[0] + s.V.m[1][i] * gtr[1] + s.V.m[2][i] * gtr[2];
#pragma unroll
    for (int j = 0; j < 3; ++j) gV.m[i][j] = gtr[i] * wu[3 + j];
  }
  float gA = dot9(gR, s.wx);
  float gB = dot9(gR, s.wx2) + dot9(gV, s.wx);
  float gC = dot9(gV, s.wx2);
  M3 wxT = tr(s.wx);
  M3 a1 = mm(gR, wxT), a2 = mm(wxT, gR), b1 = mm(gV, wxT), b2 = mm(wxT, gV);
  M3 Gw;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      Gw.m[i][j] = s.A * gR.m[i][j] + s.B * (a1.m[i][j] + a2.m[i][j]) + s.B * gV.m[i][j] +
                   s.C * (b1.m[i][j] + b2.m[i][j]);
  float gw[3];
  gw[0] = Gw.m[2][1] - Gw.m[1][2];
  gw[1] = Gw.m[0][2] - Gw.m[2][0];
  gw[2] = Gw.m[1][0] - Gw.m[0][1];
  // through theta^2 = |w|^2 : d(theta^2)/dw = 2 w
  float gx = gA * s.dA + gB * s.dB + gC * s.dC;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    g_se3[b * 6 + i] = gw[i] + 2.f * gx * wu[i];
    g_se3[b * 6 + 3 + i] = gu[i];
  }
}

// ---- ray generation ---------------------------------------------------------------------------
struct PixRay {
  float g[3];   // K^-1 [x+.5, y+.5, 1]
  float c0[3];  // un-shifted centre
  float d[3];   // world direction
};

__device__ inline PixRay pixel_ray(const float* P, const float* Ki, long idx, int W) {
  PixRay r;
  float x = (float)(idx % W) + 0.5f, y = (float)(idx / W) + 0.5f;
#pragma unroll
  for (int i = 0; i < 3; ++i) r.g[i] = x * Ki[i * 3] + y * Ki[i * 3 + 1] + Ki[i * 3 + 2];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    r.d[j] = r.g[0] * P[j] + r.g[1] * P[4 + j] + r.g[2] * P[8 + j];
    r.c0[j] = -(P[3] * P[j] + P[7] * P[4 + j] + P[11] * P[8 + j]);
  }
  return r;
}

// view of ray t in a RAGGED batch (view b owns rays view_offset[b] .. view_offset[b + 1] - 1): the last b with offset <= t
__device__ inline int ragged_view(const int* __restrict__ voff, int V, int t) {
  int lo = 0, hi = V - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (voff[mid] <= t) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// RAGGED = false: B views x the SAME r pixels (ray_idx [r]).  RAGGED = true: every view its own pixel list, concatenated
// (ray_idx [n_total], view_offset [B + 1]; `r` is n_total) -- per ray the same arithmetic, so a view's rays are bit-identical
// whether it is rendered alone or in a batch of views (batched test-time pose optimisation).
template <bool RAGGED>
__global__ __launch_bounds__(256) void k_raygen_fwd(const float* __restrict__ pose, const float* __restrict__ intr_inv,
                                                    const float* __restrict__ intr,
                                                    const int64_t* __restrict__ ray_idx, const int* __restrict__ voff, int B,
                                                    int r, int W, int ndc, float near, float* __restrict__ rays_o,
                                                    float* __restrict__ rays_d) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (RAGGED ? r : B * r)) return;
  int b, k;
  if (RAGGED) {
    b = ragged_view(voff, B, t);
    k = t;
  } else {
    b = t / r;
    k = t - b * r;
  }
  PixRay pr = pixel_ray(pose + b * 12, intr_inv + b * 9, ray_idx[k], W);
  float o[3] = {pr.c0[0], pr.c0[1], pr.c0[2]}, d[3] = {pr.d[0], pr.d[1], pr.d[2]};
  if (ndc) {
    float s = (near - o[2]) / d[2];
    float c[3] = {o[0] + s * d[0], o[1] + s * d[1], o[2] + s * d[2]};
    float sx = intr[b * 9] / intr[b * 9 + 2], sy = intr[b * 9 + 4] / intr[b * 9 + 5];
    float cxoz = c[0] / c[2], cyoz = c[1] / c[2], rxoz = d[0] / d[2], ryoz = d[1] / d[2];
    o[0] = sx * cxoz;
    o[1] = sy * cyoz;
    o[2] = 1.f - 2.f * near / c[2];
    d[0] = sx * (rxoz - cxoz);
    d[1] = sy * (ryoz - cyoz);
    d[2] = 2.f * near / c[2];
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    rays_o[(size_t)t * 3 + a] = o[a];
    rays_d[(size_t)t * 3 + a] = d[a];
  }
}

// one workgroup per view: deterministic reduction of the 12 pose-gradient entries over the view's rays
template <bool RAGGED>
__global__ __launch_bounds__(256) void k_raygen_bwd(const float* __restrict__ pose, const float* __restrict__ intr_inv,
                                                    const float* __restrict__ intr,
                                                    const int64_t* __restrict__ ray_idx, const int* __restrict__ voff, int B,
                                                    int r_all, int W, int ndc, float near, const float* __restrict__ g_o,
                                                    const float* __restrict__ g_d, float* __restrict__ g_pose) {
  __shared__ float red[4][12];
  const int b = blockIdx.x;
  const float* P = pose + b * 12;
  // (ragged: the view's own rays in the same thread order as a single-view launch: the same sums bit for bit)
  const int base = RAGGED ? voff[b] : 0, r = RAGGED ? voff[b + 1] - voff[b] : r_all;
  float acc[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) acc[i] = 0.f;
  for (int k = threadIdx.x; k < r; k += blockDim.x) {
    PixRay pr = pixel_ray(P, intr_inv + b * 9, ray_idx[base + k], W);
    size_t t = RAGGED ? (size_t)base + k : (size_t)b * r + k;
    float go[3] = {g_o[t * 3], g_o[t * 3 + 1], g_o[t * 3 + 2]};
    float gd[3] = {g_d[t * 3], g_d[t * 3 + 1], g_d[t * 3 + 2]};
    if (ndc) {
      const float* d = pr.d;
      float s = (near - pr.c0[2]) / d[2];
      float c[3] = {pr.c0[0] + s * d[0], pr.c0[1] + s * d[1], pr.c0[2] + s * d[2]};
      float sx = intr[b * 9] / intr[b * 9 + 2], sy = intr[b * 9 + 4] / intr[b * 9 + 5];
      float icz = 1.f / c[2], idz = 1.f / d[2];
      float gc[3], gdd[3];
      gc[0] = (go[0] - gd[0]) * sx * icz;
      gc[1] = (go[1] - gd[1]) * sy * icz;
      gc[2] = (-(go[0] - gd[0]) * sx * c[0] - (go[1] - gd[1]) * sy * c[1] + (go[2] - gd[2]) * 2.f * near) * icz * icz;
      gdd[0] = gd[0] * sx * idz;
      gdd[1] = gd[1] * sy * idz;
      gdd[2] = (-gd[0] * sx * d[0] - gd[1] * sy * d[1]) * idz * idz;
      // c = c0 + s d ; s = (near - c0z)/dz
      float gs = gc[0] * d[0] + gc[1] * d[1] + gc[2] * d[2];
#pragma unroll
      for (int a = 0; a < 3; ++a) gdd[a] += s * gc[a];
      gc[2] += -gs * idz;
      gdd[2] += -gs * s * idz;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        go[a] = gc[a];
        gd[a] = gdd[a];
      }
    }
    // d_j = sum_i g_i R_ij ; c0_j = -sum_i t_i R_ij
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      float ti = P[i * 4 + 3];
      float gt = 0.f;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        acc[i * 4 + j] += pr.g[i] * gd[j] - ti * go[j];
        gt -= P[i * 4 + j] * go[j];
      }
      acc[i * 4 + 3] += gt;
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    float v = wave_sum(acc[i]);
    if (lane == 0) red[wv][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < 12) g_pose[b * 12 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

}  // namespace jt

using namespace jt;

extern "C" int jt_pose_forward(const float* se3, const float* noise, const float* gt, int gt_stride, int n_views,
                               float* pose, void* stream) {
  if (!se3 || !gt || !pose || n_views < 1 || (gt_stride != 0 && gt_stride != 12)) return JT_ERR_ARG;
  hipLaunchKernelGGL(k_pose_fwd, dim3((n_views + 63) / 64), dim3(64), 0, (hipStream_t)stream, se3, noise, gt,
                     gt_stride, n_views, pose);
  JT_LAUNCH_CHECK();
  return JT_OK;
}

extern "C" int jt_pose_backward(const float* se3, const float* noise, const float* gt, int gt_stride, int n_views,
                                const float* g_pose, float* g_se3, void* stream) {
  if (!se3 || !gt || !g_pose || !g_se3 || n_views < 1 || (gt_stride != 0 && gt_stride != 12)) return JT_ERR_ARG;
  hipLaunchKernelGGL(k_pose_bwd, dim3((n_views + 63) / 64), dim3(64), 0, (hipStream_t)stream, se3, noise, gt,
                     gt_stride, n_views, g_pose, g_se3);
  JT_LAUNCH_CHECK();
  return JT_OK;
}

// pixel indices of the all_view_rand_grid lattice (model/nerf.py:660-667) from offsets that live in device memory: what a
// replayed hipGraph computes per iteration (the host pokes the two draws in front of the replay)
__global__ __launch_bounds__(256) void k_lattice(const int32_t* __restrict__ offsets, int step, int nx, int ny, int width,
                                                 int64_t* __restrict__ ray_idx) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= nx * ny) return;
  const int ix = k % nx, iy = k / nx;
  ray_idx[k] = (int64_t)(offsets[0] + ix * step) + (int64_t)(offsets[1] + iy * step) * width;
}

extern "C" int jt_lattice_indices(const int32_t* offsets, int step, int nx, int ny, int image_w, int64_t* ray_idx,
                                  void* stream) {
  if (!offsets || !ray_idx || step < 1 || nx < 1 || ny < 1 || image_w < 1 || (long)nx * ny > (1l << 30)) return JT_ERR_ARG;
  hipLaunchKernelGGL(k_lattice, dim3((nx * ny + 255) / 256), dim3(256), 0, (hipStream_t)stream, offsets, step, nx, ny,
                     image_w, ray_idx);
  JT_LAUNCH_CHECK();
  return JT_OK;
}

extern "C" int jt_raygen_forward(const float* pose, const float* intr_inv, const float* intr,
                                 const int64_t* ray_idx, int n_views, int rays_per_view, int image_w, int ndc,
                                 float ndc_near, float* rays_o, float* rays_d, void* stream) {
  if (!pose || !intr_inv || !ray_idx || !rays_o || !rays_d || n_views < 1 || rays_per_view < 1 || image_w < 1)
    return JT_ERR_ARG;
  if (ndc && !intr) return JT_ERR_ARG;
  int n = n_views * rays_per_view;
  hipLaunchKernelGGL(k_raygen_fwd<false>, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, pose, intr_inv, intr,
                     ray_idx, (const int*)nullptr, n_views, rays_per_view, image_w, ndc, ndc_near, rays_o, rays_d);
  JT_LAUNCH_CHECK();
  return JT_OK;
}

extern "C" int jt_raygen_forward_ragged(const float* pose, const float* intr_inv, const float* intr, const int64_t* ray_idx,
                                        const int32_t* view_offset, int n_views, int n_rays, int image_w, int ndc,
                                        float ndc_near, float* rays_o, float* rays_d, void* stream) {
  if (!pose || !intr_inv || !ray_idx || !view_offset || !rays_o || !rays_d || n_views < 1 || n_rays < 1 || image_w < 1)
    return JT_ERR_ARG;
  if (ndc && !intr) return JT_ERR_ARG;
  hipLaunchKernelGGL(k_raygen_fwd<true>, dim3((n_rays + 255) / 256), dim3(256), 0, (hipStream_t)stream, pose, intr_inv, intr,
                     ray_idx, view_offset, n_views, n_rays, image_w, ndc, ndc_near, rays_o, rays_d);
  JT_LAUNCH_CHECK();
  return JT_OK;
}

extern "C" int jt_raygen_backward_ragged(const float* pose, const float* intr_inv, const float* intr, const int64_t* ray_idx,
                                         const int32_t* view_offset, int n_views, int n_rays, int image_w, int ndc,
                                         float ndc_near, const float* g_rays_o, const float* g_rays_d, float* g_pose,
                                         void* stream) {
  if (!pose || !intr_inv || !ray_idx || !view_offset || !g_rays_o || !g_rays_d || !g_pose || n_views < 1 || n_rays < 1 ||
      image_w < 1)
    return JT_ERR_ARG;
  if (ndc && !intr) return JT_ERR_ARG;
  hipLaunchKernelGGL(k_raygen_bwd<true>, dim3(n_views), dim3(256), 0, (hipStream_t)stream, pose, intr_inv, intr, ray_idx,
                     view_offset, n_views, n_rays, image_w, ndc, ndc_near, g_rays_o, g_rays_d, g_pose);
  JT_LAUNCH_CHECK();
  return JT_OK;
}

extern "C" int jt_raygen_backward(const float* pose, const float* intr_inv, const float* intr,
                                  const int64_t* ray_idx, int n_views, int rays_per_view, int image_w, int ndc,
                                  float ndc_near, const float* g_rays_o, const float* g_rays_d, float* g_pose,
                                  void* stream) {
  if (!pose || !intr_inv || !ray_idx || !g_rays_o || !g_rays_d || !g_pose || n_views < 1 || rays_per_view < 1 ||
      image_w < 1)
    return JT_ERR_ARG;
  if (ndc && !intr) return JT_ERR_ARG;
  hipLaunchKernelGGL(k_raygen_bwd<false>, dim3(n_views), dim3(256), 0, (hipStream_t)stream, pose, intr_inv, intr, ray_idx,
                     (const int*)nullptr, n_views, rays_per_view, image_w, ndc, ndc_near, g_rays_o, g_rays_d, g_pose);
  JT_LAUNCH_CHECK();
  return JT_OK;
}
