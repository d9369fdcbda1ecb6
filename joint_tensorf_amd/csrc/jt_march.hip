// Ray marching kernels (gfx950): sampling, density interpolation, transmittance scan, compositing,
// and their backward.  One 64-lane wavefront owns one ray; lanes stride over the ray's samples so
// the transmittance product is a wave prefix scan with a carry between 64-sample chunks.
#include <atomic>
#include <cstdlib>

#include "jt_common.h"
#include "jt_walk.h"

namespace jt {

// ---------------------------------------------------------------------------------------------
// density feature of one point: sum_i sum_c bilinear(P_i^c) * linear(L_i^c)   (bateRF.py:41-94)
// lane-per-sample form: every lane walks its own 16-channel texels with 16-byte loads.
// ---------------------------------------------------------------------------------------------
__device__ inline float density_feature(const Dev& D, const float n[3]) {
  float feat = 0.f;
  const int C = D.Cd;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    PlaneTaps t = plane_taps(n[kM0(i)], n[kM1(i)], D.ph[i], D.pw[i], C);
    Axis l = axis_taps(n[kV(i)], D.ll[i]);
    const float* P = D.dP[i];
    const float* L = D.dL[i];
    const int l0 = l.c0 * C, l1 = l.c1 * C;
    float s = 0.f;
    for (int q = 0; q < C; q += 4) {
      float4 a = ld4(P + t.o00 + q), b = ld4(P + t.o10 + q), c = ld4(P + t.o01 + q), d = ld4(P + t.o11 + q);
      float4 u = ld4(L + l0 + q), v = ld4(L + l1 + q);
      float px = t.w00 * a.x + t.w10 * b.x + t.w01 * c.x + t.w11 * d.x;
      float py = t.w00 * a.y + t.w10 * b.y + t.w01 * c.y + t.w11 * d.y;
      float pz = t.w00 * a.z + t.w10 * b.z + t.w01 * c.z + t.w11 * d.z;
      float pw = t.w00 * a.w + t.w10 * b.w + t.w01 * c.w + t.w11 * d.w;
      s += px * (l.w0 * u.x + l.w1 * v.x) + py * (l.w0 * u.y + l.w1 * v.y) + pz * (l.w0 * u.z + l.w1 * v.z) +
           pw * (l.w0 * u.w + l.w1 * v.w);
    }
    feat += s;
  }
  return feat;
}

// d(density feature) / d(normalised coordinates) of one point (bateRF.py:41-94 differentiated; grid_sample's coordinate
// backward: taps outside the factor count as zero values, the fractional weights stay as they are), lane-per-sample form:
// what the pose-only backward needs of the density path -- no walkers, no factor gradients
__device__ inline void density_feature_grad(const Dev& D, const float n[3], float gn[3]) {
  gn[0] = gn[1] = gn[2] = 0.f;
  const int C = D.Cd;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const PlaneTaps t = plane_taps(n[kM0(i)], n[kM1(i)], D.ph[i], D.pw[i], C);
    const Axis l = axis_taps(n[kV(i)], D.ll[i]);
    const float* P = D.dP[i];
    const float* L = D.dL[i];
    const int l0 = l.c0 * C, l1 = l.c1 * C;
    const float m00 = t.ax.m0 * t.ay.m0, m10 = t.ax.m1 * t.ay.m0, m01 = t.ax.m0 * t.ay.m1, m11 = t.ax.m1 * t.ay.m1;
    const float fx = t.ax.f, fy = t.ay.f;
    float sx = 0.f, sy = 0.f, sl = 0.f;
    for (int q = 0; q < C; q += 4) {
      const float4 a = ld4(P + t.o00 + q), b = ld4(P + t.o10 + q), c = ld4(P + t.o01 + q), d = ld4(P + t.o11 + q);
      const float4 u = ld4(L + l0 + q), v = ld4(L + l1 + q);
      const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w}, cv[4] = {c.x, c.y, c.z, c.w},
                  dv[4] = {d.x, d.y, d.z, d.w}, uv[4] = {u.x, u.y, u.z, u.w}, vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float am = av[k] * m00, bm = bv[k] * m10, cm = cv[k] * m01, dm = dv[k] * m11;
        const float um = uv[k] * l.m0, vm = vv[k] * l.m1;
        const float pv = t.w00 * av[k] + t.w10 * bv[k] + t.w01 * cv[k] + t.w11 * dv[k];
        const float lv = l.w0 * uv[k] + l.w1 * vv[k];
        sx += lv * ((1.f - fy) * (bm - am) + fy * (dm - cm));
        sy += lv * ((1.f - fx) * (cm - am) + fx * (dm - bm));
        sl += pv * (vm - um);
      }
    }
    gn[kM0(i)] += sx * t.ax.scale;
    gn[kM1(i)] += sy * t.ay.scale;
    gn[kV(i)] += sl * l.scale;
  }
}

// density feature AND its coordinate gradient from one gather of the taps (the forward march of a pose-only render: the backward
// then needs no second walk over the density factors) -- density_feature's sum and density_feature_grad's sums, term by term
__device__ inline float density_feature_with_grad(const Dev& D, const float n[3], float gn[3]) {
  gn[0] = gn[1] = gn[2] = 0.f;
  float feat = 0.f;
  const int C = D.Cd;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const PlaneTaps t = plane_taps(n[kM0(i)], n[kM1(i)], D.ph[i], D.pw[i], C);
    const Axis l = axis_taps(n[kV(i)], D.ll[i]);
    const float* P = D.dP[i];
    const float* L = D.dL[i];
    const int l0 = l.c0 * C, l1 = l.c1 * C;
    const float m00 = t.ax.m0 * t.ay.m0, m10 = t.ax.m1 * t.ay.m0, m01 = t.ax.m0 * t.ay.m1, m11 = t.ax.m1 * t.ay.m1;
    const float fx = t.ax.f, fy = t.ay.f;
    float s = 0.f, sx = 0.f, sy = 0.f, sl = 0.f;
    for (int q = 0; q < C; q += 4) {
      const float4 a = ld4(P + t.o00 + q), b = ld4(P + t.o10 + q), c = ld4(P + t.o01 + q), d = ld4(P + t.o11 + q);
      const float4 u = ld4(L + l0 + q), v = ld4(L + l1 + q);
      const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w}, cv[4] = {c.x, c.y, c.z, c.w},
                  dv[4] = {d.x, d.y, d.z, d.w}, uv[4] = {u.x, u.y, u.z, u.w}, vv[4] = {v.x, v.y, v.z, v.w};
      float part = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float am = av[k] * m00, bm = bv[k] * m10, cm = cv[k] * m01, dm = dv[k] * m11;
        const float um = uv[k] * l.m0, vm = vv[k] * l.m1;
        const float pv = t.w00 * av[k] + t.w10 * bv[k] + t.w01 * cv[k] + t.w11 * dv[k];
        const float lv = l.w0 * uv[k] + l.w1 * vv[k];
        part += pv * lv;
        sx += lv * ((1.f - fy) * (bm - am) + fy * (dm - cm));
        sy += lv * ((1.f - fx) * (cm - am) + fx * (dm - bm));
        sl += pv * (vm - um);
      }
      s += part;
    }
    feat += s;
    gn[kM0(i)] += sx * t.ax.scale;
    gn[kM1(i)] += sy * t.ay.scale;
    gn[kV(i)] += sl * l.scale;
  }
  return feat;
}

// alpha_i = 1 - exp(-sigma_i * (delta_i * distance_scale))    (tensorBase.py:59, batBase.py:122)
__device__ inline float sample_alpha(const Dev& D, float feat, bool valid, float delta, float* sigma_out) {
  float sigma = valid ? density_act(D.act, feat + D.shift) : 0.f;
  *sigma_out = sigma;
  return 1.f - expf(-sigma * (delta * D.dist_scale));
}

// inclusive wave product scan; returns the exclusive value in *excl and the wave total
__device__ inline float wave_prod_scan(float f, int lane, float* excl) {
  float inc = f;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    float t = __shfl_up(inc, o);
    if (lane >= o) inc *= t;
  }
  float e = __shfl_up(inc, 1);
  *excl = (lane == 0) ? 1.f : e;
  return __shfl(inc, 63);
}

// ---------------------------------------------------------------------------------------------
// K1: march forward
// ---------------------------------------------------------------------------------------------
// GRAD (a pose-only render): d feature / d normalised coordinates of every in-box sample goes to dfeat_dn, three planes of
// [R * S] floats -- the taps are in registers here; k_march_bwd_scan<2> reads them back instead of gathering a second time
template <bool GRAD>
__global__ __launch_bounds__(256) void k_march_fwd(Dev D, const float* __restrict__ rays_o,
                                                   const float* __restrict__ rays_d,
                                                   const float* __restrict__ jitter,
                                                   const float* __restrict__ zvals, int R,
                                                   float* __restrict__ sigma_feat, float* __restrict__ weight,
                                                   float* __restrict__ tmin_out, int* __restrict__ count,
                                                   uint16_t* __restrict__ sidx, float* __restrict__ opacity,
                                                   float* __restrict__ depth, float* __restrict__ dfeat_dn) {
  const int lane = threadIdx.x & 63;
  // (workgroups that share an XCD take neighbouring rays: in a full-image render those are neighbouring pixels)
  const int ray = xcd_swizzle(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6);
  if (ray >= R) return;
  Ray r;
  load_ray(D, rays_o, rays_d, jitter, nullptr, ray, r);
  const int S = D.S;
  float carry = 1.f, acc = 0.f, dep = 0.f;
  int cnt = 0;
  const size_t row = (size_t)ray * S;
  for (int base = 0; base < S; base += 64) {
    const int i = base + lane;
    const bool live = i < S;
    float z0 = 0.f, delta = 0.f, feat = 0.f;
    bool valid = false;
    if (live) {
      z0 = sample_z(D, r, zvals, i);
      if (i < S - 1) delta = (sample_z(D, r, zvals, i + 1) - z0) * r.norm;
      float p[3], n[3];
      valid = sample_valid(D, r, z0, p);
      if (valid) {
        normalize(D, p, n);
        if (GRAD) {
          float gn[3];
          feat = density_feature_with_grad(D, n, gn);
          const size_t RS = (size_t)R * S;
          dfeat_dn[row + i] = gn[0];
          dfeat_dn[RS + row + i] = gn[1];
          dfeat_dn[2 * RS + row + i] = gn[2];
        } else {
          feat = density_feature(D, n);
        }
      }
    }
    float sigma;
    float alpha = sample_alpha(D, feat, valid, delta, &sigma);
    float f = live ? (1.f - alpha + 1e-10f) : 1.f;
    float excl;
    float total = wave_prod_scan(f, lane, &excl);
    float T = carry * excl;
    carry *= total;
    float w = live ? alpha * T : 0.f;
    if (live) {
      sigma_feat[row + i] = feat;
      weight[row + i] = w;
    }
    acc += w;
    dep += w * z0;
    const bool shade = live && (w > D.thres);
    unsigned long long bal = __ballot(shade);
    if (shade) {
      int rank = __popcll(bal & ((1ull << lane) - 1ull));
      sidx[row + cnt + rank] = (uint16_t)i;
    }
    cnt += __popcll(bal);
  }
  acc = wave_sum(acc);
  dep = wave_sum(dep);
  if (lane == 0) {
    tmin_out[ray] = r.tmin;
    count[ray] = cnt;
    opacity[ray] = acc;
    // depth = sum w z + (1-acc) * ray_dir_z - near + 0.05        (batBase.py:147-150)
    depth[ray] = dep + (1.f - acc) * r.d[2] - (D.near_dev ? *D.near_dev : D.near_) + 0.05f;
  }
}

// exclusive scan of per-ray shade counts: one workgroup walks the rays in coalesced chunks of 1 024 (wave scan by
// shuffles, the sixteen wave totals through LDS, a running carry), 3 us for a training batch, ~1 us per further chunk
__global__ __launch_bounds__(1024) void k_scan_counts(const int* __restrict__ count, int* __restrict__ offset,
                                                      int R) {
  __shared__ int wsum[16];
  __shared__ int carry_s;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  int carry = 0;
  for (int base = 0; base < R; base += 1024) {
    const int i = base + t;
    const int v = (i < R) ? count[i] : 0;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int u = __shfl_up(inc, o);
      if (lane >= o) inc += u;
    }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    if (wv == 0) {
      int w = (lane < 16) ? wsum[lane] : 0;
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        const int u = __shfl_up(w, o);
        if (lane >= o) w += u;
      }
      if (lane < 16) wsum[lane] = w;  // inclusive over the waves
      if (lane == 15) carry_s = w;
    }
    __syncthreads();
    const int before = carry + (wv ? wsum[wv - 1] : 0);
    if (i < R) offset[i] = before + inc - v;
    carry += carry_s;
    __syncthreads();
  }
  if (t == 0) offset[R] = carry;
}

// entry -> (ray, sample) map, one wave per ray
__global__ __launch_bounds__(256) void k_shade_list(Dev D, const float* __restrict__ rays_d, int R,
                                                    const int* __restrict__ offset,
                                                    const uint16_t* __restrict__ sidx, int* __restrict__ eray,
                                                    int* __restrict__ esmp, float* __restrict__ vdir, int cap) {
  const int lane = threadIdx.x & 63;
  const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= R) return;
  const int off = offset[ray], n = offset[ray + 1] - off;
  float d0 = rays_d[ray * 3], d1 = rays_d[ray * 3 + 1], d2 = rays_d[ray * 3 + 2];
  if (D.ndc) {
    float nr = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
    d0 /= nr;
    d1 /= nr;
    d2 /= nr;
  }
  for (int k = lane; k < n; k += 64) {
    int e = off + k;
    if (e >= cap) break;
    eray[e] = ray;
    esmp[e] = sidx[(size_t)ray * D.S + k];
    vdir[e * 3] = d0;
    vdir[e * 3 + 1] = d1;
    vdir[e * 3 + 2] = d2;
  }
}

// ---------------------------------------------------------------------------------------------
// K3: composite forward (batBase.py:142-159)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_composite_fwd(Dev D, int R, const int* __restrict__ offset,
                                                       const uint16_t* __restrict__ sidx,
                                                       const float* __restrict__ weight,
                                                       const float* __restrict__ rgb_s,
                                                       const float* __restrict__ opacity, float* __restrict__ rgb,
                                                       int* __restrict__ clamp_mask) {
  const int lane = threadIdx.x & 63;
  const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= R) return;
  const int off = offset[ray], n = offset[ray + 1] - off;
  const size_t row = (size_t)ray * D.S;
  float c0 = 0.f, c1 = 0.f, c2 = 0.f;
  for (int k = lane; k < n; k += 64) {
    float w = weight[row + sidx[row + k]];
    const float* c = rgb_s + (size_t)(off + k) * 3;
    c0 += w * c[0];
    c1 += w * c[1];
    c2 += w * c[2];
  }
  c0 = wave_sum(c0);
  c1 = wave_sum(c1);
  c2 = wave_sum(c2);
  if (lane == 0) {
    float bg = D.white_bg ? (1.f - opacity[ray]) : 0.f;
    float v[3] = {c0 + bg, c1 + bg, c2 + bg};
    int m = 0;
    for (int ch = 0; ch < 3; ++ch) {
      if (v[ch] >= 0.f && v[ch] <= 1.f) m |= (1 << ch);  // clamp passes gradient on [0,1]
      rgb[ray * 3 + ch] = fminf(fmaxf(v[ch], 0.f), 1.f);
    }
    clamp_mask[ray] = m;
  }
}

// g_rgb_s[e] = weight_e * g_rgb[ray] (clamp-masked)
__global__ __launch_bounds__(256) void k_composite_bwd(Dev D, const int* __restrict__ offset, int R,
                                                       const int* __restrict__ eray, const int* __restrict__ esmp,
                                                       const float* __restrict__ weight,
                                                       const int* __restrict__ clamp_mask,
                                                       const float* __restrict__ g_rgb, float* __restrict__ g_rgb_s,
                                                       int cap) {
  const int total = min(offset[R], cap);
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    int ray = eray[e];
    float w = weight[(size_t)ray * D.S + esmp[e]];
    int m = clamp_mask[ray];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) g_rgb_s[(size_t)e * 3 + ch] = ((m >> ch) & 1) ? w * g_rgb[ray * 3 + ch] : 0.f;
  }
}

// ---------------------------------------------------------------------------------------------
// march backward, two launches.
//
// k_march_bwd_scan -- one wave per ray:
//   A (forward order)  alpha_i, T_i from the stored sigma_feat; G_i = dL/dw_i
//   B (reverse order)  suffix sums -> dL/dalpha_i -> dL/dsigma_feat_i, written to gfeat[R][S]; the in-box
//                      samples with a non-zero gradient are listed in vlist[R][S] (ascending), nvalid[R]
//   D                  the appearance path's coordinate gradients of the ray's shaded samples and the NDC
//                      |d| term seed g_rays_o / g_rays_d
// k_march_bwd_walk -- one 16-lane group per run of 32 listed samples (lane = density channel):
//   re-gathers the density taps of the three planes, run-length scatters the plane / line gradients
//   (jt_walk.h) and adds the run's coordinate gradients to g_rays_o / g_rays_d.
// LDS of the scan kernel per wave: 2 float arrays of S entries + one float per 64-sample chunk.
// ---------------------------------------------------------------------------------------------
// POSE: the pose-only form (density coordinate gradients taken here, 179 registers); the training form keeps its occupancy
//       POSE = 2: the same from the derivatives the pose-only forward march stored (dfeat_dn): no gather here at all
template <int POSE>
__global__ __launch_bounds__(256) void k_march_bwd_scan(Dev D, const float* __restrict__ rays_o,
                                                        const float* __restrict__ rays_d,
                                                        const float* __restrict__ jitter,
                                                        const float* __restrict__ zvals, int R,
                                                        const float* __restrict__ sigma_feat,
                                                        const float* __restrict__ weight,
                                                        const float* __restrict__ tmin_in,
                                                        const int* __restrict__ offset,
                                                        const uint16_t* __restrict__ sidx,
                                                        const float* __restrict__ rgb_s,
                                                        const int* __restrict__ clamp_mask,
                                                        const float* __restrict__ g_rgb,
                                                        const float* __restrict__ g_opacity,
                                                        const float* __restrict__ g_xyz_app,
                                                        float* __restrict__ gfeat, uint16_t* __restrict__ vlist,
                                                        int* __restrict__ nvalid_out, float* __restrict__ g_rays_o,
                                                        float* __restrict__ g_rays_d, long long* __restrict__ rays_fixed,
                                                        int Spad, const float* __restrict__ dfeat_dn) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const int ray = blockIdx.x * 4 + wv;
  if (ray >= R) return;
  // per wave: alpha [Spad], G [Spad] and the transmittance carried INTO each 64-sample chunk [Spad / 64]: pass B rebuilds a sample's
  // transmittance as (carry of its chunk) x (exclusive wave product of the chunk's factors) -- the forward's own two factors, bit
  // for bit -- instead of reading a third array back: 8 instead of 12 KB per ray at S = 1 000, five workgroups per CU instead
  // of three, so that the 4 071 rays of the forward-facing configuration are resident at once (one round instead of two)
  const int wstride = 2 * Spad + (Spad >> 6);
  float* s_alpha = reinterpret_cast<float*>(smem) + (size_t)wv * wstride;
  float* s_G = s_alpha + Spad;
  float* s_carry = s_G + Spad;

  Ray r;
  load_ray(D, rays_o, rays_d, jitter, tmin_in, ray, r);
  const int S = D.S;
  const size_t row = (size_t)ray * S;
  const int off = offset[ray];
  const int m = clamp_mask[ray];
  const float gr0 = (m & 1) ? g_rgb[ray * 3] : 0.f;
  const float gr1 = (m & 2) ? g_rgb[ray * 3 + 1] : 0.f;
  const float gr2 = (m & 4) ? g_rgb[ray * 3 + 2] : 0.f;
  const float gacc = g_opacity ? g_opacity[ray] : 0.f;
  // d(rgb_map)/dw_i = c_i - bg  with bg = 1 when the white background is composited
  const float bgsum = D.white_bg ? (gr0 + gr1 + gr2) : 0.f;

  // ---- pass A ----
  float carry = 1.f;
  int cnt = 0;
  for (int base = 0; base < S; base += 64) {
    const int i = base + lane;
    const bool live = i < S;
    float delta = 0.f, feat = 0.f, wst = 0.f;
    bool valid = false;
    if (live) {
      float z0 = sample_z(D, r, zvals, i);
      if (i < S - 1) delta = (sample_z(D, r, zvals, i + 1) - z0) * r.norm;
      float p[3];
      valid = sample_valid(D, r, z0, p);
      feat = sigma_feat[row + i];
      wst = weight[row + i];
    }
    float sigma;
    float alpha = sample_alpha(D, feat, valid, delta, &sigma);
    float f = live ? (1.f - alpha + 1e-10f) : 1.f;
    float excl;
    float total = wave_prod_scan(f, lane, &excl);
    if (lane == 0) s_carry[base >> 6] = carry;
    carry *= total;
    const bool shade = live && (wst > D.thres);
    unsigned long long bal = __ballot(shade);
    float Gw = gacc - bgsum;
    if (shade) {
      int e = off + cnt + __popcll(bal & ((1ull << lane) - 1ull));
      const float* c = rgb_s + (size_t)e * 3;
      Gw += gr0 * c[0] + gr1 * c[1] + gr2 * c[2];
    }
    cnt += __popcll(bal);
    if (live) {
      s_alpha[i] = alpha;
      s_G[i] = Gw;
    }
  }
  // ---- pass B (reverse) ----
  float suffix = 0.f;  // sum_{j>i} G_j w_j carried from later chunks
  float gnorm = 0.f;   // NDC: dL/d|d|
  float pgo[3] = {0.f, 0.f, 0.f}, pgd[3] = {0.f, 0.f, 0.f};  // pose_density: this lane's share of the density path's ray gradient
  const int nchunk = (S + 63) / 64;
  for (int c = nchunk - 1; c >= 0; --c) {
    const int i = c * 64 + lane;
    const bool live = i < S;
    float alpha = 0.f, T = 0.f, Gw = 0.f, delta = 0.f, feat = 0.f;
    bool valid = false;
    if (live) {
      alpha = s_alpha[i];
      Gw = s_G[i];
      float z0 = sample_z(D, r, zvals, i);
      if (i < S - 1) delta = (sample_z(D, r, zvals, i + 1) - z0) * r.norm;
      float p[3];
      valid = sample_valid(D, r, z0, p);
      feat = sigma_feat[row + i];
    }
    {
      float excl_t;
      (void)wave_prod_scan(live ? (1.f - alpha + 1e-10f) : 1.f, lane, &excl_t);
      T = s_carry[c] * excl_t;   // = the forward's carry * excl of this sample
    }
    float v = live ? Gw * (alpha * T) : 0.f;
    float inc = v;  // inclusive suffix scan over lanes
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      float t = __shfl_down(inc, o);
      if (lane + o < 64) inc += t;
    }
    float excl_s = __shfl_down(inc, 1);
    if (lane == 63) excl_s = 0.f;
    float after = excl_s + suffix;  // sum over j > i
    suffix += __shfl(inc, 0);
    float f = 1.f - alpha + 1e-10f;
    float g_alpha = Gw * T - after / f;
    float one_m = 1.f - alpha;  // = exp(-sigma delta scale)
    float dsc = delta * D.dist_scale;
    float g_sigma = g_alpha * dsc * one_m;
    float x = feat + D.shift;
    float g_feat = valid ? g_sigma * density_act_grad(D.act, x) : 0.f;
    if (D.ndc && valid && live) {
      // delta = dz * |d|  ->  dalpha/d|d| = sigma * dz * scale * (1-alpha)
      float sigma = density_act(D.act, x);
      float dz = (r.norm > 0.f) ? delta / r.norm : 0.f;
      gnorm += g_alpha * sigma * dz * D.dist_scale * one_m;
    }
    if (live) {
      gfeat[row + i] = g_feat;
      s_G[i] = g_feat;
    }
    // pose-only backward: the density path's coordinate gradient of this sample right here (lane per sample, the taps gathered
    // as in the forward march) -- the walk, whose only product would be these sums, is not launched
    if (POSE && live && g_feat != 0.f) {
      const float z0 = sample_z(D, r, zvals, i);
      float gn[3];
      if (POSE == 2) {
        const size_t RS = (size_t)R * S;
        gn[0] = dfeat_dn[row + i];
        gn[1] = dfeat_dn[RS + row + i];
        gn[2] = dfeat_dn[2 * RS + row + i];
      } else {
        float p[3], nrm[3];
        sample_point(D, r, z0, p);
        normalize(D, p, nrm);
        density_feature_grad(D, nrm, gn);
      }
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float g = g_feat * gn[a] * D.inv[a];
        pgo[a] += g;
        pgd[a] += g * z0;
      }
    }
  }
  // ascending list of the samples that carry a density gradient
  int nvalid = 0;
  for (int base = 0; base < S; base += 64) {
    const int i = base + lane;
    const bool keep = (i < S) && (s_G[i] != 0.f);
    unsigned long long bal = __ballot(keep);
    if (keep) vlist[row + nvalid + __popcll(bal & ((1ull << lane) - 1ull))] = (uint16_t)i;
    nvalid += __popcll(bal);
  }
  // ---- pass D: appearance coordinate gradients of this ray's shaded samples ----
  float go[3] = {pgo[0], pgo[1], pgo[2]}, gd[3] = {pgd[0], pgd[1], pgd[2]};
  const int n = offset[ray + 1] - off;
  if (g_xyz_app) {
    for (int k = lane; k < n; k += 64) {
      int i = sidx[row + k];
      float z = sample_z(D, r, zvals, i);
      const float* gx = g_xyz_app + (size_t)(off + k) * 3;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        go[a] += gx[a];
        gd[a] += gx[a] * z;
      }
    }
  }
  gnorm = wave_sum(gnorm);
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    go[a] = wave_sum(go[a]);
    gd[a] = wave_sum(gd[a]);
  }
  if (lane < 6) rays_fixed[(size_t)ray * 6 + lane] = 0;  // the walk's sums for this ray start from zero
  if (lane == 0) {
    nvalid_out[ray] = nvalid;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      float g = gd[a];
      if (D.ndc && r.norm > 0.f) g += gnorm * r.d[a] / r.norm;
      g_rays_o[ray * 3 + a] = go[a];
      g_rays_d[ray * 3 + a] = g;
    }
  }
}

constexpr int kWalkRun = 32;   // listed samples per 16-lane group
constexpr int kWalkSub = 16;   // samples whose step records are built at a time: one per lane of the group
constexpr int kWalkRecW = kRecWords;  // per sample: the plane's step record; its two spare words carry (g_feat, z)

__device__ inline int sel3(int i, int a, int b, int c) { return i == 0 ? a : (i == 1 ? b : c); }
__device__ inline float sel3f(int i, float a, float b, float c) { return i == 0 ? a : (i == 1 ? b : c); }

// Work item = (ray, PLANE, run of 32 listed samples), one 16-lane group each (lane = density channel); the four groups
// of a wave hold four consecutive runs of the same (ray, plane) (the host pads runs_per_ray to a multiple of 4).
//
// Why one plane per group (round 3; before, a group walked its run through all three planes, 96 dependent steps): the
// walk is LATENCY-bound, not bandwidth- or atomic-rate-bound -- on the LLFF grid its HBM-side traffic is 0.2 x the
// algorithmic bytes and its atomic segments would take 0.39 ms of the 0.93 ms launch (profiles/round3_llff_*_before*).
// A wave's vector-memory operations retire IN ORDER (loads and atomics share vmcnt), so the tap load of step q + 1
// cannot return before the flush atomics of step q - 1 have retired, ~3 000 cycles with every CU issuing; what hides
// that is more independent chains per CU.  Per-plane items are three times as many chains of a third of the length, and
// one walker instead of three fits 6 waves per SIMD instead of 4 (and nothing spills).
#ifndef JT_WALK_PF
#define JT_WALK_PF 2  // tap prefetch distance of the density walk in steps (4 at five waves per SIMD, 5 at four: the same time on
                      // both grids -- 0.43 ms Blender, 0.82-0.84 ms LLFF -- the walk does not wait for its taps)
#endif
// Round 4: PERSISTENT and PLANE-MAJOR, with the LINE gradients of the pass's plane summed in a workgroup-private LDS copy of
// the line.  Measured with the line flushes compiled out (profiles/round4_line_flush_cost.txt): they are 18 % of the walk's
// time on the Blender grid and 32 % on LLFF's, where the rays march along the line axis and leave a line cell every step or
// two.  One workgroup of sixteen waves per CU (four per SIMD) loops over the (ray, four runs) wave items of plane 0, adds its
// LDS line to the real gradient with 64-byte-contiguous float atomics, then plane 1, plane 2.  LLINE = false (deterministic
// mode, or a line that does not fit the LDS): the line goes out through global atomics / fixed point as before.
constexpr int kWalkPrefixRays = 16384;
// kWalkWaves = 16: one workgroup per CU (a long line: LLFF's 55 KB); 8: two per CU when their LDS fits twice (finer grains for
// whatever runs beside the walk: the weight-gradient GEMMs on the auxiliary stream)
template <int CD, bool DET, int LLINE, int kWalkWaves>
__global__ __launch_bounds__(kWalkWaves * 64) void k_march_bwd_walk(Dev D, JtFactors G, const float* __restrict__ rays_o,
                                                        const float* __restrict__ rays_d,
                                                        const float* __restrict__ jitter,
                                                        const float* __restrict__ zvals,
                                                        const float* __restrict__ tmin_in, int R,
                                                        const float* __restrict__ gfeat,
                                                        const uint16_t* __restrict__ vlist,
                                                        const int* __restrict__ nvalid, int runs_per_ray,
                                                        float* __restrict__ g_rays_o, float* __restrict__ g_rays_d,
                                                        long long* __restrict__ rays_fixed, unsigned* __restrict__ bad,
                                                        int line_floats, int prefix) {
  constexpr int NCH = (CD + 15) / 16;
  // + 16 words per group: the four groups of a wave read their own records in the same instruction, and a group
  // stride that is a multiple of the 64 LDS banks would put all four on the same banks
  constexpr int kGroupWords = kWalkSub * kWalkRecW + 16;
  extern __shared__ __align__(16) float s_dyn[];
  float* sline = s_dyn + kWalkWaves * 4 * kGroupWords;  // [line cell][channel] of the pass's plane (LLINE)
  const int cl = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int qpr = runs_per_ray >> 2;               // wave items (four runs) per (ray, plane)
  const int witems = R * qpr;
  // every workgroup owns ONE plane (blockIdx % 3): one LDS line, one pass, one flush, no barrier between planes (a pass per
  // plane in every workgroup cost three tails: +15 %).  Rays differ in length and half of the padded wave items are empty: the
  // workgroups of a plane stride over the NON-EMPTY items only -- every workgroup builds the inclusive prefix of the rays' item
  // counts in LDS (PREFIX: R <= kWalkPrefixRays) and finds an item's ray by bisection.  (Drawing items from a global counter
  // instead was twice as slow: one word serves ~88 returning atomics per microsecond, MI355X_MICROARCH.md "dequeue".)
  const int nblk = (int)gridDim.x;
  const int pl = (int)blockIdx.x % 3, pb = (int)blockIdx.x / 3, npeers = (nblk - pl + 2) / 3;
  // (LLINE = 2: the line copy holds doubles, line_floats counts its 4-byte words)
  int* s_pre = reinterpret_cast<int*>(sline + (LLINE ? line_floats : 0));   // [R] inclusive prefix of ceil(nvalid / 128)
  int n_items = witems;
  if (prefix) {
    __shared__ int s_wsum[kWalkWaves];
    const int per = (R + kWalkWaves * 64 - 1) / (kWalkWaves * 64), r0 = (int)threadIdx.x * per;
    int loc = 0;
    for (int k = 0; k < per; ++k)
      if (r0 + k < R) loc += (nvalid[r0 + k] + 4 * kWalkRun - 1) / (4 * kWalkRun);
    int inc = loc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(inc, o);
      if ((int)(threadIdx.x & 63) >= o) inc += t;
    }
    if ((threadIdx.x & 63) == 63) s_wsum[wv] = inc;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wv; ++w) base += s_wsum[w];
    int run_sum = base + inc - loc;
    for (int k = 0; k < per; ++k)
      if (r0 + k < R) {
        run_sum += (nvalid[r0 + k] + 4 * kWalkRun - 1) / (4 * kWalkRun);
        s_pre[r0 + k] = run_sum;
      }
    __syncthreads();
    n_items = s_pre[R - 1];
  }
  {
  const int H = sel3(pl, D.ph[0], D.ph[1], D.ph[2]), W = sel3(pl, D.pw[0], D.pw[1], D.pw[2]),
            LL = sel3(pl, D.ll[0], D.ll[1], D.ll[2]);
  float* gline = pl == 0 ? G.density_line[0] : (pl == 1 ? G.density_line[1] : G.density_line[2]);
  if (LLINE) {
    for (int i = threadIdx.x; i < LL * CD * (LLINE == 2 ? 2 : 1); i += kWalkWaves * 64) sline[i] = 0.f;
    __syncthreads();
  }
#pragma unroll 1
  for (int it = pb * kWalkWaves + wv; it < n_items; it += npeers * kWalkWaves) {
  int ray, quad;
  if (prefix) {  // first ray whose inclusive prefix exceeds the item index (uniform over the wave)
    int lo = 0, hi = R - 1;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (s_pre[mid] > it) hi = mid; else lo = mid + 1;
    }
    ray = __builtin_amdgcn_readfirstlane(lo);
    quad = it - (ray > 0 ? s_pre[ray - 1] : 0);
  } else {
    ray = __builtin_amdgcn_readfirstlane(it / qpr);
    quad = it - ray * qpr;
  }
  const int rem = __builtin_amdgcn_readfirstlane(pl * runs_per_ray + 4 * quad);
  const int run = rem - pl * runs_per_ray + (grp & 3);
  const int nv = nvalid[ray];
  const int k0 = run * kWalkRun;
  if ((rem - pl * runs_per_ray) * kWalkRun >= nv) continue;  // the whole wave is past the ray's list
  const bool has_run = k0 < nv;
  const int k1 = min(k0 + kWalkRun, nv);
  Ray r;
  load_ray(D, rays_o, rays_d, jitter, tmin_in, ray, r);
  const size_t row = (size_t)ray * D.S;
  float* rec = s_dyn + grp * kGroupWords;
  const float* P = pl == 0 ? D.dP[0] : (pl == 1 ? D.dP[1] : D.dP[2]);
  const float* L = pl == 0 ? D.dL[0] : (pl == 1 ? D.dL[1] : D.dL[2]);
  // axes of the plane: x <-> kM0(pl), y <-> kM1(pl), line <-> kV(pl)
  const int a0 = pl == 2 ? 1 : 0, a1 = pl == 0 ? 1 : 2, a2 = 2 - pl;
  RecWalker<NCH, CD, DET ? 1 : 0, LLINE> wk;
  wk.init(pl == 0 ? G.density_plane[0] : (pl == 1 ? G.density_plane[1] : G.density_plane[2]), LLINE ? sline : gline, cl, DET,
          bad);
  const float sx = 0.5f * (float)(W - 1) * sel3f(a0, D.inv[0], D.inv[1], D.inv[2]),
              sy = 0.5f * (float)(H - 1) * sel3f(a1, D.inv[0], D.inv[1], D.inv[2]),
              sl = 0.5f * (float)(LL - 1) * sel3f(a2, D.inv[0], D.inv[1], D.inv[2]);
  // the lane's un-reduced partials of dL/d(o, d) along the plane's x / y / line axes
  float gox = 0.f, goy = 0.f, gol = 0.f, gdx = 0.f, gdy = 0.f, gdl = 0.f;

  if (has_run) {
    for (int kb = k0; kb < k1; kb += kWalkSub) {
      const int ns = min(kWalkSub, k1 - kb);
      // step record of the listed sample kb + cl on this plane; lanes past the end of the run mirror the last live
      // sample with a zero gradient (a no-op for the walker)
      {
        const int kk = kb + min(cl, ns - 1);
        const bool has_prev = (cl < ns) && (kk > k0);
        const int i = vlist[row + kk], ip = vlist[row + (has_prev ? kk - 1 : kk)];
        const float z = sample_z(D, r, zvals, i), zp = sample_z(D, r, zvals, ip);
        float p[3], n[3], q[3], m[3];
        sample_point(D, r, z, p);
        normalize(D, p, n);
        sample_point(D, r, zp, q);
        normalize(D, q, m);
        float* rc = rec + cl * kWalkRecW;
        make_step_rec(sel3f(a0, n[0], n[1], n[2]), sel3f(a1, n[0], n[1], n[2]), sel3f(a2, n[0], n[1], n[2]),
                      sel3f(a0, m[0], m[1], m[2]), sel3f(a1, m[0], m[1], m[2]), sel3f(a2, m[0], m[1], m[2]), has_prev, H,
                      W, LL, CD, rc);
        rc[18] = (cl < ns) ? gfeat[row + i] : 0.f;
        rc[19] = z;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      auto step = [&](TapBuf<NCH>& tv, int q) {
        const float* rq = rec + q * kWalkRecW;
        const float gc = rq[18], zc = rq[19];
        wk.advance(rq);
        float g[NCH];
#pragma unroll
        for (int k = 0; k < NCH; ++k) g[k] = wk.live[k] ? gc : 0.f;
        float aix = 0.f, aiy = 0.f, ail = 0.f;
        wk.add(tv, rq, g, aix, aiy, ail);
        gox += aix;
        gdx += aix * zc;
        goy += aiy;
        gdy += aiy * zc;
        gol += ail;
        gdl += ail * zc;
      };
      // taps JT_WALK_PF steps ahead of the step that consumes them (JT_WALK_PF + 1 buffers in rotation)
      constexpr int PF = JT_WALK_PF;
      TapBuf<NCH> buf[PF + 1];
#pragma unroll
      for (int u = 0; u < PF; ++u) wk.load(buf[u], P, L, rec + u * kWalkRecW);
#pragma unroll 1
      for (int q0 = 0; q0 < kWalkSub; q0 += PF + 1) {
#pragma unroll
        for (int u = 0; u <= PF; ++u) {
          const int q = q0 + u;
          if (q + PF < kWalkSub) wk.load(buf[(u + PF) % (PF + 1)], P, L, rec + (q + PF) * kWalkRecW);
          if (q < kWalkSub) step(buf[u], q);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    }
    wk.finish();
  }
  // reduce over the group's channels, then over the wave's four groups (same ray, same plane): six atomics per wave
  float v6[6] = {gox * sx, goy * sy, gol * sl, gdx * sx, gdy * sy, gdl * sl};
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    float v = row16_sum(v6[c]);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    v6[c] = v;
  }
  const int lane = threadIdx.x & 63;
  if (lane < 6) {
    const int c = lane;
    const float v = c == 0 ? v6[0] : c == 1 ? v6[1] : c == 2 ? v6[2] : c == 3 ? v6[3] : c == 4 ? v6[4] : v6[5];
    const int ax = (c % 3) == 0 ? a0 : ((c % 3) == 1 ? a1 : a2);
    // ALWAYS in fixed point (integer atomics commute): the pose gradient does not depend on the order in which the runs of
    // a ray arrive -- test-time pose optimisation and the camera trajectory of a training run are reproducible bit for bit
    // in their density part, for one extra 5 us launch (k_rays_fixed_add)
    fixed_add(rays_fixed + (size_t)ray * 6 + (c < 3 ? 0 : 3) + ax, v, bad);
  }
  }  // wave items of the plane
  if (LLINE) {
    __syncthreads();
    if (gline != nullptr)
      for (int i = threadIdx.x; i < LL * CD; i += kWalkWaves * 64) {
        const float v = (LLINE == 2) ? (float)reinterpret_cast<const double*>(sline)[i] : sline[i];
        if (v != 0.f) atomicAdd(gline + i, v);
      }
    __syncthreads();
  }
  }  // the workgroup's plane
}

// g_rays_o/d [R][3] += fixed-point sums [R][6].  A sum out of the format's safe range becomes NaN, and so does EVERY ray
// gradient while the sticky bad-addend flag is up (a NaN / infinite / huge addend was dropped somewhere in this process's
// fixed-point sums since the flag was last cleared: what float atomics would have carried into the pose as a NaN must not
// turn into a finite number); either way the FINITE_GRAD bit goes into the bound status word (jt_status_bind).
__global__ void k_rays_fixed_add(const long long* __restrict__ f, int R, float* __restrict__ g_rays_o,
                                 float* __restrict__ g_rays_d, const unsigned* __restrict__ bad,
                                 int32_t* __restrict__ status) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R * 6) return;
  const int ray = i / 6, c = i - ray * 6;
  const long long w = f[i];
  const bool poisoned = (*bad != 0u) || w >= kFixedLimit || w <= -kFixedLimit;
  const float v = poisoned ? __builtin_nanf("") : (float)((double)w / kFixedScale);
  if (c < 3) g_rays_o[ray * 3 + c] += v;
  else g_rays_d[ray * 3 + (c - 3)] += v;
  if (poisoned && status) atomicOr(status, JT_STATUS_FINITE_GRAD);
}

// the sticky flag of fixed_add (jt_common.h) and the status word the library reports into
__device__ unsigned g_fixed_bad;
__global__ void k_status_clear(int32_t* status) {
  g_fixed_bad = 0u;
  if (status) *status = 0;
}

// opacity of one step of `length` at arbitrary points (BatBase.compute_alpha, batBase.py:27-41): the dense
// lattice of TensorBase.getDenseAlpha (tensorBase.py:618-634) goes through this
__global__ __launch_bounds__(256) void k_dense_alpha(Dev D, const float* __restrict__ xyz, long n, float length,
                                                     float* __restrict__ alpha) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float p[3] = {xyz[i * 3], xyz[i * 3 + 1], xyz[i * 3 + 2]};
  float sigma = 0.f;
  if (!D.mask || mask_keep(D, p)) {
    float nrm[3];
    normalize(D, p, nrm);
    sigma = density_act(D.act, density_feature(D, nrm) + D.shift);
  }
  alpha[i] = 1.f - expf(-sigma * length);
}

}  // namespace jt

using namespace jt;

static int check_density_shape(const Dev& D) {
  if (D.Cd < 4 || (D.Cd % 4) != 0) return JT_ERR_UNSUPPORTED;
  return JT_OK;
}

extern "C" int jt_version(void) { return JT_VERSION; }

unsigned* jt::fixed_bad_flag() {
  static unsigned* p = nullptr;  // (one GPU per process: SURVEY 8(e))
  if (!p && hipGetSymbolAddress(reinterpret_cast<void**>(&p), HIP_SYMBOL(g_fixed_bad)) != hipSuccess) p = nullptr;
  return p;
}
static std::atomic<int32_t*> g_status_word{nullptr};
extern "C" int jt_status_bind(int32_t* status_word) {
  g_status_word.store(status_word, std::memory_order_relaxed);
  return JT_OK;
}
extern "C" int jt_status_clear(void* stream) {
  if (!jt::fixed_bad_flag()) return JT_ERR_ARG;
  hipLaunchKernelGGL(k_status_clear, dim3(1), dim3(1), 0, (hipStream_t)stream, g_status_word.load(std::memory_order_relaxed));
  JT_LAUNCH_CHECK();
  return JT_OK;
}

// geometry of the current device (jt_common.h: Chip), queried once per device index; a failed query falls back to MI355X's
const jt::Chip& jt::chip() {
  static jt::Chip table[64];
  static std::atomic<int> known[64];
  static const jt::Chip mi355x = {256, 8};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return mi355x;
  if (!known[dev].load(std::memory_order_acquire)) {
    int cus = 0, xcds = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    if (hipDeviceGetAttribute(&xcds, hipDeviceAttributeNumberOfXccs, dev) != hipSuccess || xcds <= 0) xcds = cus >= 64 ? 8 : 1;
    if (cus % xcds != 0) xcds = 1;
    table[dev] = jt::Chip{cus, xcds};     // (two threads racing here write the same values)
    known[dev].store(1, std::memory_order_release);
  }
  return table[dev];
}
/* the persistent grids as this device gets them: {compute units, XCDs} */
extern "C" int jt_chip_geometry(int32_t* out2) {
  if (!out2) return JT_ERR_ARG;
  const jt::Chip& c = jt::chip();
  out2[0] = c.cus, out2[1] = c.xcds;
  return JT_OK;
}

static std::atomic<int> g_deterministic{-1};  // -1: not yet read from the environment
int jt::jt_deterministic() {
  int v = g_deterministic.load(std::memory_order_relaxed);
  if (v < 0) {
    const char* e = getenv("JT_DETERMINISTIC");
    const int from_env = (e && atoi(e) != 0) ? 1 : 0;
    int expected = -1;  // whoever gets here first decides; a concurrent jt_set_deterministic wins over the environment
    g_deterministic.compare_exchange_strong(expected, from_env, std::memory_order_relaxed);
    v = g_deterministic.load(std::memory_order_relaxed);
  }
  return v;
}
extern "C" int jt_set_deterministic(int on) {
  const int prev = jt_deterministic();
  if (on == 0 || on == 1) g_deterministic.store(on, std::memory_order_relaxed);
  return prev;
}

extern "C" int jt_dense_alpha(const JtScene* scene, const JtFactors* factors, const float* xyz, long n, float length,
                              float* alpha, void* stream) {
  Dev D;
  int rc = make_dev(scene, factors, &D);
  if (rc) return rc;
  if (!factors || !xyz || !alpha || n < 0) return JT_ERR_ARG;
  if ((rc = check_density_shape(D))) return rc;
  for (int a = 0; a < 3; ++a)
    if (!D.dP[a] || !D.dL[a]) return JT_ERR_ARG;
  if (n == 0) return JT_OK;
  hipLaunchKernelGGL(k_dense_alpha, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, D, xyz, n,
                     length, alpha);
  JT_LAUNCH_CHECK();
  return JT_OK;
}


static int march_forward(const JtScene* scene, const JtFactors* factors, const float* rays_o,
                         const float* rays_d, const float* jitter, const float* zvals, int n_rays,
                         float* sigma_feat, float* weight, float* tmin, int32_t* shade_count,
                         int32_t* shade_offset, uint16_t* shade_idx, float* opacity, float* depth,
                         float* dfeat_dn, void* stream) {
  Dev D;
  int rc = make_dev(scene, factors, &D);
  if (rc) return rc;
  if (!factors || !rays_o || !rays_d || !sigma_feat || !weight || !tmin || !shade_count || !shade_offset ||
      !shade_idx || !opacity || !depth || n_rays < 1)
    return JT_ERR_ARG;
  if (D.ndc && !zvals) return JT_ERR_ARG;
  if ((rc = check_density_shape(D))) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (dfeat_dn)
    hipLaunchKernelGGL(k_march_fwd<true>, dim3((n_rays + 3) / 4), dim3(256), 0, st, D, rays_o, rays_d, jitter, zvals,
                       n_rays, sigma_feat, weight, tmin, shade_count, shade_idx, opacity, depth, dfeat_dn);
  else
    hipLaunchKernelGGL(k_march_fwd<false>, dim3((n_rays + 3) / 4), dim3(256), 0, st, D, rays_o, rays_d, jitter, zvals,
                       n_rays, sigma_feat, weight, tmin, shade_count, shade_idx, opacity, depth, dfeat_dn);
  JT_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, st, shade_count, shade_offset, n_rays);
  JT_LAUNCH_CHECK();
  return JT_OK;
}

extern "C" int jt_march_forward(const JtScene* scene, const JtFactors* factors, const float* rays_o,
                                const float* rays_d, const float* jitter, const float* zvals, int n_rays,
                                float* sigma_feat, float* weight, float* tmin, int32_t* shade_count,
                                int32_t* shade_offset, uint16_t* shade_idx, float* opacity, float* depth,
                                void* stream) {
  return march_forward(scene, factors, rays_o, rays_d, jitter, zvals, n_rays, sigma_feat, weight, tmin, shade_count,
                       shade_offset, shade_idx, opacity, depth, nullptr, stream);
}

extern "C" int jt_march_forward_pose(const JtScene* scene, const JtFactors* factors, const float* rays_o,
                                     const float* rays_d, const float* jitter, const float* zvals, int n_rays,
                                     float* sigma_feat, float* weight, float* tmin, int32_t* shade_count,
                                     int32_t* shade_offset, uint16_t* shade_idx, float* opacity, float* depth,
                                     float* dfeat_dn, void* stream) {
  if (!dfeat_dn) return JT_ERR_ARG;
  return march_forward(scene, factors, rays_o, rays_d, jitter, zvals, n_rays, sigma_feat, weight, tmin, shade_count,
                       shade_offset, shade_idx, opacity, depth, dfeat_dn, stream);
}

extern "C" int jt_shade_list(const JtScene* scene, const float* rays_d, int n_rays, const int32_t* shade_offset,
                             const uint16_t* shade_idx, int32_t* entry_ray, int32_t* entry_smp, float* viewdirs,
                             int n_entries_max, void* stream) {
  Dev D;
  int rc = make_dev(scene, nullptr, &D);
  if (rc) return rc;
  if (!rays_d || !shade_offset || !shade_idx || !entry_ray || !entry_smp || !viewdirs || n_rays < 1)
    return JT_ERR_ARG;
  if (n_entries_max < 1) return JT_OK;
  hipLaunchKernelGGL(k_shade_list, dim3((n_rays + 3) / 4), dim3(256), 0, (hipStream_t)stream, D, rays_d, n_rays,
                     shade_offset, shade_idx, entry_ray, entry_smp, viewdirs, n_entries_max);
  JT_LAUNCH_CHECK();
  return JT_OK;
}

extern "C" int jt_composite_forward(const JtScene* scene, int n_rays, const int32_t* shade_offset,
                                    const uint16_t* shade_idx, const float* weight, const float* rgb_s,
                                    const float* opacity, float* rgb, int32_t* clamp_mask, void* stream) {
  Dev D;
  int rc = make_dev(scene, nullptr, &D);
  if (rc) return rc;
  if (!shade_offset || !shade_idx || !weight || !opacity || !rgb || !clamp_mask || n_rays < 1) return JT_ERR_ARG;
  hipLaunchKernelGGL(k_composite_fwd, dim3((n_rays + 3) / 4), dim3(256), 0, (hipStream_t)stream, D, n_rays,
                     shade_offset, shade_idx, weight, rgb_s, opacity, rgb, clamp_mask);
  JT_LAUNCH_CHECK();
  return JT_OK;
}

extern "C" int jt_composite_backward(const JtScene* scene, int n_rays, const int32_t* shade_offset,
                                     const int32_t* entry_ray, const int32_t* entry_smp, const float* weight,
                                     const int32_t* clamp_mask, const float* g_rgb, float* g_rgb_s,
                                     int n_entries_max, void* stream) {
  Dev D;
  int rc = make_dev(scene, nullptr, &D);
  if (rc) return rc;
  if (!shade_offset || !entry_ray || !entry_smp || !weight || !clamp_mask || !g_rgb || !g_rgb_s)
    return JT_ERR_ARG;
  if (n_entries_max < 1) return JT_OK;
  int blocks = min((n_entries_max + 255) / 256, 2048);
  hipLaunchKernelGGL(k_composite_bwd, dim3(blocks), dim3(256), 0, (hipStream_t)stream, D, shade_offset, n_rays,
                     entry_ray, entry_smp, weight, clamp_mask, g_rgb, g_rgb_s, n_entries_max);
  JT_LAUNCH_CHECK();
  return JT_OK;
}

static size_t march_bwd_ws_layout(int S, int R, size_t* o_vlist, size_t* o_nvalid, size_t* o_fixed = nullptr) {
  size_t off = 0;
  off += (size_t)R * S * sizeof(float);                 // gfeat
  off = (off + 255) & ~(size_t)255;
  *o_vlist = off;
  off += (size_t)R * S * sizeof(uint16_t);
  off = (off + 255) & ~(size_t)255;
  *o_nvalid = off;
  off += (size_t)R * sizeof(int);
  off = (off + 255) & ~(size_t)255;
  if (o_fixed) *o_fixed = off;
  off += (size_t)R * 6 * sizeof(long long);             // ray-gradient sums in fixed point
  return (off + 255) & ~(size_t)255;
}

extern "C" size_t jt_march_backward_workspace_bytes(const JtScene* scene, int n_rays) {
  if (!scene || n_rays < 1 || scene->n_samples < 1) return 0;
  size_t a, b;
  return march_bwd_ws_layout(scene->n_samples, n_rays, &a, &b);
}

static int march_backward(const JtScene* scene, const JtFactors* factors, const float* rays_o,
                          const float* rays_d, const float* jitter, const float* zvals, int n_rays,
                          const float* sigma_feat, const float* weight, const float* tmin,
                          const int32_t* shade_offset, const uint16_t* shade_idx, const float* rgb_s,
                          const int32_t* clamp_mask, const float* g_rgb, const float* g_opacity,
                          const float* g_xyz_app, const JtFactors* g_factors, float* g_rays_o,
                          float* g_rays_d, void* workspace, size_t workspace_bytes, const float* dfeat_dn, void* stream) {
  Dev D;
  int rc = make_dev(scene, factors, &D);
  if (rc) return rc;
  if (!factors || !rays_o || !rays_d || !sigma_feat || !weight || !tmin || !shade_offset ||
      !shade_idx || !clamp_mask || !g_rgb || !g_rays_o || !g_rays_d || !workspace || n_rays < 1)
    return JT_ERR_ARG;
  for (int a = 0; a < 3; ++a)
    if (g_factors && (!g_factors->density_plane[a] || !g_factors->density_line[a])) return JT_ERR_ARG;
  if (D.ndc && !zvals) return JT_ERR_ARG;
  if ((rc = check_density_shape(D))) return rc;
  size_t o_vlist, o_nvalid, o_fixed;
  if (workspace_bytes < march_bwd_ws_layout(D.S, n_rays, &o_vlist, &o_nvalid, &o_fixed)) return JT_ERR_ARG;
  // JT_DETERMINISTIC with factor gradients wanted: g_factors points at int64 shadow buffers, ray sums in fixed point
  const bool det = jt_deterministic() != 0;
  long long* rays_fixed = reinterpret_cast<long long*>(reinterpret_cast<char*>(workspace) + o_fixed);
  JtFactors no_grads = {};  // g_factors == NULL: gradients w.r.t. the rays only (the walk writes no factor gradient)
  const JtFactors& GF = g_factors ? *g_factors : no_grads;
  float* gfeat = reinterpret_cast<float*>(workspace);
  uint16_t* vlist = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(workspace) + o_vlist);
  int* nvalid = reinterpret_cast<int*>(reinterpret_cast<char*>(workspace) + o_nvalid);
  const int Spad = (D.S + 63) & ~63;
  const size_t lds = (size_t)4 * (2 * Spad + (Spad >> 6)) * sizeof(float);
  if (lds > 160 * 1024) return JT_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  // nothing but the rays wants a gradient (test-time pose optimisation): the density path's coordinate gradient is taken in the
  // scan kernel, lane per sample, and the walk is skipped (JT_POSE_BWD=0, read once: the walk with its targets switched off)
  static const bool pose_env = [] { const char* e = getenv("JT_POSE_BWD"); return !e || atoi(e) != 0; }();
  const bool pose_density = pose_env && !g_factors && !det;
  if (!pose_density) dfeat_dn = nullptr;   // (the stored derivatives serve the pose-only form; the walk gathers for itself)
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_march_bwd_scan<0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_march_bwd_scan<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_march_bwd_scan<2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
#define JT_SCAN_LAUNCH(POSE)                                                                                                   \
  hipLaunchKernelGGL(k_march_bwd_scan<POSE>, dim3((n_rays + 3) / 4), dim3(256), lds, st, D, rays_o, rays_d, jitter, zvals,     \
                     n_rays, sigma_feat, weight, tmin, shade_offset, shade_idx, rgb_s, clamp_mask, g_rgb, g_opacity,           \
                     g_xyz_app, gfeat, vlist, nvalid, g_rays_o, g_rays_d, rays_fixed, Spad, dfeat_dn)
  if (pose_density && dfeat_dn)
    JT_SCAN_LAUNCH(2);
  else if (pose_density)
    JT_SCAN_LAUNCH(1);
  else
    JT_SCAN_LAUNCH(0);
#undef JT_SCAN_LAUNCH
  JT_LAUNCH_CHECK();
  if (pose_density) return JT_OK;  // the ray gradients are complete: no walk, no fixed-point sums to add
  unsigned* bad = jt::fixed_bad_flag();
  if (!bad) return JT_ERR_ARG;
  const int runs = ((D.S + kWalkRun - 1) / kWalkRun + 3) & ~3;  // a wave's four groups: four runs of ONE (ray, plane)
  const long witems = (long)n_rays * (runs / 4);
  int line_floats = 0;
  for (int a = 0; a < 3; ++a) line_floats = std::max(line_floats, D.ll[a] * D.Cd);
  // JT_WALK_LDS_LINE (read once): 0 no LDS line, 1 a float copy, 2 a copy of doubles, unset: doubles where they fit the
  // preferred workgroup shape, floats otherwise.  JT_WALK_WAVES = 8 / 16 forces the workgroup size.
  static const int lline_env = [] { const char* e = getenv("JT_WALK_LDS_LINE"); return e ? atoi(e) : -1; }();
  static const int waves_env = [] { const char* e = getenv("JT_WALK_WAVES"); return e ? atoi(e) : 0; }();
  int nw = 8, lline = 0, prefix = 0;
  size_t wlds = 0;
  // a candidate (line mode, waves, two workgroups per CU wanted): its LDS bytes, or 0 when it does not fit
  auto shape = [&](int lm, int w, bool two, int* pre) -> size_t {
    const size_t rec_bytes = (size_t)w * 4 * (kWalkSub * kWalkRecW + 16) * sizeof(float);
    size_t b = rec_bytes + (size_t)line_floats * sizeof(float) * lm;
    if (b > 158 * 1024) return 0;
    *pre = (n_rays <= kWalkPrefixRays && b + (size_t)n_rays * sizeof(int) <= 158 * 1024) ? 1 : 0;
    if (*pre) b += (size_t)n_rays * sizeof(int);
    if (two && 2 * (b + 256) > 160 * 1024) return 0;
    return b;
  };
  {
    // eight-wave workgroups only where two of them fit a CU (otherwise sixteen waves: four per SIMD either way).  A shape that
    // keeps the prefix table of the rays' item counts beats one that does not (without it the items come from a global counter,
    // twice as slow): doubles with the table, floats with the table, then the same without it
    const int modes[2] = {2, 1};
    bool found = false;
    const bool want_prefix = n_rays <= kWalkPrefixRays;
    for (int need_pre = want_prefix ? 1 : 0; need_pre >= 0 && !found; --need_pre)
      for (int mi = 0; mi < 2 && !found; ++mi) {
        const int lm = modes[mi];
        if (det || lline_env == 0 || (lline_env > 0 && lline_env != lm)) continue;
        for (int w = 8; w <= 16 && !found; w += 8) {
          if (waves_env == 8 || waves_env == 16) {
            if (w != waves_env) continue;
          }
          int pre = 0;
          const size_t b = shape(lm, w, w == 8 && waves_env != 8, &pre);
          if (b && pre >= need_pre) nw = w, lline = lm, prefix = pre, wlds = b, found = true;
        }
      }
    if (!found) {
      for (int w = 8; w <= 16 && !found; w += 8) {
        if ((waves_env == 8 || waves_env == 16) && w != waves_env) continue;
        int pre = 0;
        const size_t b = shape(0, w, w == 8 && waves_env != 8, &pre);
        if (b) nw = w, lline = 0, prefix = pre, wlds = b, found = true;
      }
    }
    if (!found) return JT_ERR_UNSUPPORTED;
  }
  // (JT_WALK_WGS, read once: fewer workgroups -- a multiple of three -- leave CUs to whatever runs beside the walk)
  static const int wgs_env = [] { const char* e = getenv("JT_WALK_WGS"); return e ? atoi(e) : 0; }();
  // (one sixteen-wave / two eight-wave workgroups per CU, a multiple of three: 255 / 510 on MI355X's 256 CUs)
  const long wg_full = (long)(chip().cus / 3 * 3) * (16 / nw);
  const long wg_cap = wgs_env > 0 ? std::min<long>(wgs_env / 3 * 3, wg_full) : wg_full;
  const int blocks = (int)std::min<long>(3 * ((witems + nw - 1) / nw), std::max<long>(wg_cap, 3));  // 85 / 170 workgroups per plane
#define JT_WALK_ONE(CD_, DET_, LL_, NW_)                                                                                    \
  do {                                                                                                                      \
    static bool attr = false;                                                                                               \
    if (!attr) {                                                                                                            \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_march_bwd_walk<CD_, DET_, LL_, NW_>),                       \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);  /* + 64 B static */               \
      attr = true;                                                                                                          \
    }                                                                                                                       \
    hipLaunchKernelGGL((k_march_bwd_walk<CD_, DET_, LL_, NW_>), dim3(blocks), dim3(NW_ * 64), wlds, st, D, GF, rays_o,      \
                       rays_d, jitter, zvals, tmin, n_rays, gfeat, vlist, nvalid, runs, g_rays_o, g_rays_d, rays_fixed,     \
                       bad, lline * line_floats, prefix);                                                               \
  } while (0)
#define JT_WALK_NW(CD_, DET_, LL_)              \
  do {                                          \
    if (nw == 8) JT_WALK_ONE(CD_, DET_, LL_, 8); \
    else JT_WALK_ONE(CD_, DET_, LL_, 16);       \
  } while (0)
#define JT_WALK(CD_)                               \
  do {                                             \
    if (det) JT_WALK_NW(CD_, true, 0);             \
    else if (lline == 2) JT_WALK_NW(CD_, false, 2); \
    else if (lline == 1) JT_WALK_NW(CD_, false, 1); \
    else JT_WALK_NW(CD_, false, 0);                \
  } while (0)
  if (D.Cd == 16) JT_WALK(16);
  else if (D.Cd == 8) JT_WALK(8);
  // (32 density components: no configuration of the reference's yamls has them, and two of that width's shapes spilled; the
  //  instantiations were removed in round 6 -- such a scene is JT_ERR_UNSUPPORTED here as it is in the shade kernels)
  else return JT_ERR_UNSUPPORTED;
#undef JT_WALK
#undef JT_WALK_NW
#undef JT_WALK_ONE
  JT_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_rays_fixed_add, dim3((n_rays * 6 + 255) / 256), dim3(256), 0, st, rays_fixed, n_rays, g_rays_o,
                     g_rays_d, bad, g_status_word.load(std::memory_order_relaxed));
  JT_LAUNCH_CHECK();
  return JT_OK;
}
extern "C" int jt_march_backward(const JtScene* scene, const JtFactors* factors, const float* rays_o,
                                 const float* rays_d, const float* jitter, const float* zvals, int n_rays,
                                 const float* sigma_feat, const float* weight, const float* tmin,
                                 const int32_t* shade_offset, const uint16_t* shade_idx, const float* rgb_s,
                                 const int32_t* clamp_mask, const float* g_rgb, const float* g_opacity,
                                 const float* g_xyz_app, const JtFactors* g_factors, float* g_rays_o,
                                 float* g_rays_d, void* workspace, size_t workspace_bytes, void* stream) {
  return march_backward(scene, factors, rays_o, rays_d, jitter, zvals, n_rays, sigma_feat, weight, tmin, shade_offset, shade_idx,
                        rgb_s, clamp_mask, g_rgb, g_opacity, g_xyz_app, g_factors, g_rays_o, g_rays_d, workspace, workspace_bytes,
                        nullptr, stream);
}

extern "C" int jt_march_backward_pose(const JtScene* scene, const JtFactors* factors, const float* rays_o,
                                      const float* rays_d, const float* jitter, const float* zvals, int n_rays,
                                      const float* sigma_feat, const float* weight, const float* tmin,
                                      const int32_t* shade_offset, const uint16_t* shade_idx, const float* rgb_s,
                                      const int32_t* clamp_mask, const float* g_rgb, const float* g_opacity,
                                      const float* g_xyz_app, const float* dfeat_dn, float* g_rays_o,
                                      float* g_rays_d, void* workspace, size_t workspace_bytes, void* stream) {
  if (!dfeat_dn) return JT_ERR_ARG;
  return march_backward(scene, factors, rays_o, rays_d, jitter, zvals, n_rays, sigma_feat, weight, tmin, shade_offset, shade_idx,
                        rgb_s, clamp_mask, g_rgb, g_opacity, g_xyz_app, nullptr, g_rays_o, g_rays_d, workspace, workspace_bytes,
                        dfeat_dn, stream);
}
