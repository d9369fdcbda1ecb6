// Device-side helpers shared by the gfx950 kernels of the TensoRF-VM renderer.
// Math follows the reference line by line where the result is DISCRETE (in-box test, sample
// positions): those expressions use non-contracted mul_rn/add_rn so that they round exactly
// like the torch elementwise ops they replace (tensorBase.py:572-612).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "jt_render.h"

namespace jt {

// matMode / vecMode of the reference (tensorBase.py:405-406)
__host__ __device__ constexpr int kM0(int i) { return i == 2 ? 1 : 0; }  // {0, 0, 1}
__host__ __device__ constexpr int kM1(int i) { return i == 0 ? 1 : 2; }  // {1, 2, 2}
__host__ __device__ constexpr int kV(int i) { return 2 - i; }            // {2, 1, 0}

struct Dev {
  float lo[3], hi[3], inv[3];  // inv = 2/(hi-lo)  (tensorBase.py:481)
  int ph[3], pw[3], ll[3];
  int Cd, Ca;
  float step, near_, far_, dist_scale, shift, thres;
  const float* near_dev;  // JtScene.near_plane_dev (depth term only)
  int act, S, ndc, white_bg;
  const float* dP[3];
  const float* dL[3];
  const float* aP[3];
  const float* aL[3];
  const float* mask;  // alpha-mask volume [z][y][x] or nullptr
  int md[3];
  float mlo[3], minv[3];
};

inline int make_dev(const JtScene* s, const JtFactors* f, Dev* d) {
  if (!s) return JT_ERR_ARG;
  for (int a = 0; a < 3; ++a) {
    d->lo[a] = s->aabb_lo[a];
    d->hi[a] = s->aabb_hi[a];
    d->inv[a] = 2.0f / (s->aabb_hi[a] - s->aabb_lo[a]);
    d->ph[a] = s->plane_h[a];
    d->pw[a] = s->plane_w[a];
    d->ll[a] = s->line_len[a];
    if (s->plane_h[a] < 1 || s->plane_w[a] < 1 || s->line_len[a] < 1) return JT_ERR_ARG;
    d->dP[a] = f ? f->density_plane[a] : nullptr;
    d->dL[a] = f ? f->density_line[a] : nullptr;
    d->aP[a] = f ? f->app_plane[a] : nullptr;
    d->aL[a] = f ? f->app_line[a] : nullptr;
  }
  d->Cd = s->n_comp_density;
  d->Ca = s->n_comp_app;
  d->step = s->step_size;
  d->near_ = s->near_plane;
  d->near_dev = s->near_plane_dev;
  d->far_ = s->far_plane;
  d->dist_scale = s->distance_scale;
  d->shift = s->density_shift;
  d->thres = s->weight_thres;
  d->act = s->density_act;
  d->S = s->n_samples;
  d->ndc = s->ndc;
  d->white_bg = s->white_bg;
  d->mask = f ? f->alpha_volume : nullptr;
  for (int a = 0; a < 3; ++a) {
    d->md[a] = s->mask_dims[a];
    d->mlo[a] = s->mask_lo[a];
    d->minv[a] = s->mask_inv[a];
    if (d->mask && d->md[a] < 1) return JT_ERR_ARG;
  }
  if (d->S < 1 || d->S > 65535) return JT_ERR_ARG;
  return JT_OK;
}

// JT_DETERMINISTIC mode (jt_set_deterministic / the environment variable, read once): every order-dependent float
// accumulation of the path is replaced by an order-independent one -- the factor / ray gradients of the scatters are
// summed as 64-bit fixed point (integer atomics commute), the small reductions run in one workgroup or one pass -- so
// that two runs from the same state produce bit-identical gradients.  A debugging aid (race detection: SURVEY section 5);
// slower, and the callers hand int64 shadow buffers where the header says so.
int jt_deterministic();                      // defined in jt_march.hip
// 2^48: 3.6e-15 resolution, +-32 768 range (a gradient element of this path is far below 1; round 2 used 2^56, whose
// +-128 wrapped silently).  An addend that is non-finite or out of range adds NOTHING and raises the library's sticky
// "bad fixed-point addend" flag instead (round 3 added a poison magnitude of 2^61 to the sum, which wrapped: eight poisoned
// addends are 0 mod 2^64); a sum that leaves the safe range through in-range addends is caught by its magnitude (>= 2^60,
// value 4 096).  Whoever converts a sum back (k_rays_fixed_add, ops.py's deterministic path) writes NaN / sets the
// FINITE_GRAD bit of the bound status word when either is seen, so neither can pass as a wrong-but-plausible gradient.
constexpr double kFixedScale = 281474976710656.0;
constexpr long long kFixedLimit = 1ll << 60;
// device address of the sticky flag (one word per process, owned by jt_march.hip; cleared by jt_status_clear)
unsigned* fixed_bad_flag();

// Chip geometry of the CURRENT device, queried once per device (hipDeviceGetAttribute: compute units, XCDs).  The persistent
// kernels of this library run one workgroup per CU (or a fixed fraction of the CUs while another kernel has the rest): their
// grids come from here, not from MI355X's 256 / 8 written into the launchers (a partitioned -- CPX --
// or differently binned part would have run wrong-sized persistent grids).  wgs(n256): the count that is n256 of 256 on a full
// MI355X, scaled to this device's CUs and rounded down to a whole number per XCD (at least one per XCD).
struct Chip {
  int cus, xcds;
  int wgs(int n256) const {
    const int per = (int)((long)n256 * cus / 256) / (xcds > 0 ? xcds : 1);
    return (per > 0 ? per : 1) * (xcds > 0 ? xcds : 1);
  }
};
const Chip& chip();                          // defined in jt_march.hip

#define JT_LAUNCH_CHECK()                      \
  do {                                         \
    hipError_t e__ = hipGetLastError();        \
    if (e__ != hipSuccess) return -(int)e__;   \
  } while (0)

// ---- per-ray geometry ------------------------------------------------------------------------
struct Ray {
  float o[3], d[3];
  float tmin;  // AABB entry distance, clamped to [near, far]   (tensorBase.py:587-589)
  float u;     // per-ray jitter (0 when not training)
  float norm;  // |d| (NDC only, batBase.py:64)
};

__device__ inline float ray_tmin(const Dev& D, const float o[3], const float d[3]) {
  float t = -INFINITY;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float v = (d[a] == 0.f) ? 1e-6f : d[a];
    float ra = (D.hi[a] - o[a]) / v;
    float rb = (D.lo[a] - o[a]) / v;
    t = fmaxf(t, fminf(ra, rb));
  }
  return fminf(fmaxf(t, D.near_), D.far_);
}

// separately rounded multiply / add.  HIP's __fmul_rn/__fadd_rn are plain `*`/`+` and get contracted
// into an FMA under the default -ffp-contract=fast; the pragma keeps these two un-fused.
__device__ inline float mul_rn(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ inline float add_rn(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}

// z_i = t_min + stepSize * (i + u)     (tensorBase.py:591-596); NDC: z_i = zvals[i]
__device__ inline float sample_z(const Dev& D, const Ray& r, const float* zvals, int i) {
  if (D.ndc) return zvals[i];
  float rng = add_rn((float)i, r.u);
  return add_rn(r.tmin, mul_rn(D.step, rng));
}

// pts = o + d*z (un-fused, tensorBase.py:598) and the in-box test (tensorBase.py:610)
__device__ inline bool sample_point(const Dev& D, const Ray& r, float z, float p[3]) {
  bool inside = true;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    p[a] = add_rn(r.o[a], mul_rn(r.d[a], z));
    inside = inside && !(D.lo[a] > p[a]) && !(p[a] > D.hi[a]);
  }
  return inside;
}

// AlphaGridMask.sample_alpha(p) > 0 (tensorBase.py:92-99, grid_sample trilinear, align_corners=True, zero
// padding) for a 0/1 volume: true iff a corner that is inside the volume, set, and has a non-zero weight exists.
__device__ inline bool mask_keep(const Dev& D, const float p[3]) {
  int i0[3];
  float f[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float g = add_rn(mul_rn(add_rn(p[a], -D.mlo[a]), D.minv[a]), -1.f);  // (xyz - aabb[0]) * invgridSize - 1
    const float ix = ((g + 1.f) * 0.5f) * (float)(D.md[a] - 1);
    const float fl = floorf(ix);
    i0[a] = (int)fl;
    f[a] = ix - fl;
  }
  bool keep = false;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int dx = c & 1, dy = (c >> 1) & 1, dz = c >> 2;
    const int x = i0[0] + dx, y = i0[1] + dy, z = i0[2] + dz;
    const float w = (dx ? f[0] : 1.f - f[0]) * (dy ? f[1] : 1.f - f[1]) * (dz ? f[2] : 1.f - f[2]);
    if (x >= 0 && x < D.md[0] && y >= 0 && y < D.md[1] && z >= 0 && z < D.md[2] && w > 0.f)
      keep = keep || (D.mask[((size_t)z * D.md[1] + y) * D.md[0] + x] > 0.f);
  }
  return keep;
}

// sample_point + the alpha mask: the validity of a sample as BatBase.forward sees it (batBase.py:76-82)
__device__ inline bool sample_valid(const Dev& D, const Ray& r, float z, float p[3]) {
  bool v = sample_point(D, r, z, p);
  if (v && D.mask) v = mask_keep(D, p);
  return v;
}

// normalize_coord (tensorBase.py:502-503): (xyz - aabb[0]) * invaabbSize - 1 as three separately rounded torch ops.
// Un-fused on purpose: the texel cell of a sample is floor() of this value scaled by (size - 1) / 2, and a sample within
// one ulp of a cell border would otherwise land in the neighbouring cell -- same interpolated value, but a different
// slope, i.e. a different coordinate (pose) gradient for that sample (measured at 400^3: a few rays per batch).
__device__ inline void normalize(const Dev& D, const float p[3], float n[3]) {
#pragma unroll
  for (int a = 0; a < 3; ++a) n[a] = add_rn(mul_rn(add_rn(p[a], -D.lo[a]), D.inv[a]), -1.f);
}

// ---- bilinear / linear taps: grid_sample(bilinear, align_corners=True, zeros) ------------------
struct Axis {
  int i0;        // floor cell (clamped copy in c0/c1 for addressing)
  int c0, c1;    // clamped indices
  float w0, w1;  // tap weights, zeroed when the tap is out of range
  float f;       // fractional part
  float m0, m1;  // 1/0 in-range masks
  float scale;   // d(ix)/d(normalised coord) = (size-1)/2
};

__device__ inline Axis axis_taps(float g, int size) {
  Axis a;
  float ix = ((g + 1.f) * 0.5f) * (float)(size - 1);
  float fl = floorf(ix);
  a.f = ix - fl;
  a.i0 = (int)fl;
  bool in0 = (a.i0 >= 0) && (a.i0 < size);
  bool in1 = (a.i0 + 1 >= 0) && (a.i0 + 1 < size);
  a.m0 = in0 ? 1.f : 0.f;
  a.m1 = in1 ? 1.f : 0.f;
  a.c0 = min(max(a.i0, 0), size - 1);
  a.c1 = min(max(a.i0 + 1, 0), size - 1);
  a.w0 = in0 ? (1.f - a.f) : 0.f;
  a.w1 = in1 ? a.f : 0.f;
  a.scale = 0.5f * (float)(size - 1);
  return a;
}

struct PlaneTaps {
  int o00, o10, o01, o11;  // element offsets (texel * C) of (x0,y0) (x1,y0) (x0,y1) (x1,y1)
  float w00, w10, w01, w11;
  Axis ax, ay;
};

__device__ inline PlaneTaps plane_taps(float gx, float gy, int H, int W, int C) {
  PlaneTaps t;
  t.ax = axis_taps(gx, W);
  t.ay = axis_taps(gy, H);
  t.o00 = (t.ay.c0 * W + t.ax.c0) * C;
  t.o10 = (t.ay.c0 * W + t.ax.c1) * C;
  t.o01 = (t.ay.c1 * W + t.ax.c0) * C;
  t.o11 = (t.ay.c1 * W + t.ax.c1) * C;
  t.w00 = t.ax.w0 * t.ay.w0;
  t.w10 = t.ax.w1 * t.ay.w0;
  t.w01 = t.ax.w0 * t.ay.w1;
  t.w11 = t.ax.w1 * t.ay.w1;
  return t;
}

// density activation (tensorBase.py:696-700; F.softplus beta=1 threshold=20)
__device__ inline float density_act(int act, float x) {
  if (act == JT_ACT_RELU) return fmaxf(x, 0.f);
  return (x > 20.f) ? x : log1pf(expf(x));
}
__device__ inline float density_act_grad(int act, float x) {
  if (act == JT_ACT_RELU) return (x > 0.f) ? 1.f : 0.f;
  return (x > 20.f) ? 1.f : 1.f / (1.f + expf(-x));
}

__device__ inline float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
// 16-byte load at (uniform base) + (32-bit byte offset) + 16 * q: the form the hardware addresses directly (scalar base
// register pair, 32-bit lane offset, immediate) -- no 64-bit address arithmetic per load
__device__ inline float4 ld4q(const float* base, unsigned byte_off, int q) {
  return reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + byte_off)[q];
}

// XCD-aware work placement (cdna_hip_programming.md T1): consecutive workgroup ids are dealt round-robin to the eight XCDs, each
// with its own 4 MB L2, so "blockIdx % 8" labels the workgroups that share an L2.  xcd_swizzle gives every label a CONTIGUOUS
// range of the n work items (bijective for any n): neighbouring rays / tiles -- which gather neighbouring texels -- then meet
// in one L2 instead of being spread over all eight.  A speed choice only: any placement computes the same thing.
#ifndef JT_XCD_SWIZZLE
#define JT_XCD_SWIZZLE 1
#endif
__device__ inline int xcd_swizzle(int id, int n) {
#if JT_XCD_SWIZZLE
  const int q = n >> 3, r = n & 7, x = id & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);
#else
  return id;
#endif
}
// the same for a persistent kernel: this workgroup's label, its rank among the workgroups of that label, how many of them there
// are, and the [lo, hi) range of the n items the label owns
struct XcdShare {
  int lo, hi, rank, peers;
};
// `blocks`: the workgroups that take part (ids 0 .. blocks - 1; a kernel whose grid was sized for a worst case lets the rest
// return before they load anything)
__device__ inline XcdShare xcd_share(int n, int blocks) {
  XcdShare s;
#if JT_XCD_SWIZZLE
  const int g = min(blocks, 8), x = (int)blockIdx.x % g;
  const int q = n / g, r = n - q * g;
  s.lo = x * q + min(x, r);
  s.hi = s.lo + q + (x < r ? 1 : 0);
  s.rank = (int)blockIdx.x / g;
  s.peers = (blocks - x + g - 1) / g;
#else
  s.lo = 0, s.hi = n, s.rank = blockIdx.x, s.peers = blocks;
#endif
  return s;
}

// order-independent accumulation: v as 2^48 fixed point into a 64-bit word
__device__ inline void fixed_add(long long* p, float v, unsigned* bad) {
  if (fabsf(v) < 8192.f)  // NaN fails the test
    atomicAdd(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double2ll_rn((double)v * kFixedScale));
  else
    atomicOr(bad, 1u);
}

__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__device__ inline void load_ray(const Dev& D, const float* rays_o, const float* rays_d, const float* jitter,
                                const float* tmin_in, int ray, Ray& r) {
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    r.o[a] = rays_o[ray * 3 + a];
    r.d[a] = rays_d[ray * 3 + a];
  }
  r.u = (jitter && !D.ndc) ? jitter[ray] : 0.f;
  r.norm = 1.f;
  if (D.ndc) {
    r.norm = sqrtf(r.d[0] * r.d[0] + r.d[1] * r.d[1] + r.d[2] * r.d[2]);
    r.tmin = 0.f;
  } else {
    r.tmin = tmin_in ? tmin_in[ray] : ray_tmin(D, r.o, r.d);
  }
}

}  // namespace jt
