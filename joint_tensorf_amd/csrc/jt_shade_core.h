// Device code of the fused appearance chain shared by the kernels of jt_shade.hip (training / inference forward, backward,
// weight gradients) and jt_fused.hip (single-launch forward + pose-only backward): shapes, the LDS weight image, the
// transposed fp32-MFMA layers, positional encoding, the record layout.  See jt_shade.hip's header comment for the tiling.
#pragma once
#include "jt_common.h"
#include "jt_walk.h"

namespace jt {

// Record traffic is streaming (1.2 GB written per launch, read once by the backward a millisecond later, far more
// than L2 + Infinity Cache hold): non-temporal accesses keep it from evicting the factor set the gathers live on.
#ifndef JT_REC_NT
#define JT_REC_NT 1
#endif
__device__ inline void rec_st(float* p, float v) {
#if JT_ABL_NOST
  return;
#endif
#if JT_REC_NT
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}
__device__ inline float rec_ld(const float* p) {
#if JT_REC_NT
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}
// record slot of tile block `rt` (uniform: it depends on the wave, not on the lane), row `row` (a constant at every use),
// lane part `lane_bytes` (4 * sample [+ 512 * lane half for rows rowmap(r, 0) + 4 h]).
// Address of (record row, lane part) of a tile's record block.  The block base is wave-uniform: pinning the 4 KB window of
// the row into a scalar register pair makes every access (scalar base) + (32-bit lane offset) + immediate, the form the
// hardware addresses by itself -- no 64-bit vector address arithmetic per 4 KB of rows
typedef __attribute__((address_space(1))) char rec_gchar;   // (the pin must not cost the pointer its address space)
__device__ inline char* rec_window(const float* rt, int row) {
  rec_gchar* win = (rec_gchar*)(const_cast<char*>(reinterpret_cast<const char*>(rt)) + (((size_t)row * 128) & ~(size_t)4095));
  asm("" : "+s"(win));
  return (char*)win;
}
__device__ inline float* rec_at(float* rt, int row, unsigned lane_bytes) {
  return reinterpret_cast<float*>(rec_window(rt, row) + lane_bytes + (((unsigned)row * 128u) & 4095u));
}
__device__ inline const float* rec_at(const float* rt, int row, unsigned lane_bytes) {
  return reinterpret_cast<const float*>(rec_window(rt, row) + lane_bytes + (((unsigned)row * 128u) & 4095u));
}
__device__ inline float4 rec_ld4(const float* p) {
#if JT_REC_NT
  typedef float v4f __attribute__((ext_vector_type(4)));
  const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
#else
  return ld4(p);
#endif
}


typedef float f32x16 __attribute__((ext_vector_type(16)));

__host__ __device__ constexpr int rowmap(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

struct MlpDev {
  const float* basis;
  const float* w1;
  const float* b1;
  const float* w2;
  const float* b2;
  const float* w3;
  const float* b3;
};

template <int CA_, int APP_, int HID_, int KIND_>
struct ShadeCfg {
  static constexpr int CA = CA_, APP = APP_, HID = HID_, KIND = KIND_;
  static constexpr int NC = 3 * CA;                                  // basis_mat input width
  static constexpr int LDB = NC | 1;                                 // odd LDS row stride
  static constexpr int IN1 = (KIND == JT_MLP_FEA) ? APP * 5 + 15 : APP * 5;  // 150 / 100
  static constexpr int LD1 = (IN1 + 1) | 1;                          // >= IN1+1: column IN1 is a zero pad
  static constexpr int MT = HID / 32;                                // M tiles of the hidden layers
  static constexpr int LD2 = HID | 1;
  static constexpr int IN3 = (KIND == JT_MLP_FEA) ? HID : HID + 12;  // 64 / 44
  static constexpr int NSLOT = (CA + 7) / 8;                         // channel-quad slots per half per plane
  // LDS carve (floats)
  static constexpr int O_BASIS = 0;
  static constexpr int O_W1 = O_BASIS + 32 * LDB;
  static constexpr int O_W2 = O_W1 + HID * LD1;
  static constexpr int O_W3 = O_W2 + HID * LD2;   // stored [k][4] (c = 0..2, pad)
  static constexpr int O_B1 = O_W3 + IN3 * 4;
  static constexpr int O_B2 = O_B1 + HID;
  static constexpr int O_B3 = O_B2 + HID;
  static constexpr int LDS_FLOATS = O_B3 + 4;
  static_assert(HID % 32 == 0 && APP <= 32 && CA % 4 == 0 && CA <= 64, "shape");
};

// Column of W1 (torch layout [HID][IN1]) that lane-half h consumes in k-step (r, t), t = 0..4:
//   t = 0: the raw feature, t = 1..4: its positional encoding [sin x, sin 2x, cos x, cos 2x]
// (tensorBase.py:43-55: per channel [sin 2^0, sin 2^1, cos 2^0, cos 2^1]).  Feature a = rowmap(r,h) of
// basis_mat's output lives at column a, its encoding at APP+3 + 4a (MLP_Fea: [f, d, PE(f), PE(d)],
// tensorBase.py:117-122) or APP + 4a (WeakView: [f, PE(f)], :199-205).  For MLP_Fea the three view
// direction components ride in the slots of half 1 whose feature row is padding (rows 28..30).
template <class C>
__host__ __device__ constexpr int w1_col(int h, int r, int t) {
  const int a = rowmap(r, h);
  if (a < C::APP) {
    if (C::KIND == JT_MLP_FEA) return t == 0 ? a : C::APP + 3 + 4 * a + (t - 1);
    return t == 0 ? a : C::APP + 4 * a + (t - 1);
  }
  if (C::KIND == JT_MLP_FEA && h == 1 && a >= 28 && a <= 30) {
    const int v = a - 28;
    return t == 0 ? C::APP + v : C::APP + 3 + 4 * C::APP + 4 * v + (t - 1);
  }
  return C::IN1;  // zero pad column
}

// number of r-steps of layer 1 that carry at least one live half
template <class C>
__host__ __device__ constexpr int l1_rsteps() {
  int n = 0;
  for (int r = 0; r < 16; ++r)
    if (w1_col<C>(0, r, 0) != C::IN1 || w1_col<C>(1, r, 0) != C::IN1) n = r + 1;
  return n;
}

// The weight image of a workgroup (75 KB for MLP_Fea): one wave per matrix row, lanes along the row -- coalesced 256-byte
// reads, no index division, every load independent of the others (the first version walked a flat index with a division
// and a modulo per element: ~60 us per launch, which is most of k_shade_fwd / k_shade_bwd on a trained, sparsely shaded
// scene where a launch has one tile per wave).
template <class C>
__device__ inline void load_weights_lds(float* s, const MlpDev& M) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int a = wv; a < 32; a += nw) {            // basis [APP][NC] -> [32][LDB], rows >= APP and the pad column zero
    const bool row = a < C::APP;
    for (int c = lane; c < C::LDB; c += 64) s[C::O_BASIS + a * C::LDB + c] = (row && c < C::NC) ? M.basis[a * C::NC + c] : 0.f;
  }
  for (int u = wv; u < C::HID; u += nw) {        // W1 [HID][IN1] -> [HID][LD1]; W2 [HID][HID] -> [HID][LD2]
    for (int c = lane; c < C::LD1; c += 64) s[C::O_W1 + u * C::LD1 + c] = (c < C::IN1) ? M.w1[u * C::IN1 + c] : 0.f;
    for (int c = lane; c < C::LD2; c += 64) s[C::O_W2 + u * C::LD2 + c] = (c < C::HID) ? M.w2[u * C::HID + c] : 0.f;
  }
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int i = tid; i < C::IN3 * 4; i += nt) {   // W3 [3][IN3] -> [IN3][4] (c = 0..2, pad)
    const int k = i >> 2, c = i & 3;
    s[C::O_W3 + i] = (c < 3) ? M.w3[c * C::IN3 + k] : 0.f;
  }
  for (int i = tid; i < C::HID; i += nt) {
    s[C::O_B1 + i] = M.b1[i];
    s[C::O_B2 + i] = M.b2[i];
  }
  if (tid < 4) s[C::O_B3 + tid] = (tid < 3) ? M.b3[tid] : 0.f;
}

struct PeMask {
  float f0, f1, v0, v1;  // (progress*freqs - level) clamped to [0,1], freqs = 2 (tensorBase.py:48)
};

// Position of entry e: normalised coordinates + taps are recomputed from the ray, exactly as the
// march kernel did (same expressions => same sample).
struct EntryGeom {
  float n[3];
  float z;
  int ray;
};

__device__ inline EntryGeom entry_geom(const Dev& D, const float* rays_o, const float* rays_d, const float* jitter,
                                       const float* zvals, const float* tmin, const int* eray, const int* esmp,
                                       int e) {
  EntryGeom g;
  g.ray = eray[e];
  Ray r;
  load_ray(D, rays_o, rays_d, jitter, tmin, g.ray, r);
  g.z = sample_z(D, r, zvals, esmp[e]);
  float p[3];
  sample_point(D, r, g.z, p);
  normalize(D, p, g.n);
  return g;
}

template <class C>
struct BwdCfg;

// ---- stage 1: gather + products + basis_mat  -> feature accumulator (16 regs: rows rowmap(r,h)) -----
template <class C, bool REC>
__device__ inline f32x16 gather_basis(const Dev& D, const float* s, const float n[3], int j, int h, float* rt,
                                      bool onrec) {
  f32x16 facc;
#pragma unroll
  for (int r = 0; r < 16; ++r) facc[r] = 0.f;
  const float* sb = s + C::O_BASIS + j * C::LDB;  // this lane's basis row (A operand: unit a = j)
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    PlaneTaps t = plane_taps(n[kM0(i)], n[kM1(i)], D.ph[i], D.pw[i], C::CA);
    Axis l = axis_taps(n[kV(i)], D.ll[i]);
    const float* P = D.aP[i];
    const float* L = D.aL[i];
    // byte offsets of the six taps for this lane half's first channel quad; quad slot m is 32 bytes further
    const unsigned hb = 16u * (unsigned)h;
    const unsigned b00 = 4u * (unsigned)t.o00 + hb, b10 = 4u * (unsigned)t.o10 + hb, b01 = 4u * (unsigned)t.o01 + hb,
                   b11 = 4u * (unsigned)t.o11 + hb, bl0 = 4u * (unsigned)(l.c0 * C::CA) + hb,
                   bl1 = 4u * (unsigned)(l.c1 * C::CA) + hb;
#pragma unroll
    for (int m = 0; m < C::NSLOT; ++m) {
      const int q = 2 * m + h;
      const bool live = q * 4 < C::CA;
      const int c0 = live ? q * 4 : 0;
      float4 a, b, c, d, u, v;
      if (C::CA % 8 == 0 || m + 1 < C::NSLOT) {  // both halves of the slot exist
        a = ld4q(P, b00, 2 * m), b = ld4q(P, b10, 2 * m), c = ld4q(P, b01, 2 * m), d = ld4q(P, b11, 2 * m);
        u = ld4q(L, bl0, 2 * m), v = ld4q(L, bl1, 2 * m);
      } else {
        // last slot of an odd quad count (VM-20: five quads): half 1 has no quad there and re-reads half 0's (its
        // products are zeroed below) instead of reading past the texel
        a = ld4q(P, b00 - hb, 2 * m), b = ld4q(P, b10 - hb, 2 * m), c = ld4q(P, b01 - hb, 2 * m);
        d = ld4q(P, b11 - hb, 2 * m), u = ld4q(L, bl0 - hb, 2 * m), v = ld4q(L, bl1 - hb, 2 * m);
      }
      float pr[4];
      pr[0] = (t.w00 * a.x + t.w10 * b.x + t.w01 * c.x + t.w11 * d.x) * (l.w0 * u.x + l.w1 * v.x);
      pr[1] = (t.w00 * a.y + t.w10 * b.y + t.w01 * c.y + t.w11 * d.y) * (l.w0 * u.y + l.w1 * v.y);
      pr[2] = (t.w00 * a.z + t.w10 * b.z + t.w01 * c.z + t.w11 * d.z) * (l.w0 * u.z + l.w1 * v.z);
      pr[3] = (t.w00 * a.w + t.w10 * b.w + t.w01 * c.w + t.w11 * d.w) * (l.w0 * u.w + l.w1 * v.w);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float bv = live ? pr[k] : 0.f;
        // row = R_PROD + i CA + 8 m + 4 h + k: the lane half rides in the lane part
        if (REC && live && onrec)
          rec_st(rec_at(rt, BwdCfg<C>::R_PROD + i * C::CA + 8 * m + k, 4u * (unsigned)j + 512u * (unsigned)h), pr[k]);
        float av = sb[i * C::CA + c0 + k];
        facc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, facc, 0, 0, 0);
      }
      // at most two quad slots of taps in flight (3, 6 or all 18 measured the same 0.734 ms: the forward is not waiting for
      // its gathers)
      if (m & 1) __builtin_amdgcn_sched_barrier(0);
    }
  }
  return facc;
}

// sin and cos of x in one go: 3-term Cody-Waite reduction by pi/2 and the cephes single-precision minimax
// kernels on [-pi/4, pi/4] (abs error ~1e-7 for |x| < 1e4; the MLP inputs are O(1) features).
__device__ inline void sincos_f(float x, float* sn, float* cs) {
  const float k = rintf(x * 0.636619772367581343f);  // x * 2/pi
  const int q = (int)k;
  float y = fmaf(k, -1.5703125f, x);
  y = fmaf(k, -4.837512969970703125e-4f, y);
  y = fmaf(k, -7.54978995489188216e-8f, y);
  const float z = y * y;
  float s = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, y, y);
  float c = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f) * z, z,
                 fmaf(-0.5f, z, 1.f));
  const float ss = (q & 1) ? c : s;
  const float cc = (q & 1) ? s : c;
  *sn = (q & 2) ? -ss : ss;
  *cs = ((q + 1) & 2) ? -cc : cc;
}

// Hardware sine / cosine (v_sin_f32 / v_cos_f32: two quarter-rate instructions and a multiply instead of ~35): absolute
// error 1.4e-7 for |x| <= 1, 4e-7 for |x| <= 4, 1.4e-6 for |x| <= 16 (measured on MI355X against double precision).
// Used where the encoding enters a GRADIENT only -- the derivative factors of the backward chain and the layer-1 input
// of the weight-gradient GEMM -- never for the forward's values (sincos_f).
__device__ inline void sincos_grad(float x, float* sn, float* cs) {
  // the instructions take revolutions and are defined on [-256, 256] only: the fraction keeps any finite x in range
  const float rev = __builtin_amdgcn_fractf(x * 0.15915494309189535f);
  *sn = __builtin_amdgcn_sinf(rev);
  *cs = __builtin_amdgcn_cosf(rev);
}

// the five values (x, PE(x)) lane-half h feeds into layer 1 for accumulator row r:
//   [x, sin x * m0, sin 2x * m1, cos x * m0, cos 2x * m1]     (tensorBase.py:43-55)
template <class C>
__device__ inline void l1_inputs(const f32x16& facc, const float vd[3], const PeMask& pm, int h, int r,
                                 float out[5], float* sn_out, float* cs_out) {
  const int a0 = rowmap(r, 0), a1 = rowmap(r, 1);
  const bool feat0 = a0 < C::APP, feat1 = a1 < C::APP;
  const bool dir1 = (C::KIND == JT_MLP_FEA) && a1 >= 28 && a1 <= 30;
  float x = facc[r], m0 = pm.f0, m1 = pm.f1;
  bool live = h ? (feat1 || dir1) : feat0;
  if (dir1 && h) {
    x = vd[a1 >= 28 ? a1 - 28 : 0];
    m0 = pm.v0;
    m1 = pm.v1;
  }
  float sn, cs;
  sincos_f(x, &sn, &cs);
  *sn_out = sn;
  *cs_out = cs;
  out[0] = live ? x : 0.f;
  out[1] = live ? sn * m0 : 0.f;
  out[2] = live ? 2.f * sn * cs * m1 : 0.f;
  out[3] = live ? cs * m0 : 0.f;
  out[4] = live ? (1.f - 2.f * sn * sn) * m1 : 0.f;
}

template <class C>
struct Hidden {
  f32x16 v[C::MT];
};

// ---- layer 1: IN1 -> HID, + bias, ReLU -----------------------------------------------------------------
template <class C>
__device__ inline Hidden<C> layer1(const float* s, const f32x16& facc, const float vd[3], const PeMask& pm, int j,
                                   int h, float* stash = nullptr, int lane = 0) {
  Hidden<C> acc;
#pragma unroll
  for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc.v[mt][r] = s[C::O_B1 + mt * 32 + rowmap(r, h)];
  constexpr int RS = l1_rsteps<C>();
#pragma unroll
  for (int r = 0; r < RS; ++r) {
    float in[5], sn, cs;
    l1_inputs<C>(facc, vd, pm, h, r, in, &sn, &cs);
    if (stash) {  // lane-private slots, conflict-free
      stash[(2 * r) * 64 + lane] = sn;
      stash[(2 * r + 1) * 64 + lane] = cs;
    }
#pragma unroll
    for (int t = 0; t < 5; ++t) {
      const int col = h ? w1_col<C>(1, r, t) : w1_col<C>(0, r, t);
#pragma unroll
      for (int mt = 0; mt < C::MT; ++mt) {
        float av = s[C::O_W1 + (mt * 32 + j) * C::LD1 + col];
        acc.v[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, in[t], acc.v[mt], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);  // keep the encodings of later rows from being hoisted (VGPR pressure)
  }
  return acc;
}

// ---- layer 2: HID -> HID, + bias (ReLU applied by the caller on input and output) ------------------------
template <class C>
__device__ inline Hidden<C> layer2(const float* s, const Hidden<C>& h1, int j, int h) {
  Hidden<C> acc;
#pragma unroll
  for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc.v[mt][r] = s[C::O_B2 + mt * 32 + rowmap(r, h)];
#pragma unroll
  for (int mk = 0; mk < C::MT; ++mk) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int col = mk * 32 + rowmap(r, 0) + 4 * h;
      const float bv = h1.v[mk][r];
#pragma unroll
      for (int mt = 0; mt < C::MT; ++mt) {
        float av = s[C::O_W2 + (mt * 32 + j) * C::LD2 + col];
        acc.v[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc.v[mt], 0, 0, 0);
      }
    }
  }
  return acc;
}

template <class C>
__device__ inline void relu_(Hidden<C>& x) {
#pragma unroll
  for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) x.v[mt][r] = fmaxf(x.v[mt][r], 0.f);
}

// view-direction encoding used by the WeakView last layer: [sin d (3x2) , cos d (3x2)] per channel
__device__ inline void view_pe(const float vd[3], const PeMask& pm, float out[12]) {
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float sn, cs;
    sincos_f(vd[a], &sn, &cs);
    out[4 * a + 0] = sn * pm.v0;
    out[4 * a + 1] = 2.f * sn * cs * pm.v1;
    out[4 * a + 2] = cs * pm.v0;
    out[4 * a + 3] = (1.f - 2.f * sn * sn) * pm.v1;
  }
}

// ---- layer 3 (3 outputs): VALU dot over the lane's own units, halves combined with one swap -------------
template <class C>
__device__ inline void layer3(const float* s, const Hidden<C>& h2, const float vd[3], const PeMask& pm, int h,
                              float out[3]) {
  float o0 = 0.f, o1 = 0.f, o2 = 0.f;
  constexpr int HOFF = (C::KIND == JT_MLP_FEA) ? 0 : 12;  // WeakView: [PE(d) (12), h (HID)]
#pragma unroll
  for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = HOFF + mt * 32 + rowmap(r, 0) + 4 * h;
      const float4 w = *reinterpret_cast<const float4*>(s + C::O_W3 + k * 4);
      const float x = h2.v[mt][r];
      o0 += x * w.x;
      o1 += x * w.y;
      o2 += x * w.z;
    }
  if (C::KIND != JT_MLP_FEA) {
    float pe[12];
    view_pe(vd, pm, pe);
    if (h == 0) {
#pragma unroll
      for (int k = 0; k < 12; ++k) {
        const float4 w = *reinterpret_cast<const float4*>(s + C::O_W3 + k * 4);
        o0 += pe[k] * w.x;
        o1 += pe[k] * w.y;
        o2 += pe[k] * w.z;
      }
    }
  }
  o0 += __shfl_xor(o0, 32);
  o1 += __shfl_xor(o1, 32);
  o2 += __shfl_xor(o2, 32);
  out[0] = o0 + s[C::O_B3 + 0];
  out[1] = o1 + s[C::O_B3 + 1];
  out[2] = o2 + s[C::O_B3 + 2];
}

static inline PeMask pe_masks(float fea_progress, float view_progress, int fea_pe, int view_pe) {
  PeMask pm;
  pm.f0 = fminf(fmaxf(fea_progress * fea_pe - 0.f, 0.f), 1.f);
  pm.f1 = fminf(fmaxf(fea_progress * fea_pe - 1.f, 0.f), 1.f);
  pm.v0 = fminf(fmaxf(view_progress * view_pe - 0.f, 0.f), 1.f);
  pm.v1 = fminf(fmaxf(view_progress * view_pe - 1.f, 0.f), 1.f);
  return pm;
}

// =================================================================================================
// Split-bf16 forward chain ("bf16x3").  gfx950 runs fp32-input MFMA at 64 FLOP / clk / SIMD and bf16-input MFMA at 1024:
// an fp32 product sum is emulated at 2.7 x the fp32 matrix rate by splitting BOTH operands into three bf16 pieces
// (x = p0 + p1 + p2 exactly to 2^-27 |x|: bf16 carries 8 significant bits) and issuing six v_mfma_f32_32x32x16_bf16 per
// 16-deep K step -- a0 b0 + a0 b1 + a1 b0 + a1 b1 + a0 b2 + a2 b0, fp32 accumulation; the dropped terms are below 2^-27.
// Measured against double precision (tools/mfma_bf16_probe.hip, K = 160): max error 9.1e-6 on sums of magnitude 33, against
// 1.2e-5 for an fp32 FMA chain -- fp32-level products, not a reduced-precision mode.
// Layout: the 32x32x16 instruction takes 8 consecutive-k values per lane (lane half = k block) and leaves D in the SAME
// register layout as the fp32 32x32x2 form (row = rowmap(r, half)), so the transposed chain keeps its property: the
// accumulator of a layer is, after ReLU / encoding, the B operand of the next one, in registers.  Which logical k the 8
// values of a (step, half) stand for is free as long as A agrees: the weights are pre-split into an LDS image
// [stage][step][M tile][piece][lane] of 16-byte vectors that a lane reads with one ds_read_b128 per piece.
// =================================================================================================
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

struct B3 {
  bf8 p0, p1, p2;
};

// x = p0 + p1 + p2 to 2^-26 |x|: each piece a bf16 (round to nearest even: the remainders are signed and at most 2^-9 and
// 2^-18 of x, which is what makes the three dropped cross terms of mfma6 smaller than an fp32 rounding).  Two values at a
// time: one packed conversion per piece, a shift and a mask to get the pieces back as fp32, one packed subtraction (nine
// VALU instructions per pair).
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));

__device__ inline unsigned pack_bf16(f32x2 x) { return __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf2)); }
__device__ inline f32x2 unpack_bf16(unsigned w) {
  f32x2 o;
  o.x = __uint_as_float(w << 16);
  o.y = __uint_as_float(w & 0xffff0000u);
  return o;
}

__device__ inline B3 split8(const float v[8]) {
  unsigned w0[4], w1[4], w2[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    f32x2 x;
    x.x = v[2 * e], x.y = v[2 * e + 1];
    w0[e] = pack_bf16(x);
    const f32x2 r1 = x - unpack_bf16(w0[e]);
    w1[e] = pack_bf16(r1);
    const f32x2 r2 = r1 - unpack_bf16(w1[e]);
    w2[e] = pack_bf16(r2);
  }
  B3 o;
  o.p0 = __builtin_bit_cast(bf8, make_uint4(w0[0], w0[1], w0[2], w0[3]));
  o.p1 = __builtin_bit_cast(bf8, make_uint4(w1[0], w1[1], w1[2], w1[3]));
  o.p2 = __builtin_bit_cast(bf8, make_uint4(w2[0], w2[1], w2[2], w2[3]));
  return o;
}

// acc += A B over one 16-deep K step, fp32-level: six bf16 MFMAs, small terms first
__device__ inline f32x16 mfma6(const B3& a, const B3& b, f32x16 acc) {
#if JT_ABL_NOMFMA
  asm volatile("" ::"v"(a.p0), "v"(a.p1), "v"(a.p2), "v"(b.p0), "v"(b.p1), "v"(b.p2));
  return acc;
#endif
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p2, b.p0, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p0, b.p2, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p1, b.p1, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p1, b.p0, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p0, b.p1, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p0, b.p0, acc, 0, 0, 0);
  return acc;
}

template <class C>
struct B16Cfg {
  static constexpr int SPP = (C::NSLOT + 1) / 2;          // K steps per plane of the basis product (two quad slots each)
  static constexpr int NSB = 3 * SPP;
  static constexpr int RS = l1_rsteps<C>();
  static constexpr int NS1 = (RS * 5 + 7) / 8;            // layer 1: five values per accumulator row and half
  static constexpr int NS2 = 2 * C::MT;                   // layer 2: 16 MT values per half
  // LDS carve in 16-byte vectors: [step][M tile][piece][64 lanes]
  static constexpr int V_BASIS = 0;
  static constexpr int V_W1 = V_BASIS + NSB * 3 * 64;
  static constexpr int V_W2 = V_W1 + NS1 * C::MT * 3 * 64;
  static constexpr int V_END = V_W2 + NS2 * C::MT * 3 * 64;
  // fp32 tail (floats, behind the vectors): W3 [IN3][4], b1, b2, b3
  static constexpr int F_W3 = 0;
  static constexpr int F_B1 = F_W3 + C::IN3 * 4;
  static constexpr int F_B2 = F_B1 + C::HID;
  static constexpr int F_B3 = F_B2 + C::HID;
  static constexpr int F_END = F_B3 + 4;
  static constexpr size_t LDS_BYTES = (size_t)V_END * 16 + (size_t)F_END * 4;
};

__device__ inline void store_b3(uint4* img, int vec0, int lane, const float w[8]) {
  const B3 p = split8(w);
  img[vec0 + lane] = __builtin_bit_cast(uint4, p.p0);
  img[vec0 + 64 + lane] = __builtin_bit_cast(uint4, p.p1);
  img[vec0 + 128 + lane] = __builtin_bit_cast(uint4, p.p2);
}

template <class C>
__device__ inline void load_weights_lds_b16(unsigned char* smem, const MlpDev& M) {
  typedef B16Cfg<C> Q;
  uint4* img = reinterpret_cast<uint4*>(smem);
  float* tail = reinterpret_cast<float*>(smem + (size_t)Q::V_END * 16);
  const int items = (Q::NSB + Q::NS1 * C::MT + Q::NS2 * C::MT) * 64;
  for (int it = threadIdx.x; it < items; it += blockDim.x) {
    const int lane = it & 63, blk = it >> 6, i = lane & 31, h = lane >> 5;
    float w[8];
    if (blk < Q::NSB) {                                   // basis: unit a = i, channels of two quad slots of one plane
      const int pl = blk / Q::SPP, sp = blk - pl * Q::SPP;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int m = 2 * sp + (e >> 2), c = (2 * m + h) * 4 + (e & 3);
        w[e] = (i < C::APP && m < C::NSLOT && c < C::CA) ? M.basis[i * C::NC + pl * C::CA + c] : 0.f;
      }
      store_b3(img, Q::V_BASIS + blk * 3 * 64, lane, w);
    } else if (blk < Q::NSB + Q::NS1 * C::MT) {           // layer 1: unit u, the (row, encoding slot) pairs 8 s .. 8 s + 7
      const int b = blk - Q::NSB, st = b / C::MT, mt = b - st * C::MT, u = mt * 32 + i;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int idx = 8 * st + e, r = idx / 5, t = idx - 5 * r;
        const int col = (r < Q::RS) ? (h ? w1_col<C>(1, r, t) : w1_col<C>(0, r, t)) : C::IN1;
        w[e] = (col < C::IN1) ? M.w1[u * C::IN1 + col] : 0.f;
      }
      store_b3(img, Q::V_W1 + b * 3 * 64, lane, w);
    } else {                                              // layer 2: unit u, hidden units of (tile mk, rows r) 8 s .. 8 s + 7
      const int b = blk - Q::NSB - Q::NS1 * C::MT, st = b / C::MT, mt = b - st * C::MT, u = mt * 32 + i;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int idx = 8 * st + e, mk = idx >> 4, r = idx & 15;
        w[e] = M.w2[u * C::HID + mk * 32 + rowmap(r, 0) + 4 * h];
      }
      store_b3(img, Q::V_W2 + b * 3 * 64, lane, w);
    }
  }
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int i = tid; i < C::IN3 * 4; i += nt) {
    const int k = i >> 2, c = i & 3;
    tail[Q::F_W3 + i] = (c < 3) ? M.w3[c * C::IN3 + k] : 0.f;
  }
  for (int i = tid; i < C::HID; i += nt) {
    tail[Q::F_B1 + i] = M.b1[i];
    tail[Q::F_B2 + i] = M.b2[i];
  }
  if (tid < 4) tail[Q::F_B3 + tid] = (tid < 3) ? M.b3[tid] : 0.f;
}

__device__ inline B3 load_b3(const uint4* img, int vec0, int lane) {
  B3 a;
  a.p0 = __builtin_bit_cast(bf8, img[vec0 + lane]);
  a.p1 = __builtin_bit_cast(bf8, img[vec0 + 64 + lane]);
  a.p2 = __builtin_bit_cast(bf8, img[vec0 + 128 + lane]);
  return a;
}

template <class C>
struct BwdCfg;

// gather + products + basis_mat on the bf16 matrix cores (gather_basis's twin; the records, if any, get the same fp32
// products).  The taps of the NEXT pair of quad slots are in flight while the current pair is multiplied and split: the
// nine (VM-48) K steps of the basis product form one software pipeline across the three planes.
#ifndef JT_B16_PIPE
#define JT_B16_PIPE 1
#endif
#ifndef JT_B16_PIN_TAPS
#define JT_B16_PIN_TAPS 0
#endif
#ifndef JT_B16_PF_INFER
#define JT_B16_PF_INFER 1  // (2 pairs ahead measured the same 174 ms per 800 x 800 image)
#endif
#ifndef JT_B16_THREADS
#define JT_B16_THREADS 512
#endif
struct TapGeo {
  unsigned b00, b10, b01, b11, bl0, bl1;   // byte offsets of the six taps (this lane half's first channel quad)
  float w00, w10, w01, w11, lw0, lw1;
};
struct TapSlot {
  float4 a, b, c, d, u, v;
};

template <class C, bool REC>
__device__ inline f32x16 gather_basis_b16(const Dev& D, const uint4* img, const float n[3], int j, int h, int lane,
                                          float* rt, bool onrec) {
  typedef B16Cfg<C> Q;
  f32x16 facc;
#pragma unroll
  for (int r = 0; r < 16; ++r) facc[r] = 0.f;
  const unsigned hb = 16u * (unsigned)h;
  TapGeo geo[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const PlaneTaps t = plane_taps(n[kM0(i)], n[kM1(i)], D.ph[i], D.pw[i], C::CA);
    const Axis l = axis_taps(n[kV(i)], D.ll[i]);
    geo[i].b00 = 4u * (unsigned)t.o00 + hb;
    geo[i].b10 = 4u * (unsigned)t.o10 + hb;
    geo[i].b01 = 4u * (unsigned)t.o01 + hb;
    geo[i].b11 = 4u * (unsigned)t.o11 + hb;
    geo[i].bl0 = 4u * (unsigned)(l.c0 * C::CA) + hb;
    geo[i].bl1 = 4u * (unsigned)(l.c1 * C::CA) + hb;
    geo[i].w00 = t.w00;
    geo[i].w10 = t.w10;
    geo[i].w01 = t.w01;
    geo[i].w11 = t.w11;
    geo[i].lw0 = l.w0;
    geo[i].lw1 = l.w1;
  }
  auto load_slot = [&](int i, int m, TapSlot& s) {
    const TapGeo& g = geo[i];
    const float* P = D.aP[i];
    const float* L = D.aL[i];
#if JT_ABL_NOLD
    s.a = s.b = s.c = s.d = make_float4(g.w00, g.w10, g.w01, g.w11), s.u = s.v = make_float4(g.lw0, g.lw1, g.w00, g.w11);
    return;
#endif
#if JT_ABL_SAMETEX  // profiling knob: every tap of every sample reads texel 0 (same instructions, all L1 hits)
    {
      const unsigned z = hb & 16u;
      s.a = ld4q(P, z, 2 * m), s.b = ld4q(P, z, 2 * m), s.c = ld4q(P, z, 2 * m), s.d = ld4q(P, z, 2 * m);
      s.u = ld4q(L, z, 2 * m), s.v = ld4q(L, z, 2 * m);
      asm volatile("" : "+v"(s.a.x), "+v"(s.b.x), "+v"(s.c.x), "+v"(s.d.x), "+v"(s.u.x), "+v"(s.v.x));
      return;
    }
#endif
    if (C::CA % 8 == 0 || m + 1 < C::NSLOT) {
      s.a = ld4q(P, g.b00, 2 * m), s.b = ld4q(P, g.b10, 2 * m), s.c = ld4q(P, g.b01, 2 * m), s.d = ld4q(P, g.b11, 2 * m);
      s.u = ld4q(L, g.bl0, 2 * m), s.v = ld4q(L, g.bl1, 2 * m);
    } else {  // last slot of an odd quad count (VM-20): half 1 has no quad there and re-reads half 0's (zeroed below)
      s.a = ld4q(P, g.b00 - hb, 2 * m), s.b = ld4q(P, g.b10 - hb, 2 * m), s.c = ld4q(P, g.b01 - hb, 2 * m);
      s.d = ld4q(P, g.b11 - hb, 2 * m), s.u = ld4q(L, g.bl0 - hb, 2 * m), s.v = ld4q(L, g.bl1 - hb, 2 * m);
    }
  };
  constexpr int NP = 3 * Q::SPP;   // pairs of quad slots = K steps
  // taps PF pairs ahead of the pair that is multiplied (PF + 1 buffers in rotation; 0 = fetched where they are used).  The
  // training forward has registers for one pair ahead; the inference / pose-only forwards could afford two, which measures the same.
  constexpr int PF = !JT_B16_PIPE ? 0 : (REC ? 1 : JT_B16_PF_INFER);
  TapSlot buf[PF + 1][2];
  auto load_pair = [&](int p, TapSlot* b) {
    const int ip = p / Q::SPP, s2 = p - ip * Q::SPP;
    load_slot(ip, 2 * s2, b[0]);
    if (2 * s2 + 1 < C::NSLOT) load_slot(ip, 2 * s2 + 1, b[1]);
  };
#pragma unroll
  for (int p = 0; p < PF && p < NP; ++p) load_pair(p, buf[p]);
#pragma unroll
  for (int pp = 0; pp < NP; ++pp) {
    const int i = pp / Q::SPP, sp = pp - i * Q::SPP;
    if (pp + PF < NP) load_pair(pp + PF, buf[(pp + PF) % (PF + 1)]);
    __builtin_amdgcn_sched_barrier(0);
    // JT_B16_PIN_TAPS: pin the consumption of this pair's taps BEHIND the loads just issued for the next pair.  Without it
    // the instruction selector, which orders pure arithmetic by data dependence only, places the products of a pair directly
    // behind that pair's own loads -- one region earlier -- and the "prefetch" waits for every load right after issuing it
    // (s_waitcnt vmcnt(11) ... vmcnt(0) behind twelve loads; with the pins: vmcnt(23) ... vmcnt(12)).  Measured with the
    // pins: training forward 0.502 ms against 0.485 without, eval render 155 against 157 ms -- the gather phase is not
    // waiting for the latency of its own loads (profiles/round3_forward_phase_stamps.txt), so the pins stay off.
    if (JT_B16_PIN_TAPS && PF > 0) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        if (2 * sp + half < C::NSLOT) {
          TapSlot& t = buf[pp % (PF + 1)][half];
          asm volatile("" : "+v"(t.a.x), "+v"(t.a.y), "+v"(t.a.z), "+v"(t.a.w));
          asm volatile("" : "+v"(t.b.x), "+v"(t.b.y), "+v"(t.b.z), "+v"(t.b.w));
          asm volatile("" : "+v"(t.c.x), "+v"(t.c.y), "+v"(t.c.z), "+v"(t.c.w));
          asm volatile("" : "+v"(t.d.x), "+v"(t.d.y), "+v"(t.d.z), "+v"(t.d.w));
          asm volatile("" : "+v"(t.u.x), "+v"(t.u.y), "+v"(t.u.z), "+v"(t.u.w));
          asm volatile("" : "+v"(t.v.x), "+v"(t.v.y), "+v"(t.v.z), "+v"(t.v.w));
        }
      }
    }
    const TapGeo& g = geo[i];
    float bv8[8];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int m = 2 * sp + half;
      if (m < C::NSLOT) {
        const TapSlot& s = buf[pp % (PF + 1)][half];
        const bool live = C::CA % 8 == 0 || (2 * m + h) * 4 < C::CA;
        float pr[4];
        pr[0] = (g.w00 * s.a.x + g.w10 * s.b.x + g.w01 * s.c.x + g.w11 * s.d.x) * (g.lw0 * s.u.x + g.lw1 * s.v.x);
        pr[1] = (g.w00 * s.a.y + g.w10 * s.b.y + g.w01 * s.c.y + g.w11 * s.d.y) * (g.lw0 * s.u.y + g.lw1 * s.v.y);
        pr[2] = (g.w00 * s.a.z + g.w10 * s.b.z + g.w01 * s.c.z + g.w11 * s.d.z) * (g.lw0 * s.u.z + g.lw1 * s.v.z);
        pr[3] = (g.w00 * s.a.w + g.w10 * s.b.w + g.w01 * s.c.w + g.w11 * s.d.w) * (g.lw0 * s.u.w + g.lw1 * s.v.w);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (REC && live && onrec)
            rec_st(rec_at(rt, BwdCfg<C>::R_PROD + i * C::CA + 8 * m + k, 4u * (unsigned)j + 512u * (unsigned)h), pr[k]);
          bv8[4 * half + k] = live ? pr[k] : 0.f;
        }
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) bv8[4 * half + k] = 0.f;
      }
    }
    const B3 bb = split8(bv8);
    const B3 aa = load_b3(img, Q::V_BASIS + pp * 3 * 64, lane);
    facc = mfma6(aa, bb, facc);
    __builtin_amdgcn_sched_barrier(0);
  }
  return facc;
}

// scheduling hint for one K step whose matrix instructions overlap the NEXT step's operand preparation: alternate one MFMA
// (32 cycles on the matrix pipe, 4 to issue) with up to `valu` vector instructions.  A wave issues in order, so the overlap has
// to be in the program order; the compiler left to itself emits the twelve MFMAs of a step back to back.
template <int N, int VALU, int DS, int LEAD>
__device__ inline void interleave_mfma() {
  __builtin_amdgcn_sched_group_barrier(0x100, DS, 0);      // the step's A operands leave the LDS first ...
  __builtin_amdgcn_sched_group_barrier(0x002, LEAD, 0);    // ... and have LEAD vector instructions to arrive
#pragma unroll
  for (int k = 0; k < N; ++k) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // 1 MFMA
    __builtin_amdgcn_sched_group_barrier(0x002, VALU, 0);  // VALU
  }
}

#ifndef JT_B16_IL1
#define JT_B16_IL1 8
#endif
#ifndef JT_B16_LEAD
#define JT_B16_LEAD 16
#endif
#ifndef JT_B16_IL2
#define JT_B16_IL2 3
#endif

template <class C>
__device__ inline Hidden<C> layer1_b16(const uint4* img, const float* tail, const f32x16& facc, const float vd[3],
                                       const PeMask& pm, int h, int lane) {
  typedef B16Cfg<C> Q;
  Hidden<C> acc;
#pragma unroll
  for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc.v[mt][r] = tail[Q::F_B1 + mt * 32 + rowmap(r, h)];
  float v8[8];
  B3 cur;   // the split operand of the previous K step: its MFMAs are issued among this step's encodings and split
#pragma unroll
  for (int r = 0; r < (Q::NS1 * 8 + 4) / 5; ++r) {
    float in[5] = {0.f, 0.f, 0.f, 0.f, 0.f}, sn, cs;
    if (r < Q::RS) l1_inputs<C>(facc, vd, pm, h, r, in, &sn, &cs);
#pragma unroll
    for (int t = 0; t < 5; ++t) {
      const int idx = 5 * r + t;
      if (idx < Q::NS1 * 8) {
        v8[idx & 7] = in[t];
        if ((idx & 7) == 7) {
          const int st = idx >> 3;
          const B3 nb = split8(v8);
          if (st > 0) {
#pragma unroll
            for (int mt = 0; mt < C::MT; ++mt) {
              const B3 aa = load_b3(img, Q::V_W1 + ((st - 1) * C::MT + mt) * 3 * 64, lane);
              acc.v[mt] = mfma6(aa, cur, acc.v[mt]);
            }
            interleave_mfma<6 * C::MT, JT_B16_IL1, 3 * C::MT, JT_B16_LEAD>();
          }
          cur = nb;
          __builtin_amdgcn_sched_barrier(0);  // one step's encodings at a time (VGPR pressure)
        }
      }
    }
  }
#pragma unroll
  for (int mt = 0; mt < C::MT; ++mt) {
    const B3 aa = load_b3(img, Q::V_W1 + ((Q::NS1 - 1) * C::MT + mt) * 3 * 64, lane);
    acc.v[mt] = mfma6(aa, cur, acc.v[mt]);
  }
  return acc;
}

template <class C>
__device__ inline Hidden<C> layer2_b16(const uint4* img, const float* tail, const Hidden<C>& h1, int h, int lane) {
  typedef B16Cfg<C> Q;
  Hidden<C> acc;
#pragma unroll
  for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc.v[mt][r] = tail[Q::F_B2 + mt * 32 + rowmap(r, h)];
  B3 cur;
#pragma unroll
  for (int st = 0; st <= Q::NS2; ++st) {
    B3 nb;
    if (st < Q::NS2) {
      float v8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v8[e] = h1.v[(8 * st + e) >> 4][(8 * st + e) & 15];
      nb = split8(v8);
    }
    if (st > 0) {
#pragma unroll
      for (int mt = 0; mt < C::MT; ++mt) {
        const B3 aa = load_b3(img, Q::V_W2 + ((st - 1) * C::MT + mt) * 3 * 64, lane);
        acc.v[mt] = mfma6(aa, cur, acc.v[mt]);
      }
      if (st < Q::NS2) interleave_mfma<6 * C::MT, JT_B16_IL2, 3 * C::MT, 8>();
    }
    cur = nb;
    __builtin_amdgcn_sched_barrier(0);
  }
  return acc;
}

template <class C>
struct BwdCfg {
  static constexpr int TP_ROWS = 68;  // 48 rows of product gradients + 32 step records (jt_walk.h);
                                      // also the 2*16*64-float sin/cos stash
  static_assert(C::CA <= 48 && (TP_ROWS - 48) * 33 >= 32 * kRecWords, "LDS tile");
  static constexpr int TP_LD = 33;
  static constexpr int WAVE_FLOATS = TP_ROWS * TP_LD + 32 * 4 + 32 * 4;
  static constexpr int NWAVE = 8;
  static constexpr int LDS_FLOATS = C::LDS_FLOATS + NWAVE * WAVE_FLOATS;
  static constexpr int STASH_FLOATS = 2 * 16 * 64;                               // split backward: the chain's sin / cos stash only
  static constexpr int LDS_FLOATS_SPLIT = C::LDS_FLOATS + NWAVE * STASH_FLOATS;
  static constexpr int PT = (C::CA + 31) / 32;  // M tiles of one plane's channels in the basis backward
  // TILE-BLOCKED records: rec[tile][row][32 samples]; every accumulator register goes out as one coalesced
  // 128-byte-per-half store, no transposition (k_wgrad reads rows, one per lane).  The training forward
  // writes the layer inputs (PROD, F, VD, GEO, H1, MID, MASK), the backward adds the gradients (GO, G2, G1, GF).
  // Round 6 ("lean tape"): the last two row groups exist only in the full layout.  A lean tile is R_LEAN rows: the products are
  // not recorded (dBasis is formed in k_shade_scatter) and neither is G2 (the dW2 GEMM derives it from GO and the sign words).
  static constexpr int R_G1 = 0;
  static constexpr int R_H1 = R_G1 + C::HID;
  static constexpr int R_MID = R_H1 + C::HID;
  static constexpr int R_F = R_MID + C::IN3;
  static constexpr int R_GF = R_F + 32;
  static constexpr int R_GO = R_GF + 32;
  static constexpr int R_VD = R_GO + 4;    // view direction (3 rows + pad)
  static constexpr int R_MASK = R_VD + 4;  // ReLU sign bits of h1 / h2: row 2*layer + lane half, one word per sample
  static constexpr int R_GEO = R_MASK + 4; // normalised sample coordinates (3 rows + pad)
  static constexpr int R_LEAN = R_GEO + 4; // rows of a lean tile (VM-48: 272 = 1 088 bytes per sample; 20-channel scene: 188 = 752)
  static constexpr int R_G2 = R_LEAN;
  static constexpr int R_PROD = R_G2 + C::HID;
  static constexpr int REC_FLOATS = R_PROD + C::NC;   // rows of a full tile (480 / 280); a tile is [rows][32 samples]
};

// record rows row0 + t*32 + rowmap(r,h) of the tile <- accumulator registers (lane = sample j)
template <int NT>
__device__ inline void rec_store(float* rt, int row0, const f32x16* v, int j, int h, bool on) {
  if (!on) return;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      rec_st(rec_at(rt, row0 + t * 32 + rowmap(r, 0), 4u * (unsigned)j + 512u * (unsigned)h), v[t][r]);
}

}  // namespace jt
