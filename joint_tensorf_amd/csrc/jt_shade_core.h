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
// lane part `lane_bytes` (4 * sample [+ 512 * lane half for rows rowmap(r, 0) + 4 h]): scalar base + 32-bit lane offset,
// the form global loads / stores address without 64-bit vector arithmetic
__device__ inline float* rec_at(float* rt, int row, unsigned lane_bytes) {
  return reinterpret_cast<float*>(reinterpret_cast<char*>(rt) + (size_t)row * 128 + lane_bytes);
}
__device__ inline const float* rec_at(const float* rt, int row, unsigned lane_bytes) {
  return reinterpret_cast<const float*>(reinterpret_cast<const char*>(rt) + (size_t)row * 128 + lane_bytes);
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

template <class C>
__device__ inline void load_weights_lds(float* s, const MlpDev& M) {
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int i = tid; i < 32 * C::LDB; i += nt) {
    int a = i / C::LDB, c = i - a * C::LDB;
    s[C::O_BASIS + i] = (a < C::APP && c < C::NC) ? M.basis[a * C::NC + c] : 0.f;
  }
  for (int i = tid; i < C::HID * C::LD1; i += nt) {
    int u = i / C::LD1, c = i - u * C::LD1;
    s[C::O_W1 + i] = (c < C::IN1) ? M.w1[u * C::IN1 + c] : 0.f;
  }
  for (int i = tid; i < C::HID * C::LD2; i += nt) {
    int u = i / C::LD2, c = i - u * C::LD2;
    s[C::O_W2 + i] = (c < C::HID) ? M.w2[u * C::HID + c] : 0.f;
  }
  for (int i = tid; i < C::IN3 * 4; i += nt) {
    int k = i >> 2, c = i & 3;
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
      if (m & 1) __builtin_amdgcn_sched_barrier(0);  // at most two quad slots of taps in flight
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

template <class C>
struct BwdCfg {
  static constexpr int TP_ROWS = 68;  // 48 rows of product gradients + 32 step records (jt_walk.h);
                                      // also the 2*16*64-float sin/cos stash
  static_assert(C::CA <= 48 && (TP_ROWS - 48) * 33 >= 32 * kRecWords, "LDS tile");
  static constexpr int TP_LD = 33;
  static constexpr int WAVE_FLOATS = TP_ROWS * TP_LD + 32 * 4 + 32 * 4;
  static constexpr int NWAVE = 8;
  static constexpr int LDS_FLOATS = C::LDS_FLOATS + NWAVE * WAVE_FLOATS;
  static constexpr int PT = (C::CA + 31) / 32;  // M tiles of one plane's channels in the basis backward
  // TILE-BLOCKED records: rec[tile][row][32 samples]; every accumulator register goes out as one coalesced
  // 128-byte-per-half store, no transposition (k_wgrad reads rows, one per lane).  The training forward
  // writes the layer inputs (PROD, F, VD, GEO, H1, MID, MASK), the backward adds the gradients (GO, G2, G1, GF).
  static constexpr int R_G1 = 0;
  static constexpr int R_G2 = R_G1 + C::HID;
  static constexpr int R_H1 = R_G2 + C::HID;
  static constexpr int R_MID = R_H1 + C::HID;
  static constexpr int R_F = R_MID + C::IN3;
  static constexpr int R_GF = R_F + 32;
  static constexpr int R_GO = R_GF + 32;
  static constexpr int R_VD = R_GO + 4;    // view direction (3 rows + pad)
  static constexpr int R_MASK = R_VD + 4;  // ReLU sign bits of h1 / h2: row 2*layer + lane half, one word per sample
  static constexpr int R_GEO = R_MASK + 4; // normalised sample coordinates (3 rows + pad)
  static constexpr int R_PROD = R_GEO + 4;
  static constexpr int REC_FLOATS = R_PROD + C::NC;   // rows of one tile; a tile is [REC_FLOATS][32 samples]
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
