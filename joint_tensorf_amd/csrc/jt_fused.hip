// Single-launch render + photometric loss + backward to the RAYS (gfx950): the renderer of test-time pose optimisation
// (model/bat.py:265-292: 400-600 iterations per held-out view in which only a 6-vector is trained) with forward and backward
// fused -- the "forward and backward fused so pose gradients come out of a single launch" of the north star, on the path
// where no factor or weight gradient exists.
//
// Why this fuses and the training step does not: the photometric loss is separable per ray (d loss / d rgb_r needs ray r and
// the constant 1 / (3 R) only: model/tensorf.py:96-124, base.py:259-261), and without factor / weight gradients nothing of a
// ray's backward leaves the ray.  So ONE WAVE owns one ray from its first sample to its gradient:
//   A  march        sample, density features (16-byte tap gathers), activation, transmittance scan  (k_march_fwd's code)
//   B  shade        the ray's shaded samples in tiles of 32 through the transposed fp32-MFMA chain (jt_shade_core.h)
//   C  loss         composite, clamp, (rgb - target)^2, d loss / d rgb; reverse scan -> d loss / d sigma_feat per sample
//   D1 density      d feat / d position of every in-box sample (tap differences), summed into d loss / d (o, d)
//   D2 appearance   per tile: the MLP backwards on the matrix cores -> basis^T -> product gradients, which land in the
//                   lane that gathered those channel quads: the taps are gathered once more and the position gradient is
//                   a lane-local sum (no walkers, no LDS transposition, no atomics)
// and writes g_rays_o / g_rays_d [ray] with plain stores: the gradient is bit-reproducible run to run.  There is no tape:
// what B leaves for D2 (basis output, ReLU sign words, colours: 160 bytes per shaded sample) sits in a per-wave scratch slot
// that is reused ray after ray.  Work split: ~2 000 rays of a test-time lattice against 2 048 resident waves (8 per CU).
#include <algorithm>
#include <cstdlib>

#include "jt_shade_core.h"

namespace jt {

constexpr int kFusedWaves = 8;          // waves per workgroup (two per SIMD)
constexpr int kTileRows = 40;           // scratch rows per 32-sample tile: F 32, ReLU sign words 4, rgb 3 (+ pad)
constexpr int kRowMask = 32, kRowRgb = 36;

__device__ inline void wave_sync_mem() {
  // LDS and global scratch written by some lanes of this wave are read by others: drain both queues, keep the compiler
  // from moving accesses across (one wave: no barrier instruction needed, its memory operations retire in order)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// d(density feature) / d(normalised coordinates) of one point (bateRF.py:41-94 differentiated; grid_sample's coordinate
// backward: taps outside the factor count as zero values, the fractional weights stay as they are)
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

struct FusedArgs {
  const float* rays_o;
  const float* rays_d;
  const float* zvals;     // NDC: the shared row of sample depths [S]
  const float* image;     // [views][3][HW]
  const long* ray_idx;    // [rays_per_view] pixel of lattice point k (the same lattice in every view)
  int R, rays_per_view, HW;
  float loss_scale;       // w_render / (3 R): loss = loss_scale * sum (rgb - target)^2
  float* rgb;             // [R][3]
  float* depth;           // [R]
  float* opacity;         // [R]
  float* sqerr;           // [R] sum over the ray's three channels of (rgb - target)^2
  float* loss;            // [1] loss_scale * sum of sqerr, written by the workgroup that finishes last
  float* loss_acc;        // workspace head: running sum + arrival counter, left zeroed for the next launch
  unsigned* loss_cnt;
  float* g_rays_o;        // [R][3]
  float* g_rays_d;        // [R][3]
  float* scratch;
  long slot_floats;
  int Sp;                 // S rounded up to 64
  int ablate;             // profiling knob (JT_FUSED_ABLATE): 1 no shade forward, 2 no density gradient pass, 4 no appearance
                          // backward, 8 no reverse scan, 16 no march (nothing at all per ray)
};

// ---- per-tile bodies of the two matrix-core phases.  `ow` = the slot (wave index) of the workgroup whose ray the tile
//      belongs to: the eight rays a workgroup holds pool their tiles and all eight waves pop from the pool (phase B, D2),
//      so a long ray does not keep seven waves waiting (a wave per ray from start to end measured 1.45 x the time of the
//      same chains in k_shade_fwd / k_shade_bwd: the launch lasts as long as its longest ray).
struct OwnerRay {
  Ray r;
  float vd[3];
  float* sc;              // the owner's scratch slot
  const uint16_t* sidx;   // the owner's shaded-sample list (LDS)
  int nsh;
};

// The two tile bodies are real functions (not inlined): inlined into the kernel, the backward body + the tap re-gather
// spilled 130 registers (2.86 ms per launch on the dense bench scene against 2.65 ms with the calls)
#ifdef JT_FUSED_INLINE
#define JT_FUSED_FN __device__ inline
#else
#define JT_FUSED_FN __device__ __attribute__((noinline))
#endif
template <class C>
JT_FUSED_FN void fused_tile_fwd(const Dev& D, const FusedArgs& A, const PeMask& pm, const float* smem,
                                      const OwnerRay& o, int t, int j, int h, float comp[3]) {
  const int Sp = A.Sp;
  const int k = t * 32 + j;
  const bool on = k < o.nsh;
  const int i = o.sidx[on ? k : o.nsh - 1];
  const float z = sample_z(D, o.r, A.zvals, i);
  float p[3], n[3];
  sample_point(D, o.r, z, p);
  normalize(D, p, n);
  float* tb = o.sc + 6 * Sp + (size_t)t * kTileRows * 32;
  f32x16 facc = gather_basis<C, false>(D, smem, n, j, h, nullptr, false);
#pragma unroll
  for (int rr = 0; rr < 16; ++rr) tb[(rowmap(rr, 0) + 4 * h) * 32 + j] = facc[rr];
  Hidden<C> h1 = layer1<C>(smem, facc, o.vd, pm, j, h);
  relu_<C>(h1);
  unsigned mask1 = 0u;
#pragma unroll
  for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) mask1 |= (h1.v[mt][rr] > 0.f) ? (1u << (mt * 16 + rr)) : 0u;
  Hidden<C> h2 = layer2<C>(smem, h1, j, h);
  relu_<C>(h2);
  unsigned mask2 = 0u;
#pragma unroll
  for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) mask2 |= (h2.v[mt][rr] > 0.f) ? (1u << (mt * 16 + rr)) : 0u;
  tb[(kRowMask + h) * 32 + j] = __uint_as_float(mask1);
  tb[(kRowMask + 2 + h) * 32 + j] = __uint_as_float(mask2);
  float out[3];
  layer3<C>(smem, h2, o.vd, pm, h, out);
  const float wi = o.sc[Sp + i];  // a_w
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float col = 1.f / (1.f + expf(-out[c]));
    if (h == 0) tb[(kRowRgb + c) * 32 + j] = col;
    comp[c] = (h == 0 && on) ? wi * col : 0.f;
  }
}

// d loss / d (sample position) of the tile's 32 samples through the appearance path, summed into go / gd of the tile
template <class C>
JT_FUSED_FN void fused_tile_bwd(const Dev& D, const FusedArgs& A, const PeMask& pm, const float* smem, float* stash,
                                      const OwnerRay& o, const float g[3], int t, int j, int h, int lane, float go[3],
                                      float gd[3]) {
  constexpr int HOFF = (C::KIND == JT_MLP_FEA) ? 0 : 12;
  constexpr int PT = (C::CA + 31) / 32;
  const int Sp = A.Sp;
  const int k = t * 32 + j;
  const bool on = k < o.nsh;
  const int i = o.sidx[on ? k : o.nsh - 1];
  const float* tb = o.sc + 6 * Sp + (size_t)t * kTileRows * 32;
  f32x16 facc;
#pragma unroll
  for (int rr = 0; rr < 16; ++rr) facc[rr] = tb[(rowmap(rr, 0) + 4 * h) * 32 + j];
  const unsigned mask1 = __float_as_uint(tb[(kRowMask + h) * 32 + j]);
  const unsigned mask2 = __float_as_uint(tb[(kRowMask + 2 + h) * 32 + j]);
#pragma unroll
  for (int rr = 0; rr < 16; ++rr) {
    if (rowmap(rr, 0) >= C::APP && rowmap(rr, 1) >= C::APP) continue;
    float sn, cs;
    sincos_grad(facc[rr], &sn, &cs);
    stash[(2 * rr) * 64 + lane] = sn;
    stash[(2 * rr + 1) * 64 + lane] = cs;
  }
  const float wi = o.sc[Sp + i];
  float gout[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float col = tb[(kRowRgb + c) * 32 + j];
    gout[c] = on ? wi * g[c] * col * (1.f - col) : 0.f;
  }
  Hidden<C> G2;
#pragma unroll
  for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
      const int kk = HOFF + mt * 32 + rowmap(rr, 0) + 4 * h;
      const float4 w = *reinterpret_cast<const float4*>(smem + C::O_W3 + kk * 4);
      const float gsum = gout[0] * w.x + gout[1] * w.y + gout[2] * w.z;
      G2.v[mt][rr] = ((mask2 >> (mt * 16 + rr)) & 1u) ? gsum : 0.f;
    }
  Hidden<C> G1;
#pragma unroll
  for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) G1.v[mt][rr] = 0.f;
#pragma unroll
  for (int mi = 0; mi < C::MT; ++mi) {
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
      const int irow = mi * 32 + rowmap(rr, 0) + 4 * h;
      const float bv = G2.v[mi][rr];
#pragma unroll
      for (int mk = 0; mk < C::MT; ++mk) {
        const float av = smem[C::O_W2 + irow * C::LD2 + mk * 32 + j];
        G1.v[mk] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, G1.v[mk], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) G1.v[mt][rr] = ((mask1 >> (mt * 16 + rr)) & 1u) ? G1.v[mt][rr] : 0.f;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the stash: written above, read below by the same lanes
  f32x16 gf;
#pragma unroll
  for (int rr = 0; rr < 16; ++rr) gf[rr] = 0.f;
#pragma unroll
  for (int tt = 0; tt < 5; ++tt) {
    int col;
    if (j < C::APP) {
      if (C::KIND == JT_MLP_FEA) col = (tt == 0) ? j : C::APP + 3 + 4 * j + (tt - 1);
      else col = (tt == 0) ? j : C::APP + 4 * j + (tt - 1);
    } else {
      col = C::IN1;
    }
    f32x16 gin;
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) gin[rr] = 0.f;
#pragma unroll
    for (int mi = 0; mi < C::MT; ++mi) {
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        const int irow = mi * 32 + rowmap(rr, 0) + 4 * h;
        const float av = smem[C::O_W1 + irow * C::LD1 + col];
        gin = __builtin_amdgcn_mfma_f32_32x32x2f32(av, G1.v[mi][rr], gin, 0, 0, 0);
      }
    }
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
      if (rowmap(rr, 0) >= C::APP && rowmap(rr, 1) >= C::APP) continue;
      const float sn = stash[(2 * rr) * 64 + lane], cs = stash[(2 * rr + 1) * 64 + lane];
      float dv;
      if (tt == 0) dv = 1.f;
      else if (tt == 1) dv = cs * pm.f0;
      else if (tt == 2) dv = 2.f * (1.f - 2.f * sn * sn) * pm.f1;
      else if (tt == 3) dv = -sn * pm.f0;
      else dv = -4.f * sn * cs * pm.f1;
      gf[rr] += gin[rr] * dv;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int rr = 0; rr < 16; ++rr)
    if (rowmap(rr, 0) + 4 * h >= C::APP) gf[rr] = 0.f;
  // ---- basis^T, then the position gradient of the plane from its taps (gathered once more: they are cache-hot) ----
  __builtin_amdgcn_sched_barrier(0);
  const float z = sample_z(D, o.r, A.zvals, i);   // (the sample's geometry only now: nothing of it lives across the chain)
  float p[3], n[3];
  sample_point(D, o.r, z, p);
  normalize(D, p, n);
  float gn3[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) {
    __builtin_amdgcn_sched_barrier(0);  // planes one after the other: nothing of the next plane's loads is hoisted up here
    // (the plane loop is unrolled: every index below is a compile-time constant.  A rolled loop indexed n[] / the sums
    //  dynamically, through scratch memory, and came back wrong for one axis -- measured against the staged path,
    //  tools/diag_fused.py)
    const PlaneTaps tp = plane_taps(n[kM0(pl)], n[kM1(pl)], D.ph[pl], D.pw[pl], C::CA);
    const Axis l = axis_taps(n[kV(pl)], D.ll[pl]);
    const float* P = D.aP[pl];
    const float* L = D.aL[pl];
    const unsigned hb = 16u * (unsigned)h;
    const unsigned b00 = 4u * (unsigned)tp.o00 + hb, b10 = 4u * (unsigned)tp.o10 + hb, b01 = 4u * (unsigned)tp.o01 + hb,
                   b11 = 4u * (unsigned)tp.o11 + hb, bl0 = 4u * (unsigned)(l.c0 * C::CA) + hb,
                   bl1 = 4u * (unsigned)(l.c1 * C::CA) + hb;
    const float m00 = tp.ax.m0 * tp.ay.m0, m10 = tp.ax.m1 * tp.ay.m0, m01 = tp.ax.m0 * tp.ay.m1, m11 = tp.ax.m1 * tp.ay.m1;
    const float fx = tp.ax.f, fy = tp.ay.f;
    float sx = 0.f, sy = 0.f, sl = 0.f;
    // one M tile of product gradients (32 channels of the plane) at a time: 16 registers live, not 16 PT
#pragma unroll
    for (int TT = 0; TT < PT; ++TT) {
    f32x16 gpt;
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) gpt[rr] = 0.f;
    {
      const int ch = TT * 32 + j;
      const int col = (ch < C::CA) ? pl * C::CA + ch : C::NC;
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        const int arow = rowmap(rr, 0) + 4 * h;
        const float av = smem[C::O_BASIS + arow * C::LDB + col];
        gpt = __builtin_amdgcn_mfma_f32_32x32x2f32(av, gf[rr], gpt, 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 4 * TT; m < C::NSLOT && m < 4 * TT + 4; ++m) {
      const int q = 2 * m + h;
      const bool live = q * 4 < C::CA;
      float4 a, b, c, d, u, v;
      if (C::CA % 8 == 0 || m + 1 < C::NSLOT) {
        a = ld4q(P, b00, 2 * m), b = ld4q(P, b10, 2 * m), c = ld4q(P, b01, 2 * m), d = ld4q(P, b11, 2 * m);
        u = ld4q(L, bl0, 2 * m), v = ld4q(L, bl1, 2 * m);
      } else {  // last slot of an odd quad count: half 1 has no quad there (its gradients are zero)
        a = ld4q(P, b00 - hb, 2 * m), b = ld4q(P, b10 - hb, 2 * m), c = ld4q(P, b01 - hb, 2 * m);
        d = ld4q(P, b11 - hb, 2 * m), u = ld4q(L, bl0 - hb, 2 * m), v = ld4q(L, bl1 - hb, 2 * m);
      }
      const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w}, cv[4] = {c.x, c.y, c.z, c.w},
                  dv[4] = {d.x, d.y, d.z, d.w}, uv[4] = {u.x, u.y, u.z, u.w}, vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        // product gradient of channel 8 m + 4 h + kk of this plane: row rowmap(4 (m & 3) + kk, h) of M tile m >> 2
        const float gpr = live ? gpt[4 * (m & 3) + kk] : 0.f;
        const float am = av[kk] * m00, bm = bv[kk] * m10, cm = cv[kk] * m01, dm = dv[kk] * m11;
        const float um = uv[kk] * l.m0, vm = vv[kk] * l.m1;
        const float pv = tp.w00 * av[kk] + tp.w10 * bv[kk] + tp.w01 * cv[kk] + tp.w11 * dv[kk];
        const float lv = l.w0 * uv[kk] + l.w1 * vv[kk];
        const float gl = gpr * lv;
        sx += gl * ((1.f - fy) * (bm - am) + fy * (dm - cm));
        sy += gl * ((1.f - fx) * (cm - am) + fx * (dm - bm));
        sl += gpr * pv * (vm - um);
      }
      if (m & 1) __builtin_amdgcn_sched_barrier(0);
    }
    }
    gn3[kM0(pl)] += sx * tp.ax.scale;
    gn3[kM1(pl)] += sy * tp.ay.scale;
    gn3[kV(pl)] += sl * l.scale;
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    // the two lane halves hold the sums over their own channel quads of the same sample
    const float tot = gn3[a] + __shfl_xor(gn3[a], 32);
    const float gpos = (on && h == 0) ? tot * D.inv[a] : 0.f;
    go[a] = gpos;
    gd[a] = gpos * z;
  }
}

// Tables of the workgroup's eight rays (LDS)
struct FusedTabs {
  int ray[kFusedWaves];
  int nsh[kFusedWaves];
  int tile0[kFusedWaves + 1];
  float g[kFusedWaves][4];
  float acc[kFusedWaves];
  int queue[2];
};

template <class C>
__global__ __launch_bounds__(512, 2) void k_pose_fused(Dev D, MlpDev M, PeMask pm, FusedArgs A) {
  extern __shared__ __align__(16) float smem[];
  load_weights_lds<C>(smem, M);
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j_ = lane & 31, h_ = lane >> 5;
  const int Sp = A.Sp, S = D.S;
  constexpr int WAVE_FLOATS = 2 * 16 * 64;  // sin / cos stash of the layer-1 backward
  float* stash = smem + C::LDS_FLOATS + wv * WAVE_FLOATS;
  uint16_t* sidx_all = reinterpret_cast<uint16_t*>(smem + C::LDS_FLOATS + kFusedWaves * WAVE_FLOATS);
  FusedTabs* tabs = reinterpret_cast<FusedTabs*>(sidx_all + (size_t)kFusedWaves * Sp);
  uint16_t* s_idx = sidx_all + (size_t)wv * Sp;
  float* sc0 = A.scratch + (size_t)blockIdx.x * kFusedWaves * A.slot_floats;  // the workgroup's eight scratch slots
  float* sc = sc0 + (size_t)wv * A.slot_floats;
  float* a_feat = sc;
  float* a_w = sc + Sp;
  float* a_alpha = sc + 2 * Sp;   // after the reverse scan: d loss / d sigma_feat
  float* a_T = sc + 3 * Sp;
  float* a_valid = sc + 4 * Sp;
  int* a_rank = reinterpret_cast<int*>(sc + 5 * Sp);
  float* tiles = sc + 6 * Sp;
  // per-tile partial sums of the cooperative phases (composited colour in B, position gradients in D2), summed by the
  // owner in tile order: the result does not depend on which wave took which tile
  const long part_off = 6 * (long)Sp + (long)(Sp / 32) * kTileRows * 32;
  float* part = sc + part_off;   // [tile][8]; row Sp / 32 holds the owner's D1 sums
  float sq_wave = 0.f;           // lane 0: squared error of the rays this wave rendered
  if (threadIdx.x < 2) tabs->queue[threadIdx.x] = 0;
  __syncthreads();
  const int nslots = gridDim.x * kFusedWaves;
  // rays of a workgroup are spread over the batch (slot s of workgroup b renders ray s * gridDim.x + b of a round):
  // neighbours in the lattice have similar lengths and would make long and short workgroups
  for (int base = 0; base < A.R; base += nslots) {
    int j = j_, h = h_;  // re-materialised per round (see k_shade_bwd: keeps per-lane address math out of loop-invariant VGPRs)
    asm volatile("" : "+v"(j), "+v"(h));
    const int ray = base + wv * (int)gridDim.x + (int)blockIdx.x;
    const bool active = ray < A.R;
    int nsh = 0;
    // ================= A: march (own ray) =================
    if (active) {
      Ray r;
      load_ray(D, A.rays_o, A.rays_d, nullptr, nullptr, ray, r);
      float carry = 1.f, acc = 0.f, dep = 0.f;
      int cnt = 0;
      for (int sb = 0; sb < ((A.ablate & 16) ? 0 : S); sb += 64) {
        const int i = sb + lane;
        const bool live = i < S;
        float z0 = 0.f, delta = 0.f, feat = 0.f;
        bool valid = false;
        if (live) {
          z0 = sample_z(D, r, A.zvals, i);
          if (i < S - 1) delta = (sample_z(D, r, A.zvals, i + 1) - z0) * r.norm;
          float p[3], n[3];
          valid = sample_valid(D, r, z0, p);
          if (valid) {
            normalize(D, p, n);
            float f = 0.f;
            const int Cd = D.Cd;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
              const PlaneTaps t = plane_taps(n[kM0(pl)], n[kM1(pl)], D.ph[pl], D.pw[pl], Cd);
              const Axis l = axis_taps(n[kV(pl)], D.ll[pl]);
              const float* P = D.dP[pl];
              const float* L = D.dL[pl];
              const int l0 = l.c0 * Cd, l1 = l.c1 * Cd;
              float s = 0.f;
              for (int q = 0; q < Cd; q += 4) {
                const float4 a = ld4(P + t.o00 + q), b = ld4(P + t.o10 + q), c = ld4(P + t.o01 + q), d = ld4(P + t.o11 + q);
                const float4 u = ld4(L + l0 + q), v = ld4(L + l1 + q);
                const float px = t.w00 * a.x + t.w10 * b.x + t.w01 * c.x + t.w11 * d.x;
                const float py = t.w00 * a.y + t.w10 * b.y + t.w01 * c.y + t.w11 * d.y;
                const float pz = t.w00 * a.z + t.w10 * b.z + t.w01 * c.z + t.w11 * d.z;
                const float pw = t.w00 * a.w + t.w10 * b.w + t.w01 * c.w + t.w11 * d.w;
                s += px * (l.w0 * u.x + l.w1 * v.x) + py * (l.w0 * u.y + l.w1 * v.y) + pz * (l.w0 * u.z + l.w1 * v.z) +
                     pw * (l.w0 * u.w + l.w1 * v.w);
              }
              f += s;
            }
            feat = f;
          }
        }
        const float sigma = valid ? density_act(D.act, feat + D.shift) : 0.f;
        const float alpha = 1.f - expf(-sigma * (delta * D.dist_scale));
        const float f1 = live ? (1.f - alpha + 1e-10f) : 1.f;
        float inc = f1;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const float tt = __shfl_up(inc, o);
          if (lane >= o) inc *= tt;
        }
        const float e1 = __shfl_up(inc, 1);
        const float T = carry * ((lane == 0) ? 1.f : e1);
        carry *= __shfl(inc, 63);
        const float w = live ? alpha * T : 0.f;
        acc += w;
        dep += w * z0;
        const bool shade = live && (w > D.thres);
        const unsigned long long bal = __ballot(shade);
        const int rank = cnt + __popcll(bal & ((1ull << lane) - 1ull));
        if (shade) s_idx[rank] = (uint16_t)i;
        cnt += __popcll(bal);
        if (live) {
          a_feat[i] = feat;
          a_w[i] = w;
          a_alpha[i] = alpha;
          a_T[i] = T;
          a_valid[i] = valid ? 1.f : 0.f;
          a_rank[i] = shade ? rank : -1;
        }
      }
      acc = wave_sum(acc);
      dep = wave_sum(dep);
      nsh = (A.ablate & 1) ? 0 : cnt;
      if (lane == 0) {
        tabs->acc[wv] = acc;
        A.opacity[ray] = acc;
        A.depth[ray] = dep + (1.f - acc) * r.d[2] - (D.near_dev ? *D.near_dev : D.near_) + 0.05f;   // batBase.py:147-150
      }
    }
    if (lane == 0) {
      tabs->ray[wv] = active ? ray : -1;
      tabs->nsh[wv] = nsh;
    }
    __threadfence_block();
    __syncthreads();
    if (threadIdx.x == 0) {
      int run = 0;
      for (int w = 0; w < kFusedWaves; ++w) {
        tabs->tile0[w] = run;
        run += (tabs->nsh[w] + 31) >> 5;
      }
      tabs->tile0[kFusedWaves] = run;
    }
    __syncthreads();
    const int pool = tabs->tile0[kFusedWaves];
    // the owner of pooled tile gt: which of the eight rays, which of its tiles
    auto owner_of = [&](int gt, OwnerRay& o, int& ow, int& t) {
      ow = 0;
#pragma unroll
      for (int w = 1; w < kFusedWaves; ++w) ow += (gt >= tabs->tile0[w]) ? 1 : 0;
      t = gt - tabs->tile0[ow];
      load_ray(D, A.rays_o, A.rays_d, nullptr, nullptr, tabs->ray[ow], o.r);
      o.vd[0] = o.r.d[0];
      o.vd[1] = o.r.d[1];
      o.vd[2] = o.r.d[2];
      if (D.ndc) {  // the view direction the MLP sees is normalised for NDC rays only (batBase.py:62-66)
        o.vd[0] /= o.r.norm;
        o.vd[1] /= o.r.norm;
        o.vd[2] /= o.r.norm;
      }
      o.sc = sc0 + (size_t)ow * A.slot_floats;
      o.sidx = sidx_all + (size_t)ow * Sp;
      o.nsh = tabs->nsh[ow];
    };
    // ================= B: shade forward (pooled tiles) =================
    for (;;) {
      int gt = 0;
      if (lane == 0) gt = atomicAdd(&tabs->queue[0], 1);
      gt = __builtin_amdgcn_readfirstlane(gt);
      if (gt >= pool) break;
      asm volatile("" : "+v"(j), "+v"(h));  // per tile: keeps per-lane LDS addresses from becoming loop invariants
      OwnerRay o;
      int ow, t;
      owner_of(gt, o, ow, t);
      float comp[3];
      fused_tile_fwd<C>(D, A, pm, smem, o, t, j, h, comp);
#pragma unroll
      for (int c = 0; c < 3; ++c) comp[c] = wave_sum(comp[c]);
      if (lane == 0) {
        float* pp = o.sc + part_off + t * 8;
        pp[0] = comp[0];
        pp[1] = comp[1];
        pp[2] = comp[2];
      }
    }
    __threadfence_block();
    __syncthreads();
    if (threadIdx.x == 0) tabs->queue[0] = 0;  // everybody has left the loop above; next use is a barrier away
    // ================= C: composite, loss, d loss / d rgb, reverse scan; D1: density path (own ray) =================
    float g[3] = {0.f, 0.f, 0.f};
    if (active) {
      Ray r;
      load_ray(D, A.rays_o, A.rays_d, nullptr, nullptr, ray, r);
      const float acc = tabs->acc[wv];
      const int ntile = (nsh + 31) >> 5;
      {
        float sq = 0.f;
        const int view = ray / A.rays_per_view;
        const long pix = A.ray_idx[ray - view * A.rays_per_view];
        const float bg = D.white_bg ? (1.f - acc) : 0.f;
        float out[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float v = 0.f;
          for (int t = 0; t < ntile; ++t) v += part[t * 8 + c];   // tile order: deterministic
          v += bg;
          const float cl = fminf(fmaxf(v, 0.f), 1.f);
          const float gt = A.image[((size_t)view * 3 + c) * A.HW + pix];
          const float df = cl - gt;
          sq += df * df;
          g[c] = (v >= 0.f && v <= 1.f) ? 2.f * df * A.loss_scale : 0.f;  // clamp passes the gradient on [0, 1]
          out[c] = cl;
        }
        if (lane == 0) {
          A.rgb[ray * 3 + 0] = out[0];
          A.rgb[ray * 3 + 1] = out[1];
          A.rgb[ray * 3 + 2] = out[2];
          A.sqerr[ray] = sq;
          sq_wave += sq;
          tabs->g[wv][0] = g[0];
          tabs->g[wv][1] = g[1];
          tabs->g[wv][2] = g[2];
        }
      }
      const float bgsum = D.white_bg ? (g[0] + g[1] + g[2]) : 0.f;
      float suffix = 0.f, gnorm = 0.f;
      for (int cch = ((A.ablate & 8) ? 0 : (S + 63) / 64) - 1; cch >= 0; --cch) {
        const int i = cch * 64 + lane;
        const bool live = i < S;
        float alpha = 0.f, T = 0.f, Gw = 0.f, delta = 0.f, feat = 0.f;
        bool valid = false;
        if (live) {
          alpha = a_alpha[i];
          T = a_T[i];
          feat = a_feat[i];
          valid = a_valid[i] != 0.f;
          const float z0 = sample_z(D, r, A.zvals, i);
          if (i < S - 1) delta = (sample_z(D, r, A.zvals, i + 1) - z0) * r.norm;
          Gw = -bgsum;
          const int rk = a_rank[i];
          if (rk >= 0 && nsh > 0) {
            const float* tb = tiles + (size_t)(rk >> 5) * kTileRows * 32 + (rk & 31);
            Gw += g[0] * tb[(kRowRgb + 0) * 32] + g[1] * tb[(kRowRgb + 1) * 32] + g[2] * tb[(kRowRgb + 2) * 32];
          }
        }
        const float v = live ? Gw * (alpha * T) : 0.f;
        float inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const float tt = __shfl_down(inc, o);
          if (lane + o < 64) inc += tt;
        }
        float excl = __shfl_down(inc, 1);
        if (lane == 63) excl = 0.f;
        const float after = excl + suffix;
        suffix += __shfl(inc, 0);
        const float f1 = 1.f - alpha + 1e-10f;
        const float g_alpha = Gw * T - after / f1;
        const float one_m = 1.f - alpha;
        const float dsc = delta * D.dist_scale;
        const float x = feat + D.shift;
        const float g_feat = valid ? g_alpha * dsc * one_m * density_act_grad(D.act, x) : 0.f;
        if (D.ndc && valid && live) {
          const float sigma = density_act(D.act, x);
          const float dz = (r.norm > 0.f) ? delta / r.norm : 0.f;
          gnorm += g_alpha * sigma * dz * D.dist_scale * one_m;
        }
        // D1, in the same pass: the density path's position gradient of this sample
        float gpo[3] = {0.f, 0.f, 0.f}, gpd[3] = {0.f, 0.f, 0.f};
        if (g_feat != 0.f && !(A.ablate & 2)) {
          const float z = sample_z(D, r, A.zvals, i);
          float p[3], n[3], gn[3];
          sample_point(D, r, z, p);
          normalize(D, p, n);
          density_feature_grad(D, n, gn);
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            gpo[a] = g_feat * gn[a] * D.inv[a];
            gpd[a] = gpo[a] * z;
          }
        }
        // the chunk's sums go to the partial row of the chunk (summed in chunk order at the end)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          gpo[a] = wave_sum(gpo[a]);
          gpd[a] = wave_sum(gpd[a]);
        }
        if (lane == 0) {
          float* pp = part + ((Sp / 32) + cch) * 8;   // rows behind the tile rows (slot layout: fused_slot_floats)
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            pp[a] = gpo[a];
            pp[3 + a] = gpd[a];
          }
        }
      }
      gnorm = wave_sum(gnorm);
      if (lane == 0) tabs->g[wv][3] = gnorm;
    }
    __threadfence_block();
    __syncthreads();
    // ================= D2: appearance backward (pooled tiles) =================
    for (;;) {
      int gt = 0;
      if (lane == 0) gt = atomicAdd(&tabs->queue[1], 1);
      gt = __builtin_amdgcn_readfirstlane(gt);
      if (gt >= ((A.ablate & 4) ? 0 : pool)) break;
      asm volatile("" : "+v"(j), "+v"(h));
      OwnerRay o;
      int ow, t;
      owner_of(gt, o, ow, t);
      const float gown[3] = {tabs->g[ow][0], tabs->g[ow][1], tabs->g[ow][2]};
      float go[3], gd[3];
      fused_tile_bwd<C>(D, A, pm, smem, stash, o, gown, t, j, h, lane, go, gd);
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        go[a] = wave_sum(go[a]);
        gd[a] = wave_sum(gd[a]);
      }
      if (lane == 0) {
        float* pp = o.sc + part_off + t * 8;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          pp[a] = go[a];
          pp[3 + a] = gd[a];
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the next tile overwrites the stash
    }
    __threadfence_block();
    __syncthreads();
    if (threadIdx.x == 0) tabs->queue[1] = 0;
    // ================= the ray's gradient: tiles in order, then chunks in order =================
    if (active && lane == 0) {
      Ray r;
      load_ray(D, A.rays_o, A.rays_d, nullptr, nullptr, ray, r);
      float s6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const int ntile = (A.ablate & 4) ? 0 : (nsh + 31) >> 5;
      for (int t = 0; t < ntile; ++t)
#pragma unroll
        for (int a = 0; a < 6; ++a) s6[a] += part[t * 8 + a];
      const int nch = (A.ablate & 8) ? 0 : (S + 63) / 64;
      for (int c = 0; c < nch; ++c)
#pragma unroll
        for (int a = 0; a < 6; ++a) s6[a] += part[((Sp / 32) + c) * 8 + a];
      const float gnorm = tabs->g[wv][3];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        float gdir = s6[3 + a];
        if (D.ndc && r.norm > 0.f) gdir += gnorm * r.d[a] / r.norm;
        A.g_rays_o[ray * 3 + a] = s6[a];
        A.g_rays_d[ray * 3 + a] = gdir;
      }
    }
    __syncthreads();  // tables, lists and scratch slots are reused by the next round
  }
  // ---- the loss value: one atomic per workgroup; the workgroup that arrives last publishes the sum and leaves the
  //      accumulator and the counter zeroed for the next launch (all accesses are device-scope atomics)
  __syncthreads();
  if (lane == 0) stash[0] = sq_wave;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int w = 0; w < kFusedWaves; ++w) s += smem[C::LDS_FLOATS + w * WAVE_FLOATS];
    atomicAdd(A.loss_acc, s * A.loss_scale);
    __threadfence();
    const unsigned arrived = atomicAdd(A.loss_cnt, 1u);
    if (arrived == gridDim.x - 1) {
      A.loss[0] = atomicExch(A.loss_acc, 0.f);
      atomicExch(A.loss_cnt, 0u);
    }
  }
}

}  // namespace jt

using namespace jt;

typedef ShadeCfg<48, 27, 64, JT_MLP_FEA> CfgBlender;
typedef ShadeCfg<20, 20, 32, JT_MLP_WEAKVIEW> CfgLlff;

static int fused_kind(const JtScene* s) {
  if (!s) return -1;
  if (s->view_pe != 2 || s->fea_pe != 2) return -1;
  if (s->n_comp_app == 48 && s->app_dim == 27 && s->mlp_hidden == 64 && s->mlp_kind == JT_MLP_FEA) return 0;
  if (s->n_comp_app == 20 && s->app_dim == 20 && s->mlp_hidden == 32 && s->mlp_kind == JT_MLP_WEAKVIEW) return 1;
  return -1;
}

static const int kFusedBlocks = 256;  // one workgroup of eight waves per CU: 2 048 ray slots

static const long kFusedHeadFloats = 64;  // loss accumulator + arrival counter in front of the slots (zero before first use)

static long fused_slot_floats(int S) {
  const long Sp = (S + 63) & ~63;
  // six per-sample arrays | tile rows | partial sums: one row of 8 per tile, then one per 64-sample chunk
  return 6 * Sp + (Sp / 32) * (long)kTileRows * 32 + (Sp / 32 + Sp / 64) * 8;
}

extern "C" size_t jt_pose_fused_workspace_bytes(const JtScene* scene) {
  if (fused_kind(scene) < 0 || scene->n_samples < 1) return 0;
  return ((size_t)fused_slot_floats(scene->n_samples) * kFusedBlocks * kFusedWaves + kFusedHeadFloats) * sizeof(float);
}

template <class C>
static int launch_fused(const Dev& D, const MlpDev& M, const PeMask& pm, const FusedArgs& A, hipStream_t st) {
  const size_t lds = (size_t)C::LDS_FLOATS * sizeof(float) + (size_t)kFusedWaves * 2 * 16 * 64 * sizeof(float) +
                     (size_t)kFusedWaves * A.Sp * sizeof(uint16_t) + sizeof(FusedTabs);
  if (lds > 160 * 1024) return JT_ERR_UNSUPPORTED;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_pose_fused<C>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
  const int blocks = std::min(kFusedBlocks, (A.R + kFusedWaves - 1) / kFusedWaves);
  hipLaunchKernelGGL((k_pose_fused<C>), dim3(blocks), dim3(64 * kFusedWaves), lds, st, D, M, pm, A);
  JT_LAUNCH_CHECK();
  return JT_OK;
}

extern "C" int jt_pose_fused(const JtScene* scene, const JtFactors* factors, const JtMlp* mlp, const float* rays_o,
                             const float* rays_d, const float* zvals, int n_rays, const float* image,
                             const int64_t* ray_idx, int rays_per_view, int image_pixels, float loss_scale, float* rgb,
                             float* depth, float* opacity, float* sqerr, float* loss, float* g_rays_o, float* g_rays_d,
                             void* workspace, size_t workspace_bytes, void* stream) {
  Dev D;
  int rc = make_dev(scene, factors, &D);
  if (rc) return rc;
  if (!factors || !mlp || !rays_o || !rays_d || !image || !ray_idx || !rgb || !depth || !opacity || !sqerr || !loss ||
      !g_rays_o || !g_rays_d || !workspace || n_rays < 1 || rays_per_view < 1 || image_pixels < 1 ||
      n_rays % rays_per_view != 0)
    return JT_ERR_ARG;
  if (!mlp->basis || !mlp->w1 || !mlp->b1 || !mlp->w2 || !mlp->b2 || !mlp->w3 || !mlp->b3) return JT_ERR_ARG;
  for (int a = 0; a < 3; ++a)
    if (!D.dP[a] || !D.dL[a] || !D.aP[a] || !D.aL[a]) return JT_ERR_ARG;
  if (D.ndc && !zvals) return JT_ERR_ARG;
  if (D.Cd < 4 || (D.Cd % 4) != 0) return JT_ERR_UNSUPPORTED;
  const int kind = fused_kind(scene);
  if (kind < 0) return JT_ERR_UNSUPPORTED;
  if (workspace_bytes < jt_pose_fused_workspace_bytes(scene)) return JT_ERR_ARG;
  MlpDev M = {mlp->basis, mlp->w1, mlp->b1, mlp->w2, mlp->b2, mlp->w3, mlp->b3};
  PeMask pm = pe_masks(scene->fea_pe_progress, scene->view_pe_progress, scene->fea_pe, scene->view_pe);
  FusedArgs A;
  A.rays_o = rays_o;
  A.rays_d = rays_d;
  A.zvals = zvals;
  A.image = image;
  A.ray_idx = reinterpret_cast<const long*>(ray_idx);
  A.R = n_rays;
  A.rays_per_view = rays_per_view;
  A.HW = image_pixels;
  A.loss_scale = loss_scale;
  A.rgb = rgb;
  A.depth = depth;
  A.opacity = opacity;
  A.sqerr = sqerr;
  A.loss = loss;
  A.g_rays_o = g_rays_o;
  A.g_rays_d = g_rays_d;
  A.loss_acc = reinterpret_cast<float*>(workspace);
  A.loss_cnt = reinterpret_cast<unsigned*>(workspace) + 1;
  A.scratch = reinterpret_cast<float*>(workspace) + kFusedHeadFloats;
  A.slot_floats = fused_slot_floats(D.S);
  A.Sp = (D.S + 63) & ~63;
  const char* abl = getenv("JT_FUSED_ABLATE");  // profiling / diagnostics only
  A.ablate = abl ? atoi(abl) : 0;
  hipStream_t st = (hipStream_t)stream;
  return kind == 0 ? launch_fused<CfgBlender>(D, M, pm, A, st) : launch_fused<CfgLlff>(D, M, pm, A, st);
}
