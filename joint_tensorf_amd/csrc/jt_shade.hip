// Fused appearance path on the CDNA4 matrix cores (fp32-input MFMA, exact f32 numerics):
//   gather(app planes/lines) -> plane*line products -> basis_mat -> positional encoding -> MLP -> sigmoid
// for every shaded sample, without materialising the [n][3*Ca] product matrix.
// Replaces compute_appfeature (bateRF.py:97-130), basis_mat (tensoRF.py:156),
// MLPRender_Fea.forward (tensorBase.py:116-126) / MLPRender_Fea_WeakView.forward (:198-214).
//
// Tiling.  One wavefront owns a tile of 32 consecutive shaded samples.  Every product of the chain is
// computed TRANSPOSED, D[unit][sample] = W[unit][k] * X[k][sample], with v_mfma_f32_32x32x2_f32:
// the sample index sits on the lane (lane & 31) for the B operand and for the accumulator, so the
// accumulator of one layer (16 registers = 16 units of the lane's sample) is directly the B operand
// of the next layer -- no LDS round trip between layers.  The two lane halves (lane >> 5) supply the
// two k-slices of each MFMA step; which unit a half supplies in a step is fixed by the accumulator
// map  row(r, h) = (r & 3) + 8 (r >> 2) + 4 h, and the A operand (weights, read from LDS with an odd
// row stride => conflict-free) simply reads the matching column.  The gather is split the same way:
// half h loads the channel quads q with q % 2 == h as 16-byte vectors.
#include <atomic>
#include <cstdlib>

#include "jt_common.h"
#include "jt_walk.h"

#include "jt_shade_core.h"

#ifndef JT_B16_PINGPONG
#define JT_B16_PINGPONG 1
#endif
#ifndef JT_BF16X3_DEFAULT
#define JT_BF16X3_DEFAULT 7
#endif

namespace jt {

// REC = 1 (training): besides rgb the kernel leaves the tile-blocked records of the layer inputs (see BwdCfg) so
// that the backward does not have to gather and run the forward chain a second time.  REC = 2: only what a
// pose-only backward reads (basis output, ReLU sign words, sample coordinates).  REC = 3 (training, "lean tape", round 6):
// as 1 without the 3 Ca product rows -- dBasis is then formed inside k_shade_scatter from the plane x line products the
// walkers hold anyway, and a tile's record block is `rrows` = R_LEAN rows instead of REC_FLOATS (1 088 instead of 1 920 bytes
// per shaded sample for VM-48).  `rrows` is a launch argument in every kernel that addresses records.
template <class C, int REC>
__global__ __launch_bounds__(256, 2) void k_shade_fwd(Dev D, MlpDev M, PeMask pm, const float* __restrict__ rays_o,
                                                      const float* __restrict__ rays_d,
                                                      const float* __restrict__ jitter,
                                                      const float* __restrict__ zvals,
                                                      const float* __restrict__ tmin,
                                                      const int* __restrict__ offset, int R,
                                                      const int* __restrict__ eray, const int* __restrict__ esmp,
                                                      const float* __restrict__ vdir, float* __restrict__ rgb_s,
                                                      float* __restrict__ rec, int cap, int rrows) {
  typedef BwdCfg<C> B;
  extern __shared__ __align__(16) float smem[];
  const int total = min(offset[R], cap);
  const int ntiles = (total + 31) >> 5;
  const int nblk = min((int)gridDim.x, (ntiles + 3) / 4);  // the grid is sized for the worst case
  if ((int)blockIdx.x >= nblk) return;
  load_weights_lds<C>(smem, M);
  __syncthreads();
  // (the wave index as a scalar: everything derived from it -- tile, record block -- stays in scalar registers)
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j_ = lane & 31, h_ = lane >> 5;
  constexpr int HOFF = (C::KIND == JT_MLP_FEA) ? 0 : 12;
  const XcdShare xs = xcd_share(ntiles, nblk);  // tiles of neighbouring samples meet in one XCD's L2
  for (int tile = xs.lo + xs.rank * 4 + wv; tile < xs.hi; tile += xs.peers * 4) {
    int j = j_, h = h_;  // see k_shade_bwd: keeps per-lane address math from being hoisted out of the loop
    asm volatile("" : "+v"(j), "+v"(h));
    const int e = tile * 32 + j;
    const bool on = e < total;
    const int ee = on ? e : total - 1;
    float* rt = REC ? rec + (size_t)tile * (size_t)rrows * 32 : nullptr;
    EntryGeom g = entry_geom(D, rays_o, rays_d, jitter, zvals, tmin, eray, esmp, ee);
    float vd[3] = {vdir[(size_t)ee * 3], vdir[(size_t)ee * 3 + 1], vdir[(size_t)ee * 3 + 2]};
    if (REC && on && h == 0) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        if (REC & 1) rec_st(rec_at(rt, B::R_VD + c, 4u * (unsigned)j), vd[c]);
        rec_st(rec_at(rt, B::R_GEO + c, 4u * (unsigned)j), g.n[c]);
      }
    }
    f32x16 facc = gather_basis<C, REC == 1>(D, smem, g.n, j, h, rt, on);
    if (REC) rec_store<1>(rt, B::R_F, &facc, j, h, on);
    Hidden<C> h1 = layer1<C>(smem, facc, vd, pm, j, h);
    relu_<C>(h1);
    if (REC) {
      unsigned mask1 = 0u;
#pragma unroll
      for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) mask1 |= (h1.v[mt][r] > 0.f) ? (1u << (mt * 16 + r)) : 0u;
      if (on) rec_st(rec_at(rt, B::R_MASK, 4u * (unsigned)j + 128u * (unsigned)h), __uint_as_float(mask1));
      if (REC & 1) rec_store<C::MT>(rt, B::R_H1, h1.v, j, h, on);
    }
    Hidden<C> h2 = layer2<C>(smem, h1, j, h);
    relu_<C>(h2);
    if (REC) {
      unsigned mask2 = 0u;
#pragma unroll
      for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) mask2 |= (h2.v[mt][r] > 0.f) ? (1u << (mt * 16 + r)) : 0u;
      if (on) rec_st(rec_at(rt, B::R_MASK + 2, 4u * (unsigned)j + 128u * (unsigned)h), __uint_as_float(mask2));
      if (REC & 1) rec_store<C::MT>(rt, B::R_MID + HOFF, h2.v, j, h, on);
      if ((REC & 1) && C::KIND != JT_MLP_FEA) {
        float pe[12];
        view_pe(vd, pm, pe);
        if (on && h == 0) {
#pragma unroll
          for (int k = 0; k < 12; ++k) rec_st(rec_at(rt, B::R_MID + k, 4u * (unsigned)j), pe[k]);
        }
      }
    }
    float o[3];
    layer3<C>(smem, h2, vd, pm, h, o);
    if (on && h == 0) {
#pragma unroll
      for (int c = 0; c < 3; ++c) rgb_s[(size_t)e * 3 + c] = 1.f / (1.f + expf(-o[c]));
    }
  }
}


#if JT_STAMP
// profiling build only (tools/build_variant.py -DJT_STAMP=1): cycles a wave of k_shade_fwd_b16 spends per phase of a tile,
// summed over all waves and tiles; read and cleared by jt_debug_read_stamps (tools/stamp_fwd.py)
__device__ unsigned long long g_stamps[8];
#define JT_STAMP_T() (__builtin_readcyclecounter())
#endif

// The same forward with the three matrix stages on the bf16 matrix cores at fp32-level accuracy (jt_shade_core.h, "bf16x3"):
// one workgroup of eight waves per CU around a 114 KB pre-split weight image.  Selected by JT_BF16X3 (launch_shade_fwd_t).
template <class C, int REC>
__global__ __launch_bounds__(JT_B16_THREADS) void k_shade_fwd_b16(Dev D, MlpDev M, PeMask pm, const float* __restrict__ rays_o,
                                                      const float* __restrict__ rays_d,
                                                      const float* __restrict__ jitter,
                                                      const float* __restrict__ zvals,
                                                      const float* __restrict__ tmin,
                                                      const int* __restrict__ offset, int R,
                                                      const int* __restrict__ eray, const int* __restrict__ esmp,
                                                      const float* __restrict__ vdir, float* __restrict__ rgb_s,
                                                      float* __restrict__ rec, int cap, int rrows) {
  typedef BwdCfg<C> B;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  typedef B16Cfg<C> Q;
  const uint4* img = reinterpret_cast<const uint4*>(smem_raw);
  const float* tail = reinterpret_cast<const float*>(smem_raw + (size_t)Q::V_END * 16);
  const float* smem = tail - C::O_W3;  // layer3 reads W3 / b3 at their fp32-image offsets: the tail keeps that relative layout
  const int total = min(offset[R], cap);
  const int ntiles = (total + 31) >> 5;
  constexpr int NW = JT_B16_THREADS / 64;
  const int nblk = min((int)gridDim.x, (ntiles + NW - 1) / NW);  // the grid is sized for the worst case
  if ((int)blockIdx.x >= nblk) return;
  load_weights_lds_b16<C>(smem_raw, M);
  __syncthreads();
  // (the wave index as a scalar: everything derived from it -- tile, record block -- stays in scalar registers)
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j_ = lane & 31, h_ = lane >> 5;
  constexpr int HOFF = (C::KIND == JT_MLP_FEA) ? 0 : 12;
  const XcdShare xs = xcd_share(ntiles, nblk);  // tiles of neighbouring samples meet in one XCD's L2
#if JT_SETPRIO
  if (wv >= NW / 2) __builtin_amdgcn_s_setprio(1);  // static priority for the later-dispatched wave of every SIMD: measured, no effect
#endif
  // A tile is a GATHER step (taps, products, basis product: half of a wave's time, most of it waiting for the texture path)
  // and a COMPUTE step (encodings, layers, records: vector and matrix instructions).  JT_B16_PINGPONG (inference only, below):
  // the two waves of a SIMD (w and w + NW / 2) take the steps in opposite order with a workgroup barrier between steps, so
  // that one of them computes while the other gathers instead of both queueing for the same unit.
  const int stride = xs.peers * NW;
  const int first = xs.lo + xs.rank * NW;
  const int iters0 = first < xs.hi ? (xs.hi - first + stride - 1) / stride : 0;  // tiles of the workgroup's wave 0: uniform
  f32x16 facc;
  float vd0 = 0.f, vd1 = 0.f, vd2 = 0.f;  // (scalars: an array captured by reference below ends up in scratch memory)
  int e = 0;
  bool on = false, have = false;
  float* rt = nullptr;
#pragma unroll
  for (int r = 0; r < 16; ++r) facc[r] = 0.f;
  // (the two steps as macros, not lambdas: captured by reference, the 20-channel instantiation kept 192 bytes of its state in
  //  scratch memory and its forward went from 0.117 to 0.144 ms)
#define JT_FWD_GATHER_STEP \
  {                                                                                                                    \
    int j = j_, h = h_; \
    asm volatile("" : "+v"(j), "+v"(h)); \
    e = tile * 32 + j; \
    on = e < total; \
    const int ee = on ? e : total - 1; \
    rt = REC ? rec + (size_t)tile * (size_t)rrows * 32 : nullptr; \
    EntryGeom g = entry_geom(D, rays_o, rays_d, jitter, zvals, tmin, eray, esmp, ee); \
    vd0 = vdir[(size_t)ee * 3], vd1 = vdir[(size_t)ee * 3 + 1], vd2 = vdir[(size_t)ee * 3 + 2]; \
    const float vd[3] = {vd0, vd1, vd2}; \
    if (REC && on && h == 0) { \
_Pragma("unroll") \
      for (int c = 0; c < 3; ++c) { \
        if (REC & 1) rec_st(rec_at(rt, B::R_VD + c, 4u * (unsigned)j), vd[c]); \
        rec_st(rec_at(rt, B::R_GEO + c, 4u * (unsigned)j), g.n[c]); \
      } \
    } \
    facc = gather_basis_b16<C, REC == 1>(D, img, g.n, j, h, lane, rt, on); \
    if (REC) rec_store<1>(rt, B::R_F, &facc, j, h, on); \
  }
#define JT_FWD_COMPUTE_STEP \
  {                                                                                                                    \
    int j = j_, h = h_; \
    asm volatile("" : "+v"(j), "+v"(h)); \
    const float vd[3] = {vd0, vd1, vd2}; \
    Hidden<C> h1 = layer1_b16<C>(img, tail, facc, vd, pm, h, lane); \
    relu_<C>(h1); \
    if (REC) { \
      unsigned mask1 = 0u; \
_Pragma("unroll") \
      for (int mt = 0; mt < C::MT; ++mt) \
_Pragma("unroll") \
        for (int r = 0; r < 16; ++r) mask1 |= (h1.v[mt][r] > 0.f) ? (1u << (mt * 16 + r)) : 0u; \
      if (on) rec_st(rec_at(rt, B::R_MASK, 4u * (unsigned)j + 128u * (unsigned)h), __uint_as_float(mask1)); \
      if (REC & 1) rec_store<C::MT>(rt, B::R_H1, h1.v, j, h, on); \
    } \
    Hidden<C> h2 = layer2_b16<C>(img, tail, h1, h, lane); \
    relu_<C>(h2); \
    if (REC) { \
      unsigned mask2 = 0u; \
_Pragma("unroll") \
      for (int mt = 0; mt < C::MT; ++mt) \
_Pragma("unroll") \
        for (int r = 0; r < 16; ++r) mask2 |= (h2.v[mt][r] > 0.f) ? (1u << (mt * 16 + r)) : 0u; \
      if (on) rec_st(rec_at(rt, B::R_MASK + 2, 4u * (unsigned)j + 128u * (unsigned)h), __uint_as_float(mask2)); \
      if (REC & 1) rec_store<C::MT>(rt, B::R_MID + HOFF, h2.v, j, h, on); \
      if ((REC & 1) && C::KIND != JT_MLP_FEA) { \
        float pe[12]; \
        view_pe(vd, pm, pe); \
        if (on && h == 0) { \
_Pragma("unroll") \
          for (int k = 0; k < 12; ++k) rec_st(rec_at(rt, B::R_MID + k, 4u * (unsigned)j), pe[k]); \
        } \
      } \
    } \
    float o[3]; \
    layer3<C>(smem, h2, vd, pm, h, o); \
    if (on && h == 0) { \
_Pragma("unroll") \
      for (int c = 0; c < 3; ++c) rgb_s[(size_t)e * 3 + c] = 1.f / (1.f + expf(-o[c])); \
    } \
  }
  // in lock step only where it pays: the inference forward of the 48-channel scene (800 x 800 eval render 168 -> 157 ms); the
  // training forward is unchanged by it (0.485 / 0.479 ms) and the 20-channel one, whose gather step is short, loses
  // (0.117 -> 0.153 ms)
  if (JT_B16_PINGPONG && REC == 0 && C::CA >= 48) {
    const int half = (wv >= NW / 2) ? 1 : 0;
    for (int step = 0; step < 2 * iters0 + 1; ++step) {
      if ((step & 1) == half) {
        const int tile = first + wv + ((step - half) >> 1) * stride;
        have = tile < xs.hi;
        if (have) JT_FWD_GATHER_STEP
      } else if (have) {
        JT_FWD_COMPUTE_STEP
        have = false;
      }
      __syncthreads();
    }
  } else {
    for (int tile = first + wv; tile < xs.hi; tile += stride) {
      JT_FWD_GATHER_STEP
      JT_FWD_COMPUTE_STEP
    }
  }
#undef JT_FWD_GATHER_STEP
#undef JT_FWD_COMPUTE_STEP
}


// =================================================================================================
// backward
// =================================================================================================
// Per tile of 32 samples a wave (1) reads back what the training forward left in the tile's records (basis
// output, ReLU sign words, sample coordinates), (2) walks the MLP backwards with the same
// transposed MFMA chain (the gradient w.r.t. a layer's input comes out in exactly the register layout the
// forward consumed, so the chain rule through ReLU / positional encoding is lane-local), (3) turns the
// feature gradient into per-channel product gradients (basis_mat^T), hands them through a small LDS tile
// to a channel-parallel scatter that walks the samples of the tile in ray order and accumulates the four
// plane corners / two line taps in registers, flushing a texel with ONE float atomic per channel only
// when the walk leaves it (samples are half a voxel apart, so consecutive samples share texels), and
// (4) adds the pre-activation gradients to the tile's records for the weight-gradient kernel k_wgrad,
// which is a skinny GEMM over the sample axis on the same MFMA instruction.

// LDS hand-off between the lanes of ONE wave: the LDS executes a wave's instructions in order, so it is
// enough to stop the compiler from moving LDS accesses across this point and to wait for the LDS queue
// (lgkmcnt); global stores / atomics in flight (vmcnt) are NOT waited for.
__device__ inline void wave_lds_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// ---- channel-parallel scatter of one plane's product gradients ------------------------------------------
// lanes: group = lane >> 4 (4 groups), cl = lane & 15.  Group g walks samples g*8 .. g*8+7 of the tile in
// order (jt_walk.h); the step records of the 32 samples (tap addresses, weights, cells) were computed once per
// sample and sit in LDS; the factor values of the next sample are loaded while the current one is accumulated.
// slot of sample j of a tile in the scatter's LDS arrays: the odd 8-sample runs are stored back to front
__device__ inline int walk_slot(int j) { return j ^ (((j >> 3) & 1) * 7); }

template <class C, bool DET>
__device__ inline void scatter_plane(const Dev& D, const JtFactors& G, int pl, const float* tp, const float* recs,
                                     float* gxyz, int lane, unsigned* bad) {
  constexpr int NCH = (C::CA + 15) / 16;
  const int grp = lane >> 4, cl = lane & 15;
  const float* P = D.aP[pl];
  const float* Ln = D.aL[pl];
  RecWalker<NCH, C::CA, DET ? 1 : 0> wk;  // (instantiated per accumulation mode: float atomics / 2^56 fixed point)
  wk.init(G.app_plane[pl], G.app_line[pl], cl, DET, bad);
  const int m0 = kM0(pl), m1 = kM1(pl), mv = kV(pl);
  // lanes 0..2 of a group add the x / y / line coordinate gradient of the step to gxyz[sample][axis]
  const int my_axis = (cl == 0) ? m0 : (cl == 1) ? m1 : mv;
  const float my_scale = ((cl == 0) ? 0.5f * (float)(D.pw[pl] - 1) : (cl == 1) ? 0.5f * (float)(D.ph[pl] - 1)
                                                                                : 0.5f * (float)(D.ll[pl] - 1)) *
                         D.inv[my_axis];
  // odd groups walk their eight samples in DESCENDING order: groups 0 | 1 and 2 | 3 then end on neighbouring samples
  // (7 | 8, 23 | 24) and merge their last texels before the flush (RecWalker::finish_pair).  The caller stores the
  // tile's samples in WALK order -- slot grp * 8 + q holds sample grp * 8 + (7 - q) for the odd groups (walk_slot) --
  // so every group reads slots q = 0 .. 7 upwards and the offsets below stay instruction immediates
  const float* rec0 = recs + (grp * 8) * kRecWords;
  constexpr int rstep = kRecWords;
  TapBuf<NCH> bufA, bufB;
  auto step = [&](TapBuf<NCH>& tv, int q) {
    const int sidx = grp * 8 + q;
    const float* rec = rec0 + q * rstep;
    wk.advance(rec);
    float g[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) g[k] = wk.live[k] ? tp[(cl + 16 * k) * 33 + sidx] : 0.f;
    float aix = 0.f, aiy = 0.f, ail = 0.f;
    wk.add(tv, rec, g, aix, aiy, ail);
    aix = row16_sum(aix);
    aiy = row16_sum(aiy);
    ail = row16_sum(ail);
    if (cl < 3) atomicAdd(&gxyz[sidx * 4 + my_axis], ((cl == 0) ? aix : (cl == 1) ? aiy : ail) * my_scale);
  };
  // the factor values are fetched two steps ahead of their use (three rotating buffers, steps fully unrolled)
  TapBuf<NCH> bufC;
  wk.load(bufA, P, Ln, rec0);
  wk.load(bufB, P, Ln, rec0 + rstep);
  wk.load(bufC, P, Ln, rec0 + 2 * rstep);
  step(bufA, 0);
  wk.load(bufA, P, Ln, rec0 + 3 * rstep);
  step(bufB, 1);
  wk.load(bufB, P, Ln, rec0 + 4 * rstep);
  step(bufC, 2);
  wk.load(bufC, P, Ln, rec0 + 5 * rstep);
  step(bufA, 3);
  wk.load(bufA, P, Ln, rec0 + 6 * rstep);
  step(bufB, 4);
  wk.load(bufB, P, Ln, rec0 + 7 * rstep);
  step(bufC, 5);
  step(bufA, 6);
  step(bufB, 7);
  wk.finish_pair(grp);
}

// ---- the backward chain on the bf16 matrix cores with three-piece operands (round 5; SPLIT kernels only) ------------------
// G1 = W2^T G2 and, per encoding slot t, gin_t = W1_t^T G1 are products of the same shape as the forward's layers with the
// weight matrices TRANSPOSED: the A operands come from pre-split images [stage][K step][M tile][piece][lane] of 16-byte
// vectors (jt_shade_core.h: split8 / mfma6 / load_b3), the B operands are the gradient registers of the previous stage split
// in place -- G1's split is shared by the five slots.  120 + 24 MT bf16 MFMAs of 32 cycles replace 160 + 32 MT fp32 ones of 64.
// LDS: W2^T 12 MT^2 KB + W1^T 15 MT... (VM-48: 24.6 + 61.4 KB) + the fp32 tail; it fits beside the eight 8 KB sin / cos stashes
// only because the SPLIT kernel has no scatter tiles (the fused kernel: 164.2 KB against 163.8).
template <class C>
struct BwdB16Cfg {
  static constexpr int NSK = 2 * C::MT;                      // K steps of 16 over the HID hidden units
  static constexpr int V_W2T = 0;                            // [step][M tile of h1 units][piece][lane]
  static constexpr int V_W1T = V_W2T + NSK * C::MT * 3 * 64;  // [slot t][step][piece][lane]  (one M tile: the APP <= 32 features)
  static constexpr int V_END = V_W1T + 5 * NSK * 3 * 64;
  static constexpr int F_W3 = 0;                             // fp32 tail (floats): W3 [IN3][4]
  static constexpr int F_END = F_W3 + C::IN3 * 4;
  static constexpr size_t IMG_BYTES = (size_t)V_END * 16 + (size_t)F_END * 4;
};

template <class C>
__device__ inline void load_bwd_images_b16(unsigned char* smem, const MlpDev& M) {
  typedef BwdB16Cfg<C> Q;
  uint4* img = reinterpret_cast<uint4*>(smem);
  float* tail = reinterpret_cast<float*>(smem + (size_t)Q::V_END * 16);
  const int items = (Q::NSK * C::MT + 5 * Q::NSK) * 64;
  for (int it = threadIdx.x; it < items; it += blockDim.x) {
    const int lane = it & 63, blk = it >> 6, i = lane & 31, h = lane >> 5;
    float w[8];
    if (blk < Q::NSK * C::MT) {   // W2^T: row = h1 unit k, the K values are h2 units in the order the G2 registers hold them
      const int st = blk / C::MT, mt = blk - st * C::MT, k = mt * 32 + i;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int idx = 8 * st + e, mi = idx >> 4, r = idx & 15;
        w[e] = M.w2[(mi * 32 + rowmap(r, 0) + 4 * h) * C::HID + k];
      }
      store_b3(img, Q::V_W2T + blk * 3 * 64, lane, w);
    } else {                      // W1_t^T: row = feature j, the K values are h1 units in the order the G1 registers hold them
      const int b = blk - Q::NSK * C::MT, t = b / Q::NSK, st = b - t * Q::NSK;
      int col = C::IN1;
      if (i < C::APP) {
        if (C::KIND == JT_MLP_FEA) col = (t == 0) ? i : C::APP + 3 + 4 * i + (t - 1);
        else col = (t == 0) ? i : C::APP + 4 * i + (t - 1);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int idx = 8 * st + e, mi = idx >> 4, r = idx & 15;
        w[e] = (col < C::IN1) ? M.w1[(mi * 32 + rowmap(r, 0) + 4 * h) * C::IN1 + col] : 0.f;
      }
      store_b3(img, Q::V_W1T + b * 3 * 64, lane, w);
    }
  }
  for (int i = threadIdx.x; i < C::IN3 * 4; i += blockDim.x) {
    const int k = i >> 2, c = i & 3;
    tail[Q::F_W3 + i] = (c < 3) ? M.w3[c * C::IN3 + k] : 0.f;
  }
}

// SPLIT: the chain only -- the feature gradients leave through the GF record rows (always written then) and the scatter runs
// as its own launch (k_shade_scatter below); the wave's LDS is then just the 8 KB sin / cos stash.
// B16 (with SPLIT): layers 2 and 1 of the chain on the bf16 matrix cores (above).
template <class C, bool DET, bool SPLIT = false, bool B16 = false>
__global__ __launch_bounds__(512, 2) void k_shade_bwd(Dev D, MlpDev M, PeMask pm, JtFactors G,
                                                      const int* __restrict__ offset, int R,
                                                      const float* __restrict__ rgb_s,
                                                      const float* __restrict__ g_rgb_s, float* __restrict__ g_xyz,
                                                      float* __restrict__ rec, int chunk_start, int chunk_cap,
                                                      int cap, int ablate, unsigned* __restrict__ bad, int rrows) {
  typedef BwdCfg<C> B;
  extern __shared__ __align__(16) float smem[];
  const int total = min(offset[R], cap);
  const int n_chunk = min(total - chunk_start, chunk_cap);
  const int ntiles = (n_chunk + 31) >> 5;
  const int nblk = min((int)gridDim.x, (ntiles + B::NWAVE - 1) / B::NWAVE);  // the grid is sized for the worst case
  if ((int)blockIdx.x >= nblk) return;  // chunk beyond the shaded samples: nothing to do
  typedef BwdB16Cfg<C> QB;
  static_assert(!B16 || SPLIT, "the bf16 chain exists for the split backward only");
  const uint4* img = reinterpret_cast<const uint4*>(smem);
  if (B16) load_bwd_images_b16<C>(reinterpret_cast<unsigned char*>(smem), M);
  else load_weights_lds<C>(smem, M);
  __syncthreads();
  // (the wave index as a scalar: everything derived from it -- tile, record block -- stays in scalar registers)
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j_ = lane & 31, h_ = lane >> 5;
  const float* w3tab = B16 ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(smem) + (size_t)QB::V_END * 16)
                           : smem + C::O_W3;
  float* tp = B16 ? reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(smem) + ((QB::IMG_BYTES + 15) & ~(size_t)15)) +
                        wv * B::STASH_FLOATS
                  : smem + C::LDS_FLOATS + wv * (SPLIT ? B::STASH_FLOATS : B::WAVE_FLOATS);
  float* geo = tp + B::TP_ROWS * B::TP_LD;
  float* gxyz = geo + 32 * 4;
  const size_t RC = (size_t)rrows;
  constexpr int HOFF = (C::KIND == JT_MLP_FEA) ? 0 : 12;
  const XcdShare xs = xcd_share(ntiles, nblk);  // tiles of neighbouring samples meet in one XCD's L2
  // (the two waves of a SIMD taking the chain and the scatter of their tiles in opposite order, in lock step by a workgroup
  //  barrier as in the inference forward, was measured here as well: 1.51 ms against 1.33 free-running, LLFF 0.68 / 0.65)
#if JT_SETPRIO
  if (wv >= B::NWAVE / 2) __builtin_amdgcn_s_setprio(1);
#endif
  for (int tile = xs.lo + xs.rank * B::NWAVE + wv; tile < xs.hi; tile += xs.peers * B::NWAVE) {
    // re-materialise the lane indices per tile: otherwise every per-lane LDS address / select that depends
    // on them is hoisted out of the tile loop as a loop invariant and the kernel spills hundreds of VGPRs
    int j = j_, h = h_;
    asm volatile("" : "+v"(j), "+v"(h));
    const int pj = walk_slot(j);             // where this lane's sample sits in the scatter's LDS arrays
    const int l0 = tile * 32;                // first record row of the tile (chunk-local)
    const int e = chunk_start + l0 + j;      // global entry of this lane's sample
    const int nlive = min(32, n_chunk - l0);
    const bool on = j < nlive;
    const bool onrec = on && !(ablate & 2);
    float* rt = rec + (size_t)tile * RC * 32;  // this tile's record block (layer inputs left by the forward)
    const int jj = on ? j : nlive - 1;         // padding lanes mirror the last live sample
    if (!SPLIT && h == 0) {
      geo[j * 4 + 0] = rec_ld(rec_at(rt, B::R_GEO + 0, 4u * (unsigned)jj));
      geo[j * 4 + 1] = rec_ld(rec_at(rt, B::R_GEO + 1, 4u * (unsigned)jj));
      geo[j * 4 + 2] = rec_ld(rec_at(rt, B::R_GEO + 2, 4u * (unsigned)jj));
      gxyz[j * 4 + 0] = gxyz[j * 4 + 1] = gxyz[j * 4 + 2] = 0.f;
    }
    // basis_mat output of the forward (accumulator layout) and the ReLU sign bits of both hidden layers
    f32x16 facc;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      facc[r] = rec_ld(rec_at(rt, B::R_F + rowmap(r, 0), 4u * (unsigned)jj + 512u * (unsigned)h));
    const unsigned mask1 = __float_as_uint(rec_ld(rec_at(rt, B::R_MASK, 4u * (unsigned)jj + 128u * (unsigned)h)));
    const unsigned mask2 = __float_as_uint(rec_ld(rec_at(rt, B::R_MASK + 2, 4u * (unsigned)jj + 128u * (unsigned)h)));
    // sin / cos of the features, parked in the wave's LDS scratch for the layer-1 backward
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (rowmap(r, 0) >= C::APP && rowmap(r, 1) >= C::APP) continue;
      float sn, cs;
      sincos_grad(facc[r], &sn, &cs);
      tp[(2 * r) * 64 + lane] = sn;
      tp[(2 * r + 1) * 64 + lane] = cs;
    }
    // ---- output layer backward (VALU) ----
    float go[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float rgb = rgb_s[(size_t)(on ? e : chunk_start + l0 + nlive - 1) * 3 + c];
      go[c] = on ? g_rgb_s[(size_t)e * 3 + c] * rgb * (1.f - rgb) : 0.f;
    }
    if (onrec && h == 0) {
#pragma unroll
      for (int c = 0; c < 3; ++c) rec_st(rec_at(rt, B::R_GO + c, 4u * (unsigned)j), go[c]);
    }
    Hidden<C> G2;
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = HOFF + mt * 32 + rowmap(r, 0) + 4 * h;
        const float4 w = *reinterpret_cast<const float4*>(w3tab + k * 4);
        float gsum = go[0] * w.x + go[1] * w.y + go[2] * w.z;
        G2.v[mt][r] = ((mask2 >> (mt * 16 + r)) & 1u) ? gsum : 0.f;
      }
    rec_store<C::MT>(rt, B::R_G2, G2.v, j, h, onrec && !(ablate & 32));   // (lean tape: the dW2 GEMM derives G2)
    // ---- layer 2 backward: g_h1[k] = sum_i W2[i][k] G2[i] ; masked by relu(h1) ----
    Hidden<C> G1;
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) G1.v[mt][r] = 0.f;
    if (B16) {
#pragma unroll
      for (int st = 0; st < QB::NSK; ++st) {
        float v8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v8[e] = G2.v[(8 * st + e) >> 4][(8 * st + e) & 15];
        const B3 bb = split8(v8);
#pragma unroll
        for (int mk = 0; mk < C::MT; ++mk)
          G1.v[mk] = mfma6(load_b3(img, QB::V_W2T + (st * C::MT + mk) * 3 * 64, lane), bb, G1.v[mk]);
      }
    } else {
#pragma unroll
    for (int mi = 0; mi < C::MT; ++mi) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int irow = mi * 32 + rowmap(r, 0) + 4 * h;  // unit i this half supplies
        const float bv = G2.v[mi][r];
#pragma unroll
        for (int mk = 0; mk < C::MT; ++mk) {
          float av = smem[C::O_W2 + irow * C::LD2 + mk * 32 + j];
          G1.v[mk] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, G1.v[mk], 0, 0, 0);
        }
      }
    }
    }
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) G1.v[mt][r] = ((mask1 >> (mt * 16 + r)) & 1u) ? G1.v[mt][r] : 0.f;
    rec_store<C::MT>(rt, B::R_G1, G1.v, j, h, onrec);
    // ---- layer 1 backward, one M tile per encoding slot t: row rowmap(r,h) of tile t is the gradient of the
    //      very value this lane fed forward in k-step (r, t)  =>  chain rule through the encoding is lane-local
    f32x16 gf;
    const float* stash = tp;
#pragma unroll
    for (int r = 0; r < 16; ++r) gf[r] = 0.f;
    B3 g1s[B16 ? QB::NSK : 1];   // G1 split once, the B operand of all five slots
    if (B16) {
#pragma unroll
      for (int st = 0; st < QB::NSK; ++st) {
        float v8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v8[e] = G1.v[(8 * st + e) >> 4][(8 * st + e) & 15];
        g1s[st] = split8(v8);
      }
    }
#pragma unroll
    for (int t = 0; t < 5; ++t) {
      // column of W1 for output row m = j of tile t (feature rows only; view-direction inputs are detached)
      int col;
      if (j < C::APP) {
        if (C::KIND == JT_MLP_FEA) col = (t == 0) ? j : C::APP + 3 + 4 * j + (t - 1);
        else col = (t == 0) ? j : C::APP + 4 * j + (t - 1);
      } else {
        col = C::IN1;
      }
      f32x16 gin;
#pragma unroll
      for (int r = 0; r < 16; ++r) gin[r] = 0.f;
      if (B16) {
#pragma unroll
        for (int st = 0; st < QB::NSK; ++st)
          gin = mfma6(load_b3(img, QB::V_W1T + (t * QB::NSK + st) * 3 * 64, lane), g1s[st], gin);
      } else {
#pragma unroll
      for (int mi = 0; mi < C::MT; ++mi) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int irow = mi * 32 + rowmap(r, 0) + 4 * h;
          float av = smem[C::O_W1 + irow * C::LD1 + col];
          gin = __builtin_amdgcn_mfma_f32_32x32x2f32(av, G1.v[mi][r], gin, 0, 0, 0);
        }
      }
      }
      // d/dx of [x, sin x m0, sin 2x m1, cos x m0, cos 2x m1]
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (rowmap(r, 0) >= C::APP && rowmap(r, 1) >= C::APP) continue;
        // (keeping the 30 sin/cos values in registers across the backward chain spilled to scratch memory)
        const float sn = stash[(2 * r) * 64 + lane], cs = stash[(2 * r + 1) * 64 + lane];
        float dv;
        if (t == 0) dv = 1.f;
        else if (t == 1) dv = cs * pm.f0;
        else if (t == 2) dv = 2.f * (1.f - 2.f * sn * sn) * pm.f1;
        else if (t == 3) dv = -sn * pm.f0;
        else dv = -4.f * sn * cs * pm.f1;
        gf[r] += gin[r] * dv;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (rowmap(r, 0) + 4 * h >= C::APP) gf[r] = 0.f;
    rec_store<1>(rt, B::R_GF, &gf, j, h, SPLIT ? on : onrec);
    // ---- basis_mat^T and the scatter, plane by plane ----
#pragma unroll 1
    for (int pl = 0; pl < (SPLIT ? 0 : 3); ++pl) {
      f32x16 gp[B::PT];
#pragma unroll
      for (int T = 0; T < B::PT; ++T) {
#pragma unroll
        for (int r = 0; r < 16; ++r) gp[T][r] = 0.f;
        const int ch = T * 32 + j;
        const int col = (ch < C::CA) ? pl * C::CA + ch : C::NC;  // NC = zero pad column
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int arow = rowmap(r, 0) + 4 * h;
          float av = smem[C::O_BASIS + arow * C::LDB + col];
          gp[T] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, gf[r], gp[T], 0, 0, 0);
        }
      }
      // tp[channel][sample]
#pragma unroll
      for (int T = 0; T < B::PT; ++T)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ch = T * 32 + rowmap(r, 0) + 4 * h;
          if (ch < C::CA) tp[ch * 33 + pj] = gp[T][r];
        }
      // step records of the tile's samples for this plane (one lane per sample), in the rows of tp past
      // the product gradients
      float* recs = tp + 48 * 33;
      if (h == 0) {
        // previous sample of the same 8-sample run in WALK order: runs 1 and 3 of a tile are walked downwards
        const bool down = (j >> 3) & 1;
        const bool has_prev = down ? (j & 7) != 7 : (j & 7) != 0;
        const int jp = has_prev ? (down ? j + 1 : j - 1) : j;
        make_step_rec(geo[j * 4 + kM0(pl)], geo[j * 4 + kM1(pl)], geo[j * 4 + kV(pl)], geo[jp * 4 + kM0(pl)],
                      geo[jp * 4 + kM1(pl)], geo[jp * 4 + kV(pl)], has_prev, D.ph[pl], D.pw[pl], D.ll[pl], C::CA,
                      recs + pj * kRecWords);
      }
      wave_lds_sync();
      if (!(ablate & 1)) scatter_plane<C, DET>(D, G, pl, tp, recs, gxyz, lane, bad);
      wave_lds_sync();
    }
    if (!SPLIT && on && h == 0) {
#pragma unroll
      for (int a = 0; a < 3; ++a) g_xyz[(size_t)e * 3 + a] = gxyz[pj * 4 + a];
    }
    wave_lds_sync();
  }
}

// ---- split backward: the scatter as its own launch -------------------------------------------------------------------
// k_shade_bwd<SPLIT> leaves the feature gradients GF [APP][32 samples] of every tile in the records; this kernel turns them
// into the factor gradients.  A wave takes a BATCH of 4 RUN consecutive shaded samples (RUN = 8: one 32-sample tile, as the
// fused kernel; 16 / 32: two / four tiles); its four 16-lane groups (lane = channel) each walk one run of RUN samples in
// order (jt_walk.h), odd groups downwards so that runs 0 | 1 and 2 | 3 end on neighbouring samples (finish_pair): the longer
// the run, the fewer texels are flushed twice.  The product gradients of a step are not staged anywhere: basis^T GF comes out
// of v_mfma_f32_16x16x4_f32 (A = GF of sixteen samples, four per group; B = basis^T of the plane's channels; D: lane (group,
// channel), register r = the group's step 4 b + r) directly in the registers of the lane that scatters them.  Nothing of the
// MLP chain lives here: 16 KB of basis operands + 3.5 / 7 / 14 KB per wave of step records in LDS, ~130 registers -- the
// latency-bound walk runs at its own occupancy instead of the chain's two waves per SIMD.
template <int RUN>
__device__ inline int slot_sample(int t) {  // walk slot t = group * RUN + step  ->  sample of the batch
  const int g = t / RUN, q = t - g * RUN;
  return g * RUN + ((g & 1) ? RUN - 1 - q : q);
}

template <class C>
struct ScatCfg {
  static constexpr int NCH = (C::CA + 15) / 16;
  static constexpr int KS = (C::APP + 3) / 4;             // K steps of the 16x16x4 product over basis_mat's rows
  static constexpr int BT_FLOATS = NCH * KS * 64;         // basis^T operand image of ONE plane [channel group][K step][lane]
  static constexpr int GT_FLOATS = 32 * 16;               // (dBasis) a wave's GF rows of a block, transposed: [row a][16 samples]
  static constexpr int RED_FLOATS = 32 * C::CA;           // (dBasis) the workgroup's sum of one plane's slice [row a][channel]
  // geo, gxyz, step records (+ the transposition tile)
  static constexpr int wave_floats(int run, bool db) { return 4 * run * (4 + 4 + kRecWords) + (db ? GT_FLOATS : 0); }
  static constexpr int DB_SLAB = 3 * 32 * C::CA;          // floats a workgroup leaves in its dBasis slab [plane][row a][channel]
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

// FLAGS bit 0 (not with DET): the line gradients of the pass's plane are summed in a workgroup-private LDS copy of the line
// (line_floats = the longest line x channels; LDS float atomics) and added to the real gradient once per workgroup and plane --
// a fifth of the scatter's global atomic segments.
// FLAGS bit 1 (round 6, "lean tape"): dBasis is formed HERE.  dBasis[a][pl Ca + ch] = sum over samples of GF[a][s] prod[ch][s],
// and the walker of lane (group, channel) holds prod = (plane value) x (line value) of its group's current sample: per step
// that is the B operand of a v_mfma_f32_16x16x4_f32 whose K axis is the four groups' samples.  The A operand -- GF[a][sample of
// group k] in lane (a, k) -- is the block's GF rows, which loadA already fetched for basis^T GF in the OTHER orientation
// (lane = sample, K = row): they go through a 2 KB LDS tile per wave once per block (seven 4-byte writes, two 16-byte reads per
// lane) instead of being loaded again per step -- round 4 formed dBasis here with two GF loads per step, which queue behind the
// step's flush atomics (a wave's vector-memory operations retire in order), and paid + 0.6 ms for it.  2 NCH accumulator quads
// per lane; a plane's slice is summed over the waves in LDS in a fixed order and left in the workgroup's slab for
// k_dbasis_reduce.  With it the forward does not record the 3 Ca products (576 of 1 920 bytes per sample for VM-48) and the
// dBasis GEMM (k_wgrad_b16<1, NTB, 0>, 184 us alone) is gone.
// The plane loop is OUTSIDE the batch loop (one LDS line at a time): the coordinate gradients of a sample go to g_xyz as
// store (plane 0) / load-add-store (planes 1, 2) by the same lane.
template <class C, bool DET, int RUN, int WAVES, int FLAGS>
__global__ __launch_bounds__(WAVES * 64) void k_shade_scatter(Dev D, MlpDev M, JtFactors G, const int* __restrict__ offset,
                                                              int R, float* __restrict__ g_xyz,
                                                              const float* __restrict__ rec, int chunk_start,
                                                              int chunk_cap, int cap, unsigned* __restrict__ bad,
                                                              int line_floats, int rrows, float* __restrict__ dbslab) {
  typedef BwdCfg<C> B;
  typedef ScatCfg<C> Q;
  constexpr bool LLINE = (FLAGS & 1) && !DET;
  constexpr bool DB = (FLAGS & 2) != 0;
  constexpr int NS = 4 * RUN, NB = RUN / 4, NCH = Q::NCH, KS = Q::KS;
  extern __shared__ __align__(16) float smem[];
  const int total = min(offset[R], cap);
  const int n_chunk = min(total - chunk_start, chunk_cap);
  const int nbatch = (n_chunk + NS - 1) / NS;
  const int nblk = min((int)gridDim.x, (nbatch + WAVES - 1) / WAVES);  // the grid is sized for the worst case
  if ((int)blockIdx.x >= nblk) return;
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* sline = smem + Q::BT_FLOATS;                                    // [line cell][channel] of the pass's plane
  float* red = sline + (LLINE ? line_floats : 0);                        // (DB) [row a][channel]
  float* geo = red + (DB ? Q::RED_FLOATS : 0) + wv * Q::wave_floats(RUN, DB);
  float* gxyz = geo + NS * 4;
  float* recs = gxyz + NS * 4;
  float* gft = recs + NS * kRecWords;                                    // (DB) [32 rows][16 samples]
  if (DB) {  // rows past 4 KS are never written: they must read as zeros
#pragma unroll
    for (int i = 0; i < Q::GT_FLOATS / 64; ++i) gft[lane + 64 * i] = 0.f;
  }
  const size_t RC = (size_t)rrows;
  const XcdShare xs = xcd_share(nbatch, nblk);
#pragma unroll 1
  for (int pl = 0; pl < 3; ++pl) {
    // the plane's basis^T operand image [channel group][K step][lane]: lane (cl, grp) of step k holds basis[4 k + grp][ch]
    for (int it = threadIdx.x; it < Q::BT_FLOATS; it += WAVES * 64) {
      const int ln = it & 63, blk = it >> 6, k = blk % KS, c = blk / KS;
      const int a = 4 * k + (ln >> 4), ch = 16 * c + (ln & 15);
      smem[it] = (a < C::APP && ch < C::CA) ? M.basis[a * C::NC + pl * C::CA + ch] : 0.f;
    }
    if (LLINE)
      for (int i = threadIdx.x; i < D.ll[pl] * C::CA; i += WAVES * 64) sline[i] = 0.f;
    __syncthreads();
    f32x4 dB[DB ? 2 : 1][NCH];
    if (DB) {
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int c = 0; c < NCH; ++c) dB[m][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int batch = xs.lo + xs.rank * WAVES + wv; batch < xs.hi; batch += xs.peers * WAVES) {
      int ln = lane;
      asm volatile("" : "+v"(ln));  // (keeps per-lane address math inside the loop, see k_shade_bwd)
      const int grp = ln >> 4, cl = ln & 15;
      const int L0 = batch * NS;                  // first sample of the batch (chunk-local)
      const int nlive = min(NS, n_chunk - L0);
      // normalised coordinates of the batch's samples, in WALK order (padding slots mirror the last live sample)
#pragma unroll
      for (int ti = 0; ti < (NS + 63) / 64; ++ti) {
        const int t = ln + 64 * ti;
        if (NS % 64 != 0 && t >= NS) continue;
        const int L = L0 + min(slot_sample<RUN>(t), nlive - 1);
        const float* rt = rec + (size_t)(L >> 5) * RC * 32 + (L & 31);
        geo[t * 4 + 0] = rec_ld(rt + (B::R_GEO + 0) * 32);
        geo[t * 4 + 1] = rec_ld(rt + (B::R_GEO + 1) * 32);
        geo[t * 4 + 2] = rec_ld(rt + (B::R_GEO + 2) * 32);
        gxyz[t * 4 + 0] = gxyz[t * 4 + 1] = gxyz[t * 4 + 2] = 0.f;
      }
      // A operand of block b (steps 4 b .. 4 b + 3 of every group): row i = cl is step 4 b + (cl & 3) of group cl >> 2, the
      // lane's K slice is basis row 4 k + grp
      auto loadA = [&](int b, float* av) {
        const int s = slot_sample<RUN>((cl >> 2) * RUN + 4 * b + (cl & 3));
        const bool live = s < nlive;
        const int L = L0 + (live ? s : nlive - 1);
        const float* rt = rec + (size_t)(L >> 5) * RC * 32 + (size_t)(B::R_GF + grp) * 32 + (L & 31);
#pragma unroll
        for (int k = 0; k < KS; ++k) {
          const float v = rec_ld(rt + 4 * k * 32);
          av[k] = live ? v : 0.f;
        }
      };
      wave_lds_sync();
      // step records of the batch's samples for this plane, one lane per walk slot
#pragma unroll
      for (int ti = 0; ti < (NS + 63) / 64; ++ti) {
        const int t = ln + 64 * ti;
        if (NS % 64 != 0 && t >= NS) continue;
        const bool has_prev = (t % RUN) != 0;
        const int tq = has_prev ? t - 1 : t;
        make_step_rec(geo[t * 4 + kM0(pl)], geo[t * 4 + kM1(pl)], geo[t * 4 + kV(pl)], geo[tq * 4 + kM0(pl)],
                      geo[tq * 4 + kM1(pl)], geo[tq * 4 + kV(pl)], has_prev, D.ph[pl], D.pw[pl], D.ll[pl], C::CA,
                      recs + t * kRecWords);
      }
      // (twelve-wave workgroups have 168 registers per lane: there the basis^T operands are read from LDS where they are used,
      //  once per block of four steps, instead of living in 21 registers across the batch)
      constexpr bool BOP_LDS = WAVES == 12 && C::CA >= 48;
      float bop[BOP_LDS ? 1 : NCH][BOP_LDS ? 1 : KS];
      if (!BOP_LDS) {
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
          for (int k = 0; k < KS; ++k) bop[BOP_LDS ? 0 : c][BOP_LDS ? 0 : k] = smem[(c * KS + k) * 64 + ln];
      }
      wave_lds_sync();
      const float* P = D.aP[pl];
      const float* Ln = D.aL[pl];
      RecWalker<NCH, C::CA, DET ? 1 : 0, LLINE> wk;
      wk.init(G.app_plane[pl], LLINE ? sline : G.app_line[pl], cl, DET, bad);
      const int m0 = kM0(pl), m1 = kM1(pl), mv = kV(pl);
      const int my_axis = (cl == 0) ? m0 : (cl == 1) ? m1 : mv;
      const float my_scale = ((cl == 0) ? 0.5f * (float)(D.pw[pl] - 1) : (cl == 1) ? 0.5f * (float)(D.ph[pl] - 1)
                                                                                    : 0.5f * (float)(D.ll[pl] - 1)) *
                             D.inv[my_axis];
      const float* rec0 = recs + (grp * RUN) * kRecWords;
      float* gx0 = gxyz + (grp * RUN) * 4 + my_axis;
      // (DB) gfa[m][r]: GF[cl + 16 m][the sample group grp walks at step r of the block] -- the A operand of the dBasis product
      f32x4 gfa[2];
      auto step = [&](TapBuf<NCH>& tv, int q, const float* g, int r) {
        const float* sr = rec0 + q * kRecWords;
        wk.advance(sr);
        float aix = 0.f, aiy = 0.f, ail = 0.f;
        float prod[NCH];
        wk.template add<DB>(tv, sr, g, aix, aiy, ail, prod);
        if (DB) {
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int c = 0; c < NCH; ++c) dB[m][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(gfa[m][r], prod[c], dB[m][c], 0, 0, 0);
        }
        aix = row16_sum(aix);
        aiy = row16_sum(aiy);
        ail = row16_sum(ail);
        if (cl < 3) atomicAdd(gx0 + q * 4, ((cl == 0) ? aix : (cl == 1) ? aiy : ail) * my_scale);
      };
      TapBuf<NCH> bufA, bufB;
      float av[KS];
      loadA(0, av);
      wk.load(bufA, P, Ln, rec0);
      auto block = [&](int b) {
        f32x4 dv[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          dv[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int k = 0; k < KS; ++k)
            dv[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[k], BOP_LDS ? smem[(c * KS + k) * 64 + ln] : bop[BOP_LDS ? 0 : c][BOP_LDS ? 0 : k],
                                                         dv[c], 0, 0, 0);
        }
        if (DB) {
          // the block's GF rows once more, transposed through the wave's LDS tile: lane (cl, grp) holds GF[4 k + grp][sample cl]
          wave_lds_sync();  // (the previous block's reads are done)
#pragma unroll
          for (int k = 0; k < KS; ++k) gft[(4 * k + grp) * 16 + cl] = av[k];
          wave_lds_sync();
          gfa[0] = *reinterpret_cast<const f32x4*>(gft + cl * 16 + 4 * grp);
          gfa[1] = *reinterpret_cast<const f32x4*>(gft + (cl + 16) * 16 + 4 * grp);
        }
        if (b + 1 < NB) loadA(b + 1, av);  // the next block's rows are in flight while this block is walked
        float g[4][NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          g[0][c] = dv[c].x, g[1][c] = dv[c].y, g[2][c] = dv[c].z, g[3][c] = dv[c].w;
        }
        const int q = 4 * b;
        wk.load(bufB, P, Ln, rec0 + (q + 1) * kRecWords);
        step(bufA, q, g[0], 0);
        wk.load(bufA, P, Ln, rec0 + (q + 2) * kRecWords);
        step(bufB, q + 1, g[1], 1);
        wk.load(bufB, P, Ln, rec0 + (q + 3) * kRecWords);
        step(bufA, q + 2, g[2], 2);
        if (b + 1 < NB) wk.load(bufA, P, Ln, rec0 + (q + 4) * kRecWords);
        step(bufB, q + 3, g[3], 3);
      };
      if (NB <= 2) {
#pragma unroll
        for (int b = 0; b < NB; ++b) block(b);
      } else {
#pragma unroll 1
        for (int b = 0; b < NB; ++b) block(b);
      }
      wk.finish_pair(grp);
      wave_lds_sync();
#pragma unroll
      for (int ti = 0; ti < (NS + 63) / 64; ++ti) {
        const int t = ln + 64 * ti;
        if (NS % 64 != 0 && t >= NS) continue;
        const int s = slot_sample<RUN>(t);
        if (s < nlive) {
          float* o = g_xyz + (size_t)(chunk_start + L0 + s) * 3;
          if (pl == 0) {
            o[0] = gxyz[t * 4 + 0], o[1] = gxyz[t * 4 + 1], o[2] = gxyz[t * 4 + 2];
          } else {
            o[0] += gxyz[t * 4 + 0], o[1] += gxyz[t * 4 + 1], o[2] += gxyz[t * 4 + 2];
          }
        }
      }
      wave_lds_sync();
    }
    __syncthreads();
    if (LLINE) {
      float* gl = G.app_line[pl];
      if (gl != nullptr)
        for (int i = threadIdx.x; i < D.ll[pl] * C::CA; i += WAVES * 64) {
          const float v = sline[i];
          if (v != 0.f) atomicAdd(gl + i, v);
        }
    }
    if (DB) {
      // the plane's dBasis slice: the waves add their accumulators into `red` one after the other (a fixed order), then the
      // workgroup's sum goes to its slab.  dB[m][c][r] is row a = 16 m + 4 grp + r, channel 16 c + cl
      const int grp = lane >> 4, cl = lane & 15;
#pragma unroll 1
      for (int w = 0; w < WAVES; ++w) {
        if (wv == w) {
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int ch = 16 * c + cl;
                if (C::CA % 16 != 0 && ch >= C::CA) continue;
                float* t = red + (16 * m + 4 * grp + r) * C::CA + ch;
                *t = (w == 0) ? dB[m][c][r] : *t + dB[m][c][r];
              }
        }
        __syncthreads();
      }
      float* my = dbslab + ((size_t)blockIdx.x * 3 + pl) * Q::RED_FLOATS;
      for (int i = threadIdx.x; i < Q::RED_FLOATS; i += WAVES * 64) my[i] = red[i];
    }
    __syncthreads();
  }
}

// dBasis += the slabs k_shade_scatter<..., FLAGS & 2> left: [chunk][workgroup][plane][row a][channel].  The workgroups that ran
// are recomputed from the shaded count exactly as the scatter did (grid `sgrid`, batches of `ns` samples, `waves` per
// workgroup); blockIdx.y strides over them, ONE y block = a fixed summation order (JT_DETERMINISTIC).
template <class C>
__global__ __launch_bounds__(256) void k_dbasis_reduce(const float* __restrict__ slabs, size_t chunk_stride, int chunk_entries,
                                                       int sgrid, int ns, int waves, const int* __restrict__ offset, int R,
                                                       int cap, float* __restrict__ dBasis) {
  typedef ScatCfg<C> Q;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= Q::DB_SLAB) return;
  const int pl = idx / Q::RED_FLOATS, a = (idx / C::CA) % 32, ch = idx % C::CA;
  if (a >= C::APP) return;
  const int total = min(offset[R], cap);
  const int nchunks = (total + chunk_entries - 1) / chunk_entries;
  float s0 = 0.f, s1 = 0.f;
  const int gy = gridDim.y;
  for (int c = 0; c < nchunks; ++c) {
    const int n_chunk = min(total - c * chunk_entries, chunk_entries);
    const int nbatch = (n_chunk + ns - 1) / ns;
    const int active = min(sgrid, (nbatch + waves - 1) / waves);
    const float* base = slabs + (size_t)c * chunk_stride + idx;
    int q = blockIdx.y;
    for (; q + gy < active; q += 2 * gy) {
      s0 += base[(size_t)q * Q::DB_SLAB];
      s1 += base[(size_t)(q + gy) * Q::DB_SLAB];
    }
    for (; q < active; q += gy) s0 += base[(size_t)q * Q::DB_SLAB];
  }
  const float sum = s0 + s1;
  if (sum != 0.f) atomicAdd(dBasis + (size_t)a * C::NC + pl * C::CA + ch, sum);
}

// ---- pose-only backward: the coordinate gradients without walkers ---------------------------------------------------------
// Test-time pose optimisation (model/bat.py:265-292) wants nothing but the rays' gradient.  k_shade_bwd<SPLIT> leaves the feature
// gradients GF of every tile in the records; this kernel forms basis^T GF one M tile (32 channels of a plane) at a time and takes
// the coordinate gradient of every plane from its taps GATHERED ONCE MORE in the forward's layout: the product gradients of an M
// tile land in the lane half that gathers those channel quads, so the sum over the channels is lane-local -- no walkers, no step
// records, no scatter tile.  On its own (not behind the chain in one kernel: the chain's 190 registers plus the taps in flight
// spill, jt_fused.hip) it needs 18 KB of LDS for basis_mat and runs at its own occupancy.
// B16 (round 6, with matrix-mode bit 2 as the chain): basis^T GF on the bf16 matrix cores with three-piece operands -- per plane
// and M tile two 16-deep K steps of six v_mfma_f32_32x32x16_bf16 (12 MFMAs of 32 cycles) instead of sixteen fp32 MFMAs of 64;
// the A operands are pre-split images of basis^T [plane][M tile][K step][piece][lane] (36 KB for VM-48), GF is split once per tile.
template <class C, bool B16 = false>
__global__ __launch_bounds__(256, 3) void k_pose_gather(Dev D, MlpDev M, const int* __restrict__ offset, int R,
                                                     float* __restrict__ g_xyz, const float* __restrict__ rec, int chunk_start,
                                                     int chunk_cap, int cap, int rrows) {
  typedef BwdCfg<C> B;
  constexpr int IMG_VECS = 3 * B::PT * 2 * 3 * 64;
  __shared__ __align__(16) float smem[B16 ? IMG_VECS * 4 : 32 * C::LDB];
  const uint4* img = reinterpret_cast<const uint4*>(smem);
  const int total = min(offset[R], cap);
  const int n_chunk = min(total - chunk_start, chunk_cap);
  const int ntiles = (n_chunk + 31) >> 5;
  const int nblk = min((int)gridDim.x, (ntiles + 3) / 4);
  if ((int)blockIdx.x >= nblk) return;
  if (B16) {
    // block (pl, TT, st): lane (i, h) holds basis[a][col] for the eight K values a = rowmap(8 st + e, 0) + 4 h of its half --
    // the order in which split8 of the GF registers 8 st .. 8 st + 7 supplies the B operand; col = channel 32 TT + i of plane pl
    for (int it = threadIdx.x; it < 3 * B::PT * 2 * 64; it += blockDim.x) {
      const int ln = it & 63, blk = it >> 6, i = ln & 31, h = ln >> 5;
      const int st = blk & 1, TT = (blk >> 1) % B::PT, pl = (blk >> 1) / B::PT;
      const int ch = TT * 32 + i;
      float w[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int a = rowmap(8 * st + e, 0) + 4 * h;
        w[e] = (a < C::APP && ch < C::CA) ? M.basis[a * C::NC + pl * C::CA + ch] : 0.f;
      }
      store_b3(reinterpret_cast<uint4*>(smem), blk * 3 * 64, ln, w);
    }
  } else {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int a = wv; a < 32; a += 4) {  // basis [APP][NC] -> [32][LDB], rows >= APP and the pad column zero
      const bool row = a < C::APP;
      for (int c = lane; c < C::LDB; c += 64) smem[C::O_BASIS + a * C::LDB + c] = (row && c < C::NC) ? M.basis[a * C::NC + c] : 0.f;
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const XcdShare xs = xcd_share(ntiles, nblk);
  for (int tile = xs.lo + xs.rank * 4 + wv; tile < xs.hi; tile += xs.peers * 4) {
    int j = lane & 31, h = lane >> 5;
    asm volatile("" : "+v"(j), "+v"(h));
    const int l0 = tile * 32;
    const int e = chunk_start + l0 + j;
    const int nlive = min(32, n_chunk - l0);
    const bool on = j < nlive;
    const int jj = on ? j : nlive - 1;
    const float* rt = rec + (size_t)tile * (size_t)rrows * 32;
    f32x16 gf;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = rec_ld(rec_at(rt, B::R_GF + rowmap(r, 0), 4u * (unsigned)jj + 512u * (unsigned)h));
      gf[r] = on ? v : 0.f;
    }
    // ---- basis^T one M tile (32 channels of a plane) at a time, then the plane's position gradient from its taps ----
    constexpr int PT = B::PT;
    // (no register ARRAYS indexed by the plane: a rolled loop that indexed n[] / the sums dynamically went through scratch
    //  memory and came back wrong for one axis in jt_fused.hip -- the plane's operands are picked with selects)
    const float n0 = rec_ld(rec_at(rt, B::R_GEO + 0, 4u * (unsigned)jj)), n1 = rec_ld(rec_at(rt, B::R_GEO + 1, 4u * (unsigned)jj)),
                n2 = rec_ld(rec_at(rt, B::R_GEO + 2, 4u * (unsigned)jj));
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    B3 gs[B16 ? 2 : 1];   // (B16) GF split once per tile: the B operand of both K steps of every plane and M tile
    if (B16) {
      float v8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v8[e] = gf[e];
      gs[0] = split8(v8);
#pragma unroll
      for (int e = 0; e < 8; ++e) v8[e] = gf[8 + e];
      gs[B16 ? 1 : 0] = split8(v8);
    }
#pragma unroll 1
    for (int pl = 0; pl < 3; ++pl) {
      // matMode / vecMode: plane 0 (x, y | z), plane 1 (x, z | y), plane 2 (y, z | x)
      const float nx = pl == 2 ? n1 : n0, ny = pl == 0 ? n1 : n2, nl = pl == 0 ? n2 : (pl == 1 ? n1 : n0);
      const int PH = pl == 0 ? D.ph[0] : (pl == 1 ? D.ph[1] : D.ph[2]), PW = pl == 0 ? D.pw[0] : (pl == 1 ? D.pw[1] : D.pw[2]),
                LLn = pl == 0 ? D.ll[0] : (pl == 1 ? D.ll[1] : D.ll[2]);
      const PlaneTaps tq = plane_taps(nx, ny, PH, PW, C::CA);
      const Axis l = axis_taps(nl, LLn);
      const float* P = pl == 0 ? D.aP[0] : (pl == 1 ? D.aP[1] : D.aP[2]);
      const float* L = pl == 0 ? D.aL[0] : (pl == 1 ? D.aL[1] : D.aL[2]);
      const unsigned hb = 16u * (unsigned)h;
      const unsigned b00 = 4u * (unsigned)tq.o00 + hb, b10 = 4u * (unsigned)tq.o10 + hb, b01 = 4u * (unsigned)tq.o01 + hb,
                     b11 = 4u * (unsigned)tq.o11 + hb, bl0 = 4u * (unsigned)(l.c0 * C::CA) + hb,
                     bl1 = 4u * (unsigned)(l.c1 * C::CA) + hb;
      const float m00 = tq.ax.m0 * tq.ay.m0, m10 = tq.ax.m1 * tq.ay.m0, m01 = tq.ax.m0 * tq.ay.m1,
                  m11 = tq.ax.m1 * tq.ay.m1;
      const float fx = tq.ax.f, fy = tq.ay.f;
      float sx = 0.f, sy = 0.f, sl = 0.f;
      const bool oor = __builtin_amdgcn_ballot_w64(m00 * m10 * m01 * m11 * l.m0 * l.m1 == 0.f) != 0ull;  // (wave-uniform)
#pragma unroll
      for (int TT = 0; TT < PT; ++TT) {
        f32x16 gpt;
#pragma unroll
        for (int r = 0; r < 16; ++r) gpt[r] = 0.f;
        if (B16) {
#pragma unroll
          for (int st = 0; st < 2; ++st)
            gpt = mfma6(load_b3(img, (((pl * B::PT + TT) * 2) + st) * 3 * 64, lane), gs[B16 ? st : 0], gpt);
        } else {
          const int ch = TT * 32 + j;
          const int col = (ch < C::CA) ? pl * C::CA + ch : C::NC;  // (NC: the zero pad column)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int arow = rowmap(r, 0) + 4 * h;
            const float av = smem[C::O_BASIS + arow * C::LDB + col];
            gpt = __builtin_amdgcn_mfma_f32_32x32x2f32(av, gf[r], gpt, 0, 0, 0);
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
          if (!oor) {
            // every tap inside its factor (all but samples exactly on a far border): the interpolations in their nested form,
            // 16 vector instructions per channel instead of 26
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
              const float gpr = live ? gpt[4 * (m & 3) + kk] : 0.f;
              const float ba = bv[kk] - av[kk], dc = dv[kk] - cv[kk], vu = vv[kk] - uv[kk];
              const float top = av[kk] + fx * ba, bot = cv[kk] + fx * dc;
              const float dpy = bot - top;
              const float pv = top + fy * dpy;
              const float dpx = ba + fy * (dc - ba);
              const float lv = uv[kk] + l.f * vu;
              const float gl = gpr * lv;
              sx += gl * dpx;
              sy += gl * dpy;
              sl += (gpr * pv) * vu;
            }
          } else {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
              // product gradient of channel 8 m + 4 h + kk of this plane: row rowmap(4 (m & 3) + kk, h) of M tile m >> 2
              const float gpr = live ? gpt[4 * (m & 3) + kk] : 0.f;
              const float am = av[kk] * m00, bm = bv[kk] * m10, cm = cv[kk] * m01, dm = dv[kk] * m11;
              const float um = uv[kk] * l.m0, vm = vv[kk] * l.m1;
              const float pv = tq.w00 * av[kk] + tq.w10 * bv[kk] + tq.w01 * cv[kk] + tq.w11 * dv[kk];
              const float lv = l.w0 * uv[kk] + l.w1 * vv[kk];
              const float gl = gpr * lv;
              sx += gl * ((1.f - fy) * (bm - am) + fy * (dm - cm));
              sy += gl * ((1.f - fx) * (cm - am) + fx * (dm - bm));
              sl += gpr * pv * (vm - um);
            }
          }
          if (m & 1) __builtin_amdgcn_sched_barrier(0);
        }
      }
      const float vx = sx * tq.ax.scale, vy = sy * tq.ay.scale, vl = sl * l.scale;
      g0 += pl == 2 ? vl : vx;
      g1 += pl == 0 ? vy : (pl == 1 ? vl : vx);
      g2 += pl == 0 ? vl : vy;
    }
    // the two lane halves hold the sums over their own channel quads of the same sample
    g0 += __shfl_xor(g0, 32), g1 += __shfl_xor(g1, 32), g2 += __shfl_xor(g2, 32);
    if (on && h == 0) {
      float* o = g_xyz + (size_t)e * 3;
      o[0] = g0 * D.inv[0], o[1] = g1 * D.inv[1], o[2] = g2 * D.inv[2];
    }

  }
}

}  // namespace jt (jt_tile.h opens its own)
#include "jt_tile.h"
namespace jt {

// ---- weight gradients: dW[m][n] += sum_p A[p][m] * B[p][n],  db[m] += sum_p A[p][m] ------------------------
// A skinny GEMM over the sample axis with v_mfma_f32_32x32x2_f32: the two lane halves take two consecutive
// samples per step, lane & 31 is the unit index for both operands (coalesced 128-byte row reads).
// XF = 1 / 2 builds the layer-1 input row [f, d, PE(f), PE(d)] / [f, PE(f)] on the fly from the F record.
// Column of the layer-1 weight matrix that N-tile b, lane m of the weight-gradient GEMM stands for.
// Tile 0 holds the raw inputs (features, then the view direction for MLP_Fea), tile t = 1..4 holds
// positional-encoding function t of the SAME source scalar, so a lane evaluates sin/cos once per sample
// and feeds all five tiles.  Returns -1 for lanes that carry nothing.
template <int XF>
__device__ inline int l1_column(int b, int m, int APP) {
  if (m < APP) return (b == 0) ? m : ((XF == 1) ? APP + 3 : APP) + 4 * m + (b - 1);
  if (XF == 1 && m < APP + 3) return (b == 0) ? m : APP + 3 + 4 * APP + 4 * (m - APP) + (b - 1);
  return -1;
}

// ---- weight gradients: dW[m][n] += sum_p A[p][m] * B[p][n],  db[m] += sum_p A[p][m] ------------------------
// A skinny GEMM over the sample axis with v_mfma_f32_32x32x2_f32.  The records are tile-blocked
// ([row][32 samples]), and the MFMA wants the unit on the lane (lane & 31) and the sample on the k axis
// (lane half + step): every lane therefore reads ITS unit's 128-byte row of the tile with eight 16-byte
// loads (the same access shape as the factor gather) and feeds its 16 same-parity samples step by step --
// no LDS, no transposition.  XF = 1 / 2: B is the layer-1 input [f, d, PE(f), PE(d)] / [f, PE(f)]: lane m
// loads the row of source scalar m once, takes sin/cos once per sample and serves all five N tiles.
template <int XF>
__device__ inline float pe_pick(int b, float x, float sn, float cs, float m0, float m1) {
  return (b == 0) ? x : (b == 1) ? sn * m0 : (b == 2) ? 2.f * sn * cs * m1 : (b == 3) ? cs * m0
                                                                                    : (1.f - 2.f * sn * sn) * m1;
}

// the 16 samples [16 h, 16 h + 16) of one record row (zeros for a dead row): four 16-byte loads
__device__ inline void load_row_half(const float* __restrict__ row, bool live, int h, float out[16]) {
#pragma unroll
  // (the load itself is unconditional -- callers clamp the row -- and a dead row is zeroed by selects: a load under a per-lane
  //  condition is a branch around it, and with loads under control flow the wait-count pass waits for everything)
  for (int q = 0; q < 4; ++q) {
    const float4 v = ld4(row + 16 * h + 4 * q);
    out[4 * q] = live ? v.x : 0.f;
    out[4 * q + 1] = live ? v.y : 0.f;
    out[4 * q + 2] = live ? v.z : 0.f;
    out[4 * q + 3] = live ? v.w : 0.f;
  }
}

// zero the samples past the nl live ones of a partial tile
__device__ inline void mask_tail(float v[16], int h, int nl) {
#pragma unroll
  for (int q = 0; q < 16; ++q) v[q] = (16 * h + q < nl) ? v[q] : 0.f;
}

// Work split of a weight-gradient GEMM over `blocks` workgroups of four waves: every wave takes `per` consecutive 32-sample
// tiles, at least kWgradMinTiles of them, so a chunk with few shaded samples (a trained scene shades a few per cent of what
// the random-init scene does) occupies only as many workgroups as it can feed -- the others return at once, write no slab,
// and k_wgrad_reduce4, which recomputes the same split, does not read theirs.  Measured on the converged synthetic scene
// (1.9 k tiles, replayed step): 1 tile per wave 0.963 ms, 2: 0.960, 4: 0.951, 8: 0.988 -- a wave's tiles are a chain of
// dependent row loads (~12 us each for dBasis), so few workgroups with long chains lose what the smaller epilogue wins.
#ifndef JT_WGRAD_MIN_TILES
#define JT_WGRAD_MIN_TILES 4
#endif
constexpr int kWgradMinTiles = JT_WGRAD_MIN_TILES;
__device__ inline int wgrad_tiles_per_wave(int n, int blocks) {
  const int ntiles = (n + 31) >> 5, nwaves = blocks * 4;
  return max((ntiles + nwaves - 1) / nwaves, kWgradMinTiles);
}
__device__ inline int wgrad_active_blocks(int n, int blocks) {
  if (n <= 0) return 0;
  const int ntiles = (n + 31) >> 5, per4 = wgrad_tiles_per_wave(n, blocks) * 4;
  return (ntiles + per4 - 1) / per4;
}

// epilogue of the weight-gradient GEMMs: sum the four waves' tiles through LDS and park the block's partial result in its
// slab (plain 256-byte stores).  Many blocks atomically adding into the same few-KB weight matrix would run at a fraction of
// the float-atomic rate, so the cross-block sum is a second, deterministic pass (k_wgrad_reduce4).
template <int MT, int NT>
__device__ inline void wgrad_epilogue(f32x16 (*acc)[NT], const float* asum, float (*s_red)[16][64], float* slab, int lane,
                                      int wv, int m, int h) {
  float* my = slab + (size_t)blockIdx.x * (MT * NT * 1024 + MT * 32);
#pragma unroll
  for (int a = 0; a < MT; ++a) {
#pragma unroll
    for (int b = 0; b < NT; ++b) {
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 16; ++r) s_red[wv][r][lane] = acc[a][b][r];
      __syncthreads();
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int r = wv * 4 + rr;
        my[(a * NT + b) * 1024 + r * 64 + lane] =
            s_red[0][r][lane] + s_red[1][r][lane] + s_red[2][r][lane] + s_red[3][r][lane];
      }
    }
    __syncthreads();
    s_red[wv][0][lane] = asum[a];
    __syncthreads();
    if (wv == 0 && h == 0)
      my[MT * NT * 1024 + a * 32 + m] = s_red[0][0][m] + s_red[0][0][m + 32] + s_red[1][0][m] + s_red[1][0][m + 32] +
                                       s_red[2][0][m] + s_red[2][0][m + 32] + s_red[3][0][m] + s_red[3][0][m + 32];
  }
}

// XA = 1 (round 6, lean tape): the A operand is G2, the gradient at the second hidden layer's pre-activations, and it is NOT
// read from records: G2[i][s] = relu'(h2[i][s]) (w3[0][i] go0[s] + w3[1][i] go1[s] + w3[2][i] go2[s]) is three fused
// multiply-adds on what the tile already holds per sample -- the three GO rows and the layer's ReLU sign word (row
// mask_row0 + lane half of the chain that owned the unit, bit (M tile) * 16 + r: jt_shade_record_layout) -- so the chain kernel
// does not store the HID rows of G2 (256 of 1 344 bytes per sample for VM-48) and this GEMM streams 5 + HID rows instead of
// 2 HID.  The four source rows are the same for all 32 units of a lane half: one fetch per tile, broadcast.
struct G2Src {
  const float* w3;   // [3][in3] (torch layout), the unit's column is hoff + i
  int go_row0, mask_row0, in3, hoff;
};
__device__ inline void g2_load_raw(const float* tile, const G2Src& S, int m, int h, float (*raw)[16]) {
  const int hh = (m >> 2) & 1;   // lane half of the chain wave that held unit m of an M tile (rowmap)
#pragma unroll
  for (int c = 0; c < 3; ++c) load_row_half(tile + (size_t)(S.go_row0 + c) * 32, true, h, raw[c]);
  load_row_half(tile + (size_t)(S.mask_row0 + hh) * 32, true, h, raw[3]);
}
template <int MT>
__device__ inline void g2_derive(const float (*raw)[16], const float (*w)[3], int m, float (*av)[16]) {
  const int r = (m & 3) + 4 * (m >> 3);
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float gsum = raw[0][q] * w[a][0] + raw[1][q] * w[a][1] + raw[2][q] * w[a][2];
      av[a][q] = ((__float_as_uint(raw[3][q]) >> (a * 16 + r)) & 1u) ? gsum : 0.f;
    }
}

template <int MT, int NT, int XF, int XA = 0>
__global__ __launch_bounds__(256) void k_wgrad(const float* __restrict__ rec, int a_row0, int M, int b_row0, int N,
                                               int f_row0, int vd_row0, int rec_rows, PeMask pm, int APP,
                                               const int* __restrict__ offset, int R, int cap, int chunk_start,
                                               int chunk_cap, float* __restrict__ slab, G2Src gs) {
  __shared__ float s_red[4][16][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int m = lane & 31, h = lane >> 5;
  const int total = min(offset[R], cap);
  const int n = min(total - chunk_start, chunk_cap);
  if (n <= 0) return;
  const int ntiles = (n + 31) >> 5;
  const int per = wgrad_tiles_per_wave(n, gridDim.x);
  if ((int)blockIdx.x >= wgrad_active_blocks(n, gridDim.x)) return;
  const int w = blockIdx.x * 4 + wv;
  const int t_begin = min(w * per, ntiles), t_end = min(t_begin + per, ntiles);
  f32x16 acc[MT][NT];
  float asum[MT];
#pragma unroll
  for (int a = 0; a < MT; ++a) {
    asum[a] = 0.f;
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  }
  // The contraction runs over the samples of a tile in any order: lane half h takes samples 16 h .. 16 h + 15
  // (MFMA step q pairs sample q with sample 16 + q), so a lane reads 64 contiguous bytes of its unit's row.
  // Rows are fetched one step ahead of the MFMAs that consume them (next B row, or the next tile's A rows).
  const size_t tstride = (size_t)rec_rows * 32;
  auto loadA = [&](int t, float (*dst)[16]) {
#pragma unroll
    for (int a = 0; a < MT; ++a) {
      const int c = a * 32 + m;
      load_row_half(rec + (size_t)t * tstride + (size_t)(a_row0 + min(c, M - 1)) * 32, c < M, h, dst[a]);
    }
  };
  float av[MT][16], an[MT][16];
  float gw[MT][3], raw[XA ? 4 : 1][16], rawn[XA ? 4 : 1][16];
  if (XA) {
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int c = 0; c < 3; ++c) gw[a][c] = (a * 32 + m < M) ? gs.w3[c * gs.in3 + gs.hoff + a * 32 + m] : 0.f;
  }
  if (XF == 0) {
    auto loadB = [&](int t, int b, float* dst) {
      const int c = b * 32 + m;
      load_row_half(rec + (size_t)t * tstride + (size_t)(b_row0 + min(c, N - 1)) * 32, c < N, h, dst);
    };
    float bv[16], bn[16];
    if (t_begin < t_end) {
      if (XA) g2_load_raw(rec + (size_t)t_begin * tstride, gs, m, h, raw);
      else loadA(t_begin, av);
      loadB(t_begin, 0, bv);
    }
    for (int t = t_begin; t < t_end; ++t) {
      const int nl = min(32, n - t * 32);
#pragma unroll
      for (int b = 0; b < NT; ++b) {
        if (b + 1 < NT) {
          loadB(t, b + 1, bn);
        } else if (t + 1 < t_end) {
          if (XA) g2_load_raw(rec + (size_t)(t + 1) * tstride, gs, m, h, rawn);
          else loadA(t + 1, an);
          loadB(t + 1, 0, bn);
        }
        if (b == 0) {
          if (XA) g2_derive<MT>(raw, gw, m, av);
          if (nl < 32) {
#pragma unroll
            for (int a = 0; a < MT; ++a) mask_tail(av[a], h, nl);
          }
#pragma unroll
          for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int q = 0; q < 16; ++q) asum[a] += av[a][q];
        }
        if (nl < 32) mask_tail(bv, h, nl);  // never-written record slots may hold anything, 0 * NaN is NaN
#pragma unroll
        for (int q = 0; q < 16; ++q)
#pragma unroll
          for (int a = 0; a < MT; ++a)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a][q], bv[q], acc[a][b], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 16; ++q) bv[q] = bn[q];
      }
      if (XA) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int q = 0; q < 16; ++q) raw[XA ? c : 0][q] = rawn[XA ? c : 0][q];
      } else {
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int q = 0; q < 16; ++q) av[a][q] = an[a][q];
      }
    }
  } else {
    const bool feat = m < APP, view = (XF == 1) && !feat && (m < APP + 3);
    const int srow = feat ? f_row0 + m : vd_row0 + (view ? m - APP : 0);
    const float m0 = view ? pm.v0 : pm.f0, m1 = view ? pm.v1 : pm.f1;
    const bool live = feat || view;
    float x[16], xn[16];
    if (t_begin < t_end) {
      loadA(t_begin, av);
      load_row_half(rec + (size_t)t_begin * tstride + (size_t)srow * 32, live, h, x);
    }
    for (int t = t_begin; t < t_end; ++t) {
      const int nl = min(32, n - t * 32);
      if (t + 1 < t_end) {
        loadA(t + 1, an);
        load_row_half(rec + (size_t)(t + 1) * tstride + (size_t)srow * 32, live, h, xn);
      }
      if (nl < 32) {
#pragma unroll
        for (int a = 0; a < MT; ++a) mask_tail(av[a], h, nl);
        mask_tail(x, h, nl);  // never-written record slots may hold anything, 0 * NaN is NaN
      }
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int q = 0; q < 16; ++q) asum[a] += av[a][q];
      float sn[16], cs[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) sincos_grad(x[q], &sn[q], &cs[q]);
#pragma unroll
      for (int b = 0; b < NT; ++b) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          // the A operand is zero past nl, so the (finite) encoding of a padding sample never contributes
          const float bv = live ? pe_pick<XF>(b, x[q], sn[q], cs[q], m0, m1) : 0.f;
#pragma unroll
          for (int a = 0; a < MT; ++a)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a][q], bv, acc[a][b], 0, 0, 0);
        }
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) x[q] = xn[q];
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int q = 0; q < 16; ++q) av[a][q] = an[a][q];
    }
  }
  wgrad_epilogue<MT, NT>(acc, asum, s_red, slab, lane, wv, m, h);
}

// Keep the use of sixteen prefetched values BEHIND whatever loads were issued in front of this point.  The instruction selector
// orders pure arithmetic by data dependence only: without a pin it places the consumer of a row directly behind that row's own
// loads, i.e. in front of the NEXT row's prefetch, and the "one step ahead" of the loops below turns into a full wait
// (s_waitcnt vmcnt(0)) behind every load (JT_WGRAD_PIN = 0 shows it in the listing).
#ifndef JT_WGRAD_PIN
#define JT_WGRAD_PIN 1
#endif
__device__ inline void pin16(float v[16]) {
#if JT_WGRAD_PIN
#pragma unroll
  for (int q = 0; q < 16; q += 4) asm volatile("" : "+v"(v[q]), "+v"(v[q + 1]), "+v"(v[q + 2]), "+v"(v[q + 3]));
#endif
}

// The same skinny GEMM on the bf16 matrix cores at fp32-level accuracy (jt_shade_core.h, "bf16x3"): both operands are
// per-sample data here, so both are split in registers -- (MT + NT) x 16 values per lane and tile -- and a tile's 32 samples
// are two 16-deep K steps (lane half h feeds samples 16 h + 8 s .. + 7 to step s: the contraction order is free as long as
// A and B agree).  Per (M tile, N tile) and tile of samples: 12 MFMAs of 32 cycles instead of 16 of 64.
template <int MT, int NT, int XF, int XA = 0>
__global__ __launch_bounds__(256) void k_wgrad_b16(const float* __restrict__ rec, int a_row0, int M, int b_row0, int N,
                                                   int f_row0, int vd_row0, int rec_rows, PeMask pm, int APP,
                                                   const int* __restrict__ offset, int R, int cap, int chunk_start,
                                                   int chunk_cap, float* __restrict__ slab, G2Src gs) {
  __shared__ float s_red[4][16][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int m = lane & 31, h = lane >> 5;
  const int total = min(offset[R], cap);
  const int n = min(total - chunk_start, chunk_cap);
  if (n <= 0) return;
  const int ntiles = (n + 31) >> 5;
  const int per = wgrad_tiles_per_wave(n, gridDim.x);
  if ((int)blockIdx.x >= wgrad_active_blocks(n, gridDim.x)) return;
  const int w = blockIdx.x * 4 + wv;
  const int t_begin = min(w * per, ntiles), t_end = min(t_begin + per, ntiles);
  f32x16 acc[MT][NT];
  float asum[MT];
#pragma unroll
  for (int a = 0; a < MT; ++a) {
    asum[a] = 0.f;
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  }
  const size_t tstride = (size_t)rec_rows * 32;
  auto loadA = [&](int t, float (*dst)[16]) {
#pragma unroll
    for (int a = 0; a < MT; ++a) {
      const int c = a * 32 + m;
      load_row_half(rec + (size_t)t * tstride + (size_t)(a_row0 + min(c, M - 1)) * 32, c < M, h, dst[a]);
    }
  };
  // A of the current tile: masked, summed for the bias gradient, split once for all N tiles
  auto prepA = [&](float (*av)[16], int nl, B3 (*a3)[2]) {
#pragma unroll
    for (int a = 0; a < MT; ++a) {
      if (nl < 32) mask_tail(av[a], h, nl);
#pragma unroll
      for (int q = 0; q < 16; ++q) asum[a] += av[a][q];
      a3[a][0] = split8(av[a]);
      a3[a][1] = split8(av[a] + 8);
    }
  };
  float av[MT][16], an[MT][16];
  B3 a3[MT][2];
  float gw[MT][3], raw[XA ? 4 : 1][16], rawn[XA ? 4 : 1][16];
  if (XA) {
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int c = 0; c < 3; ++c) gw[a][c] = (a * 32 + m < M) ? gs.w3[c * gs.in3 + gs.hoff + a * 32 + m] : 0.f;
  }
  if (XF == 0) {
    auto loadB = [&](int t, int b, float* dst) {
      const int c = b * 32 + m;
      load_row_half(rec + (size_t)t * tstride + (size_t)(b_row0 + min(c, N - 1)) * 32, c < N, h, dst);
    };
    float bv[16], bn[16];
    if (t_begin < t_end) {
      if (XA) g2_load_raw(rec + (size_t)t_begin * tstride, gs, m, h, raw);
      else loadA(t_begin, av);
      loadB(t_begin, 0, bv);
    }
    for (int t = t_begin; t < t_end; ++t) {
      const int nl = min(32, n - t * 32);
#pragma unroll
      for (int b = 0; b < NT; ++b) {
        // (unconditionally -- the last step of the last tile fetches its own rows again: with the loads under control flow the
        //  wait-count pass gives up and waits for everything)
        if (b + 1 < NT) {
          loadB(t, b + 1, bn);
        } else {
          const int tn = min(t + 1, t_end - 1);
          if (XA) g2_load_raw(rec + (size_t)tn * tstride, gs, m, h, rawn);
          else loadA(tn, an);
          loadB(tn, 0, bn);
        }
        if (b == 0) {
          if (XA) {
#pragma unroll
            for (int c = 0; c < 4; ++c) pin16(raw[XA ? c : 0]);
            g2_derive<MT>(raw, gw, m, av);
          } else {
#pragma unroll
          for (int a = 0; a < MT; ++a) pin16(av[a]);
          }
          prepA(av, nl, a3);
        }
        pin16(bv);
        if (nl < 32) mask_tail(bv, h, nl);  // never-written record slots may hold anything, 0 * NaN is NaN
        const B3 b0 = split8(bv), b1 = split8(bv + 8);
#pragma unroll
        for (int a = 0; a < MT; ++a) {
          acc[a][b] = mfma6(a3[a][0], b0, acc[a][b]);
          acc[a][b] = mfma6(a3[a][1], b1, acc[a][b]);
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) bv[q] = bn[q];
      }
      if (XA) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int q = 0; q < 16; ++q) raw[XA ? c : 0][q] = rawn[XA ? c : 0][q];
      } else {
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int q = 0; q < 16; ++q) av[a][q] = an[a][q];
      }
    }
  } else {
    const bool feat = m < APP, view = (XF == 1) && !feat && (m < APP + 3);
    const int srow = feat ? f_row0 + m : vd_row0 + (view ? m - APP : 0);
    const float m0 = view ? pm.v0 : pm.f0, m1 = view ? pm.v1 : pm.f1;
    const bool live = feat || view;
    float x[16], xn[16];
    if (t_begin < t_end) {
      loadA(t_begin, av);
      load_row_half(rec + (size_t)t_begin * tstride + (size_t)srow * 32, live, h, x);
    }
    for (int t = t_begin; t < t_end; ++t) {
      const int nl = min(32, n - t * 32);
      {
        const int tn = min(t + 1, t_end - 1);  // (unconditionally, see above)
        loadA(tn, an);
        load_row_half(rec + (size_t)tn * tstride + (size_t)srow * 32, live, h, xn);
      }
      pin16(x);
#pragma unroll
      for (int a = 0; a < MT; ++a) pin16(av[a]);
      if (nl < 32) mask_tail(x, h, nl);  // never-written record slots may hold anything, 0 * NaN is NaN
      prepA(av, nl, a3);
      float sn[16], cs[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) sincos_grad(x[q], &sn[q], &cs[q]);
#pragma unroll
      for (int b = 0; b < NT; ++b) {
        float bv[16];
        // the A operand is zero past nl, so the (finite) encoding of a padding sample never contributes
#pragma unroll
        for (int q = 0; q < 16; ++q) bv[q] = live ? pe_pick<XF>(b, x[q], sn[q], cs[q], m0, m1) : 0.f;
        const B3 b0 = split8(bv), b1 = split8(bv + 8);
#pragma unroll
        for (int a = 0; a < MT; ++a) {
          acc[a][b] = mfma6(a3[a][0], b0, acc[a][b]);
          acc[a][b] = mfma6(a3[a][1], b1, acc[a][b]);
        }
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) x[q] = xn[q];
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int q = 0; q < 16; ++q) av[a][q] = an[a][q];
    }
  }
  wgrad_epilogue<MT, NT>(acc, asum, s_red, slab, lane, wv, m, h);
}

// dW[i][k] += sum over the live chunks' block slabs; grid.y slab groups, a few atomics per element.  `xblock` = the block
// index inside this GEMM's range of the launch (k_wgrad_reduce4 serves the four GEMMs with one launch).
template <int MT, int NT, int XF>
__device__ inline void wgrad_reduce_body(int xblock, const float* __restrict__ slabs, int blocks_per_chunk,
                                         size_t chunk_stride, int chunk_entries, int total, int M, int N, int APP,
                                         float* __restrict__ dW, int ldw, float* __restrict__ db) {
  const int nchunks = (total + chunk_entries - 1) / chunk_entries;
  constexpr int PER = MT * NT * 1024 + MT * 32;
  const int idx = xblock * blockDim.x + threadIdx.x;
  if (idx >= PER) return;
  // blockIdx.y strides over the slabs of every chunk's ACTIVE blocks; four independent partial sums keep loads in flight
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  const int gy = gridDim.y;
  for (int c = 0; c < nchunks; ++c) {
    const int active = wgrad_active_blocks(min(total - c * chunk_entries, chunk_entries), blocks_per_chunk);
    const float* base = slabs + (size_t)c * chunk_stride + idx;
    int q = blockIdx.y;
    for (; q + 3 * gy < active; q += 4 * gy) {
      s0 += base[(size_t)q * PER];
      s1 += base[(size_t)(q + gy) * PER];
      s2 += base[(size_t)(q + 2 * gy) * PER];
      s3 += base[(size_t)(q + 3 * gy) * PER];
    }
    for (; q < active; q += gy) s0 += base[(size_t)q * PER];
  }
  const float sum = (s0 + s1) + (s2 + s3);
  if (sum == 0.f) return;
  if (idx < MT * NT * 1024) {
    const int tile = idx >> 10, r = (idx >> 6) & 15, lane = idx & 63;
    const int a = tile / NT, b = tile - a * NT, m = lane & 31, h = lane >> 5;
    const int i = a * 32 + rowmap(r, 0) + 4 * h;
    const int k = (XF == 0) ? b * 32 + m : l1_column<XF>(b, m, APP);
    if (i < M && k >= 0 && k < N) atomicAdd(dW + (size_t)i * ldw + k, sum);
  } else if (db) {
    const int i = idx - MT * NT * 1024;
    if (i < M) atomicAdd(db + i, sum);
  }
}

// tile counts and per-block slab sizes (floats) of the four weight-gradient GEMMs: dW3 / db3, dW2 / db2, dW1 / db1, dBasis
template <class C>
struct WgradDims {
  static constexpr int NT3 = (C::IN3 + 31) / 32, NT1 = 5, NTB = (C::NC + 31) / 32;
  static constexpr size_t P3 = 1 * NT3 * 1024 + 32, P2 = (size_t)C::MT * C::MT * 1024 + C::MT * 32,
                          P1 = (size_t)C::MT * NT1 * 1024 + C::MT * 32, PB = 1 * NTB * 1024 + 32;
};

// gradient tensors of the MLP as the reduce kernel's argument
struct MlpGrad {
  float *basis, *w1, *b1, *w2, *b2, *w3, *b3;
};

// the cross-block sums of the four weight-gradient GEMMs in ONE launch: grid.x is the concatenation of the four element
// ranges, 256 elements per block
template <class C>
__global__ __launch_bounds__(256) void k_wgrad_reduce4(const float* __restrict__ slabs, int blocks_per_chunk,
                                                       size_t chunk_stride, int chunk_entries,
                                                       const int* __restrict__ offset, int R, int cap, MlpGrad GM) {
  typedef WgradDims<C> W;
  constexpr int NT3 = W::NT3, NT1 = W::NT1, NTB = W::NTB;
  constexpr int XF1 = (C::KIND == JT_MLP_FEA) ? 1 : 2;
  constexpr int B3 = (int)((W::P3 + 255) / 256), B2 = (int)((W::P2 + 255) / 256), B1 = (int)((W::P1 + 255) / 256);
  const int total = min(offset[R], cap);
  const float* s3 = slabs;
  const float* s2 = s3 + W::P3 * blocks_per_chunk;
  const float* s1 = s2 + W::P2 * blocks_per_chunk;
  const float* sb = s1 + W::P1 * blocks_per_chunk;
  const int x = blockIdx.x;
  if (x < B3)
    wgrad_reduce_body<1, NT3, 0>(x, s3, blocks_per_chunk, chunk_stride, chunk_entries, total, 3, C::IN3, C::APP,
                                 GM.w3, C::IN3, GM.b3);
  else if (x < B3 + B2)
    wgrad_reduce_body<C::MT, C::MT, 0>(x - B3, s2, blocks_per_chunk, chunk_stride, chunk_entries, total, C::HID, C::HID,
                                       C::APP, GM.w2, C::HID, GM.b2);
  else if (x < B3 + B2 + B1)
    wgrad_reduce_body<C::MT, NT1, XF1>(x - B3 - B2, s1, blocks_per_chunk, chunk_stride, chunk_entries, total, C::HID,
                                       C::IN1, C::APP, GM.w1, C::IN1, GM.b1);
  else
    wgrad_reduce_body<1, NTB, 0>(x - B3 - B2 - B1, sb, blocks_per_chunk, chunk_stride, chunk_entries, total, C::APP, C::NC,
                                 C::APP, GM.basis, C::NC, (float*)nullptr);
}

}  // namespace jt

using namespace jt;

// which stages run on the bf16 matrix cores with three-piece operands (fp32-level accuracy, jt_shade_core.h): bit 0 the
// forward chain (k_shade_fwd_b16), bit 1 the weight-gradient GEMMs (k_wgrad_b16), bit 2 the chain of the SPLIT backward
// (k_shade_bwd<..., SPLIT, B16>).  JT_BF16X3 (read once) overrides the build
// default; 0 = everything on the fp32 matrix cores.
static std::atomic<int> g_matrix_mode{-1};
static int bf16x3_mode() {
  int m = g_matrix_mode.load(std::memory_order_relaxed);
  if (m < 0) {
    const char* e = getenv("JT_BF16X3");
    m = (e ? atoi(e) : JT_BF16X3_DEFAULT) & 7;
    g_matrix_mode.store(m, std::memory_order_relaxed);
  }
  return m;
}

typedef ShadeCfg<48, 27, 64, JT_MLP_FEA> CfgBlender;     // bat_blender_VM: VM-48, MLP_Fea 150->64->64->3
typedef ShadeCfg<20, 20, 32, JT_MLP_WEAKVIEW> CfgLlff;   // bat_llff_VM_MLP: VM-20, WeakView 100->32->32, 44->3

static int shade_kind(const JtScene* s) {
  if (!s) return -1;
  if (s->view_pe != 2 || s->fea_pe != 2) return -1;
  if (s->n_comp_app == 48 && s->app_dim == 27 && s->mlp_hidden == 64 && s->mlp_kind == JT_MLP_FEA) return 0;
  if (s->n_comp_app == 20 && s->app_dim == 20 && s->mlp_hidden == 32 && s->mlp_kind == JT_MLP_WEAKVIEW) return 1;
  return -1;
}

// workspace = [tile-blocked records of every chunk of shaded samples | per-(chunk, block) partial weight
// gradients].  Every chunk has its own record block so that the weight-gradient GEMMs of all chunks can run
// on an auxiliary stream, concurrently with whatever the caller enqueues next on the main stream (the
// density backward: atomics / VALU bound, while the GEMMs are MFMA bound).
// shaded samples per backward launch (and per set of weight-gradient GEMMs); JT_SHADE_CHUNK_LOG2 overrides (16..22)
static int g_chunk_log2 = 0;  // 0 = not yet initialised from the environment
static int chunk_entries() {
  if (!g_chunk_log2) {
    const char* e = getenv("JT_SHADE_CHUNK_LOG2");
    int l = e ? atoi(e) : 22;  // measured: 2^20 4.48 ms / step, 2^21 4.34, 2^22 4.33 (fewer launches, slabs and tails)
    if (l < 16 || l > 22) l = 22;
    g_chunk_log2 = l;
  }
  return 1 << g_chunk_log2;
}
#if JT_STAMP
extern "C" int jt_debug_read_stamps(unsigned long long* out8) {
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_stamps), 8 * sizeof(unsigned long long)) != hipSuccess) return JT_ERR_ARG;
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z)) != hipSuccess) return JT_ERR_ARG;
  return JT_OK;
}
#endif
// Split appearance backward: 0 = one kernel (chain + scatter in k_shade_bwd), 8 / 16 = k_shade_bwd<SPLIT> + k_shade_scatter
// with runs of that many samples per 16-lane group, 1 = k_shade_bwd<SPLIT> + the TILE-OWNED scatter of jt_tile.h (pairs binned
// by plane tile, gradient slices in matrix-core registers), -1 = per scene kind (the default): split 16 for the 20-channel
// WeakView scene (bat_llff_VM_MLP: its line gradients, privatised in LDS by the scatter kernel, are 40 % of the fused kernel's
// time: 0.49 -> 0.35 ms per launch) and, since the chain runs on the bf16 matrix cores (round 5: 508 -> 281 us), for VM-48 as
// well (281 + 994 against 1 365 us fused; with the fp32 chain the fused kernel wins: profiles/round4_bwd_split_ablation.txt).
// JT_BWD_SPLIT (read once) overrides.
static std::atomic<int> g_bwd_split{-2};
// Workgroups of k_shade_scatter (persistent, one per CU: its LDS fills the CU).  The scatter's time is inversely proportional to
// its workgroups (256: 0.96 ms, 192: 1.25, 128: 1.88): the float-atomic path is a PER-CU limit.  All the same the 48-channel
// scatter runs on 192 CUs -- 24 per XCD -- while the weight-gradient GEMMs are forked beside it: their 262-368 registers do not
// fit on a CU that holds a scatter workgroup, on 256 scatter workgroups they start when the scatter ends and the launch stream
// idles ~0.39 ms at the join; with 8 CUs per XCD to themselves they are done when the longer scatter is -- 3.00 against 3.07 ms
// per step, six alternating repeats (188 / 196 workgroups, which do not divide by the XCDs, lose 2-5 %; the 20-channel scene,
// whose GEMMs are small, loses 2 % and keeps 256; profiles/round5_scatter_beside_gemms.txt).  JT_SCATTER_WGS (read once) overrides.
// Round 6: with dBasis formed in the scatter itself only three GEMMs (1.7 instead of 2.5 GB of record rows) run beside it and the
// balance moves to 224 workgroups = 28 per XCD = 7 per shader engine: 2.88 against 2.93 (192) and 2.97 ms (256) per step
// (profiles/round6_lean_tape_ab.txt).
// (the counts are those of a full MI355X -- 256 CUs in 8 XCDs --; Chip::wgs scales them to the device the library runs on)
// Later in round 6: the twelve-wave scatter (three waves per SIMD, runs of 8) is faster per CU and hands the GEMMs 64 CUs again:
// 192 workgroups, 2.81 against 2.85-2.89 ms per step (profiles/round6_scatter_12_waves.txt).
static int scatter_wgs(bool gemms_beside, bool lean = false, bool w12 = false) {
  static const int v = [] { const char* e = getenv("JT_SCATTER_WGS"); const int n = e ? atoi(e) : 0; return n > 0 ? n : 0; }();
  return v ? std::min(v, chip().cus) : chip().wgs(gemms_beside ? ((lean && !w12) ? 224 : 192) : 256);
}
static int bwd_split_mode() {
  int m = g_bwd_split.load(std::memory_order_relaxed);
  if (m < -1) {
    const char* e = getenv("JT_BWD_SPLIT");
    m = e ? atoi(e) : -1;
    if (m != 0 && m != 1 && m != 8 && m != 16) m = -1;
    g_bwd_split.store(m, std::memory_order_relaxed);
  }
  return m;
}
#if JT_TILE_STAMP
extern "C" int jt_debug_read_tile_stamps(unsigned long long* out16) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_tstamps), 16 * sizeof(unsigned long long)) != hipSuccess) return JT_ERR_ARG;
  unsigned long long z[16] = {};
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_tstamps), z, sizeof(z)) != hipSuccess) return JT_ERR_ARG;
  return JT_OK;
}
#endif
// "Lean tape" (round 6; JT_LEAN_TAPE, read once, default 1; jt_shade_set_lean_tape): when the backward is the split form with the
// walker scatter (runs of 8 / 16), dBasis is formed inside k_shade_scatter and the training forward does not record the 3 Ca
// plane x line products -- a tile's record block shrinks from REC_FLOATS to R_LEAN rows (neither are the G2 rows stored: k_wgrad XA).  Whether a render is lean is a function
// of the library's modes alone (this switch, the split mode, matrix-mode bit 2), so the forward, the backward and the workspace
// query agree as long as no mode changes between a forward and its backward (as for jt_shade_set_chunk_log2).
static std::atomic<int> g_lean{-1};
static int lean_mode() {
  int m = g_lean.load(std::memory_order_relaxed);
  if (m < 0) {
    const char* e = getenv("JT_LEAN_TAPE");
    m = (e ? atoi(e) : 1) ? 1 : 0;
    g_lean.store(m, std::memory_order_relaxed);
  }
  return m;
}
template <class C>
static int split_default() { return (C::CA < 48 || (bf16x3_mode() & 4)) ? 16 : 0; }
template <class C>
static bool lean_tape() {
  if (!lean_mode()) return false;
  const int m = bwd_split_mode();
  return m == 8 || m == 16 || (m == -1 && split_default<C>() != 0);
}
template <class C>
static int rec_rows() { return lean_tape<C>() ? BwdCfg<C>::R_LEAN : BwdCfg<C>::REC_FLOATS; }
extern "C" int jt_shade_lean_tape(void) { return lean_mode(); }
extern "C" int jt_shade_set_lean_tape(int on) {
  const int prev = lean_mode();
  if (on == 0 || on == 1) g_lean.store(on, std::memory_order_relaxed);
  return prev;
}
extern "C" int jt_shade_bwd_split(void) { return bwd_split_mode(); }
extern "C" int jt_shade_set_bwd_split(int run) {
  const int prev = bwd_split_mode();
  if (run == -1 || run == 0 || run == 1 || run == 8 || run == 16) g_bwd_split.store(run, std::memory_order_relaxed);
  return prev;
}
extern "C" int jt_shade_chunk_entries(void) { return chunk_entries(); }
extern "C" int jt_shade_matrix_mode(void) { return bf16x3_mode(); }
extern "C" int jt_shade_set_matrix_mode(int mode) {
  const int prev = bf16x3_mode();
  if (mode >= 0 && mode <= 7) g_matrix_mode.store(mode, std::memory_order_relaxed);
  return prev;
}
// returns the previous log2; values outside 16..22 only query.  The caller re-sizes its workspace afterwards
// (jt_shade_workspace_bytes depends on the chunk size); not to be changed between a forward and its backward.
extern "C" int jt_shade_set_chunk_log2(int log2_entries) {
  (void)chunk_entries();
  const int prev = g_chunk_log2;
  if (log2_entries >= 16 && log2_entries <= 22) g_chunk_log2 = log2_entries;
  return prev;
}
#define kChunkEntries (chunk_entries())
static const int kNoGradRecords = 1 << 8;  // internal flag of launch_shade_bwd
static const int kWgradBlocks = 512;

template <class C>
struct WsLayout {
  typedef BwdCfg<C> B;
  typedef WgradDims<C> WD;
  static constexpr int NT3 = WD::NT3, NT1 = WD::NT1, NTB = WD::NTB;
  static constexpr size_t P3 = WD::P3, P2 = WD::P2, P1 = WD::P1, PB = WD::PB;
  static size_t rec_floats_per_chunk() { return (size_t)rec_rows<C>() * kChunkEntries; }
  static size_t slab_floats_per_chunk() { return (P3 + P2 + P1 + PB) * kWgradBlocks; }
  // records of all chunks are one contiguous tile-blocked array (a chunk is a whole number of 32-sample tiles), so
  // they take what `cap` samples need, not whole chunks; the slabs follow
  static size_t rec_floats(int cap) { return (size_t)rec_rows<C>() * (((size_t)std::max(cap, 1) + 31) / 32 * 32); }
  // (the tile-owned scatter's lists exist only while that variant is selected: up to 0.7 GB nobody else reads)
  static size_t bytes(int cap) {
    return main_bytes(cap) + (bwd_split_mode() == 1 ? tile_ws_bytes(cap, kChunkEntries) : 0);
  }
  // records + slabs; the tile-owned scatter's lists and counters (jt_tile.h) sit behind them
  static size_t main_bytes(int cap) {
    const int nchunks = (int)(((long)cap + kChunkEntries - 1) / kChunkEntries);
    return (rec_floats(cap) + slab_floats_per_chunk() * (size_t)std::max(nchunks, 1)) * sizeof(float);
  }
};

// Capacities above 2^30 shaded samples are refused (found by the UBSan run of tests/test_sanitizers.py: the chunk count of a
// capacity near INT_MAX overflowed an int, and the kernels index samples with 32-bit integers; 2^30 samples would be a
// 1.2 TB tape anyway).
static const int kMaxEntries = 1 << 30;
extern "C" size_t jt_shade_workspace_bytes(const JtScene* scene, int n_entries_max) {
  const int kind = shade_kind(scene);
  if (kind < 0 || n_entries_max < 1 || n_entries_max > kMaxEntries) return 0;
  return (kind == 0) ? WsLayout<CfgBlender>::bytes(n_entries_max) : WsLayout<CfgLlff>::bytes(n_entries_max);
}

// Where the pieces of a shade workspace sit, for the library's CURRENT modes (tests/test_sanitizers.py: every carved piece must
// lie inside jt_shade_workspace_bytes for boundary capacities).  Byte offsets from the workspace base / byte sizes:
//   out[0] total (= jt_shade_workspace_bytes)   out[1] records: size (offset 0)        out[2] slabs: offset
//   out[3] slabs: bytes per chunk                out[4] chunks                          out[5] rows of a tile's record block
//   out[6] dBasis slab: offset inside a chunk's slabs   out[7] dBasis slab: bytes the scatter's workgroups write at most
//   out[8] tile-owned scatter: 1 if its lists are part of the workspace, then (offset, bytes) pairs of its seven pieces
//          list, gx, items, cnt, offs, cursor, ctl in out[9..22]
template <class C>
static void ws_layout(int cap, int64_t* out) {
  typedef WsLayout<C> W;
  const int chunk = kChunkEntries;
  const int nchunks = std::max((int)(((long)cap + chunk - 1) / chunk), 1);
  out[0] = (int64_t)W::bytes(cap);
  out[1] = (int64_t)(W::rec_floats(cap) * sizeof(float));
  out[2] = out[1];
  out[3] = (int64_t)(W::slab_floats_per_chunk() * sizeof(float));
  out[4] = nchunks;
  out[5] = rec_rows<C>();
  out[6] = (int64_t)((W::P3 + W::P2 + W::P1) * kWgradBlocks * sizeof(float));
  out[7] = (int64_t)((size_t)chip().cus * ScatCfg<C>::DB_SLAB * sizeof(float));
  out[8] = bwd_split_mode() == 1 ? 1 : 0;
  for (int i = 9; i < 23; ++i) out[i] = 0;
  if (out[8]) {
    const TileWs t = tile_ws_carve(reinterpret_cast<void*>((uintptr_t)W::main_bytes(cap)), cap, chunk);
    const size_t lc = (size_t)t.list_cap, mi = (size_t)t.max_items;
    const void* ptrs[7] = {t.list, t.gx, t.items, t.cnt, t.offs, t.cursor, t.ctl};
    const size_t sizes[7] = {3 * lc * sizeof(uint4), kTileMaxClasses * lc * sizeof(float4), 3 * mi * sizeof(int4),
                             3 * (size_t)kTileMaxTiles * sizeof(int), 3 * (size_t)kTileMaxTiles * sizeof(int),
                             3 * (size_t)kTileMaxTiles * sizeof(int), 64 * sizeof(int)};
    for (int i = 0; i < 7; ++i) out[9 + 2 * i] = (int64_t)(uintptr_t)ptrs[i], out[10 + 2 * i] = (int64_t)sizes[i];
  }
}
extern "C" int jt_shade_workspace_layout(const JtScene* scene, int n_entries_max, int64_t* out23) {
  const int kind = shade_kind(scene);
  if (kind < 0 || !out23) return JT_ERR_UNSUPPORTED;
  if (n_entries_max < 1 || n_entries_max > kMaxEntries) return JT_ERR_ARG;
  if (kind == 0) ws_layout<CfgBlender>(n_entries_max, out23);
  else ws_layout<CfgLlff>(n_entries_max, out23);
  return JT_OK;
}

// layout of the per-tile records for readers outside this file (tests pin the ReLU signs the kernels took):
// out = {rows per tile, first of the four ReLU sign rows (2 * layer + lane half), hidden width, samples per tile}.
// Sign word of (layer, half h), sample j of tile t: float index (t * rows + row0 + 2 * layer + h) * 32 + j;
// bit mt * 16 + r  <->  hidden unit mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h.
extern "C" int jt_shade_record_layout(const JtScene* scene, int32_t* out) {
  const int kind = shade_kind(scene);
  if (kind < 0 || !out) return JT_ERR_UNSUPPORTED;
  if (kind == 0) {
    out[0] = rec_rows<CfgBlender>(); out[1] = BwdCfg<CfgBlender>::R_MASK; out[2] = CfgBlender::HID;
  } else {
    out[0] = rec_rows<CfgLlff>(); out[1] = BwdCfg<CfgLlff>::R_MASK; out[2] = CfgLlff::HID;
  }
  out[3] = 32;
  return JT_OK;
}

template <class C, int REC>
static int launch_shade_fwd_t(const Dev& D, const MlpDev& M, const PeMask& pm, const float* rays_o,
                              const float* rays_d, const float* jitter, const float* zvals, const float* tmin,
                              const int32_t* offset, int R, const int32_t* eray, const int32_t* esmp,
                              const float* vdir, float* rgb_s, float* rec, int cap, hipStream_t st) {
  long tiles = ((long)cap + 31) / 32;
  // JT_BF16X3 (read once): the forward's matrix stages as six bf16 MFMAs per fp32 product sum (jt_shade_core.h)
  if (bf16x3_mode() & 1) {
    const size_t lds16 = B16Cfg<C>::LDS_BYTES;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_shade_fwd_b16<C, REC>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16);
    constexpr int NW = JT_B16_THREADS / 64;
    int blocks16 = (int)std::min<long>((tiles + NW - 1) / NW, chip().cus);   // one 115 KB-LDS workgroup per CU
    hipLaunchKernelGGL((k_shade_fwd_b16<C, REC>), dim3(blocks16), dim3(JT_B16_THREADS), lds16, st, D, M, pm, rays_o, rays_d, jitter,
                       zvals, tmin, offset, R, eray, esmp, vdir, rgb_s, rec, cap, rec_rows<C>());
    JT_LAUNCH_CHECK();
    return JT_OK;
  }
  const size_t lds = C::LDS_FLOATS * sizeof(float);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_shade_fwd<C, REC>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  int blocks = (int)std::min<long>((tiles + 3) / 4, 512);
  hipLaunchKernelGGL((k_shade_fwd<C, REC>), dim3(blocks), dim3(256), lds, st, D, M, pm, rays_o, rays_d, jitter, zvals,
                     tmin, offset, R, eray, esmp, vdir, rgb_s, rec, cap, rec_rows<C>());
  JT_LAUNCH_CHECK();
  return JT_OK;
}

// workspace != NULL: training forward, the records of the layer inputs are left in the workspace for
// jt_shade_backward (which must be given the same workspace, untouched in between)
template <class C>
static int launch_shade_fwd(const Dev& D, const MlpDev& M, const PeMask& pm, const float* rays_o,
                            const float* rays_d, const float* jitter, const float* zvals, const float* tmin,
                            const int32_t* offset, int R, const int32_t* eray, const int32_t* esmp,
                            const float* vdir, float* rgb_s, int cap, float* ws, size_t ws_bytes, int flags,
                            hipStream_t st) {
  if (ws) {
    if (ws_bytes < WsLayout<C>::bytes(cap)) return JT_ERR_ARG;
    if (flags & JT_SHADE_POSE_ONLY)
      return launch_shade_fwd_t<C, 2>(D, M, pm, rays_o, rays_d, jitter, zvals, tmin, offset, R, eray, esmp, vdir,
                                      rgb_s, ws, cap, st);
    if (lean_tape<C>())
      return launch_shade_fwd_t<C, 3>(D, M, pm, rays_o, rays_d, jitter, zvals, tmin, offset, R, eray, esmp, vdir,
                                      rgb_s, ws, cap, st);
    return launch_shade_fwd_t<C, 1>(D, M, pm, rays_o, rays_d, jitter, zvals, tmin, offset, R, eray, esmp, vdir,
                                    rgb_s, ws, cap, st);
  }
  return launch_shade_fwd_t<C, 0>(D, M, pm, rays_o, rays_d, jitter, zvals, tmin, offset, R, eray, esmp, vdir,
                                  rgb_s, nullptr, cap, st);
}

extern "C" int jt_shade_forward(const JtScene* scene, const JtFactors* factors, const JtMlp* mlp, const float* rays_o,
                                const float* rays_d, const float* jitter, const float* zvals, const float* tmin,
                                const int32_t* shade_offset, int n_rays, const int32_t* entry_ray,
                                const int32_t* entry_smp, const float* viewdirs, float* rgb_s, int n_entries_max,
                                void* workspace, size_t workspace_bytes, int flags, void* stream) {
  Dev D;
  int rc = make_dev(scene, factors, &D);
  if (rc) return rc;
  if (!factors || !mlp || !rays_o || !rays_d || !tmin || !shade_offset || !entry_ray || !entry_smp || !viewdirs ||
      !rgb_s)
    return JT_ERR_ARG;
  if (!mlp->basis || !mlp->w1 || !mlp->b1 || !mlp->w2 || !mlp->b2 || !mlp->w3 || !mlp->b3) return JT_ERR_ARG;
  if (D.ndc && !zvals) return JT_ERR_ARG;
  const int kind = shade_kind(scene);
  if (kind < 0) return JT_ERR_UNSUPPORTED;
  if (n_entries_max < 1) return JT_OK;
  if (n_entries_max > kMaxEntries) return JT_ERR_ARG;
  MlpDev M = {mlp->basis, mlp->w1, mlp->b1, mlp->w2, mlp->b2, mlp->w3, mlp->b3};
  PeMask pm = pe_masks(scene->fea_pe_progress, scene->view_pe_progress, scene->fea_pe, scene->view_pe);
  hipStream_t st = (hipStream_t)stream;
  if (kind == 0)
    return launch_shade_fwd<CfgBlender>(D, M, pm, rays_o, rays_d, jitter, zvals, tmin, shade_offset, n_rays,
                                        entry_ray, entry_smp, viewdirs, rgb_s, n_entries_max, (float*)workspace,
                                        workspace_bytes, flags, st);
  return launch_shade_fwd<CfgLlff>(D, M, pm, rays_o, rays_d, jitter, zvals, tmin, shade_offset, n_rays, entry_ray,
                                   entry_smp, viewdirs, rgb_s, n_entries_max, (float*)workspace, workspace_bytes, flags,
                                   st);
}

// the tile-owned scatter (jt_tile.h) per scene kind.  Configuration 1 (the default): ONE channel class, eight-wave workgroups
// beside the plane's whole line of doubles (154 KB for VM-48) -- 252 registers per lane at two waves per SIMD, 1.0 ms at 400^3.
// Configuration 0 (JT_TILE_CFG=0, read once): sixteen-wave workgroups, VM-48's three channel groups as two classes (groups
// 0..1 | group 2, so that the class's line leaves room): the 128 registers of four waves per SIMD spill, 2.7-3.4 ms.  The
// launcher takes the other shape when the chosen one does not fit the LDS (a longer line).
template <class C>
struct TileSel {
  static constexpr int NG = (C::CA + 15) / 16;
  static constexpr int SPLIT0 = (NG >= 3) ? 2 : 0;
  static int ch_max(int cfg) { return (cfg == 0 && SPLIT0) ? 16 * SPLIT0 : C::CA; }
  static int waves(int cfg) { return cfg == 0 ? 16 : 8; }
  static size_t lds_bytes(int cfg, int line_len) { return tile_lds_bytes(line_len, ch_max(cfg), waves(cfg)); }
  static int tiles(int H, int W) { return tiles_along(W) * tiles_along(H); }
  template <int CFG>
  static int launch(const Dev& D, const MlpDev& M, const JtFactors& G, const TileWs& TW, const int32_t* offset, int R,
                    float* g_xyz, const float* rc, int start, int ccap, int cap, int line_len, hipStream_t st) {
    constexpr int SPLIT = CFG == 0 ? SPLIT0 : 0;
    constexpr int WAVES = CFG == 0 ? 16 : 8;
    constexpr int NCLS = SPLIT ? 2 : 1;
    int nt[3], ntmax = 1;
    for (int a = 0; a < 3; ++a) nt[a] = tiles(D.ph[a], D.pw[a]), ntmax = std::max(ntmax, nt[a]);
    const int nblk = (ccap + 255) / 256;
    hipLaunchKernelGGL(k_tile_zero, dim3((ntmax + 255) / 256, 3), dim3(256), 0, st, TW, nt[0], nt[1], nt[2]);
    hipLaunchKernelGGL((k_tile_bin<C, false>), dim3(nblk), dim3(256), 0, st, D, TW, offset, R, rc, start, ccap, cap);
    hipLaunchKernelGGL(k_tile_scan, dim3(3), dim3(1024), 0, st, TW, nt[0], nt[1], nt[2]);
    hipLaunchKernelGGL((k_tile_bin<C, true>), dim3(nblk), dim3(256), 0, st, D, TW, offset, R, rc, start, ccap, cap);
    const size_t lds = lds_bytes(CFG, line_len);
    static size_t attr = 0;
    if (attr < lds) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile_scatter<C, SPLIT, WAVES>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return JT_ERR_UNSUPPORTED;
      attr = lds;
    }
    // workgroups: one per CU, dealt to the (plane, class) sets; a class of two channel groups gets JT_TILE_RATIO per cent of
    // its plane's share (the per-pair set-up -- list entry, GF rows, tap records -- is the same for both classes)
    static const int nwg_env = [] { const char* e = getenv("JT_TILE_WGS"); const int v = e ? atoi(e) : 0; return v >= 6 ? v : 0; }();
    const int nwg = nwg_env ? nwg_env : std::max(chip().cus, 6);
    static const int ratio = [] { const char* e = getenv("JT_TILE_RATIO"); const int v = e ? atoi(e) : 0; return (v > 0 && v < 100) ? v : 62; }();
    TileClasses TC;
    int acc = 0;
    for (int pl = 0; pl < 3; ++pl) {
      const int share = nwg / 3 + (pl < nwg % 3 ? 1 : 0);
      const int c0 = NCLS == 2 ? std::max(1, std::min(share - 1, share * ratio / 100)) : share;
      TC.start[pl * NCLS] = acc;
      acc += c0;
      if (NCLS == 2) {
        TC.start[pl * NCLS + 1] = acc;
        acc += share - c0;
      }
    }
    for (int k = 3 * NCLS; k <= kTileMaxClasses; ++k) TC.start[k] = acc;
    hipLaunchKernelGGL((k_tile_scatter<C, SPLIT, WAVES>), dim3(acc), dim3(WAVES * 64), lds, st, D, M, G, TW, rc, TC);
    hipLaunchKernelGGL(k_tile_gxyz, dim3(nblk), dim3(256), 0, st, TW, 3 * NCLS, offset, R, g_xyz, start, ccap, cap);
    return JT_OK;
  }
};

template <class C>
static int launch_shade_bwd(const Dev& D, const MlpDev& M, const PeMask& pm, const JtFactors& G, const JtMlp& GM,
                            const int32_t* offset, int R, const float* rgb_s, const float* g_rgb_s, float* g_xyz,
                            int cap,
                            float* ws, size_t ws_bytes, int flags, hipStream_t st, hipStream_t aux,
                            hipEvent_t ev_fork, hipEvent_t ev_join) {
  typedef BwdCfg<C> B;
  typedef WsLayout<C> W;
  const size_t lds = B::LDS_FLOATS * sizeof(float);
  if (lds > 160 * 1024) return JT_ERR_UNSUPPORTED;
  if (ws_bytes < W::bytes(cap)) return JT_ERR_ARG;
  const int chunk = kChunkEntries;
  const int nchunks = (int)(((long)cap + chunk - 1) / chunk);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_shade_bwd<C, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_shade_bwd<C, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  float* recs = ws;
  float* slabs = ws + W::rec_floats(cap);
  const size_t cstride = W::slab_floats_per_chunk();
  const int nb = kWgradBlocks;
  // profiling knob, read ONCE per process: 1 = no scatter, 2 = no gradient records, 4 = no weight-gradient GEMMs
  static const int abl_env = [] { const char* e = getenv("JT_ABLATE"); return e ? atoi(e) : 0; }();
  const int det = jt_deterministic();  // 16: the appearance-factor gradients go to int64 shadow buffers (fixed point)
  // (bit 5: lean tape -- the chain does not store the G2 rows)
  const int ablate = abl_env | ((flags & JT_SHADE_SKIP_WGRAD) ? 4 : 0) |
                     ((flags & kNoGradRecords) ? 2 : 0) | (det ? 16 : 0) | (lean_tape<C>() ? 32 : 0);
  constexpr int NT3 = W::NT3, NT1 = W::NT1, NTB = W::NTB;
  constexpr int XF1 = (C::KIND == JT_MLP_FEA) ? 1 : 2;
  // ---- per chunk: the per-sample backward on the main stream, its weight-gradient GEMMs on the auxiliary stream ----
  // (the GEMMs of chunk c stream the records of chunk c while the backward of chunk c + 1 runs: the backward holds
  //  one 155 KB-LDS workgroup per CU at two waves per SIMD, the GEMMs need no LDS and fit beside it)
  const bool use_aux = aux && ev_fork && ev_join && !(ablate & 4);
  static const bool pipe = [] { const char* e = getenv("JT_WGRAD_PIPE"); return !e || atoi(e) != 0; }();
  hipStream_t ws_st = use_aux ? aux : st;
  const int RR = rec_rows<C>();   // rows of a tile's record block (lean tape: without the product rows)
  // per scene kind (-1): split 16 whenever the chain runs on the bf16 matrix cores (matrix-mode bit 2) -- the 20-channel scene
  // always did; VM-48 since round 5: chain 281 us + scatter 994 against 1 365 fused (with the fp32 chain, 508 + 994, the fused
  // kernel wins and stays)
  const int split_dflt = split_default<C>();
  int split = bwd_split_mode() >= 0 ? bwd_split_mode() : split_dflt;
  unsigned* bad = jt::fixed_bad_flag();
  if (!bad) return JT_ERR_ARG;
  // tile-owned scatter (split == 1): needs factor gradients to write, float accumulation, a scene whose tiles and LDS line fit
  // what the kernel and the workspace are sized for -- otherwise the scene kind's default takes over
  typedef TileSel<C> TS;
  int tile_line_len = 0;
  for (int a = 0; a < 3; ++a) tile_line_len = std::max(tile_line_len, D.ll[a]);
  // (one workgroup shape since round 6: eight waves, one channel class; the sixteen-wave two-class shape of round 5 spilled
  //  184-264 bytes per lane, took 2.7-3.4 ms and was removed -- a line too long for this shape falls back to the walker scatter)
  const int tile_cfg = 1;
  if (split == 1) {
    bool ok = !det && G.app_plane[0] && G.app_line[0] && TS::lds_bytes(tile_cfg, tile_line_len) <= 160 * 1024;
    for (int a = 0; a < 3 && ok; ++a) ok = TS::tiles(D.ph[a], D.pw[a]) <= kTileMaxTiles;
    if (!ok) split = split_dflt;
  }
  const bool tile = (split == 1);
  const TileWs TW = tile_ws_carve(reinterpret_cast<char*>(ws) + W::main_bytes(cap), cap, chunk);
  // fused: one kernel per chunk.  Split: the chain (launch_bwd) and the scatter (launch_scatter) -- the weight-gradient GEMMs only
  // need the chain's records, so they are forked BEHIND THE CHAIN and run next to the atomic-bound scatter.
  // nothing but the rays wants a gradient (no factor gradients, no weight gradients): the walker-free kernel
  // (JT_POSE_BWD=0, read once: the fused kernel with its scatter switched to "no targets", as before round 5)
  static const bool pose_env = [] { const char* e = getenv("JT_POSE_BWD"); return !e || atoi(e) != 0; }();
  const bool pose_only = pose_env && !det && !G.app_plane[0] && !G.app_line[0] && (flags & kNoGradRecords) && (ablate & 4);
  if (pose_only) split = 16;  // the chain alone (k_shade_bwd<SPLIT>: GF rows out), then k_pose_gather in the scatter's place
  // lean tape: the forward recorded no products, dBasis comes out of the walker scatter (k_shade_scatter FLAGS bit 1)
  const bool lean = lean_tape<C>();
  if (lean && (split != 8 && split != 16)) return JT_ERR_ARG;  // (a mode was changed between the forward and this backward)
  const bool dbs = lean && !pose_only && !(ablate & 4) && GM.basis != nullptr;
  // (the scatter's workgroups leave their dBasis slices where the fourth GEMM's slabs used to go)
  if (dbs && (size_t)chip().cus * ScatCfg<C>::DB_SLAB > W::PB * (size_t)kWgradBlocks) return JT_ERR_UNSUPPORTED;
  auto launch_bwd = [&](int ci) -> int {
    const int start = ci * chunk, ccap = std::min(chunk, cap - start);
    long tiles = ((long)ccap + 31) / 32;
    int blocks = (int)std::min<long>((tiles + B::NWAVE - 1) / B::NWAVE, chip().cus);   // one workgroup per CU (its LDS fills it)
    float* rc = recs + W::rec_floats_per_chunk() * ci;
    if (split && (bf16x3_mode() & 4)) {
      // the chain of the split backward on the bf16 matrix cores (three-piece operands): matrix-mode bit 2
      const size_t lds_b = ((BwdB16Cfg<C>::IMG_BYTES + 15) & ~(size_t)15) + (size_t)B::NWAVE * B::STASH_FLOATS * sizeof(float);
      static bool attr_b = false;
      if (!attr_b) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_shade_bwd<C, false, true, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b);
        attr_b = true;
      }
      hipLaunchKernelGGL((k_shade_bwd<C, false, true, true>), dim3(blocks), dim3(512), lds_b, st, D, M, pm, G, offset, R, rgb_s,
                         g_rgb_s, g_xyz, rc, start, ccap, cap, ablate, bad, RR);
    } else if (split) {
      const size_t lds_c = B::LDS_FLOATS_SPLIT * sizeof(float);
      static bool attr_done = false;
      if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_shade_bwd<C, false, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c);
        attr_done = true;
      }
      hipLaunchKernelGGL((k_shade_bwd<C, false, true>), dim3(blocks), dim3(512), lds_c, st, D, M, pm, G, offset, R, rgb_s,
                         g_rgb_s, g_xyz, rc, start, ccap, cap, ablate, bad, RR);
    } else if (det) {
      hipLaunchKernelGGL((k_shade_bwd<C, true>), dim3(blocks), dim3(512), lds, st, D, M, pm, G, offset, R, rgb_s,
                         g_rgb_s, g_xyz, rc, start, ccap, cap, ablate, bad, RR);
    } else {
      hipLaunchKernelGGL((k_shade_bwd<C, false>), dim3(blocks), dim3(512), lds, st, D, M, pm, G, offset, R, rgb_s,
                         g_rgb_s, g_xyz, rc, start, ccap, cap, ablate, bad, RR);
    }
    JT_LAUNCH_CHECK();
    return JT_OK;
  };
  // JT_SCATTER_FLAGS (read once): bit 0 line gradients through LDS (k_shade_scatter)
  static const int sflags_env = [] { const char* e = getenv("JT_SCATTER_FLAGS"); return e ? atoi(e) & 1 : 1; }();
  // waves per scatter workgroup, ONE workgroup per CU: two waves per SIMD (8) reach the atomic unit's rate on VM-48; the
  // 20-channel scatter runs at 40 % of that rate (latency of the walk, not atomics) and takes four per SIMD (16) where the LDS
  // line still fits beside their step records (runs of 8)
  int line_floats = 0;
  for (int a = 0; a < 3; ++a) line_floats = std::max(line_floats, D.ll[a] * C::CA);
  int sflags = split ? sflags_env : 0;
  if (det) sflags &= ~1;
  auto scatter_lds = [&](int run, int fl, int sw) {
    return (size_t)(ScatCfg<C>::BT_FLOATS + ((fl & 1) ? line_floats : 0) + (dbs ? ScatCfg<C>::RED_FLOATS : 0) +
                    sw * ScatCfg<C>::wave_floats(run, dbs)) * sizeof(float);
  };
  static const int sw_env = [] { const char* e = getenv("JT_SCATTER_WAVES"); return e ? atoi(e) : 0; }();
  int sw = 8;
  // (with dBasis formed in the scatter the sixteen-wave shape's 128 registers spill 44-96 bytes per lane -- and it is still the
  //  faster shape where it fits: LLFF stage 0, 20 480 rays, 4.01 against 4.24 ms per step with eight waves)
  if (C::CA < 48 && split && !tile && (sw_env == 16 || (sw_env == 0 && scatter_lds(split, sflags, 16) <= 160 * 1024))) sw = 16;
  // The 48-channel scatter at THREE waves per SIMD (round 6): twelve-wave workgroups, runs of 8 (the step records of runs of 16
  // do not fit beside the LDS line twelve times), 168 registers with the basis^T operands read from LDS at their use (12 bytes of
  // scratch remain).  The kernel waits on memory for half of its wave cycles (SQ counters, profiles/round6_held_flush_experiment.txt),
  // and the third wave hides more of that than the shorter runs' extra flushes cost: scatter alone 1.05 -> 0.99 ms, and with
  // 192 workgroups beside the three GEMMs the step 2.85-2.89 -> 2.81 ms.  Chosen when the split mode is left to the library
  // (or JT_BWD_SPLIT=8 JT_SCATTER_WAVES=12), dBasis is formed in the kernel, accumulation is float and the line fits;
  // JT_SCATTER_WAVES=8 keeps the eight-wave shape.
  bool w12 = false;
  // (the 20-channel scene gains nothing from it: final LLFF grid, scatter 0.288 ms in both shapes -- its instantiation was removed)
  if (C::CA >= 48 && !det && !tile && !pose_only && dbs && sflags == 1 && (sw_env == 0 || sw_env == 12) &&
      ((bwd_split_mode() == -1 && split == 16) || (bwd_split_mode() == 8 && sw_env == 12)) &&
      scatter_lds(8, sflags, 12) <= 160 * 1024) {
    split = 8, sw = 12, w12 = true;
  }
  if (split && !tile && scatter_lds(split, sflags, sw) > 160 * 1024) sflags &= ~1;  // a line too long for the LDS: global atomics as before
  auto launch_scatter = [&](int ci) -> int {
    if (!split || (ablate & 1)) return JT_OK;
    const int start = ci * chunk, ccap = std::min(chunk, cap - start);
    const float* rc = recs + W::rec_floats_per_chunk() * ci;
    if (pose_only) {
      const long tiles = ((long)ccap + 31) / 32;
      const int pblocks = (int)std::min<long>((tiles + 3) / 4, 2048L);
      // (the 20-channel scene keeps the fp32 product: its bf16 instantiation needs 168 registers + 8 bytes of scratch at three
      //  waves per SIMD, for a product that is a third of VM-48's)
      if ((bf16x3_mode() & 4) && C::CA >= 48)
        hipLaunchKernelGGL((k_pose_gather<C, (C::CA >= 48)>), dim3(pblocks), dim3(256), 0, st, D, M, offset, R, g_xyz, rc, start, ccap, cap, RR);
      else
        hipLaunchKernelGGL((k_pose_gather<C, false>), dim3(pblocks), dim3(256), 0, st, D, M, offset, R, g_xyz, rc, start, ccap, cap, RR);
      JT_LAUNCH_CHECK();
      return JT_OK;
    }
    if (tile) {
      int rc_ = TS::template launch<1>(D, M, G, TW, offset, R, g_xyz, rc, start, ccap, cap, tile_line_len, st);
      if (rc_) return rc_;
      JT_LAUNCH_CHECK();
      return JT_OK;
    }
    const size_t lds_s = scatter_lds(split, sflags, sw);
#define JT_SCATTER_LAUNCH(RUN_, DET_, FL_, SW_)                                                                         \
  {                                                                                                                     \
    static bool attr = false;                                                                                           \
    if (!attr) {                                                                                                        \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_shade_scatter<C, DET_, RUN_, SW_, FL_>),                \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                                \
      attr = true;                                                                                                      \
    }                                                                                                                   \
    const long nbatch = ((long)ccap + 4 * RUN_ - 1) / (4 * RUN_);                                                       \
    const int sblocks = (int)std::min<long>((nbatch + SW_ - 1) / SW_, (long)scatter_wgs(use_aux && C::CA >= 48, dbs, w12));                             \
    hipLaunchKernelGGL((k_shade_scatter<C, DET_, RUN_, SW_, FL_>), dim3(sblocks), dim3(SW_ * 64), lds_s, st, D, M, G,   \
                       offset, R, g_xyz, rc, start, ccap, cap, bad, line_floats, RR,                                    \
                       slabs + (size_t)ci * cstride + (W::P3 + W::P2 + W::P1) * nb);                                    \
  }
#define JT_SCATTER_FL(RUN_, SW_)                                                                \
  {                                                                                             \
    if (det && dbs) JT_SCATTER_LAUNCH(RUN_, true, 2, SW_)                                       \
    else if (det) JT_SCATTER_LAUNCH(RUN_, true, 0, SW_)                                         \
    else if (sflags == 1 && dbs) JT_SCATTER_LAUNCH(RUN_, false, 3, SW_)                         \
    else if (sflags == 1) JT_SCATTER_LAUNCH(RUN_, false, 1, SW_)                                \
    else if (dbs) JT_SCATTER_LAUNCH(RUN_, false, 2, SW_)                                        \
    else JT_SCATTER_LAUNCH(RUN_, false, 0, SW_)                                                 \
  }
#define JT_SCATTER_RUN(RUN_)                                                                    \
  {                                                                                             \
    if (C::CA < 48 && sw == 16) JT_SCATTER_FL(RUN_, (C::CA < 48 ? 16 : 8))                      \
    else JT_SCATTER_FL(RUN_, 8)                                                                 \
  }
    // (experiment, JT_SCATTER_WAVES=12 with JT_BWD_SPLIT=8: three waves per SIMD for the 48-channel scatter)
    if (w12)
      JT_SCATTER_LAUNCH(8, false, 3, (C::CA >= 48 ? 12 : 8))
    else if (split == 8) JT_SCATTER_RUN(8) else JT_SCATTER_RUN(16)
#undef JT_SCATTER_RUN
#undef JT_SCATTER_FL
#undef JT_SCATTER_LAUNCH
    JT_LAUNCH_CHECK();
    return JT_OK;
  };
  auto launch_wgrad = [&](int ci) -> int {
    const int start = ci * chunk, ccap = std::min(chunk, cap - start);
    const float* rec = recs + W::rec_floats_per_chunk() * ci;
    // dW3/db3 = GO^T MID ; dW2/db2 = G2^T H1 ; dW1/db1 = G1^T X(F, d) ; dBasis = GF^T PROD
    const G2Src gs = {M.w3, B::R_GO, B::R_MASK + 2, C::IN3, (C::KIND == JT_MLP_FEA) ? 0 : 12};
    float* s3 = slabs + (size_t)ci * cstride;
    float* s2 = s3 + W::P3 * nb;
    float* s1 = s2 + W::P2 * nb;
    float* sb = s1 + W::P1 * nb;
#define JT_WGRAD_LAUNCH(KERNEL)                                                                                          \
  hipLaunchKernelGGL((KERNEL<1, NT3, 0>), dim3(nb), dim3(256), 0, ws_st, rec, B::R_GO, 3, B::R_MID, C::IN3, B::R_F,      \
                     B::R_VD, RR, pm, C::APP, offset, R, cap, start, ccap, s3, gs);                                      \
  JT_LAUNCH_CHECK();                                                                                                     \
  if (lean) {                                                                                                            \
  hipLaunchKernelGGL((KERNEL<C::MT, C::MT, 0, 1>), dim3(nb), dim3(256), 0, ws_st, rec, B::R_G2, C::HID, B::R_H1, C::HID, \
                     B::R_F, B::R_VD, RR, pm, C::APP, offset, R, cap, start, ccap, s2, gs);                              \
  } else {                                                                                                               \
  hipLaunchKernelGGL((KERNEL<C::MT, C::MT, 0>), dim3(nb), dim3(256), 0, ws_st, rec, B::R_G2, C::HID, B::R_H1, C::HID,    \
                     B::R_F, B::R_VD, RR, pm, C::APP, offset, R, cap, start, ccap, s2, gs);                              \
  }                                                                                                                      \
  JT_LAUNCH_CHECK();                                                                                                     \
  hipLaunchKernelGGL((KERNEL<C::MT, NT1, XF1>), dim3(nb), dim3(256), 0, ws_st, rec, B::R_G1, C::HID, B::R_F, C::IN1,     \
                     B::R_F, B::R_VD, RR, pm, C::APP, offset, R, cap, start, ccap, s1, gs);                              \
  JT_LAUNCH_CHECK();                                                                                                     \
  if (!dbs) {                                                                                                            \
  hipLaunchKernelGGL((KERNEL<1, NTB, 0>), dim3(nb), dim3(256), 0, ws_st, rec, B::R_GF, C::APP, B::R_PROD, C::NC, B::R_F, \
                     B::R_VD, RR, pm, C::APP, offset, R, cap, start, ccap, sb, gs);                                      \
  JT_LAUNCH_CHECK();                                                                                                     \
  }
    if (bf16x3_mode() & 2) {
      JT_WGRAD_LAUNCH(k_wgrad_b16)
    } else {
      JT_WGRAD_LAUNCH(k_wgrad)
    }
#undef JT_WGRAD_LAUNCH
    return JT_OK;
  };
  // (JT_SCATTER_FIRST=1: the scatter in front of the fork, as in the fused kernel's order)
  static const bool scatter_first = [] { const char* e = getenv("JT_SCATTER_FIRST"); return e && atoi(e) != 0; }();
  for (int ci = 0; ci < nchunks; ++ci) {
    int rc = launch_bwd(ci);
    if (rc) return rc;
    if (scatter_first && (rc = launch_scatter(ci))) return rc;
    if (use_aux && pipe) {
      if (hipEventRecord(ev_fork, st) != hipSuccess) return JT_ERR_ARG;
      if (hipStreamWaitEvent(aux, ev_fork, 0) != hipSuccess) return JT_ERR_ARG;
      if ((rc = launch_wgrad(ci))) return rc;
    }
    if (!scatter_first && (rc = launch_scatter(ci))) return rc;
  }
  if (dbs && !(ablate & 1)) {
    // dBasis: the sum of the scatter workgroups' slabs, on the launch stream behind the last scatter
    const int ry = det ? 1 : 8;
    hipLaunchKernelGGL((k_dbasis_reduce<C>), dim3((ScatCfg<C>::DB_SLAB + 255) / 256, ry), dim3(256), 0, st,
                       slabs + (W::P3 + W::P2 + W::P1) * nb, cstride, chunk, scatter_wgs(use_aux && C::CA >= 48, dbs, w12), 4 * split,
                       sw, offset, R, cap, GM.basis);
    JT_LAUNCH_CHECK();
  }
  if (ablate & 4) return JT_OK;
  if (use_aux && !pipe) {
    if (hipEventRecord(ev_fork, st) != hipSuccess) return JT_ERR_ARG;
    if (hipStreamWaitEvent(aux, ev_fork, 0) != hipSuccess) return JT_ERR_ARG;
  }
  // no auxiliary stream but an event: it marks the end of the per-sample backward kernels on `stream` (callers time
  // k_shade_bwd inside a training step with it: bench.py's in-step roofline)
  if (!use_aux && ev_fork && hipEventRecord(ev_fork, st) != hipSuccess) return JT_ERR_ARG;
  if (!(use_aux && pipe)) {
    for (int ci = 0; ci < nchunks; ++ci) {
      int rc = launch_wgrad(ci);
      if (rc) return rc;
    }
  }
  {
    const int ry = det ? 1 : 32;  // slab groups that add into dW atomically; ONE group = a fixed summation order
    // (dBasis out of the scatter: the last range of the reduce kernel's grid, the dBasis GEMM's slabs, is left off)
    const int nblk = (int)((W::P3 + 255) / 256 + (W::P2 + 255) / 256 + (W::P1 + 255) / 256 + (dbs ? 0 : (W::PB + 255) / 256));
    const MlpGrad gm = {GM.basis, GM.w1, GM.b1, GM.w2, GM.b2, GM.w3, GM.b3};
    hipLaunchKernelGGL((k_wgrad_reduce4<C>), dim3(nblk, ry), dim3(256), 0, ws_st, slabs, nb, cstride, chunk, offset, R, cap,
                       gm);
    JT_LAUNCH_CHECK();
  }
  if (ws_st != st) {
    if (hipEventRecord(ev_join, aux) != hipSuccess) return JT_ERR_ARG;
  }
  return JT_OK;
}

extern "C" int jt_shade_backward(const JtScene* scene, const JtFactors* factors, const JtMlp* mlp,
                                 const float* rays_o, const float* rays_d, const float* jitter, const float* zvals,
                                 const float* tmin, const int32_t* shade_offset, int n_rays,
                                 const int32_t* entry_ray, const int32_t* entry_smp, const float* viewdirs,
                                 const float* rgb_s, const float* g_rgb_s, const JtFactors* g_factors, const JtMlp* g_mlp,
                                 float* g_xyz_app, int n_entries_max, void* workspace, size_t workspace_bytes,
                                 int flags, void* stream, void* aux_stream, void* ev_fork, void* ev_join) {
  Dev D;
  int rc = make_dev(scene, factors, &D);
  if (rc) return rc;
  if (!factors || !mlp || !rays_o || !rays_d || !tmin || !shade_offset || !entry_ray ||
      !entry_smp || !viewdirs || !rgb_s || !g_rgb_s || !g_xyz_app)
    return JT_ERR_ARG;
  if (!mlp->basis || !mlp->w1 || !mlp->b1 || !mlp->w2 || !mlp->b2 || !mlp->w3 || !mlp->b3) return JT_ERR_ARG;
  // g_factors == NULL / g_mlp == NULL: that group of gradients is not wanted (pose-only backward)
  if (g_mlp && (!g_mlp->basis || !g_mlp->w1 || !g_mlp->b1 || !g_mlp->w2 || !g_mlp->b2 || !g_mlp->w3 || !g_mlp->b3))
    return JT_ERR_ARG;
  for (int a = 0; a < 3; ++a)
    if (g_factors && (!g_factors->app_plane[a] || !g_factors->app_line[a])) return JT_ERR_ARG;
  JtFactors no_fac = {};
  JtMlp no_mlp = {};
  if (!g_mlp) flags |= JT_SHADE_SKIP_WGRAD | kNoGradRecords;  // nobody will read the gradient records
  const JtFactors& GFr = g_factors ? *g_factors : no_fac;
  const JtMlp& GMr = g_mlp ? *g_mlp : no_mlp;
  if (D.ndc && !zvals) return JT_ERR_ARG;
  const int kind = shade_kind(scene);
  if (kind < 0) return JT_ERR_UNSUPPORTED;
  if (n_entries_max < 1) return JT_OK;
  if (!workspace || n_entries_max > kMaxEntries) return JT_ERR_ARG;
  MlpDev M = {mlp->basis, mlp->w1, mlp->b1, mlp->w2, mlp->b2, mlp->w3, mlp->b3};
  PeMask pm = pe_masks(scene->fea_pe_progress, scene->view_pe_progress, scene->fea_pe, scene->view_pe);
  hipStream_t st = (hipStream_t)stream;
  if (kind == 0)
    return launch_shade_bwd<CfgBlender>(D, M, pm, GFr, GMr, shade_offset, n_rays, rgb_s, g_rgb_s, g_xyz_app,
                                        n_entries_max, (float*)workspace, workspace_bytes, flags, st, (hipStream_t)aux_stream,
                                        (hipEvent_t)ev_fork, (hipEvent_t)ev_join);
  return launch_shade_bwd<CfgLlff>(D, M, pm, GFr, GMr, shade_offset, n_rays, rgb_s, g_rgb_s, g_xyz_app,
                                   n_entries_max,
                                   (float*)workspace, workspace_bytes, flags, st, (hipStream_t)aux_stream,
                                   (hipEvent_t)ev_fork, (hipEvent_t)ev_join);
}
