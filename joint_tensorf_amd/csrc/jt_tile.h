// Tile-owned scatter of the appearance-factor gradients (round 5).
//
// grid_sampler_2d_backward under autograd (bateRF.py:97-130) adds four plane taps and two line taps per (sample, plane)
// wherever the sample happens to lie; the run-length walkers of jt_walk.h cut that to ~5 float-atomic segments per
// (sample, plane) -- 20 M per launch at 400^3, 0.96 ms of the chip-wide float-atomic unit whatever issues them.  Here the
// (sample, plane) pairs of a chunk are first BINNED by the plane tile their bilinear footprint falls into (counting sort:
// count, scan, fill; consecutive samples of a ray stay in a tile for tens of steps, so a wave reserves list space once
// per run of equal tiles, not once per sample).  Then one workgroup at a time OWNS a tile: the tile's gradient slice
// ((TX + 1) x (TY + 1) texels x CA channels) and the plane's whole line gradient live in LDS, every tap is an LDS float
// atomic, and the slice goes out ONCE -- ~0.4 global atomic segments per pair instead of ~5.  The tile's factor values are
// staged in LDS beside the gradient slice (STAGE), so of the six taps of a pair only the two line taps are still gathered
// from memory.
//
// Work split: a persistent workgroup belongs to ONE plane (its LDS line) and pulls items (tile, range of the tile's list)
// from the plane's counter; its waves take groups of 16 pairs of the item.  basis^T GF comes out of
// v_mfma_f32_16x16x4_f32 exactly as in k_shade_scatter: lane (group g, channel cl) holds the product gradients of pairs
// 4 g .. 4 g + 3 in its four D registers.  The coordinate gradients of a pair (three floats, one per axis) leave as one
// 16-byte store into a per-plane array indexed by the sample; k_tile_gxyz adds the three planes' arrays into g_xyz.
#pragma once
#include <type_traits>

#include "jt_shade_core.h"

#ifndef JT_TILE_ABL
#define JT_TILE_ABL 0  // profiling knob: 1 no slice atomics, 2 no line atomics, 4 no line-tap loads, 8 no slots at all
#endif

namespace jt {

constexpr int kTileMaxTiles = 16384;   // tiles per plane the workspace is sized for (a scene with more falls back)
constexpr int kTileItemCap = 2048;     // pairs per work item: a heavier tile is split (each part flushes the slice)
constexpr int kTileRec = 24;           // words of a pair's tap record in LDS

struct TileWs {
  int* cnt;       // [3][kTileMaxTiles] pairs per tile
  int* offs;      // [3][kTileMaxTiles] start of the tile's list (plane-relative)
  int* cursor;    // [3][kTileMaxTiles] fill position
  int4* items;    // [3][max_items] {tile, begin, end, -}
  int* ctl;       // [0..2] items per plane, [4..6] next item per plane
  uint4* list;    // [3][list_cap] {sample (chunk-local), n[m0], n[m1], n[v] as float bits}
  float4* gx3;    // [3][list_cap] coordinate gradients of a sample from one plane, by axis
  int max_items, list_cap;
};

__host__ inline size_t tile_ws_bytes(int cap, int chunk) {
  const size_t lc = (size_t)std::min(std::max(cap, 1), chunk);
  const size_t max_items = kTileMaxTiles + (size_t)chunk / kTileItemCap + 1;
  return 3 * kTileMaxTiles * 3 * sizeof(int) + 3 * max_items * sizeof(int4) + 64 * sizeof(int) + 3 * lc * sizeof(uint4) +
         3 * lc * sizeof(float4) + 1024;
}
__host__ inline TileWs tile_ws_carve(void* base, int cap, int chunk) {
  TileWs w;
  const size_t lc = (size_t)std::min(std::max(cap, 1), chunk);
  const size_t max_items = kTileMaxTiles + (size_t)chunk / kTileItemCap + 1;
  char* p = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(base) + 255) & ~(uintptr_t)255);
  w.list = reinterpret_cast<uint4*>(p), p += 3 * lc * sizeof(uint4);
  w.gx3 = reinterpret_cast<float4*>(p), p += 3 * lc * sizeof(float4);
  w.items = reinterpret_cast<int4*>(p), p += 3 * max_items * sizeof(int4);
  w.cnt = reinterpret_cast<int*>(p), p += 3 * kTileMaxTiles * sizeof(int);
  w.offs = reinterpret_cast<int*>(p), p += 3 * kTileMaxTiles * sizeof(int);
  w.cursor = reinterpret_cast<int*>(p), p += 3 * kTileMaxTiles * sizeof(int);
  w.ctl = reinterpret_cast<int*>(p);
  w.max_items = (int)max_items;
  w.list_cap = (int)lc;
  return w;
}

// tiles along an axis of `size` texels (size - 1 cells) with `tc` cells per tile
__host__ __device__ inline int tiles_along(int size, int tc) { return (size < 2 ? 0 : size - 2) / tc + 1; }

template <int TXC, int TYC>
__device__ inline int tile_of(float gx, float gy, int H, int W, int ntx) {
  const int cx = min(max(axis_cell(gx, W), 0), max(W - 2, 0));
  const int cy = min(max(axis_cell(gy, H), 0), max(H - 2, 0));
  return (cy / TYC) * ntx + cx / TXC;
}

__global__ void k_tile_zero(TileWs W, int nt0, int nt1, int nt2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int pl = blockIdx.y;
  const int nt = pl == 0 ? nt0 : (pl == 1 ? nt1 : nt2);
  if (i < nt) W.cnt[pl * kTileMaxTiles + i] = 0;
  if (i < 8 && pl == 0) W.ctl[i] = 0;
}

// FILL = false: pairs per tile; FILL = true: the lists.  One thread per shaded sample of the chunk; a wave's lanes hold
// consecutive samples, and a run of lanes with the same tile reserves its list space with ONE atomic (its first lane).
template <class C, int TXC, int TYC, bool FILL>
__global__ __launch_bounds__(256) void k_tile_bin(Dev D, TileWs W, const int* __restrict__ offset, int R,
                                                  const float* __restrict__ rec, int chunk_start, int chunk_cap, int cap) {
  typedef BwdCfg<C> B;
  const int total = min(offset[R], cap);
  const int n_chunk = min(total - chunk_start, chunk_cap);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if ((int)(blockIdx.x * 256) >= n_chunk) return;
  const int lane = threadIdx.x & 63;
  const bool live = i < n_chunk;
  const int L = live ? i : n_chunk - 1;
  const float* rt = rec + (size_t)(L >> 5) * B::REC_FLOATS * 32 + (L & 31);
  float n[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) n[a] = rec_ld(rt + (B::R_GEO + a) * 32);
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) {
    const int ntx = tiles_along(D.pw[pl], TXC);
    const float gx = n[kM0(pl)], gy = n[kM1(pl)], gl = n[kV(pl)];
    const int t = live ? tile_of<TXC, TYC>(gx, gy, D.ph[pl], D.pw[pl], ntx) : -1;
    const int tp = __shfl_up(t, 1);
    const bool head = (lane == 0) || (t != tp);
    const unsigned long long hm = __ballot(head);
    const unsigned long long below = hm & ((2ull << lane) - 1ull);      // heads at or below this lane (lane 63: all)
    const int hp = 63 - __builtin_clzll(lane == 63 ? hm : below);       // this lane's run head
    const unsigned long long above = (lane == 63) ? 0ull : (hm >> (lane + 1));
    const int len = above ? __builtin_ctzll(above) + 1 : 64 - lane;     // (meaningful in head lanes)
    int* slot = (FILL ? W.cursor : W.cnt) + pl * kTileMaxTiles + (t < 0 ? 0 : t);
    int base = 0;
    if (head && t >= 0) base = atomicAdd(slot, len);
    if (FILL) {
      base = __shfl(base, hp);
      if (live)
        W.list[(size_t)pl * W.list_cap + base + (lane - hp)] =
            make_uint4((unsigned)i, __float_as_uint(gx), __float_as_uint(gy), __float_as_uint(gl));
    }
  }
}

// one workgroup: per plane the exclusive scan of the tile counts (list offsets, fill cursors) and the work items -- a tile
// of n pairs becomes ceil(n / kTileItemCap) items of equal size
__device__ inline int wave_incl_scan_i(int v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(v, o);
    if (lane >= o) v += t;
  }
  return v;
}

__global__ __launch_bounds__(1024) void k_tile_scan(TileWs W, int nt0, int nt1, int nt2) {
  __shared__ int s_a[16], s_b[16];
  __shared__ int s_carry[2];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int pl = 0; pl < 3; ++pl) {
    const int nt = pl == 0 ? nt0 : (pl == 1 ? nt1 : nt2);
    if (tid == 0) s_carry[0] = s_carry[1] = 0;
    __syncthreads();
    for (int base = 0; base < nt; base += 1024) {
      const int i = base + tid;
      const int v = (i < nt) ? W.cnt[pl * kTileMaxTiles + i] : 0;
      const int ni = (v + kTileItemCap - 1) / kTileItemCap;
      const int sv = wave_incl_scan_i(v, lane), sn = wave_incl_scan_i(ni, lane);
      if (lane == 63) s_a[wv] = sv, s_b[wv] = sn;
      __syncthreads();
      int pa = s_carry[0], pb = s_carry[1];
      for (int k = 0; k < wv; ++k) pa += s_a[k], pb += s_b[k];
      const int off = pa + sv - v, io = pb + sn - ni;
      if (i < nt) {
        W.offs[pl * kTileMaxTiles + i] = off;
        W.cursor[pl * kTileMaxTiles + i] = off;
        for (int k = 0; k < ni; ++k) {
          const int b0 = off + (int)(((long long)v * k) / ni), b1 = off + (int)(((long long)v * (k + 1)) / ni);
          if (io + k < W.max_items) W.items[(size_t)pl * W.max_items + io + k] = make_int4(i, b0, b1, 0);
        }
      }
      __syncthreads();
      if (tid == 1023) {
        s_carry[0] = pa + sv;
        s_carry[1] = pb + sn;
      }
      __syncthreads();
    }
    if (tid == 0) W.ctl[pl] = min(s_carry[1], W.max_items);
    __syncthreads();
  }
}

// tap record of one pair, built by one lane, read back as broadcast LDS loads by the 16 channel lanes that handle the pair:
//   0..3   float offsets of the four taps inside the LDS tile (slice and staged values share them)
//   4..5   float offsets of the two line taps (LDS line and global line share them)
//   6      bits 0..5: tap is in range        7   the sample (chunk-local)
//   8..11  plane tap weights (zero when out of range)   12..13 line tap weights
//   14..15 d value / d ix = cxA (b - a) + cxB (d - c);   16..17 d value / d iy = cyA (c - a) + cyB (d - b)
//   18..21 byte offsets of the four plane taps in the factor itself (un-staged variant)
template <int TXC, int TYC>
__device__ inline void make_tile_rec(unsigned smp, float gx, float gy, float gl, int H, int W, int LL, int CA, int x0, int y0,
                                     float* rec) {
  constexpr int NX = TXC + 1;
  const PlaneTaps t = plane_taps(gx, gy, H, W, CA);
  const Axis l = axis_taps(gl, LL);
  const int lx0 = t.ax.c0 - x0, lx1 = t.ax.c1 - x0, ly0 = t.ay.c0 - y0, ly1 = t.ay.c1 - y0;
  const unsigned bits = (unsigned)(t.ax.m0 * t.ay.m0 != 0.f) | ((unsigned)(t.ax.m1 * t.ay.m0 != 0.f) << 1) |
                        ((unsigned)(t.ax.m0 * t.ay.m1 != 0.f) << 2) | ((unsigned)(t.ax.m1 * t.ay.m1 != 0.f) << 3) |
                        ((unsigned)(l.m0 != 0.f) << 4) | ((unsigned)(l.m1 != 0.f) << 5);
  *reinterpret_cast<int4*>(rec) =
      make_int4((ly0 * NX + lx0) * CA, (ly0 * NX + lx1) * CA, (ly1 * NX + lx0) * CA, (ly1 * NX + lx1) * CA);
  *reinterpret_cast<uint4*>(rec + 4) = make_uint4((unsigned)(l.c0 * CA), (unsigned)(l.c1 * CA), bits, smp);
  *reinterpret_cast<float4*>(rec + 8) = make_float4(t.w00, t.w10, t.w01, t.w11);
  *reinterpret_cast<float4*>(rec + 12) = make_float4(l.w0, l.w1, 1.f - t.ay.f, t.ay.f);
  *reinterpret_cast<float4*>(rec + 16) = make_float4(1.f - t.ax.f, t.ax.f, 0.f, 0.f);
  *reinterpret_cast<uint4*>(rec + 20) =
      make_uint4(4u * (unsigned)t.o00, 4u * (unsigned)t.o10, 4u * (unsigned)t.o01, 4u * (unsigned)t.o11);
}

template <class C, int TXC, int TYC, bool STAGE, int WAVES>
struct TileScatCfg {
  static constexpr int NCH = (C::CA + 15) / 16, KS = (C::APP + 3) / 4;
  static constexpr int NX = TXC + 1, NY = TYC + 1, NTEX = NX * NY;
  static constexpr int BOP_FLOATS = NCH * KS * 64;
  static constexpr int TILE_FLOATS = NTEX * C::CA;
  static size_t lds_bytes(int line_floats) {
    return (size_t)(BOP_FLOATS + line_floats + TILE_FLOATS * (STAGE ? 2 : 1) + WAVES * 16 * kTileRec + 4) * sizeof(float);
  }
};

template <class C, int TXC, int TYC, bool STAGE, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_tile_scatter(Dev D, MlpDev M, JtFactors G, TileWs W,
                                                             const float* __restrict__ rec, int line_floats) {
  typedef BwdCfg<C> B;
  typedef TileScatCfg<C, TXC, TYC, STAGE, WAVES> Q;
  constexpr int NCH = Q::NCH, KS = Q::KS, CA = C::CA, NX = Q::NX, NT = WAVES * 64;
  extern __shared__ __align__(16) float smem[];
  const int pl = (int)blockIdx.x % 3;
  const int nit = W.ctl[pl];
  if (nit == 0) return;
  const int tid = threadIdx.x;
  float* sbop = smem;
  float* sline = sbop + Q::BOP_FLOATS;
  float* tgrad = sline + line_floats;
  float* tval = tgrad + Q::TILE_FLOATS;
  const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* wrec = tgrad + Q::TILE_FLOATS * (STAGE ? 2 : 1) + wv * 16 * kTileRec;
  volatile int* s_item_p = reinterpret_cast<volatile int*>(tgrad + Q::TILE_FLOATS * (STAGE ? 2 : 1) + WAVES * 16 * kTileRec);
  const int H = D.ph[pl], Wd = D.pw[pl], LL = D.ll[pl];
  const int ntx = tiles_along(Wd, TXC);
  const float* P = D.aP[pl];
  const float* Ln = D.aL[pl];
  float* gP = G.app_plane[pl];
  for (int it = tid; it < Q::BOP_FLOATS; it += NT) {
    const int ln = it & 63, blk = it >> 6, k = blk % KS, c = blk / KS;
    const int a = 4 * k + (ln >> 4), ch = 16 * c + (ln & 15);
    sbop[it] = (a < C::APP && ch < CA) ? M.basis[a * C::NC + pl * CA + ch] : 0.f;
  }
  for (int i = tid; i < LL * CA; i += NT) sline[i] = 0.f;
  for (int i = tid; i < Q::TILE_FLOATS; i += NT) tgrad[i] = 0.f;
  const uint4* list = W.list + (size_t)pl * W.list_cap;
  float4* gxo = W.gx3 + (size_t)pl * W.list_cap;
  const size_t RC = B::REC_FLOATS;
  const int m0 = kM0(pl), m1 = kM1(pl), mv = kV(pl);
  const float sx = 0.5f * (float)(Wd - 1) * D.inv[m0], sy = 0.5f * (float)(H - 1) * D.inv[m1],
              sl = 0.5f * (float)(LL - 1) * D.inv[mv];
  for (;;) {
    if (tid == 0) *s_item_p = atomicAdd(&W.ctl[4 + pl], 1);
    __syncthreads();
    const int item_i = *s_item_p;
    if (item_i >= nit) break;
    const int4 item = W.items[(size_t)pl * W.max_items + item_i];
    const int tyi = item.x / ntx, txi = item.x - tyi * ntx;
    const int x0 = txi * TXC, y0 = tyi * TYC;
    if (STAGE) {
      constexpr int QPT = CA / 4;  // 16-byte pieces per texel
      for (int i = tid; i < Q::NTEX * QPT; i += NT) {
        const int texel = i / QPT, q = i - texel * QPT;
        const int ly = texel / NX, lx = texel - ly * NX;
        const int gx_ = min(x0 + lx, Wd - 1), gy_ = min(y0 + ly, H - 1);
        *reinterpret_cast<float4*>(tval + texel * CA + 4 * q) = ld4(P + ((size_t)gy_ * Wd + gx_) * CA + 4 * q);
      }
    }
    __syncthreads();
    const int ngroups = (item.z - item.y + 15) >> 4;
    for (int gi = wv; gi < ngroups; gi += WAVES) {
      int ln = lane;
      asm volatile("" : "+v"(ln));
      const int e0 = item.y + 16 * gi;
      const int nlive = min(16, item.z - e0);
      const int p = ln & 15;
      const uint4 ent = list[e0 + min(p, nlive - 1)];
      const int L = (int)ent.x;
      // A operand: row = pair p, the lane's K slice is basis row 4 k + grp
      float av[KS];
      {
        const float* rt = rec + (size_t)(L >> 5) * RC * 32 + (size_t)(B::R_GF + (ln >> 4)) * 32 + (L & 31);
#pragma unroll
        for (int k = 0; k < KS; ++k) {
          const float v = rec_ld(rt + 4 * k * 32);
          av[k] = (p < nlive) ? v : 0.f;
        }
      }
      if (ln < 16)
        make_tile_rec<TXC, TYC>(ent.x, __uint_as_float(ent.y), __uint_as_float(ent.z), __uint_as_float(ent.w), H, Wd, LL, CA,
                                x0, y0, wrec + p * kTileRec);
      f32x4 dv[NCH];
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        dv[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < KS; ++k)
          dv[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[k], sbop[(c * KS + k) * 64 + ln], dv[c], 0, 0, 0);
      }
      wave_lds_sync();
#if !(JT_TILE_ABL & 8)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int q = 4 * (ln >> 4) + i;
        const float* r = wrec + q * kTileRec;
        const int4 ro = *reinterpret_cast<const int4*>(r);
        const uint4 rl = *reinterpret_cast<const uint4*>(r + 4);
        const float4 w = *reinterpret_cast<const float4*>(r + 8);
        const float4 x = *reinterpret_cast<const float4*>(r + 12);  // lw0, lw1, cxA, cxB
        const float2 y = *reinterpret_cast<const float2*>(r + 16);  // cyA, cyB
        const unsigned bits = rl.z;
        float aix = 0.f, aiy = 0.f, ail = 0.f;
        // a tap outside the factor (a sample exactly on the far border) counts as zero: rare, tested for the whole wave
        auto body = [&](auto OOR) {
#pragma unroll
          for (int c = 0; c < NCH; ++c) {
            const bool livec = (CA % 16 == 0) || (16 * c + 15 < CA) || ((ln & 15) + 16 * c < CA);
            const int ch = livec ? (ln & 15) + 16 * c : 0;
            float a, b, cc, d;
            if (STAGE) {
              a = tval[ro.x + ch], b = tval[ro.y + ch], cc = tval[ro.z + ch], d = tval[ro.w + ch];
            } else {
              const uint4 go = *reinterpret_cast<const uint4*>(r + 20);
              a = ldb(P, go.x + 4u * ch), b = ldb(P, go.y + 4u * ch), cc = ldb(P, go.z + 4u * ch), d = ldb(P, go.w + 4u * ch);
            }
#if JT_TILE_ABL & 4
            float u = w.x, v = w.y;
#else
            float u = Ln[rl.x + ch], v = Ln[rl.y + ch];
#endif
            if (decltype(OOR)::value) {
              if (!(bits & 1u)) a = 0.f;
              if (!(bits & 2u)) b = 0.f;
              if (!(bits & 4u)) cc = 0.f;
              if (!(bits & 8u)) d = 0.f;
              if (!(bits & 16u)) u = 0.f;
              if (!(bits & 32u)) v = 0.f;
            }
            const float gch = livec ? dv[c][i] : 0.f;
            const float pv = w.x * a + w.y * b + w.z * cc + w.w * d;
            const float lv = x.x * u + x.y * v;
            const float gpv = gch * lv, glv = gch * pv;
            if (livec) {
#if !(JT_TILE_ABL & 1)
              atomicAdd(tgrad + ro.x + ch, w.x * gpv);
              atomicAdd(tgrad + ro.y + ch, w.y * gpv);
              atomicAdd(tgrad + ro.z + ch, w.z * gpv);
              atomicAdd(tgrad + ro.w + ch, w.w * gpv);
#else
              ail += w.x * gpv + w.y * gpv;
#endif
#if !(JT_TILE_ABL & 2)
              atomicAdd(sline + rl.x + ch, x.x * glv);
              atomicAdd(sline + rl.y + ch, x.y * glv);
#else
              ail += x.x * glv;
#endif
            }
            aix += gpv * (x.z * (b - a) + x.w * (d - cc));
            aiy += gpv * (y.x * (cc - a) + y.y * (d - b));
            ail += glv * (v - u);
          }
        };
        if (__builtin_amdgcn_ballot_w64((bits & 0x3fu) != 0x3fu) != 0ull) body(std::true_type{});
        else body(std::false_type{});
        aix = row16_sum(aix) * sx;
        aiy = row16_sum(aiy) * sy;
        ail = row16_sum(ail) * sl;
        if ((ln & 15) == 0 && q < nlive) {
          // by axis: plane 0 (x, y | z), plane 1 (x, z | y), plane 2 (y, z | x)
          const float vx = (pl == 2) ? ail : aix;
          const float vy = (pl == 0) ? aiy : ((pl == 1) ? ail : aix);
          const float vz = (pl == 0) ? ail : aiy;
          gxo[rl.w] = make_float4(vx, vy, vz, 0.f);
        }
      }
#else
      if (dv[0][0] == 123.f) gxo[0] = make_float4(dv[NCH - 1][3], 0.f, 0.f, 0.f);
#endif
      wave_lds_sync();
    }
    __syncthreads();
    // the slice leaves once: 64-byte-contiguous float atomics (halo texels are shared with the neighbouring tiles)
    for (int i = tid; i < Q::TILE_FLOATS; i += NT) {
      const float v = tgrad[i];
      if (v != 0.f) {
        const int texel = i / CA, c = i - texel * CA;
        const int ly = texel / NX, lx = texel - ly * NX;
        if (x0 + lx < Wd && y0 + ly < H) atomicAdd(gP + ((size_t)(y0 + ly) * Wd + (x0 + lx)) * CA + c, v);
        tgrad[i] = 0.f;
      }
    }
  }
  float* gl = G.app_line[pl];
  for (int i = tid; i < LL * CA; i += NT) {
    const float v = sline[i];
    if (v != 0.f) atomicAdd(gl + i, v);
  }
}

// g_xyz[e] = sum over the planes of the per-plane coordinate gradients of sample e
__global__ __launch_bounds__(256) void k_tile_gxyz(TileWs W, const int* __restrict__ offset, int R, float* __restrict__ g_xyz,
                                                   int chunk_start, int chunk_cap, int cap) {
  const int total = min(offset[R], cap);
  const int n_chunk = min(total - chunk_start, chunk_cap);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_chunk) return;
  const float4 a = W.gx3[i], b = W.gx3[(size_t)W.list_cap + i], c = W.gx3[2 * (size_t)W.list_cap + i];
  float* o = g_xyz + (size_t)(chunk_start + i) * 3;
  o[0] = a.x + b.x + c.x;
  o[1] = a.y + b.y + c.y;
  o[2] = a.z + b.z + c.z;
}

}  // namespace jt
