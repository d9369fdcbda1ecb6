// Tile-owned scatter of the appearance-factor gradients on the matrix cores (round 5).
//
// grid_sampler_2d_backward under autograd (bateRF.py:97-130) adds four plane taps and two line taps per (sample, plane)
// wherever the sample happens to lie; the run-length walkers of jt_walk.h cut that to ~5 float-atomic segments per
// (sample, plane) -- 20 M per launch at 400^3, 0.96 ms of the chip-wide float-atomic unit whatever issues them.
//
// Here the (sample, plane) pairs of a chunk are first BINNED by the 4 x 4-texel tile (3 x 3 cells) their bilinear footprint
// falls into: counting sort (count, scan, fill); consecutive samples of a ray stay in a tile for several steps, so a wave
// reserves list space once per run of equal tiles, not once per sample.  Then ONE WAVE owns a tile and keeps the tile's
// gradient slice (16 texels x CA channels) in MFMA accumulators: with W[pair][texel] the bilinear weight of a pair at a texel of
// the tile (zero outside its 2 x 2 footprint) the whole per-tap arithmetic of the backward is four small matrix products on
// v_mfma_f32_16x16x4_f32,
//     G   = GF basis^T            product gradients of 16 pairs            (as in k_shade_scatter)
//     PV  = W V,  DPX = Wx V,  DPY = Wy V      plane value and its coordinate derivatives (V: the tile's 16 texels)
//     S  += W^T (G * LV)          the gradient slice, accumulated in registers over all pairs of the tile
// and the slice leaves ONCE per tile with 64-byte-contiguous float atomics (~0.6 segments per pair instead of ~5).  What stays
// on the vector pipe is the line side (two gathered taps, the line gradient) and the three dot products over the channels
// that give the coordinate gradient.
//
// The LINE gradient of the plane is summed in LDS as DOUBLES: on gfx950 ds_add_f32 retires one lane every three cycles
// (193 cycles per wave instruction, tools/lds_atomic_rate.hip) while ds_add_f64 takes 9.  A 400 x 48 line of doubles is
// 154 KB, so a workgroup handles one plane and a CLASS of channel groups (VM-48: channels 0..31 or 32..47) and the workgroups
// are dealt to the classes in proportion to their work; all workgroups of a class pull items (tile, range of its list) from the
// class's counter.
#pragma once
#include "jt_shade_core.h"

#ifndef JT_TILE_ABL
#define JT_TILE_ABL 0  // profiling knob: 1 no slice MFMA / flush, 2 no line atomics, 4 no line-tap loads, 8 no coordinate gradients
#endif

#if JT_TILE_STAMP
// profiling build only (tools/build_variant.py -DJT_TILE_STAMP=1; tools/round5/stamp_tile.py): s_memtime ticks a wave of
// k_tile_scatter spends per phase of a group, summed over all waves; [15] counts the groups
__device__ unsigned long long g_tstamps[16];
#define JT_TS(k)                                               \
  {                                                            \
    const unsigned long long now__ = __builtin_readcyclecounter(); \
    ts_acc[k] += now__ - ts_last;                              \
    ts_last = now__;                                           \
  }
#else
#define JT_TS(k)
#endif

namespace jt {

constexpr int kTileCells = 3;           // cells per tile side: a 2 x 2 footprint that starts in the tile ends inside its 4 x 4 texels
constexpr int kTileMaxTiles = 131072;   // tiles per plane the workspace is sized for (a scene with more falls back)
constexpr int kTileItemCap = 512;       // pairs per work item: a heavier tile is split (each part flushes the slice)
constexpr int kTileRec = 12;            // words of a pair's record in LDS
constexpr int kTileGrab = 8;            // items a wave takes per visit to the class counter
constexpr int kTileMaxClasses = 8;

struct TileWs {
  int* cnt;       // [3][kTileMaxTiles] pairs per tile
  int* offs;      // [3][kTileMaxTiles] start of the tile's list (plane-relative)
  int* cursor;    // [3][kTileMaxTiles] fill position
  int4* items;    // [3][max_items] {tile, begin, end, -}
  int* ctl;       // [0..2] items per plane, [4 .. 4 + classes) next item per (plane, class)
  uint4* list;    // [3][list_cap] {sample (chunk-local), n[m0], n[m1], n[v] as float bits}
  float4* gx;     // [kTileMaxClasses][list_cap] coordinate gradients of a sample from one (plane, channel class), by axis
  int max_items, list_cap;
};

__host__ inline size_t tile_ws_bytes(int cap, int chunk) {
  const size_t lc = (size_t)std::min(std::max(cap, 1), chunk);
  const size_t max_items = kTileMaxTiles + (size_t)chunk / kTileItemCap + 1;
  return 3 * (size_t)kTileMaxTiles * 3 * sizeof(int) + 3 * max_items * sizeof(int4) + 64 * sizeof(int) + 3 * lc * sizeof(uint4) +
         kTileMaxClasses * lc * sizeof(float4) + 1024;
}
__host__ inline TileWs tile_ws_carve(void* base, int cap, int chunk) {
  TileWs w;
  const size_t lc = (size_t)std::min(std::max(cap, 1), chunk);
  const size_t max_items = kTileMaxTiles + (size_t)chunk / kTileItemCap + 1;
  char* p = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(base) + 255) & ~(uintptr_t)255);
  w.list = reinterpret_cast<uint4*>(p), p += 3 * lc * sizeof(uint4);
  w.gx = reinterpret_cast<float4*>(p), p += kTileMaxClasses * lc * sizeof(float4);
  w.items = reinterpret_cast<int4*>(p), p += 3 * max_items * sizeof(int4);
  w.cnt = reinterpret_cast<int*>(p), p += 3 * (size_t)kTileMaxTiles * sizeof(int);
  w.offs = reinterpret_cast<int*>(p), p += 3 * (size_t)kTileMaxTiles * sizeof(int);
  w.cursor = reinterpret_cast<int*>(p), p += 3 * (size_t)kTileMaxTiles * sizeof(int);
  w.ctl = reinterpret_cast<int*>(p);
  w.max_items = (int)max_items;
  w.list_cap = (int)lc;
  return w;
}

// tiles along an axis of `size` texels (size - 1 cells)
__host__ __device__ inline int tiles_along(int size) { return (size < 2 ? 0 : size - 2) / kTileCells + 1; }

__device__ inline int tile_of(float gx, float gy, int H, int W, int ntx) {
  const int cx = min(max(axis_cell(gx, W), 0), max(W - 2, 0));
  const int cy = min(max(axis_cell(gy, H), 0), max(H - 2, 0));
  return (cy / kTileCells) * ntx + cx / kTileCells;
}

__global__ void k_tile_zero(TileWs W, int nt0, int nt1, int nt2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int pl = blockIdx.y;
  const int nt = pl == 0 ? nt0 : (pl == 1 ? nt1 : nt2);
  if (i < nt) W.cnt[pl * kTileMaxTiles + i] = 0;
  if (i < 4 + kTileMaxClasses && pl == 0) W.ctl[i] = 0;
}

// FILL = false: pairs per tile; FILL = true: the lists.  One thread per shaded sample of the chunk; a wave's lanes hold
// consecutive samples, and a run of lanes with the same tile reserves its list space with ONE atomic (its first lane).
template <class C, bool FILL>
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
    const int ntx = tiles_along(D.pw[pl]);
    const float gx = n[kM0(pl)], gy = n[kM1(pl)], gl = n[kV(pl)];
    const int t = live ? tile_of(gx, gy, D.ph[pl], D.pw[pl], ntx) : -1;
    const int tp = __shfl_up(t, 1);
    const bool head = (lane == 0) || (t != tp);
    const unsigned long long hm = __ballot(head);
    const unsigned long long below = hm & ((2ull << lane) - 1ull);      // heads at or below this lane (lane 63: all)
    const int hp = 63 - __builtin_clzll(below);                         // this lane's run head
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

__device__ inline int wave_incl_scan_i(int v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(v, o);
    if (lane >= o) v += t;
  }
  return v;
}

// one workgroup per plane: the exclusive scan of the tile counts (list offsets, fill cursors) and the work items -- a tile
// of n pairs becomes ceil(n / kTileItemCap) items of equal size.  A thread takes kScanPer consecutive tiles per round.
constexpr int kScanPer = 8;
__global__ __launch_bounds__(1024) void k_tile_scan(TileWs W, int nt0, int nt1, int nt2) {
  __shared__ int s_a[16], s_b[16];
  __shared__ int s_carry[2];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int pl = blockIdx.x;
  const int nt = pl == 0 ? nt0 : (pl == 1 ? nt1 : nt2);
  if (tid == 0) s_carry[0] = s_carry[1] = 0;
  __syncthreads();
  for (int base = 0; base < nt; base += 1024 * kScanPer) {
    const int i0 = base + tid * kScanPer;
    int v[kScanPer], ni[kScanPer], tv = 0, tn = 0;
#pragma unroll
    for (int k = 0; k < kScanPer; ++k) {
      v[k] = (i0 + k < nt) ? W.cnt[pl * kTileMaxTiles + i0 + k] : 0;
      ni[k] = (v[k] + kTileItemCap - 1) / kTileItemCap;
      tv += v[k];
      tn += ni[k];
    }
    const int sv = wave_incl_scan_i(tv, lane), sn = wave_incl_scan_i(tn, lane);
    if (lane == 63) s_a[wv] = sv, s_b[wv] = sn;
    __syncthreads();
    int pa = s_carry[0], pb = s_carry[1];
    for (int k = 0; k < wv; ++k) pa += s_a[k], pb += s_b[k];
    int off = pa + sv - tv, io = pb + sn - tn;
#pragma unroll
    for (int k = 0; k < kScanPer; ++k) {
      if (i0 + k < nt) {
        W.offs[pl * kTileMaxTiles + i0 + k] = off;
        W.cursor[pl * kTileMaxTiles + i0 + k] = off;
        if (ni[k] == 1) {   // (the common case, without the 64-bit divisions of the general split)
          if (io < W.max_items) W.items[(size_t)pl * W.max_items + io] = make_int4(i0 + k, off, off + v[k], 0);
        } else {
          for (int j = 0; j < ni[k]; ++j) {
            const int b0 = off + (int)(((long long)v[k] * j) / ni[k]), b1 = off + (int)(((long long)v[k] * (j + 1)) / ni[k]);
            if (io + j < W.max_items) W.items[(size_t)pl * W.max_items + io + j] = make_int4(i0 + k, b0, b1, 0);
          }
        }
      }
      off += v[k];
      io += ni[k];
    }
    __syncthreads();
    if (tid == 1023) {
      s_carry[0] = pa + sv;
      s_carry[1] = pb + sn;
    }
    __syncthreads();
  }
  if (tid == 0) W.ctl[pl] = min(s_carry[1], W.max_items);
}

// weight of a tap pair (cell c0 -> texels c0, c0 + 1 with weights 1 - f, f) at position t of the tile, and its derivative
// with respect to the (fractional) texel coordinate
__device__ inline float tent(int t, int c0, float f) { return t == c0 ? 1.f - f : (t == c0 + 1 ? f : 0.f); }
__device__ inline float dtent(int t, int c0) { return t == c0 ? -1.f : (t == c0 + 1 ? 1.f : 0.f); }

template <class C, int CB0, int NCB, int WAVES>
struct TileScatCfg {
  static constexpr int KS = (C::APP + 3) / 4;
  static constexpr int CH = (16 * (CB0 + NCB) <= C::CA ? 16 * NCB : C::CA - 16 * CB0);  // channels of the class
};
__host__ inline size_t tile_lds_bytes(int line_len, int ch, int waves) {
  return (size_t)line_len * ch * sizeof(double) + (size_t)waves * (16 * kTileRec + 4 * kTileGrab) * sizeof(float);
}

// byte-offset addressing from a uniform base: (scalar base) + (32-bit lane offset) is what a global access takes as is
__device__ inline float ldf(const void* base, unsigned byte_off) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}

// one group of (up to) 16 pairs of a work item, as uniform values
struct TileGroup {
  int e0, nl, end, j, tile;
  bool first, last, valid;
};

// CB0 / NCB: first 16-channel group and number of groups this instantiation handles; `cls` indexes the class's counter and
// coordinate-gradient array
template <class C, int CB0, int NCB, int WAVES>
__device__ inline void tile_scatter_body(const Dev& D, const MlpDev& M, const JtFactors& G, const TileWs& W,
                                         const float* __restrict__ rec, int pl, int cls, float* smem) {
  typedef BwdCfg<C> B;
  typedef TileScatCfg<C, CB0, NCB, WAVES> Q;
  constexpr int KS = Q::KS, CA = C::CA, CH = Q::CH, NT = WAVES * 64;
  constexpr bool FULL = 16 * (CB0 + NCB) <= CA;  // every lane carries a channel in every group
  constexpr int kWaveLds = 16 * kTileRec + 4 * kTileGrab;  // floats: pair records + the batch's item descriptors
  const int nit = W.ctl[pl];
  const int tid = threadIdx.x;
  const int H = D.ph[pl], Wd = D.pw[pl], LL = D.ll[pl];
  double* sline = reinterpret_cast<double*>(smem);  // [line cell][CH]
  const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* wrec = reinterpret_cast<float*>(sline + (size_t)LL * CH) + wv * kWaveLds;
  int4* wdesc = reinterpret_cast<int4*>(wrec + 16 * kTileRec);
  for (int i = tid; i < LL * CH; i += NT) sline[i] = 0.0;
  __syncthreads();
  if (nit > 0) {
    const int ntx = tiles_along(Wd);
    const float* P = D.aP[pl];
    const float* Ln = D.aL[pl];
    float* gP = G.app_plane[pl];
    const uint4* list = W.list + (size_t)pl * W.list_cap;
    float4* gxo = W.gx + (size_t)cls * W.list_cap;
    const int4* items = W.items + (size_t)pl * W.max_items;
    const size_t RC = B::REC_FLOATS;
    const int m0 = kM0(pl), m1 = kM1(pl), mv = kV(pl);
    const float sx = 0.5f * (float)(Wd - 1) * D.inv[m0], sy = 0.5f * (float)(H - 1) * D.inv[m1],
                sl = 0.5f * (float)(LL - 1) * D.inv[mv];
    const int g = lane >> 4, n = lane & 15;
    // the lane's channel of group 0 (a partial last group -- VM-20's channels 16..19 -- parks its idle lanes on channel 0)
    const bool live_last = FULL || (16 * (CB0 + NCB - 1) + n < CA);
    const unsigned chb = 4u * (unsigned)(16 * CB0 + n);  // byte offset of the lane's channel of group 0 inside a texel
    auto chan_off = [&](int c) -> unsigned {             // ... of group c
      return (FULL || c + 1 < NCB || live_last) ? chb + 64u * (unsigned)c : 0u;
    };
    auto livec = [&](int c) -> bool { return FULL || c + 1 < NCB || live_last; };
    // B operand of G = GF basis^T, constant for the kernel: basis row 4 k + g, the lane's channel
    float bop[NCB][KS];
#pragma unroll
    for (int c = 0; c < NCB; ++c)
#pragma unroll
      for (int k = 0; k < KS; ++k) {
        const int a = 4 * k + g;
        bop[c][k] = (a < C::APP && livec(c)) ? ldf(M.basis, 4u * (unsigned)(a * C::NC + pl * CA) + chan_off(c)) : 0.f;
      }
    // ---- the wave's work: batches of kTileGrab items from the class counter; the groups of 16 pairs of a batch form ONE
    //      software pipeline (a group's list entries are fetched two groups ahead, its GF rows -- whose addresses come out of
    //      the entries -- one group ahead, the values of the next tile during the last group of the current one)
    auto load_entry = [&](const TileGroup& d) -> uint4 {
      return *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(list + d.e0) + 16u * (unsigned)min(n, d.nl - 1));
    };
    // (nothing may touch a prefetched value before its group is worked on: a select behind the load is a wait for it.  The
    //  padding pairs of a group's tail -- copies of its last pair -- are switched off through their WEIGHTS instead: zero
    //  slice weights, zero line weights, no coordinate-gradient store)
    auto load_gf = [&](const uint4& ent, float* av) {
      const unsigned L = ent.x;
      const float* rt = rec + (size_t)(L >> 5) * (RC * 32) + ((B::R_GF + g) * 32 + (L & 31u));
#pragma unroll
      for (int k = 0; k < KS; ++k) av[k] = rec_ld(rt + 4 * k * 32);
    };
    auto load_tile = [&](int tile, float (*V)[4]) {
      const int tyi = tile / ntx, txi = tile - tyi * ntx;
      const int x = txi * kTileCells + g;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int y = tyi * kTileCells + kk;
        const unsigned tb = 4u * (unsigned)((min(y, H - 1) * Wd + min(x, Wd - 1)) * CA);
#pragma unroll
        for (int c = 0; c < NCB; ++c) V[c][kk] = ldf(P, tb + chan_off(c));  // (texels outside the plane: zeroed at use)
      }
    };
#if JT_TILE_STAMP
    unsigned long long ts_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ts_last = __builtin_readcyclecounter(), ts_groups = 0;
#endif
    for (;;) {
      JT_TS(8)
      int first = 0;
      if (lane == 0) first = atomicAdd(&W.ctl[4 + cls], kTileGrab);
      first = __builtin_amdgcn_readfirstlane(first);
      if (first >= nit) break;
      const int nb = min(kTileGrab, nit - first);
      if (lane < nb) wdesc[lane] = items[first + lane];
      wave_lds_sync();
      auto item_group = [&](int j) -> TileGroup {   // first group of item j of the batch
        TileGroup d;
        d.j = j, d.valid = j < nb;
        const int4 it = wdesc[d.valid ? j : 0];
        d.tile = __builtin_amdgcn_readfirstlane(it.x);
        d.e0 = __builtin_amdgcn_readfirstlane(it.y);
        d.end = __builtin_amdgcn_readfirstlane(it.z);
        d.nl = min(16, d.end - d.e0);
        d.first = true, d.last = d.e0 + 16 >= d.end;
        return d;
      };
      auto next_group = [&](const TileGroup& d) -> TileGroup {
        if (d.last || !d.valid) return item_group(d.j + (d.valid ? 1 : 0));
        TileGroup r = d;
        r.e0 = d.e0 + 16, r.nl = min(16, d.end - r.e0), r.first = false, r.last = r.e0 + 16 >= d.end;
        return r;
      };
      TileGroup d0 = item_group(0), d1 = next_group(d0), d2 = next_group(d1);
      float Vb[NCB][4], Vn[NCB][4];
      f32x4 S[NCB];
      uint4 ent = load_entry(d0), ent1 = ent, ent2 = ent;
      float av[KS], av1[KS];
      load_tile(d0.tile, Vn);
      if (d1.valid) ent1 = load_entry(d1);
      load_gf(ent, av);
      int x0 = 0, y0 = 0;
      JT_TS(9)
      while (d0.valid) {
#if JT_TILE_STAMP
        ++ts_groups;
#endif
        // prefetches: entries of the group after next, GF rows of the next group, the values of the next item's tile
        if (d2.valid) ent2 = load_entry(d2);
        if (d1.valid) load_gf(ent1, av1);
        if (d0.first) {
          const int tyi = d0.tile / ntx, txi = d0.tile - tyi * ntx;
          x0 = txi * kTileCells, y0 = tyi * kTileCells;
#pragma unroll
          for (int c = 0; c < NCB; ++c) {
            S[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) Vb[c][kk] = (x0 + g < Wd && y0 + kk < H && livec(c)) ? Vn[c][kk] : 0.f;
          }
        }
        if (d0.last && d1.valid) load_tile(d1.tile, Vn);
        const int nlive = d0.nl;
        JT_TS(0)
        // mapping (a): the lane's pair is lane & 15 -- its cell inside the tile, fractions, line taps
        const Axis ax = axis_taps(__uint_as_float(ent.y), Wd);
        const Axis ay = axis_taps(__uint_as_float(ent.z), H);
        const Axis al = axis_taps(__uint_as_float(ent.w), LL);
        const int lx0 = ax.i0 - x0, ly0 = ay.i0 - y0;
        if (g == 0) {
          float* r = wrec + n * kTileRec;
          const bool dead = n >= nlive;  // a padding pair: off every texel of the tile, line weights zero
          *reinterpret_cast<float4*>(r) = make_float4(__int_as_float(dead ? -8 : lx0), __int_as_float(ly0), ax.f, ay.f);
          // line taps: byte offset of the cell in the factor's line, element offset in the LDS line; weights; masks
          *reinterpret_cast<float4*>(r + 4) = make_float4(__int_as_float(al.c0 * (CA * 4)), __int_as_float(al.c1 * (CA * 4)),
                                                          dead ? 0.f : al.w0, dead ? 0.f : al.w1);
          *reinterpret_cast<float4*>(r + 8) = make_float4(al.m0, al.m1, __int_as_float((int)ent.x * 16),
                                                          __int_as_float(al.c0 * CH + (al.c1 * CH << 16)));
        }
        // A operands of PV / DPX / DPY: row = pair, MFMA kk, K slice g  <->  texel (column g, row kk) of the tile
        const float wxg = tent(g, lx0, ax.f), dxg = dtent(g, lx0);
        float Aw[4], Ax[4], Ay[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const float wy = tent(kk, ly0, ay.f), dy = dtent(kk, ly0);
          Aw[kk] = wxg * wy;
          Ax[kk] = dxg * wy;
          Ay[kk] = wxg * dy;
        }
        JT_TS(1)
        wave_lds_sync();
        // mapping (b): lane (g, n) carries channel n of pairs 4 g + i in register i
        float lw0[4], lw1[4], lm0[4], lm1[4];
        unsigned lb0[4], lb1[4], ls0[4], ls1[4], gxb[4];
        float At[4];  // A operand of the slice product i: row = texel lane & 15 of the tile, K slice g <-> pair 4 g + i
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float* r = wrec + (4 * g + i) * kTileRec;
          const float4 r0 = *reinterpret_cast<const float4*>(r);
          const float4 r1 = *reinterpret_cast<const float4*>(r + 4);
          const float4 r2 = *reinterpret_cast<const float4*>(r + 8);
          lb0[i] = __float_as_uint(r1.x), lb1[i] = __float_as_uint(r1.y);
          lw0[i] = r1.z, lw1[i] = r1.w, lm0[i] = r2.x, lm1[i] = r2.y;
          gxb[i] = __float_as_uint(r2.z);
          const unsigned sp = __float_as_uint(r2.w);
          ls0[i] = 8u * ((sp & 0xffffu) + (unsigned)n), ls1[i] = 8u * ((sp >> 16) + (unsigned)n);
          At[i] = tent(n & 3, __float_as_int(r0.x), r0.z) * tent(n >> 2, __float_as_int(r0.y), r0.w);
        }
        JT_TS(2)
        // the line taps of all channel groups go out before the matrix work that does not need them
        float lu[NCB][4], lv_[NCB][4];
#pragma unroll
        for (int c = 0; c < NCB; ++c)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
#if JT_TILE_ABL & 4
            lu[c][i] = lw0[i], lv_[c][i] = lw1[i];
#else
            lu[c][i] = ldf(Ln, lb0[i] + chan_off(c)), lv_[c][i] = ldf(Ln, lb1[i] + chan_off(c));
#endif
          }
        JT_TS(3)
        float aix[4] = {0.f, 0.f, 0.f, 0.f}, aiy[4] = {0.f, 0.f, 0.f, 0.f}, ail[4] = {0.f, 0.f, 0.f, 0.f};
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NCB; ++c) {
          f32x4 Gd = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], bop[c][0], zero4, 0, 0, 0);
#pragma unroll
          for (int k = 1; k < KS; ++k) Gd = __builtin_amdgcn_mfma_f32_16x16x4f32(av[k], bop[c][k], Gd, 0, 0, 0);
          f32x4 PV = __builtin_amdgcn_mfma_f32_16x16x4f32(Aw[0], Vb[c][0], zero4, 0, 0, 0);
#if !(JT_TILE_ABL & 8)
          f32x4 DPX = __builtin_amdgcn_mfma_f32_16x16x4f32(Ax[0], Vb[c][0], zero4, 0, 0, 0);
          f32x4 DPY = __builtin_amdgcn_mfma_f32_16x16x4f32(Ay[0], Vb[c][0], zero4, 0, 0, 0);
#else
          f32x4 DPX = zero4, DPY = zero4;
#endif
#pragma unroll
          for (int kk = 1; kk < 4; ++kk) {
            PV = __builtin_amdgcn_mfma_f32_16x16x4f32(Aw[kk], Vb[c][kk], PV, 0, 0, 0);
#if !(JT_TILE_ABL & 8)
            DPX = __builtin_amdgcn_mfma_f32_16x16x4f32(Ax[kk], Vb[c][kk], DPX, 0, 0, 0);
            DPY = __builtin_amdgcn_mfma_f32_16x16x4f32(Ay[kk], Vb[c][kk], DPY, 0, 0, 0);
#endif
          }
          float gpv[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float u = lu[c][i], v = lv_[c][i];
            const float lv = lw0[i] * u + lw1[i] * v;
            const float dl = lm1[i] * v - lm0[i] * u;  // an out-of-range tap counts as zero
            gpv[i] = Gd[i] * lv;
            const float glv = Gd[i] * PV[i];
            aix[i] += gpv[i] * DPX[i];
            aiy[i] += gpv[i] * DPY[i];
            ail[i] += glv * dl;
#if !(JT_TILE_ABL & 2)
            if (livec(c)) {
              atomicAdd(reinterpret_cast<double*>(reinterpret_cast<char*>(sline) + ls0[i] + 128u * c), (double)(lw0[i] * glv));
              atomicAdd(reinterpret_cast<double*>(reinterpret_cast<char*>(sline) + ls1[i] + 128u * c), (double)(lw1[i] * glv));
            }
#else
            ail[i] += lw0[i] * glv;
#endif
          }
#if !(JT_TILE_ABL & 1)
#pragma unroll
          for (int i = 0; i < 4; ++i) S[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(At[i], gpv[i], S[c], 0, 0, 0);
#else
          S[c][0] += gpv[0] + gpv[1] + gpv[2] + gpv[3];
#endif
        }
        JT_TS(4)
#if !(JT_TILE_ABL & 8)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float vx_ = row16_sum(aix[i]) * sx, vy_ = row16_sum(aiy[i]) * sy, vl_ = row16_sum(ail[i]) * sl;
          if (n == 0 && 4 * g + i < nlive) {
            // by axis: plane 0 (x, y | z), plane 1 (x, z | y), plane 2 (y, z | x)
            const float vx = (pl == 2) ? vl_ : vx_;
            const float vy = (pl == 0) ? vy_ : ((pl == 1) ? vl_ : vx_);
            const float vz = (pl == 0) ? vl_ : vy_;
            *reinterpret_cast<float4*>(reinterpret_cast<char*>(gxo) + gxb[i]) = make_float4(vx, vy, vz, 0.f);
          }
        }
#endif
        JT_TS(5)
        wave_lds_sync();
        // the slice leaves once: register i of lane (g, n) is texel (column i, row g) of the tile, channel n
        if (d0.last) {
#if !(JT_TILE_ABL & 1)
          const int y = y0 + g;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int x = x0 + i;
            const unsigned tb = 4u * (unsigned)((y * Wd + x) * CA);
            const bool in = x < Wd && y < H;
#pragma unroll
            for (int c = 0; c < NCB; ++c) {
              const float v = S[c][i];
              if (in && livec(c) && v != 0.f)
                atomicAdd(reinterpret_cast<float*>(reinterpret_cast<char*>(gP) + tb + chan_off(c)), v);
            }
          }
#else
          if (S[0][0] == 123.f) gxo[0] = make_float4(S[NCB - 1][0], 0.f, 0.f, 0.f);
#endif
        }
        JT_TS(6)
        ent = ent1, ent1 = ent2;
#pragma unroll
        for (int k = 0; k < KS; ++k) av[k] = av1[k];
        d0 = d1, d1 = d2, d2 = next_group(d2);
        JT_TS(7)
      }
      wave_lds_sync();
    }
#if JT_TILE_STAMP
    if (lane == 0) {
      for (int k = 0; k < 10; ++k) atomicAdd(&g_tstamps[k], ts_acc[k]);
      atomicAdd(&g_tstamps[15], ts_groups);
    }
#endif
  }
  __syncthreads();
  float* gl = G.app_line[pl];
  for (int i = tid; i < LL * CH; i += NT) {
    const double v = sline[i];
    const int cell = i / CH, ch = i - cell * CH;
    if (v != 0.0) atomicAdd(gl + (size_t)cell * CA + 16 * CB0 + ch, (float)v);
  }
}

// launch layout: workgroup b belongs to the (plane, class) set k with start[k] <= b < start[k + 1]
struct TileClasses {
  int start[kTileMaxClasses + 1];
};

// SPLIT = 0: one class (all channel groups); SPLIT = s: two classes, groups [0, s) and [s, NG)
template <class C, int SPLIT, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_tile_scatter(Dev D, MlpDev M, JtFactors G, TileWs W,
                                                             const float* __restrict__ rec, TileClasses TC) {
  extern __shared__ __align__(16) float smem[];
  constexpr int NG = (C::CA + 15) / 16;
  constexpr int NCLS = (SPLIT > 0 && SPLIT < NG) ? 2 : 1;
  int k = 0;
#pragma unroll
  for (int j = 1; j < 3 * NCLS; ++j) k += ((int)blockIdx.x >= TC.start[j]) ? 1 : 0;
  const int pl = k / NCLS, cl = k - pl * NCLS;
  if constexpr (NCLS == 1) {
    tile_scatter_body<C, 0, NG, WAVES>(D, M, G, W, rec, pl, k, smem);
  } else {
    if (cl == 0) tile_scatter_body<C, 0, SPLIT, WAVES>(D, M, G, W, rec, pl, k, smem);
    else tile_scatter_body<C, SPLIT, NG - SPLIT, WAVES>(D, M, G, W, rec, pl, k, smem);
  }
}

// g_xyz[e] = sum over the (plane, class) sets of the coordinate gradients of sample e
__global__ __launch_bounds__(256) void k_tile_gxyz(TileWs W, int nsets, const int* __restrict__ offset, int R,
                                                   float* __restrict__ g_xyz, int chunk_start, int chunk_cap, int cap) {
  const int total = min(offset[R], cap);
  const int n_chunk = min(total - chunk_start, chunk_cap);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_chunk) return;
  float4 a = W.gx[i];
  for (int s = 1; s < nsets; ++s) {
    const float4 b = W.gx[(size_t)s * W.list_cap + i];
    a.x += b.x, a.y += b.y, a.z += b.z;
  }
  float* o = g_xyz + (size_t)(chunk_start + i) * 3;
  o[0] = a.x, o[1] = a.y, o[2] = a.z;
}

}  // namespace jt
