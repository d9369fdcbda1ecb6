// Run-length scatter of factor gradients along a ray.
//
// Samples of one ray are half a voxel apart, so consecutive samples hit the same plane cell / line
// segment several times.  A 16-lane group (lane = channel) walks a run of consecutive samples IN ORDER and
// keeps the gradient of the four corners of the current plane cell and of the two taps of the current
// line segment in registers.  A texel is written with ONE float atomic per channel when the walk leaves
// it; a step to an edge-adjacent cell keeps the two shared corners.  This cuts the atomic traffic of
// grid_sampler_2d_backward (one atomic per tap per sample) by the run length (measured 3-7x), and every
// atomic wave-instruction touches 64-byte-contiguous channel vectors.  How the walk is organised (step
// records, parity slots) is described below.
#pragma once
#include "jt_common.h"

// an accumulator that never received anything needs no atomic
#define JT_FLUSH_COND(x) ((x) != 0.f)

// profiling knob (tools/build_variant.py -DJT_ABL_WALK_ATOMICS=1 / 2): the flush of a walker as nothing at all (1) or as a
// plain store of the same value to the same address (2) instead of the float atomic -- what the atomic unit costs a scatter
#if JT_ABL_WALK_ATOMICS == 1
#define JT_WALK_ATOMIC(p, v) asm volatile("" ::"v"(p), "v"(v))
#elif JT_ABL_WALK_ATOMICS == 2
#define JT_WALK_ATOMIC(p, v) (*(p) = (v))
#else
#define JT_WALK_ATOMIC(p, v) atomicAdd(p, v)
#endif

namespace jt {

// ---------------------------------------------------------------------------------------------
// Step records.  The geometry of a (sample, plane) pair -- tap addresses, weights, which texels the walk
// leaves -- is the same for every channel, so it is computed ONCE by one lane (make_step_rec) and parked in
// LDS as kRecWords words; the channel lanes of the walker read it back as broadcast LDS loads instead of
// redoing ~100 VALU instructions of index arithmetic and cell bookkeeping per lane.
//
// Accumulator slots.  The four texels of a plane cell are kept in four accumulators indexed by the PARITY of
// the texel coordinates, slot = (X & 1) + 2 (Y & 1) (line: slot = Z & 1).  The texels of any cell have
// distinct parities, and when the walk steps to a neighbouring cell the texels that stay keep their slot --
// nothing is ever moved between accumulators; the slots whose texel is left behind are flushed (one float
// atomic per channel) and start over for the texel that takes their place.  Which slots to flush follows
// from the previous sample's cell, so the record carries it as a bit mask.
//   word 0..3   byte offsets of the texels of the plane cell in slot order, clamped into the plane
//        4..5   byte offsets of the two line taps in slot order, clamped
//        6      bits 0..3 / 4..5: plane / line slot holds an in-range texel;
//               bits 8..11 / 12..13: plane / line slot must be flushed before this sample is accumulated
//        7      +-1: sign that turns (line slot 1 - line slot 0) into (tap 1 - tap 0)
//        8..11  tap weights in slot order (zero when the tap is out of range), 12..13 line tap weights
//        14..17 coefficients of d(plane value)/d(ix, iy) in slot terms:
//               d/dix = cxA (s1 - s0) + cxB (s3 - s2),  d/diy = cyA (s2 - s0) + cyB (s3 - s1)
// ---------------------------------------------------------------------------------------------
constexpr int kRecWords = 20;

template <class T>
__device__ inline void swap_if(T& a, T& b, bool c) {
  const T t = c ? b : a;
  b = c ? a : b;
  a = t;
}

// cell of a normalised coordinate along an axis of `size` texels (the i0 of axis_taps)
__device__ inline int axis_cell(float g, int size) { return (int)floorf(((g + 1.f) * 0.5f) * (float)(size - 1)); }

// (gx, gy, gl): normalised coordinates of the sample on the plane axes / the line axis; (pgx, pgy, pgl) those
// of the previous sample of the run (has_prev = false for the first sample of a run: nothing to flush)
__device__ inline void make_step_rec(float gx, float gy, float gl, float pgx, float pgy, float pgl, bool has_prev,
                                     int H, int W, int LL, int CA, float* rec) {
  const PlaneTaps t = plane_taps(gx, gy, H, W, CA);
  const Axis l = axis_taps(gl, LL);
  const int cx = t.ax.i0, cy = t.ay.i0, cz = l.i0;
  const bool px = cx & 1, py = cy & 1, pz = cz & 1;
  unsigned o0 = (unsigned)t.o00 * 4u, o1 = (unsigned)t.o10 * 4u, o2 = (unsigned)t.o01 * 4u, o3 = (unsigned)t.o11 * 4u;
  float w0 = t.w00, w1 = t.w10, w2 = t.w01, w3 = t.w11;
  unsigned b0 = (t.ax.m0 * t.ay.m0 != 0.f), b1 = (t.ax.m1 * t.ay.m0 != 0.f), b2 = (t.ax.m0 * t.ay.m1 != 0.f),
           b3 = (t.ax.m1 * t.ay.m1 != 0.f);
  // slot sx + 2 sy holds corner (sx ^ px) + 2 (sy ^ py)
  swap_if(o0, o1, px); swap_if(o2, o3, px); swap_if(w0, w1, px); swap_if(w2, w3, px); swap_if(b0, b1, px); swap_if(b2, b3, px);
  swap_if(o0, o2, py); swap_if(o1, o3, py); swap_if(w0, w2, py); swap_if(w1, w3, py); swap_if(b0, b2, py); swap_if(b1, b3, py);
  unsigned l0 = (unsigned)(l.c0 * CA) * 4u, l1 = (unsigned)(l.c1 * CA) * 4u;
  float lw0 = l.w0, lw1 = l.w1;
  unsigned lb0 = (l.m0 != 0.f), lb1 = (l.m1 != 0.f);
  swap_if(l0, l1, pz); swap_if(lw0, lw1, pz); swap_if(lb0, lb1, pz);
  unsigned bits = b0 | (b1 << 1) | (b2 << 2) | (b3 << 3) | (lb0 << 4) | (lb1 << 5);
  if (has_prev) {
    const int qx = axis_cell(pgx, W), qy = axis_cell(pgy, H), qz = axis_cell(pgl, LL);
    // texel of parity s in the previous cell: q + ((s - q) & 1); it stays iff it is c or c + 1
    const int x0 = qx + ((0 - qx) & 1), x1 = qx + ((1 - qx) & 1);
    const int y0 = qy + ((0 - qy) & 1), y1 = qy + ((1 - qy) & 1);
    const int z0 = qz + ((0 - qz) & 1), z1 = qz + ((1 - qz) & 1);
    const unsigned fx0 = (x0 != cx) && (x0 != cx + 1), fx1 = (x1 != cx) && (x1 != cx + 1);
    const unsigned fy0 = (y0 != cy) && (y0 != cy + 1), fy1 = (y1 != cy) && (y1 != cy + 1);
    const unsigned fz0 = (z0 != cz) && (z0 != cz + 1), fz1 = (z1 != cz) && (z1 != cz + 1);
    bits |= ((fx0 | fy0) << 8) | ((fx1 | fy0) << 9) | ((fx0 | fy1) << 10) | ((fx1 | fy1) << 11) | (fz0 << 12) |
            (fz1 << 13);
  }
  const float wy0 = 1.f - t.ay.f, wy1 = t.ay.f, wx0 = 1.f - t.ax.f, wx1 = t.ax.f;
  const float sgx = px ? -1.f : 1.f, sgy = py ? -1.f : 1.f;
  *reinterpret_cast<uint4*>(rec) = make_uint4(o0, o1, o2, o3);
  *reinterpret_cast<uint4*>(rec + 4) = make_uint4(l0, l1, bits, __float_as_uint(pz ? -1.f : 1.f));
  *reinterpret_cast<float4*>(rec + 8) = make_float4(w0, w1, w2, w3);
  *reinterpret_cast<float4*>(rec + 12) = make_float4(lw0, lw1, sgx * (py ? wy1 : wy0), sgx * (py ? wy0 : wy1));
  *reinterpret_cast<float2*>(rec + 16) = make_float2(sgy * (px ? wx1 : wx0), sgy * (px ? wx0 : wx1));
}

// factor values at the six taps of one (sample, plane), slot order, for the lane's channels cl, cl+16, ...
template <int NCH>
struct TapBuf {
  float a[NCH], b[NCH], c[NCH], d[NCH], u[NCH], v[NCH];
};

__device__ inline float ldb(const float* base, unsigned byte_off) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}

// sum over the 16 lanes of a DPP row, result in every lane: four v_add_f32 with DPP operands
__device__ inline float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));
  return v;
}

// Run-length accumulator of one plane + its line, driven by step records.
// FX: 0 = float atomics, 1 = JT_DETERMINISTIC fixed point, 2 = chosen at run time by init()'s flag (a kernel that is
// instantiated once; the density walk, short of registers, is instantiated per mode)
// LDSL: the LINE gradients are added into a workgroup-private copy of the line in LDS (gL points there; LDS atomics, the owner
// adds the copy into the real gradient once) instead of going out as global atomics.  1: a float copy (ds_add_f32);
// 2: a copy of DOUBLES, same element indexing (ds_add_f64) -- on gfx950 ds_add_f32 retires one lane every three cycles
// (193 cycles per full wave instruction) while ds_add_f64 takes 9 (tools/lds_atomic_rate.hip), so the double copy is the one
// to use wherever twice the bytes fit
template <int NCH, int CA, int FX = 2, int LDSL = 0>
struct RecWalker {
  float acc[4][NCH];   // plane accumulators, parity slots
  float accl[2][NCH];  // line accumulators, parity slots
  unsigned o[4], lo[2];  // byte offsets of the texels the slots currently stand for
  unsigned ck[NCH];      // byte offset of the lane's k-th channel inside a texel (0 for a padding lane)
  bool live[NCH];
  float* gP;
  float* gL;
  bool fixed;  // JT_DETERMINISTIC: gP / gL are int64 shadow buffers (same element indexing), sums in 2^48 fixed point
  unsigned* bad;  // the library's sticky flag for fixed-point addends out of range (jt_common.h: fixed_add)

  __device__ inline void init(float* gP_, float* gL_, int cl, bool fixed_ = false, unsigned* bad_ = nullptr) {
    gP = gP_;
    gL = gL_;
    fixed = fixed_;
    bad = bad_;
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c] = 0u;
    lo[0] = lo[1] = 0u;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      live[k] = (CA % 16 == 0) || (cl + 16 * k < CA);
      ck[k] = live[k] ? 4u * (unsigned)(cl + 16 * k) : 0u;
      acc[0][k] = acc[1][k] = acc[2][k] = acc[3][k] = 0.f;
      accl[0][k] = accl[1][k] = 0.f;
    }
  }
  __device__ inline void load(TapBuf<NCH>& tv, const float* P, const float* L, const float* rec) const {
    const uint4 ro = *reinterpret_cast<const uint4*>(rec);
    const uint2 rl = *reinterpret_cast<const uint2*>(rec + 4);
    // channel groups k < CA / 16 exist in every lane: ONE offset add per tap, the 64-byte step between groups rides in
    // the load's immediate offset; a partial last group (VM-20: channels 16..19) keeps its own clamped offset
    constexpr int NFULL = CA / 16;
    const float* pa = reinterpret_cast<const float*>(reinterpret_cast<const char*>(P) + (ro.x + ck[0]));
    const float* pb = reinterpret_cast<const float*>(reinterpret_cast<const char*>(P) + (ro.y + ck[0]));
    const float* pc = reinterpret_cast<const float*>(reinterpret_cast<const char*>(P) + (ro.z + ck[0]));
    const float* pd = reinterpret_cast<const float*>(reinterpret_cast<const char*>(P) + (ro.w + ck[0]));
    const float* pu = reinterpret_cast<const float*>(reinterpret_cast<const char*>(L) + (rl.x + ck[0]));
    const float* pv = reinterpret_cast<const float*>(reinterpret_cast<const char*>(L) + (rl.y + ck[0]));
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      if (k < NFULL) {
        tv.a[k] = pa[16 * k];
        tv.b[k] = pb[16 * k];
        tv.c[k] = pc[16 * k];
        tv.d[k] = pd[16 * k];
        tv.u[k] = pu[16 * k];
        tv.v[k] = pv[16 * k];
      } else {
        tv.a[k] = ldb(P, ro.x + ck[k]);
        tv.b[k] = ldb(P, ro.y + ck[k]);
        tv.c[k] = ldb(P, ro.z + ck[k]);
        tv.d[k] = ldb(P, ro.w + ck[k]);
        tv.u[k] = ldb(L, rl.x + ck[k]);
        tv.v[k] = ldb(L, rl.y + ck[k]);
      }
    }
  }
  // out-of-range texels never receive anything (their weights are zero), so a clamped address is fine.
  // base == nullptr: the caller does not want factor gradients (pose-only backward), nothing is written.
  __device__ inline void flush_lds(float* base, unsigned off, float* a) {
    bool any = false;
#pragma unroll
    for (int k = 0; k < NCH; ++k) any = any || JT_FLUSH_COND(a[k]);
    if (any) {
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        if (!live[k]) continue;
        if (LDSL == 2) atomicAdd(reinterpret_cast<double*>(reinterpret_cast<char*>(base) + 2u * (off + ck[k])), (double)a[k]);
        else atomicAdd(reinterpret_cast<float*>(reinterpret_cast<char*>(base) + (off + ck[k])), a[k]);
      }
    }
#pragma unroll
    for (int k = 0; k < NCH; ++k) a[k] = 0.f;
  }
  __device__ inline void flush_line(unsigned off, float* a) {
#if JT_ABL_WALK_LINE  // profiling knob: the line gradients are dropped -- what the LINE flushes cost a walker
#pragma unroll
    for (int k = 0; k < NCH; ++k) a[k] = 0.f;
    return;
#endif
    if (LDSL) flush_lds(gL, off, a);
    else flush(gL, off, a);
  }
  __device__ inline void flush(float* base, unsigned off, float* a) {
    // (full channel groups, k < CA / 16: the texel's byte offset plus the lane's first channel once, the 64-byte group
    //  step as an immediate offset of the atomic)
    float* t0 = reinterpret_cast<float*>(reinterpret_cast<char*>(base) + (off + ck[0]));
    // ONE test per slot: a slot that received anything has (with measure-zero exceptions) all of the lane's channels
    // non-zero, and adding an exact zero changes nothing -- a test per channel was a compare, an exec-mask save and a
    // branch around every atomic
    bool any = false;
#pragma unroll
    for (int k = 0; k < NCH; ++k) any = any || JT_FLUSH_COND(a[k]);
    if (base != nullptr && any) {
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        if (!live[k]) continue;
        if (FX == 1 || (FX == 2 && fixed))  // byte offset of a float element -> the same element of the 64-bit shadow buffer
          fixed_add(reinterpret_cast<long long*>(reinterpret_cast<char*>(base) + 2 * (size_t)(off + ck[k])), a[k], bad);
        else if (k < CA / 16)
          JT_WALK_ATOMIC(t0 + 16 * k, a[k]);
        else {
#if !JT_ABL_WALK_TAIL   // profiling knob: the partial last channel group (20 channels: 16 + 4) is dropped -- what its atomics cost
          JT_WALK_ATOMIC(reinterpret_cast<float*>(reinterpret_cast<char*>(base) + (off + ck[k])), a[k]);
#endif
        }
      }
    }
#pragma unroll
    for (int k = 0; k < NCH; ++k) a[k] = 0.f;
  }
  // flush the slots whose texel the walk leaves with this sample, then adopt the sample's texels
  __device__ inline void advance(const float* rec) {
    if (gP == nullptr) return;  // pose-only backward: no accumulators to keep
    const uint4 rl = *reinterpret_cast<const uint4*>(rec + 4);
    const unsigned bits = rl.z;
    if (bits & 0x3f00u) {
      if (bits & 0x100u) flush(gP, o[0], acc[0]);
      if (bits & 0x200u) flush(gP, o[1], acc[1]);
      if (bits & 0x400u) flush(gP, o[2], acc[2]);
      if (bits & 0x800u) flush(gP, o[3], acc[3]);
      if (bits & 0x1000u) flush_line(lo[0], accl[0]);
      if (bits & 0x2000u) flush_line(lo[1], accl[1]);
    }
    const uint4 ro = *reinterpret_cast<const uint4*>(rec);
    o[0] = ro.x;
    o[1] = ro.y;
    o[2] = ro.z;
    o[3] = ro.w;
    lo[0] = rl.x;
    lo[1] = rl.y;
  }
  // accumulate one sample.  g[k] = dL/d(plane_c * line_c) for the lane's channels (zero for padding lanes).
  // Returns the lane's UN-reduced partials of dL/d(ix, iy, il) (grid_sampler backward w.r.t. the coordinates:
  // out-of-range taps count as zeros).
  // PROD: also hands back the lane's plane x line products of the sample, prod[k] = pv * lv (what the forward fed basis_mat:
  // k_shade_scatter forms dBasis from them, so the forward does not have to record them)
  template <bool PROD = false>
  __device__ inline void add(TapBuf<NCH>& tv, const float* rec, const float g[NCH], float& aix, float& aiy,
                             float& ail, float* prod = nullptr) {
#if JT_ABL_WALK_PAD  // profiling knob: N extra vector instructions per step -- is a walker bound by instruction issue?
    {
      float pad = aix;
#pragma unroll
      for (int i = 0; i < JT_ABL_WALK_PAD; ++i) asm volatile("v_add_f32 %0, %0, %0" : "+v"(pad));
      asm volatile("" ::"v"(pad));
    }
#endif
    const float4 w = *reinterpret_cast<const float4*>(rec + 8);
    const float4 x = *reinterpret_cast<const float4*>(rec + 12);  // lw0, lw1, cxA, cxB
    const float2 y = *reinterpret_cast<const float2*>(rec + 16);  // cyA, cyB
    const unsigned bits = reinterpret_cast<const unsigned*>(rec)[6];
    const float sgz = rec[7];
    // a tap outside the factor (a sample exactly on the far border): its value counts as zero.  Rare, and tested for the
    // whole wave first: as a per-lane condition the compiler turns it into 6 NCH selects on EVERY step
    if (__builtin_amdgcn_ballot_w64((bits & 0x3fu) != 0x3fu) != 0ull) {
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        if (!(bits & 1u)) tv.a[k] = 0.f;
        if (!(bits & 2u)) tv.b[k] = 0.f;
        if (!(bits & 4u)) tv.c[k] = 0.f;
        if (!(bits & 8u)) tv.d[k] = 0.f;
        if (!(bits & 16u)) tv.u[k] = 0.f;
        if (!(bits & 32u)) tv.v[k] = 0.f;
      }
    }
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const float pv = w.x * tv.a[k] + w.y * tv.b[k] + w.z * tv.c[k] + w.w * tv.d[k];
      const float lv = x.x * tv.u[k] + x.y * tv.v[k];
      if (PROD) prod[k] = pv * lv;
      const float gpv = g[k] * lv, glv = g[k] * pv;
      // (unconditionally: without factor gradients -- gP == nullptr, pose-only backward -- the slots are never flushed, and
      //  a test here comes back as one select per accumulator and step, 18 of the ~80 arithmetic instructions of a step)
      acc[0][k] += w.x * gpv;
      acc[1][k] += w.y * gpv;
      acc[2][k] += w.z * gpv;
      acc[3][k] += w.w * gpv;
      accl[0][k] += x.x * glv;
      accl[1][k] += x.y * glv;
      aix += gpv * (x.z * (tv.b[k] - tv.a[k]) + x.w * (tv.d[k] - tv.c[k]));
      aiy += gpv * (y.x * (tv.c[k] - tv.a[k]) + y.y * (tv.d[k] - tv.b[k]));
      ail += glv * (sgz * (tv.v[k] - tv.u[k]));
    }
  }
  // Two 16-lane groups whose runs END on neighbouring samples (lanes l and l ^ 16) meet before they flush: a parity
  // slot that stands for the same texel in both -- all four when they end in the same cell, two for edge-adjacent
  // cells -- is handed from the even group to the odd one, which writes the sum with one atomic instead of two.
  __device__ inline void finish_pair(int grp) {
    if (gP == nullptr) return;
    const bool taker = grp & 1;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const bool same = (unsigned)__shfl_xor((int)o[c], 16) == o[c];
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        const float pa = __shfl_xor(acc[c][k], 16);
        acc[c][k] = same ? (taker ? acc[c][k] + pa : 0.f) : acc[c][k];
      }
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const bool same = (unsigned)__shfl_xor((int)lo[c], 16) == lo[c];
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        const float pa = __shfl_xor(accl[c][k], 16);
        accl[c][k] = same ? (taker ? accl[c][k] + pa : 0.f) : accl[c][k];
      }
    }
    finish();
  }
  __device__ inline void finish() {
    if (gP == nullptr) return;
    flush(gP, o[0], acc[0]);
    flush(gP, o[1], acc[1]);
    flush(gP, o[2], acc[2]);
    flush(gP, o[3], acc[3]);
    flush_line(lo[0], accl[0]);
    flush_line(lo[1], accl[1]);
  }
};

}  // namespace jt
