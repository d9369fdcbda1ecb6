// Run-length scatter of factor gradients along a ray.
//
// Samples of one ray are half a voxel apart, so consecutive samples hit the same plane cell / line
// segment several times.  A 16-lane group (lane = channel) walks a run of consecutive samples IN ORDER and
// keeps the gradient of the four corners of the current plane cell and of the two taps of the current
// line segment in registers.  A texel is written with ONE float atomic per channel when the walk leaves
// it; a step to an edge-adjacent cell keeps the two shared corners.  This cuts the atomic traffic of
// grid_sampler_2d_backward (one atomic per tap per sample) by the run length (measured 3-7x), and every
// atomic wave-instruction touches 64-byte-contiguous channel vectors.  The bookkeeping is kept
// branch-light (retiring corners = four predicated blocks, sliding accumulators = selects, 32-bit
// offsets): the first version spent 5x more instructions on exec-mask juggling and 64-bit address
// arithmetic than on the gradient arithmetic itself.
#pragma once
#include "jt_common.h"

namespace jt {

// taps + the factor values of one (sample, plane) for the lane's channels cl, cl+16, ...
template <int NCH>
struct TapVals {
  PlaneTaps t;
  Axis l;
  float a[NCH], b[NCH], c[NCH], d[NCH], u[NCH], v[NCH];
};

template <int NCH, int CA>
__device__ inline void tap_load(TapVals<NCH>& tv, const float* __restrict__ P, const float* __restrict__ L,
                                float gx, float gy, float gl, int H, int W, int LL, int cl) {
  tv.t = plane_taps(gx, gy, H, W, CA);
  tv.l = axis_taps(gl, LL);
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int c = cl + 16 * k;
    const int cc = (c < CA) ? c : 0;
    tv.a[k] = P[(unsigned)(tv.t.o00 + cc)];
    tv.b[k] = P[(unsigned)(tv.t.o10 + cc)];
    tv.c[k] = P[(unsigned)(tv.t.o01 + cc)];
    tv.d[k] = P[(unsigned)(tv.t.o11 + cc)];
    tv.u[k] = L[(unsigned)(tv.l.c0 * CA + cc)];
    tv.v[k] = L[(unsigned)(tv.l.c1 * CA + cc)];
  }
}

template <int NCH, int CA>
struct PlaneWalker {
  float acc[4][NCH];  // corners (0,0) (1,0) (0,1) (1,1) of cell (cx, cy)
  float accl[2][NCH];
  unsigned off[4];    // element offsets of the four corner texels (clamped into the plane)
  unsigned loff[2];
  int cx, cy, cz;
  int cl;
  float* gP;
  float* gL;

  __device__ inline void init(float* gP_, float* gL_, int /*H*/, int /*W*/, int /*LL*/, int cl_) {
    gP = gP_;
    gL = gL_;
    cl = cl_;
    cx = cy = cz = -1000000;
    off[0] = off[1] = off[2] = off[3] = 0u;
    loff[0] = loff[1] = 0u;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      acc[0][k] = acc[1][k] = acc[2][k] = acc[3][k] = 0.f;
      accl[0][k] = accl[1][k] = 0.f;
    }
  }
  // one float atomic per channel of the lane; 16 lanes cover 64 contiguous bytes of the texel.
  // Out-of-range corners carry zero weight, so their accumulator is exactly 0 and their (clamped)
  // address is a valid texel: adding 0.0 there is harmless, no range / zero test is needed.
  __device__ inline void flush(float* base, unsigned o, float* a) {
    float* p = base + (o + (unsigned)cl);
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      if ((CA % 16 == 0) || (cl + 16 * k < CA)) atomicAdd(p + 16 * k, a[k]);
      a[k] = 0.f;
    }
  }
  // move the register window to the cell / segment of the next sample.  The walk case (same cell, one
  // step in +-x / +-y, anything else) decides which corners retire (one predicated block per corner) and
  // which accumulators slide into a new slot (selects, no branches).
  __device__ inline void advance(const PlaneTaps& t, const Axis& l) {
    const int nx = t.ax.i0, ny = t.ay.i0, nz = l.i0;
    const int dx = nx - cx, dy = ny - cy, dz = nz - cz;
    const bool same = (dx == 0) & (dy == 0);
    const bool px = (dy == 0) & (dx == 1), mx = (dy == 0) & (dx == -1);
    const bool py = (dx == 0) & (dy == 1), my = (dx == 0) & (dy == -1);
    const bool fresh = cx < -999999;  // first sample of a run: nothing accumulated yet, nothing to retire
    const bool other = !(same | px | mx | py | my) & !fresh;
    // retiring corners: +x {0,2}  -x {1,3}  +y {0,1}  -y {2,3}  other: all
    if (px | py | other) flush(gP, off[0], acc[0]);
    if (mx | py | other) flush(gP, off[1], acc[1]);
    if (px | my | other) flush(gP, off[2], acc[2]);
    if (mx | my | other) flush(gP, off[3], acc[3]);
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const float a0 = acc[0][k], a1 = acc[1][k], a2 = acc[2][k], a3 = acc[3][k];  // retired ones are 0 already
      acc[0][k] = same ? a0 : px ? a1 : py ? a2 : 0.f;
      acc[1][k] = same ? a1 : mx ? a0 : py ? a3 : 0.f;
      acc[2][k] = same ? a2 : px ? a3 : my ? a0 : 0.f;
      acc[3][k] = same ? a3 : mx ? a2 : my ? a1 : 0.f;
    }
    off[0] = (unsigned)t.o00;
    off[1] = (unsigned)t.o10;
    off[2] = (unsigned)t.o01;
    off[3] = (unsigned)t.o11;
    cx = nx;
    cy = ny;
    const bool lsame = dz == 0, lp = dz == 1, lm = dz == -1;
    if (!lsame & !lm & !fresh) flush(gL, loff[0], accl[0]);  // +1 or other
    if (!lsame & !lp & !fresh) flush(gL, loff[1], accl[1]);  // -1 or other
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const float b0 = accl[0][k], b1 = accl[1][k];
      accl[0][k] = lsame ? b0 : lp ? b1 : 0.f;
      accl[1][k] = lsame ? b1 : lm ? b0 : 0.f;
    }
    loff[0] = (unsigned)(l.c0 * CA);
    loff[1] = (unsigned)(l.c1 * CA);
    cz = nz;
  }
  // accumulate one sample.  g[k] = dL/d(plane_c * line_c) for the lane's channels (already zero for
  // inactive lanes).  Returns the UN-reduced coordinate-gradient partials of this lane in (aix, aiy, ail).
  __device__ inline void add(const TapVals<NCH>& tv, const float g[NCH], float& aix, float& aiy, float& ail) {
    const PlaneTaps& t = tv.t;
    const Axis& l = tv.l;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const float pv = t.w00 * tv.a[k] + t.w10 * tv.b[k] + t.w01 * tv.c[k] + t.w11 * tv.d[k];
      const float lv = l.w0 * tv.u[k] + l.w1 * tv.v[k];
      const float gpv = g[k] * lv, glv = g[k] * pv;
      acc[0][k] += t.w00 * gpv;
      acc[1][k] += t.w10 * gpv;
      acc[2][k] += t.w01 * gpv;
      acc[3][k] += t.w11 * gpv;
      accl[0][k] += l.w0 * glv;
      accl[1][k] += l.w1 * glv;
      // grid_sampler backward w.r.t. the coordinates: out-of-range taps count as zeros
      const float a_ = tv.a[k] * t.ax.m0 * t.ay.m0, b_ = tv.b[k] * t.ax.m1 * t.ay.m0,
                  c_ = tv.c[k] * t.ax.m0 * t.ay.m1, d_ = tv.d[k] * t.ax.m1 * t.ay.m1;
      aix += gpv * ((b_ - a_) * (1.f - t.ay.f) + (d_ - c_) * t.ay.f);
      aiy += gpv * ((c_ - a_) * (1.f - t.ax.f) + (d_ - b_) * t.ax.f);
      ail += glv * (tv.v[k] * l.m1 - tv.u[k] * l.m0);
    }
  }
  __device__ inline void finish() {
    flush(gP, off[0], acc[0]);
    flush(gP, off[1], acc[1]);
    flush(gP, off[2], acc[2]);
    flush(gP, off[3], acc[3]);
    flush(gL, loff[0], accl[0]);
    flush(gL, loff[1], accl[1]);
  }
};

__device__ inline float group16_sum(float v) {
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o);
  return v;
}

}  // namespace jt
