// Run-length scatter of factor gradients along a ray.
//
// Samples of one ray are half a voxel apart, so consecutive samples hit the same plane cell / line
// segment several times.  A 16-lane group (lane = channel) walks a run of consecutive samples IN ORDER and
// keeps the gradient of the four corners of the current plane cell and of the two taps of the current
// line segment in registers.  A texel is written with ONE float atomic per channel when the walk leaves
// it; a step to an edge-adjacent cell keeps the two shared corners.  This cuts the atomic traffic of
// grid_sampler_2d_backward (one atomic per tap per sample) by the run length (measured 3-7x), and every
// atomic wave-instruction touches 64-byte-contiguous channel vectors.
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
  int cx, cy, cz;
  int H, W, LL, cl;
  float* gP;
  float* gL;

  __device__ inline void init(float* gP_, float* gL_, int H_, int W_, int LL_, int cl_) {
    gP = gP_;
    gL = gL_;
    H = H_;
    W = W_;
    LL = LL_;
    cl = cl_;
    cx = cy = cz = -1000000;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      acc[0][k] = acc[1][k] = acc[2][k] = acc[3][k] = 0.f;
      accl[0][k] = accl[1][k] = 0.f;
    }
  }
  __device__ inline void flush_corner(int i, int j, float* a) {
    const int x = cx + i, y = cy + j;
    const bool ok = (x >= 0) && (x < W) && (y >= 0) && (y < H);
    if (ok) {
      float* p = gP + ((unsigned)(y * W + x) * (unsigned)CA + (unsigned)cl);  // 32-bit offset, one add
#pragma unroll
      for (int k = 0; k < NCH; ++k)
        if (((CA % 16 == 0) || (cl + 16 * k < CA)) && a[k] != 0.f) atomicAdd(p + 16 * k, a[k]);
    }
#pragma unroll
    for (int k = 0; k < NCH; ++k) a[k] = 0.f;
  }
  __device__ inline void flush_line(int i, float* a) {
    const int z = cz + i;
    const bool ok = (z >= 0) && (z < LL);
    if (ok) {
      float* p = gL + ((unsigned)z * (unsigned)CA + (unsigned)cl);
#pragma unroll
      for (int k = 0; k < NCH; ++k)
        if (((CA % 16 == 0) || (cl + 16 * k < CA)) && a[k] != 0.f) atomicAdd(p + 16 * k, a[k]);
    }
#pragma unroll
    for (int k = 0; k < NCH; ++k) a[k] = 0.f;
  }
  // move the register window to cell (nx, ny) / segment nz
  __device__ inline void advance(int nx, int ny, int nz) {
    if (nx != cx || ny != cy) {
      if (ny == cy && nx == cx + 1) {
        flush_corner(0, 0, acc[0]);
        flush_corner(0, 1, acc[2]);
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
          acc[0][k] = acc[1][k];
          acc[2][k] = acc[3][k];
          acc[1][k] = acc[3][k] = 0.f;
        }
      } else if (ny == cy && nx == cx - 1) {
        flush_corner(1, 0, acc[1]);
        flush_corner(1, 1, acc[3]);
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
          acc[1][k] = acc[0][k];
          acc[3][k] = acc[2][k];
          acc[0][k] = acc[2][k] = 0.f;
        }
      } else if (nx == cx && ny == cy + 1) {
        flush_corner(0, 0, acc[0]);
        flush_corner(1, 0, acc[1]);
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
          acc[0][k] = acc[2][k];
          acc[1][k] = acc[3][k];
          acc[2][k] = acc[3][k] = 0.f;
        }
      } else if (nx == cx && ny == cy - 1) {
        flush_corner(0, 1, acc[2]);
        flush_corner(1, 1, acc[3]);
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
          acc[2][k] = acc[0][k];
          acc[3][k] = acc[1][k];
          acc[0][k] = acc[1][k] = 0.f;
        }
      } else {
        flush_corner(0, 0, acc[0]);
        flush_corner(1, 0, acc[1]);
        flush_corner(0, 1, acc[2]);
        flush_corner(1, 1, acc[3]);
      }
      cx = nx;
      cy = ny;
    }
    if (nz != cz) {
      if (nz == cz + 1) {
        flush_line(0, accl[0]);
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
          accl[0][k] = accl[1][k];
          accl[1][k] = 0.f;
        }
      } else if (nz == cz - 1) {
        flush_line(1, accl[1]);
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
          accl[1][k] = accl[0][k];
          accl[0][k] = 0.f;
        }
      } else {
        flush_line(0, accl[0]);
        flush_line(1, accl[1]);
      }
      cz = nz;
    }
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
    flush_corner(0, 0, acc[0]);
    flush_corner(1, 0, acc[1]);
    flush_corner(0, 1, acc[2]);
    flush_corner(1, 1, acc[3]);
    flush_line(0, accl[0]);
    flush_line(1, accl[1]);
  }
};

__device__ inline float group16_sum(float v) {
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o);
  return v;
}

}  // namespace jt
