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

#ifndef JT_FLUSH_COND
#define JT_FLUSH_COND(x) ((x) != 0.f)
#endif

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
        if (((CA % 16 == 0) || (cl + 16 * k < CA)) && JT_FLUSH_COND(a[k])) atomicAdd(p + 16 * k, a[k]);
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
        if (((CA % 16 == 0) || (cl + 16 * k < CA)) && JT_FLUSH_COND(a[k])) atomicAdd(p + 16 * k, a[k]);
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

// ---------------------------------------------------------------------------------------------
// Step records.  The geometry of a (sample, plane) pair -- tap addresses, weights, cell ids -- is the same
// for every channel, so it is computed ONCE by one lane (make_step_rec) and parked in LDS as 16 words;
// the channel lanes of the walker read it back as broadcast LDS loads instead of redoing ~60 VALU
// instructions of index arithmetic per lane.
//   word 0..3  byte offsets of the plane taps (x0,y0) (x1,y0) (x0,y1) (x1,y1), clamped into the plane
//        4..5  byte offsets of the two line taps, clamped
//        6     cell x | cell y << 16   (int16 each)
//        7     line cell | in-range bits << 16  (bit0..3 plane taps in the order above, bit 4..5 line taps)
//        8..11 plane tap weights (zero when the tap is out of range), 12..13 line tap weights
//        14,15 fractional parts fx, fy
// ---------------------------------------------------------------------------------------------
constexpr int kRecWords = 16;

__device__ inline void make_step_rec(float gx, float gy, float gl, int H, int W, int LL, int CA, float* rec) {
  const PlaneTaps t = plane_taps(gx, gy, H, W, CA);
  const Axis l = axis_taps(gl, LL);
  unsigned bits = 0u;
  bits |= (t.ax.m0 * t.ay.m0 != 0.f) ? 1u : 0u;
  bits |= (t.ax.m1 * t.ay.m0 != 0.f) ? 2u : 0u;
  bits |= (t.ax.m0 * t.ay.m1 != 0.f) ? 4u : 0u;
  bits |= (t.ax.m1 * t.ay.m1 != 0.f) ? 8u : 0u;
  bits |= (l.m0 != 0.f) ? 16u : 0u;
  bits |= (l.m1 != 0.f) ? 32u : 0u;
  uint4 o = make_uint4((unsigned)t.o00 * 4u, (unsigned)t.o10 * 4u, (unsigned)t.o01 * 4u, (unsigned)t.o11 * 4u);
  uint4 m = make_uint4((unsigned)(l.c0 * CA) * 4u, (unsigned)(l.c1 * CA) * 4u,
                       ((unsigned)t.ax.i0 & 0xffffu) | ((unsigned)t.ay.i0 << 16),
                       ((unsigned)l.i0 & 0xffffu) | (bits << 16));
  *reinterpret_cast<uint4*>(rec) = o;
  *reinterpret_cast<uint4*>(rec + 4) = m;
  *reinterpret_cast<float4*>(rec + 8) = make_float4(t.w00, t.w10, t.w01, t.w11);
  *reinterpret_cast<float4*>(rec + 12) = make_float4(l.w0, l.w1, t.ax.f, t.ay.f);
}

// factor values at the six taps of one (sample, plane) for the lane's channels cl, cl+16, ...
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
template <int NCH, int CA>
struct RecWalker {
  float acc[4][NCH];  // corners (0,0) (1,0) (0,1) (1,1) of the current cell
  float accl[2][NCH];
  int cx, cy, cz;
  unsigned o[4], lo[2];  // byte offsets of the current cell's texels (from its step record)
  unsigned ck[NCH];      // byte offset of the lane's k-th channel inside a texel (0 for a padding lane)
  bool live[NCH];
  float* gP;
  float* gL;

  __device__ inline void init(float* gP_, float* gL_, int cl) {
    gP = gP_;
    gL = gL_;
    cx = cy = cz = -1000000;
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
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      tv.a[k] = ldb(P, ro.x + ck[k]);
      tv.b[k] = ldb(P, ro.y + ck[k]);
      tv.c[k] = ldb(P, ro.z + ck[k]);
      tv.d[k] = ldb(P, ro.w + ck[k]);
      tv.u[k] = ldb(L, rl.x + ck[k]);
      tv.v[k] = ldb(L, rl.y + ck[k]);
    }
  }
  // out-of-range texels never receive anything (their weights are zero), so a clamped address is fine
  __device__ inline void flush(float* base, unsigned off, float* a) {
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      if (live[k] && JT_FLUSH_COND(a[k]))
        atomicAdd(reinterpret_cast<float*>(reinterpret_cast<char*>(base) + (off + ck[k])), a[k]);
      a[k] = 0.f;
    }
  }
  // move the register window to the cell of `rec`
  __device__ inline void advance(const float* rec) {
    const uint2 cw = *reinterpret_cast<const uint2*>(rec + 6);
    const int nx = (int)(short)(cw.x & 0xffffu), ny = (int)cw.x >> 16, nz = (int)(short)(cw.y & 0xffffu);
    if (nx != cx || ny != cy) {
      if (ny == cy && nx == cx + 1) {
        flush(gP, o[0], acc[0]);
        flush(gP, o[2], acc[2]);
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
          acc[0][k] = acc[1][k];
          acc[2][k] = acc[3][k];
          acc[1][k] = acc[3][k] = 0.f;
        }
      } else if (ny == cy && nx == cx - 1) {
        flush(gP, o[1], acc[1]);
        flush(gP, o[3], acc[3]);
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
          acc[1][k] = acc[0][k];
          acc[3][k] = acc[2][k];
          acc[0][k] = acc[2][k] = 0.f;
        }
      } else if (nx == cx && ny == cy + 1) {
        flush(gP, o[0], acc[0]);
        flush(gP, o[1], acc[1]);
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
          acc[0][k] = acc[2][k];
          acc[1][k] = acc[3][k];
          acc[2][k] = acc[3][k] = 0.f;
        }
      } else if (nx == cx && ny == cy - 1) {
        flush(gP, o[2], acc[2]);
        flush(gP, o[3], acc[3]);
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
          acc[2][k] = acc[0][k];
          acc[3][k] = acc[1][k];
          acc[0][k] = acc[1][k] = 0.f;
        }
      } else {
        flush(gP, o[0], acc[0]);
        flush(gP, o[1], acc[1]);
        flush(gP, o[2], acc[2]);
        flush(gP, o[3], acc[3]);
      }
      cx = nx;
      cy = ny;
    }
    if (nz != cz) {
      if (nz == cz + 1) {
        flush(gL, lo[0], accl[0]);
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
          accl[0][k] = accl[1][k];
          accl[1][k] = 0.f;
        }
      } else if (nz == cz - 1) {
        flush(gL, lo[1], accl[1]);
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
          accl[1][k] = accl[0][k];
          accl[0][k] = 0.f;
        }
      } else {
        flush(gL, lo[0], accl[0]);
        flush(gL, lo[1], accl[1]);
      }
      cz = nz;
    }
    const uint4 ro = *reinterpret_cast<const uint4*>(rec);
    const uint2 rl = *reinterpret_cast<const uint2*>(rec + 4);
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
  __device__ inline void add(TapBuf<NCH>& tv, const float* rec, const float g[NCH], float& aix, float& aiy,
                             float& ail) {
    const float4 w = *reinterpret_cast<const float4*>(rec + 8);
    const float4 x = *reinterpret_cast<const float4*>(rec + 12);  // lw0, lw1, fx, fy
    const unsigned bits = reinterpret_cast<const unsigned*>(rec)[7] >> 16;
    if (bits != 0x3fu) {  // a tap outside the factor (exactly on the far border): its value counts as zero
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
    const float fx = x.z, fy = x.w, gx1 = 1.f - fx, gy1 = 1.f - fy;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const float pv = w.x * tv.a[k] + w.y * tv.b[k] + w.z * tv.c[k] + w.w * tv.d[k];
      const float lv = x.x * tv.u[k] + x.y * tv.v[k];
      const float gpv = g[k] * lv, glv = g[k] * pv;
      acc[0][k] += w.x * gpv;
      acc[1][k] += w.y * gpv;
      acc[2][k] += w.z * gpv;
      acc[3][k] += w.w * gpv;
      accl[0][k] += x.x * glv;
      accl[1][k] += x.y * glv;
      aix += gpv * ((tv.b[k] - tv.a[k]) * gy1 + (tv.d[k] - tv.c[k]) * fy);
      aiy += gpv * ((tv.c[k] - tv.a[k]) * gx1 + (tv.d[k] - tv.b[k]) * fx);
      ail += glv * (tv.v[k] - tv.u[k]);
    }
  }
  __device__ inline void finish() {
    flush(gP, o[0], acc[0]);
    flush(gP, o[1], acc[1]);
    flush(gP, o[2], acc[2]);
    flush(gP, o[3], acc[3]);
    flush(gL, lo[0], accl[0]);
    flush(gL, lo[1], accl[1]);
  }
};

__device__ inline float group16_sum(float v) {
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o);
  return v;
}

}  // namespace jt
