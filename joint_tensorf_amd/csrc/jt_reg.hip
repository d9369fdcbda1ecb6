// Regularisers over a channel-last VM factor in one pass: sum|x| (density_L1, tensoRF.py:212-216) and the two
// total-variation sums  sum (x[y+1,x]-x[y,x])^2, sum (x[y,x+1]-x[y,x])^2  (TVLoss, tensorBase.py:16-41), and
// their gradient accumulated into the factor's gradient buffer.  The stock-op version makes ~10 passes over
// the 123 MB factor set per iteration; this reads it once (forward) and touches only the tensors that carry a
// non-zero weight in the backward.
#include "jt_common.h"

namespace jt {

__device__ inline float4 ld4z(const float* p, bool ok) { return ok ? ld4(p) : make_float4(0.f, 0.f, 0.f, 0.f); }

constexpr int kRegSeg = 16;  // rows a thread of the TV kernels walks (the vertical neighbours stay in its registers)

// out[0] += sum |x| ; out[1] += sum (down - x)^2 ; out[2] += sum (right - x)^2
template <bool TV>
__device__ inline void factor_reg_fwd_body(const float* __restrict__ x, int H, int W, int C, float* __restrict__ out,
                                           int bid, int nblocks) {
  __shared__ float red[4][3];
  const unsigned C4 = C / 4;
  const unsigned total = (unsigned)H * W * C4;  // (< 2^31: checked by the callers)
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  if (TV && H >= 2 * kRegSeg) {
    // TV items: a thread owns (quad, column) and walks kRegSeg rows downwards -- the row below is next step's own value, so a
    // texel is fetched twice (itself / as somebody's right neighbour) instead of three times
    const unsigned nseg = ((unsigned)H + kRegSeg - 1) / kRegSeg, per_row = (unsigned)W * C4, items = nseg * per_row;
    const size_t rstride = (size_t)W * C;
    for (unsigned it = bid * blockDim.x + threadIdx.x; it < items; it += (unsigned)nblocks * blockDim.x) {
      const unsigned seg = it / per_row, q = it - seg * per_row;        // q = xx * C4 + c4: float offset 4 q inside a row
      const unsigned xx = q / C4;
      const int y0 = (int)(seg * kRegSeg), y1 = min(y0 + kRegSeg, H);
      const float* p = x + (size_t)y0 * rstride + (size_t)q * 4;
      float4 v = ld4(p);
#pragma unroll 4
      for (int yy = y0; yy < y1; ++yy) {
        s0 += fabsf(v.x) + fabsf(v.y) + fabsf(v.z) + fabsf(v.w);
        float4 d = v;
        if (yy + 1 < H) {
          d = ld4(p + rstride);
          const float a = d.x - v.x, b = d.y - v.y, c = d.z - v.z, e = d.w - v.w;
          s1 += a * a + b * b + c * c + e * e;
        }
        if (xx + 1 < (unsigned)W) {
          const float4 r = ld4(p + C);
          const float a = r.x - v.x, b = r.y - v.y, c = r.z - v.z, e = r.w - v.w;
          s2 += a * a + b * b + c * c + e * e;
        }
        v = d;
        p += rstride;
      }
    }
  } else
  // (32-bit index arithmetic: with `long` the three divisions per item were most of the kernel's instructions)
#pragma unroll 4
  for (unsigned idx = bid * blockDim.x + threadIdx.x; idx < total; idx += (unsigned)nblocks * blockDim.x) {
    int xx = 0, yy = 0;
    if (TV) {
      const unsigned tex = idx / C4;
      yy = (int)(tex / (unsigned)W);
      xx = (int)(tex - (unsigned)yy * W);
    }
    const float* p = x + (size_t)idx * 4;   // [tex][C] with C = 4 C4: quad idx starts at float 4 idx
    const float4 v = ld4(p);
    s0 += fabsf(v.x) + fabsf(v.y) + fabsf(v.z) + fabsf(v.w);
    if (TV && yy + 1 < H) {
      const float4 d = ld4(p + (long)W * C);
      const float a = d.x - v.x, b = d.y - v.y, c = d.z - v.z, e = d.w - v.w;
      s1 += a * a + b * b + c * c + e * e;
    }
    if (TV && xx + 1 < W) {
      const float4 r = ld4(p + C);
      const float a = r.x - v.x, b = r.y - v.y, c = r.z - v.z, e = r.w - v.w;
      s2 += a * a + b * b + c * c + e * e;
    }
  }
  s0 = wave_sum(s0);
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) {
    red[wv][0] = s0;
    red[wv][1] = s1;
    red[wv][2] = s2;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int k = threadIdx.x;
    atomicAdd(out + k, red[0][k] + red[1][k] + red[2][k] + red[3][k]);
  }
}

template <bool TV>
__global__ __launch_bounds__(256) void k_factor_reg_fwd(const float* __restrict__ x, int H, int W, int C,
                                                        float* __restrict__ out) {
  factor_reg_fwd_body<TV>(x, H, W, C, out, blockIdx.x, gridDim.x);
}

// g (+)= coef[0] * sign(x) + coef[1] * d/dx sum(down-x)^2 + coef[2] * d/dx sum(right-x)^2   (coef on the device)
__device__ inline void factor_reg_bwd_body(const float* __restrict__ x, int H, int W, int C,
                                           const float* __restrict__ coef, float* __restrict__ g, int accumulate,
                                           int bid, int nblocks) {
  const float c0 = coef[0], c1 = coef[1], c2 = coef[2];
  const unsigned C4 = C / 4;
  const unsigned total = (unsigned)H * W * C4;
  const bool tv = (c1 != 0.f) || (c2 != 0.f);
  if (tv && H >= 2 * kRegSeg) {
    // (quad, column) per thread, kRegSeg rows downwards with (up, v, down) sliding through registers: three fetches per texel
    // (the new row, left, right) instead of five; the expression per element is the one of the general loop below
    const unsigned nseg = ((unsigned)H + kRegSeg - 1) / kRegSeg, per_row = (unsigned)W * C4, items = nseg * per_row;
    const size_t rstride = (size_t)W * C;
    for (unsigned it = bid * blockDim.x + threadIdx.x; it < items; it += (unsigned)nblocks * blockDim.x) {
      const unsigned seg = it / per_row, q = it - seg * per_row;
      const int xx = (int)(q / C4);
      const int y0 = (int)(seg * kRegSeg), y1 = min(y0 + kRegSeg, H);
      size_t off = (size_t)y0 * rstride + (size_t)q * 4;
      float4 up = ld4z(x + off - rstride, y0 > 0), v = ld4(x + off);
      const float ml = xx > 0 ? 1.f : 0.f, mr = xx + 1 < W ? 1.f : 0.f;
#pragma unroll 2
      for (int yy = y0; yy < y1; ++yy) {
        const float4 dn = ld4z(x + off + rstride, yy + 1 < H);
        const float4 lf = ld4z(x + off - C, xx > 0), rt = ld4z(x + off + C, xx + 1 < W);
        const float mu = yy > 0 ? 1.f : 0.f, md = yy + 1 < H ? 1.f : 0.f;
        float4 r;
        r.x = c0 * ((v.x > 0.f) - (v.x < 0.f));
        r.y = c0 * ((v.y > 0.f) - (v.y < 0.f));
        r.z = c0 * ((v.z > 0.f) - (v.z < 0.f));
        r.w = c0 * ((v.w > 0.f) - (v.w < 0.f));
        r.x += 2.f * (c1 * (mu * (v.x - up.x) - md * (dn.x - v.x)) + c2 * (ml * (v.x - lf.x) - mr * (rt.x - v.x)));
        r.y += 2.f * (c1 * (mu * (v.y - up.y) - md * (dn.y - v.y)) + c2 * (ml * (v.y - lf.y) - mr * (rt.y - v.y)));
        r.z += 2.f * (c1 * (mu * (v.z - up.z) - md * (dn.z - v.z)) + c2 * (ml * (v.z - lf.z) - mr * (rt.z - v.z)));
        r.w += 2.f * (c1 * (mu * (v.w - up.w) - md * (dn.w - v.w)) + c2 * (ml * (v.w - lf.w) - mr * (rt.w - v.w)));
        if (accumulate) {
          const float4 gg = ld4(g + off);
          r.x += gg.x;
          r.y += gg.y;
          r.z += gg.z;
          r.w += gg.w;
        }
        *reinterpret_cast<float4*>(g + off) = r;
        up = v;
        v = dn;
        off += rstride;
      }
    }
    return;
  }
  for (unsigned idx = bid * blockDim.x + threadIdx.x; idx < total; idx += (unsigned)nblocks * blockDim.x) {
    int xx = 0, yy = 0;
    if (tv) {
      const unsigned tex = idx / C4;
      yy = (int)(tex / (unsigned)W);
      xx = (int)(tex - (unsigned)yy * W);
    }
    const long off = (long)idx * 4;
    const float4 v = ld4(x + off);
    float4 r;
    r.x = c0 * ((v.x > 0.f) - (v.x < 0.f));
    r.y = c0 * ((v.y > 0.f) - (v.y < 0.f));
    r.z = c0 * ((v.z > 0.f) - (v.z < 0.f));
    r.w = c0 * ((v.w > 0.f) - (v.w < 0.f));
    if (tv) {
      const float4 up = ld4z(x + off - (long)W * C, yy > 0), dn = ld4z(x + off + (long)W * C, yy + 1 < H);
      const float4 lf = ld4z(x + off - C, xx > 0), rt = ld4z(x + off + C, xx + 1 < W);
      const float mu = yy > 0 ? 1.f : 0.f, md = yy + 1 < H ? 1.f : 0.f, ml = xx > 0 ? 1.f : 0.f,
                  mr = xx + 1 < W ? 1.f : 0.f;
      r.x += 2.f * (c1 * (mu * (v.x - up.x) - md * (dn.x - v.x)) + c2 * (ml * (v.x - lf.x) - mr * (rt.x - v.x)));
      r.y += 2.f * (c1 * (mu * (v.y - up.y) - md * (dn.y - v.y)) + c2 * (ml * (v.y - lf.y) - mr * (rt.y - v.y)));
      r.z += 2.f * (c1 * (mu * (v.z - up.z) - md * (dn.z - v.z)) + c2 * (ml * (v.z - lf.z) - mr * (rt.z - v.z)));
      r.w += 2.f * (c1 * (mu * (v.w - up.w) - md * (dn.w - v.w)) + c2 * (ml * (v.w - lf.w) - mr * (rt.w - v.w)));
    }
    if (accumulate) {
      const float4 gg = ld4(g + off);
      r.x += gg.x;
      r.y += gg.y;
      r.z += gg.z;
      r.w += gg.w;
    }
    *reinterpret_cast<float4*>(g + off) = r;
  }
}

__global__ __launch_bounds__(256) void k_factor_reg_bwd(const float* __restrict__ x, int H, int W, int C,
                                                        const float* __restrict__ coef, float* __restrict__ g,
                                                        int accumulate) {
  factor_reg_bwd_body(x, H, W, C, coef, g, accumulate, blockIdx.x, gridDim.x);
}

// all tensors of a scene in ONE launch each way: block ranges per tensor, the same partition of every tensor as the
// per-tensor launches (nine launches of 4-10 us become one)
struct RegBatchItem {
  const float* x;
  float* g;
  int H, W, C, tv, slot, block0, nblocks;
};
struct RegBatch {
  RegBatchItem t[9];
  int n;
};

// the normalisation table of the combine step: (H, W, C) of the nine regularised tensors (0-2 density planes, 3-5 density
// lines, 6-8 appearance planes)
struct RegDims {
  int H[9], W[9], C[9];
};

// out3 = {L1, TV_density, TV_color} with the reference's normalisations (tensoRF.py:212-228, tensorBase.py:21-38)
__device__ inline void reg_combine(const float* sums, const RegDims& S, float* __restrict__ out3) {
  float l1 = 0.f, tvd = 0.f, tva = 0.f;
  // (unrolled: S is a kernel argument, and a rolled loop indexing it dynamically made the compiler copy nine words of it into
  //  scratch memory -- 36 bytes per lane in k_reg_batch_fused)
#pragma unroll
  for (int i = 0; i < 6; ++i) l1 += sums[i * 3] / ((float)S.H[i] * S.W[i] * S.C[i]);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float ta = 0.f, tb = 0.f;
    if (S.H[i] > 1) ta += sums[i * 3 + 1] / ((float)S.C[i] * (S.H[i] - 1) * S.W[i]);
    if (S.W[i] > 1) ta += sums[i * 3 + 2] / ((float)S.C[i] * S.H[i] * (S.W[i] - 1));
    const int b = 6 + i;
    if (S.H[b] > 1) tb += sums[b * 3 + 1] / ((float)S.C[b] * (S.H[b] - 1) * S.W[b]);
    if (S.W[b] > 1) tb += sums[b * 3 + 2] / ((float)S.C[b] * S.H[b] * (S.W[b] - 1));
    tvd += 2.f * ta * 1e-2f;
    tva += 2.f * tb * 1e-2f;
  }
  out3[0] = l1;
  out3[1] = tvd;
  out3[2] = tva;
}

// ONE launch (round 4; before: a zero fill of the scratch, this kernel, a combine kernel).  Every workgroup adds its partial sums
// into one of kRegShards copies of the 36 sums and takes a ticket; the workgroup that draws the last ticket adds the copies up,
// combines the 27 sums into out3 and puts the scratch back to zero -- the scratch must be ZERO when the first call sees it and is
// left zero by every call.  Why copies: float atomics on ONE address are serialised at the memory side (~50 ns each), and 512
// workgroups per tensor adding into the same three words took 25 us whatever the tensor's size (30 MB of density factors:
// 37.6 us; one word per 32 workgroups: 17 us).  The tickets are two-level for the same reason: one counter per tensor, and the
// workgroup that completes a tensor draws from the master counter.
// scratch (floats): [shard][36] sums, then the master counter and nine per-tensor counters; the interface asks for 640.
constexpr int kRegShards = 16;
constexpr int kRegCounters = kRegShards * 36;
__global__ __launch_bounds__(256) void k_reg_batch_fwd(RegBatch B, RegDims S, float* __restrict__ scratch,
                                                       float* __restrict__ out3) {
  int it = 0;
#pragma unroll 1
  for (int i = 1; i < B.n; ++i)
    if ((int)blockIdx.x >= B.t[i].block0) it = i;
  const RegBatchItem& T = B.t[it];
  const int bid = blockIdx.x - T.block0;
  float* sums = scratch + (bid % kRegShards) * 36 + T.slot * 3;
  if (T.tv)
    factor_reg_fwd_body<true>(T.x, T.H, T.W, T.C, sums, bid, T.nblocks);
  else
    factor_reg_fwd_body<false>(T.x, T.H, T.W, T.C, sums, bid, T.nblocks);
  // (no __threadfence(): a release fence writes the XCD's L2 back, ~2-6 us per workgroup, 4 600 of them -- measured +115 us on
  //  the LLFF grid.  The sums and the tickets are float / integer atomics, which execute at the memory side and never sit in an
  //  L2: waiting for this workgroup's own atomics to be acknowledged is all the ordering a ticket needs.)
  __shared__ int s_last;
  __shared__ float s_sums[36];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned* cnt = reinterpret_cast<unsigned*>(scratch + kRegCounters);
    bool last = false;
    if (atomicAdd(cnt + 1 + T.slot, 1u) == (unsigned)T.nblocks - 1u)   // this tensor is complete ...
      last = atomicAdd(cnt, 1u) == (unsigned)B.n - 1u;                 // ... and it was the last one
    s_last = last;
  }
  __syncthreads();
  if (s_last) {   // (uniform over the workgroup) read AND reset in one returning atomic per word
    if (threadIdx.x < 36) {
      float a = 0.f;
      for (int sh = 0; sh < kRegShards; ++sh) a += atomicExch(scratch + sh * 36 + threadIdx.x, 0.f);
      s_sums[threadIdx.x] = a;
    }
    if (threadIdx.x >= 64 && threadIdx.x < 64 + 10) atomicExch(reinterpret_cast<unsigned*>(scratch + kRegCounters) + (threadIdx.x - 64), 0u);
    __syncthreads();
    if (threadIdx.x == 0) reg_combine(s_sums, S, out3);
  }
}

// ---- value AND gradient in one pass (round 5) ---------------------------------------------------------------------------------
// The gradient of a regulariser does not depend on its value, and the upstream gradients dL/d{L1, TV_density, TV_color} of a
// training step are the loss weights, which the caller knows BEFORE the forward (host floats, or device memory under hipGraph
// replay): one launch reads every texel once (plus its neighbours for TV), adds the partial sums AND writes the gradient --
// k_reg_batch_fwd + k_reg_batch_bwd streamed the same factors back to back (LLFF final grid: 285 MB twice, 107 + 151 us).
// The gradient is WRITTEN (the render backward's atomics land on top of it: ops.RenderRays, "reg_first").
template <bool TV>
__device__ inline void factor_reg_fused_body(const float* __restrict__ x, int H, int W, int C, float c0, float c1, float c2,
                                             float* __restrict__ g, float* __restrict__ out, int bid, int nblocks) {
  __shared__ float red[4][3];
  const unsigned C4 = C / 4;
  const unsigned total = (unsigned)H * W * C4;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  if (TV && H >= 2 * kRegSeg) {
    const unsigned nseg = ((unsigned)H + kRegSeg - 1) / kRegSeg, per_row = (unsigned)W * C4, items = nseg * per_row;
    const size_t rstride = (size_t)W * C;
    for (unsigned it = bid * blockDim.x + threadIdx.x; it < items; it += (unsigned)nblocks * blockDim.x) {
      const unsigned seg = it / per_row, q = it - seg * per_row;
      const int xx = (int)(q / C4);
      const int y0 = (int)(seg * kRegSeg), y1 = min(y0 + kRegSeg, H);
      size_t off = (size_t)y0 * rstride + (size_t)q * 4;
      float4 up = ld4z(x + off - rstride, y0 > 0), v = ld4(x + off);
      const float ml = xx > 0 ? 1.f : 0.f, mr = xx + 1 < W ? 1.f : 0.f;
#pragma unroll 2
      for (int yy = y0; yy < y1; ++yy) {
        const float4 dn = ld4z(x + off + rstride, yy + 1 < H);
        const float4 lf = ld4z(x + off - C, xx > 0), rt = ld4z(x + off + C, xx + 1 < W);
        const float mu = yy > 0 ? 1.f : 0.f, md = yy + 1 < H ? 1.f : 0.f;
        s0 += fabsf(v.x) + fabsf(v.y) + fabsf(v.z) + fabsf(v.w);
        const float dx = md * (dn.x - v.x), dy = md * (dn.y - v.y), dz = md * (dn.z - v.z), dw = md * (dn.w - v.w);
        const float rx = mr * (rt.x - v.x), ry = mr * (rt.y - v.y), rz = mr * (rt.z - v.z), rw = mr * (rt.w - v.w);
        s1 += dx * dx + dy * dy + dz * dz + dw * dw;
        s2 += rx * rx + ry * ry + rz * rz + rw * rw;
        float4 r;
        r.x = c0 * ((v.x > 0.f) - (v.x < 0.f)) + 2.f * (c1 * (mu * (v.x - up.x) - dx) + c2 * (ml * (v.x - lf.x) - rx));
        r.y = c0 * ((v.y > 0.f) - (v.y < 0.f)) + 2.f * (c1 * (mu * (v.y - up.y) - dy) + c2 * (ml * (v.y - lf.y) - ry));
        r.z = c0 * ((v.z > 0.f) - (v.z < 0.f)) + 2.f * (c1 * (mu * (v.z - up.z) - dz) + c2 * (ml * (v.z - lf.z) - rz));
        r.w = c0 * ((v.w > 0.f) - (v.w < 0.f)) + 2.f * (c1 * (mu * (v.w - up.w) - dw) + c2 * (ml * (v.w - lf.w) - rw));
        *reinterpret_cast<float4*>(g + off) = r;
        up = v;
        v = dn;
        off += rstride;
      }
    }
  } else {
    for (unsigned idx = bid * blockDim.x + threadIdx.x; idx < total; idx += (unsigned)nblocks * blockDim.x) {
      int xx = 0, yy = 0;
      if (TV) {
        const unsigned tex = idx / C4;
        yy = (int)(tex / (unsigned)W);
        xx = (int)(tex - (unsigned)yy * W);
      }
      const long off = (long)idx * 4;
      const float4 v = ld4(x + off);
      s0 += fabsf(v.x) + fabsf(v.y) + fabsf(v.z) + fabsf(v.w);
      float4 r;
      r.x = c0 * ((v.x > 0.f) - (v.x < 0.f));
      r.y = c0 * ((v.y > 0.f) - (v.y < 0.f));
      r.z = c0 * ((v.z > 0.f) - (v.z < 0.f));
      r.w = c0 * ((v.w > 0.f) - (v.w < 0.f));
      if (TV) {
        const float4 up = ld4z(x + off - (long)W * C, yy > 0), dn = ld4z(x + off + (long)W * C, yy + 1 < H);
        const float4 lf = ld4z(x + off - C, xx > 0), rt = ld4z(x + off + C, xx + 1 < W);
        const float mu = yy > 0 ? 1.f : 0.f, md = yy + 1 < H ? 1.f : 0.f, ml = xx > 0 ? 1.f : 0.f,
                    mr = xx + 1 < W ? 1.f : 0.f;
        const float dx = md * (dn.x - v.x), dy = md * (dn.y - v.y), dz = md * (dn.z - v.z), dw = md * (dn.w - v.w);
        const float rx = mr * (rt.x - v.x), ry = mr * (rt.y - v.y), rz = mr * (rt.z - v.z), rw = mr * (rt.w - v.w);
        s1 += dx * dx + dy * dy + dz * dz + dw * dw;
        s2 += rx * rx + ry * ry + rz * rz + rw * rw;
        r.x += 2.f * (c1 * (mu * (v.x - up.x) - dx) + c2 * (ml * (v.x - lf.x) - rx));
        r.y += 2.f * (c1 * (mu * (v.y - up.y) - dy) + c2 * (ml * (v.y - lf.y) - ry));
        r.z += 2.f * (c1 * (mu * (v.z - up.z) - dz) + c2 * (ml * (v.z - lf.z) - rz));
        r.w += 2.f * (c1 * (mu * (v.w - up.w) - dw) + c2 * (ml * (v.w - lf.w) - rw));
      }
      *reinterpret_cast<float4*>(g + off) = r;
    }
  }
  s0 = wave_sum(s0);
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) {
    red[wv][0] = s0;
    red[wv][1] = s1;
    red[wv][2] = s2;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int k = threadIdx.x;
    atomicAdd(out + k, red[0][k] + red[1][k] + red[2][k] + red[3][k]);
  }
}

struct RegWeights {
  float w[3];          // dL/d{L1, TV_density, TV_color} as host values ...
  const float* dev;    // ... or (non-NULL) three floats in device memory
};

__global__ __launch_bounds__(256) void k_reg_batch_fused(RegBatch B, RegDims S, RegWeights Wt, float* __restrict__ scratch,
                                                         float* __restrict__ out3) {
  int it = 0;
#pragma unroll 1
  for (int i = 1; i < B.n; ++i)
    if ((int)blockIdx.x >= B.t[i].block0) it = i;
  const RegBatchItem& T = B.t[it];
  const int bid = blockIdx.x - T.block0;
  const int i = T.slot;  // 0-2 density planes, 3-5 density lines, 6-8 appearance planes
  const float g0 = Wt.dev ? Wt.dev[0] : Wt.w[0], g1 = Wt.dev ? Wt.dev[1] : Wt.w[1], g2 = Wt.dev ? Wt.dev[2] : Wt.w[2];
  // (three scalars, not an array handed on by address: the array kept a 36-byte private segment alive in the kernel descriptor)
  float cf0 = 0.f, cf1 = 0.f, cf2 = 0.f;
  if (i < 6) cf0 = g0 / ((float)T.H * T.W * T.C);
  if (T.tv) {
    const float gt = i < 3 ? g1 : g2;
    if (T.H > 1) cf1 = gt * 2e-2f / ((float)T.C * (T.H - 1) * T.W);
    if (T.W > 1) cf2 = gt * 2e-2f / ((float)T.C * T.H * (T.W - 1));
  }
  float* sums = scratch + (bid % kRegShards) * 36 + T.slot * 3;
  if (T.tv)
    factor_reg_fused_body<true>(T.x, T.H, T.W, T.C, cf0, cf1, cf2, T.g, sums, bid, T.nblocks);
  else
    factor_reg_fused_body<false>(T.x, T.H, T.W, T.C, cf0, cf1, cf2, T.g, sums, bid, T.nblocks);
  // tickets and the combine step: as k_reg_batch_fwd
  __shared__ int s_last;
  __shared__ float s_sums[36];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned* cnt = reinterpret_cast<unsigned*>(scratch + kRegCounters);
    bool last = false;
    if (atomicAdd(cnt + 1 + T.slot, 1u) == (unsigned)T.nblocks - 1u) last = atomicAdd(cnt, 1u) == (unsigned)B.n - 1u;
    s_last = last;
  }
  __syncthreads();
  if (s_last) {
    if (threadIdx.x < 36) {
      float a = 0.f;
      for (int sh = 0; sh < kRegShards; ++sh) a += atomicExch(scratch + sh * 36 + threadIdx.x, 0.f);
      s_sums[threadIdx.x] = a;
    }
    if (threadIdx.x >= 64 && threadIdx.x < 64 + 10) atomicExch(reinterpret_cast<unsigned*>(scratch + kRegCounters) + (threadIdx.x - 64), 0u);
    __syncthreads();
    if (threadIdx.x == 0) reg_combine(s_sums, S, out3);
  }
}

// (the coefficient triple of the block's tensor from the upstream gradients g3 = dL/d{L1, TV_density, TV_color} is three
//  divisions: every block works it out for itself instead of a launch of its own in front of this one)
__global__ __launch_bounds__(256) void k_reg_batch_bwd(RegBatch B, const float* __restrict__ g3, int accumulate) {
  int it = 0;
#pragma unroll 1
  for (int i = 1; i < B.n; ++i)
    if ((int)blockIdx.x >= B.t[i].block0) it = i;
  const RegBatchItem& T = B.t[it];
  const int i = T.slot;  // 0-2 density planes, 3-5 density lines, 6-8 appearance planes
  float coef[3] = {0.f, 0.f, 0.f};
  if (i < 6) coef[0] = g3[0] / ((float)T.H * T.W * T.C);
  if (i < 3 || i >= 6) {
    const float gt = i < 3 ? g3[1] : g3[2];
    if (T.H > 1) coef[1] = gt * 2e-2f / ((float)T.C * (T.H - 1) * T.W);
    if (T.W > 1) coef[2] = gt * 2e-2f / ((float)T.C * T.H * (T.W - 1));
  }
  factor_reg_bwd_body(T.x, T.H, T.W, T.C, coef, T.g, accumulate, blockIdx.x - T.block0, T.nblocks);
}

}  // namespace jt

using namespace jt;

extern "C" int jt_factor_reg_forward(const float* x, int H, int W, int C, float* out3, void* stream) {
  if (!x || !out3 || H < 1 || W < 1 || C < 4) return JT_ERR_ARG;
  if (C % 4) return JT_ERR_UNSUPPORTED;
  long total = (long)H * W * (C / 4);
  if (total >= (1l << 31)) return JT_ERR_UNSUPPORTED;  // the kernels index quads with 32 bits
  int blocks = (int)min((total + 255) / 256, 1024L);
  hipLaunchKernelGGL(k_factor_reg_fwd<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, H, W, C, out3);
  JT_LAUNCH_CHECK();
  return JT_OK;
}

extern "C" int jt_factor_reg_backward(const float* x, int H, int W, int C, const float* coef3, float* g,
                                      int accumulate, void* stream) {
  if (!x || !coef3 || !g || H < 1 || W < 1 || C < 4) return JT_ERR_ARG;
  if (C % 4) return JT_ERR_UNSUPPORTED;
  long total = (long)H * W * (C / 4);
  if (total >= (1l << 31)) return JT_ERR_UNSUPPORTED;  // the kernels index quads with 32 bits
  int blocks = (int)min((total + 255) / 256, 2048L);
  hipLaunchKernelGGL(k_factor_reg_bwd, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, H, W, C, coef3, g, accumulate);
  JT_LAUNCH_CHECK();
  return JT_OK;
}

// ---------------------------------------------------------------------------------------------
// All regularisers of one scene in one call: L1 over the six density factors, TV over the three density
// planes and over the three appearance planes.  One sums launch per tensor + one tiny combine kernel instead
// of ~150 elementwise / reduce launches of the stock-op formulation (which left the host launch-bound).
// ---------------------------------------------------------------------------------------------
namespace jt {

struct RegTensor {
  const float* x;
  float* g;
  int H, W, C;
};

struct RegSet {
  RegTensor t[12];  // 0-2 density planes, 3-5 density lines, 6-8 app planes, 9-11 app lines
};

}  // namespace jt

static int reg_set(const JtFactors* f, const JtFactors* g, const int32_t* hw, int Cd, int Ca, RegSet* S) {
  if (!f || !hw || Cd < 4 || Ca < 4 || (Cd % 4) || (Ca % 4)) return JT_ERR_ARG;
  for (int i = 0; i < 3; ++i) {
    const int H = hw[i * 3], W = hw[i * 3 + 1], L = hw[i * 3 + 2];
    if (H < 1 || W < 1 || L < 1) return JT_ERR_ARG;
    S->t[i] = {f->density_plane[i], g ? g->density_plane[i] : nullptr, H, W, Cd};
    S->t[3 + i] = {f->density_line[i], g ? g->density_line[i] : nullptr, L, 1, Cd};
    S->t[6 + i] = {f->app_plane[i], g ? g->app_plane[i] : nullptr, H, W, Ca};
    S->t[9 + i] = {f->app_line[i], g ? g->app_line[i] : nullptr, L, 1, Ca};
  }
  for (int i = 0; i < 12; ++i)
    if (!S->t[i].x) return JT_ERR_ARG;
  return JT_OK;
}

extern "C" int jt_reg_losses_forward(const JtFactors* factors, const int32_t* plane_hw_line, int n_comp_density,
                                     int n_comp_app, int with_tv_density, int with_tv_app, float* scratch640,
                                     float* out3, void* stream) {
  RegSet S;
  int rc = reg_set(factors, nullptr, plane_hw_line, n_comp_density, n_comp_app, &S);
  if (rc) return rc;
  if (!scratch640 || !out3) return JT_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  RegBatch B;
  B.n = 0;
  int nblk = 0;
  for (int i = 0; i < 9; ++i) {  // app lines (9-11) enter no regulariser
    // a TV term whose weight is zero is not evaluated (it reads every texel three times): out3 carries 0 for it
    const bool tv = (i < 3 && with_tv_density) || (i >= 6 && with_tv_app);
    if (i >= 6 && !tv) continue;  // appearance planes only enter TV_color
    const RegTensor& t = S.t[i];
    long total = (long)t.H * t.W * (t.C / 4);
    if (total >= (1l << 31)) return JT_ERR_UNSUPPORTED;  // the kernels index quads with 32 bits
    // (measured, 400^3 L1-only / LLFF final grid with both TV terms: 128 workgroups per tensor 17 / 126 us, 512: 27 / 107,
    //  1 024: 44 / 131 -- the three-load TV items want the parallelism, the one-load L1 items the shorter epilogue)
    static const long max_blocks = [] { const char* e = getenv("JT_REG_BLOCKS"); return e ? atol(e) : 0L; }();
    int blocks = (int)min((total + 255) / 256, max_blocks > 0 ? max_blocks : (tv ? 512L : 128L));
    if (jt_deterministic()) blocks = 1;                // one workgroup per tensor: a fixed summation order
    B.t[B.n++] = {t.x, nullptr, t.H, t.W, t.C, tv ? 1 : 0, i, nblk, blocks};
    nblk += blocks;
  }
  RegDims Dm;
  for (int i = 0; i < 9; ++i) Dm.H[i] = S.t[i].H, Dm.W[i] = S.t[i].W, Dm.C[i] = S.t[i].C;
  hipLaunchKernelGGL(k_reg_batch_fwd, dim3(nblk), dim3(256), 0, st, B, Dm, scratch640, out3);
  JT_LAUNCH_CHECK();
  return JT_OK;
}

extern "C" int jt_reg_losses_backward(const JtFactors* factors, const int32_t* plane_hw_line, int n_comp_density,
                                      int n_comp_app, const float* g3, int with_tv_density, int with_tv_app,
                                      const JtFactors* g_factors, int accumulate, float* scratch640, void* stream) {
  RegSet S;
  int rc = reg_set(factors, g_factors, plane_hw_line, n_comp_density, n_comp_app, &S);
  if (rc) return rc;
  if (!g3 || !scratch640 || !g_factors) return JT_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  RegBatch B;
  B.n = 0;
  int nblk = 0;
  for (int i = 0; i < 9; ++i) {
    const bool dens = i < 6;
    const bool need = dens || (i >= 6 && with_tv_app);  // density tensors always carry the L1 term
    if (!need) continue;
    const RegTensor& t = S.t[i];
    if (!t.g) return JT_ERR_ARG;
    (void)with_tv_density;  // the TV coefficient of a density plane is on the device (0 when its weight is 0)
    long total = (long)t.H * t.W * (t.C / 4);
    if (total >= (1l << 31)) return JT_ERR_UNSUPPORTED;  // the kernels index quads with 32 bits
    int blocks = (int)min((total + 255) / 256, 2048L);
    B.t[B.n++] = {t.x, t.g, t.H, t.W, t.C, 0, i, nblk, blocks};
    nblk += blocks;
  }
  hipLaunchKernelGGL(k_reg_batch_bwd, dim3(nblk), dim3(256), 0, st, B, g3, accumulate ? 1 : 0);
  JT_LAUNCH_CHECK();
  return JT_OK;
}

extern "C" int jt_reg_losses_fused(const JtFactors* factors, const int32_t* plane_hw_line, int n_comp_density, int n_comp_app,
                                   int with_tv_density, int with_tv_app, const float* w3_host, const float* w3_dev,
                                   const JtFactors* g_factors, float* scratch640, float* out3, void* stream) {
  RegSet S;
  int rc = reg_set(factors, g_factors, plane_hw_line, n_comp_density, n_comp_app, &S);
  if (rc) return rc;
  if (!scratch640 || !out3 || !g_factors || (!w3_host && !w3_dev)) return JT_ERR_ARG;
  if (jt_deterministic()) return JT_ERR_UNSUPPORTED;  // (the fixed summation order lives in the two-launch form)
  hipStream_t st = (hipStream_t)stream;
  RegBatch B;
  B.n = 0;
  int nblk = 0;
  for (int i = 0; i < 9; ++i) {
    const bool tv = (i < 3 && with_tv_density) || (i >= 6 && with_tv_app);
    if (i >= 6 && !tv) continue;  // appearance planes only enter TV_color
    const RegTensor& t = S.t[i];
    if (!t.g) return JT_ERR_ARG;
    long total = (long)t.H * t.W * (t.C / 4);
    if (total >= (1l << 31)) return JT_ERR_UNSUPPORTED;
    static const long max_blocks = [] { const char* e = getenv("JT_REG_FUSED_BLOCKS"); return e ? atol(e) : 0L; }();
    int blocks = (int)min((total + 255) / 256, max_blocks > 0 ? max_blocks : (tv ? 1024L : 256L));
    B.t[B.n++] = {t.x, t.g, t.H, t.W, t.C, tv ? 1 : 0, i, nblk, blocks};
    nblk += blocks;
  }
  RegDims Dm;
  for (int i = 0; i < 9; ++i) Dm.H[i] = S.t[i].H, Dm.W[i] = S.t[i].W, Dm.C[i] = S.t[i].C;
  RegWeights Wt;
  for (int k = 0; k < 3; ++k) Wt.w[k] = w3_host ? w3_host[k] : 0.f;
  Wt.dev = w3_host ? nullptr : w3_dev;
  hipLaunchKernelGGL(k_reg_batch_fused, dim3(nblk), dim3(256), 0, st, B, Dm, Wt, scratch640, out3);
  JT_LAUNCH_CHECK();
  return JT_OK;
}
