// Separable 1-D blur of channel-last VM factors with replicate padding (cross-correlation), and its
// exact adjoint.  Replaces BAT_VMSplit.convolute_plane / convolute_line (bateRF.py:8-39): pad by
// K//2 on both sides (replicate), correlate along W, then along H, same taps for every channel.
//
// Layout [H][W][C]: channels are the fastest axis, so a work-item owns 4 consecutive channels (16-byte
// accesses, coalesced across the channel lanes) of kBlurP consecutive positions along the blurred axis and
// slides over the inputs that reach them; all factors of a scene go through one launch per pass.
#include <algorithm>
#include <cstdlib>

#include "jt_common.h"

namespace jt {

#ifndef JT_BLUR_P
#define JT_BLUR_P 8
#endif
constexpr int kBlurP = JT_BLUR_P;  // outputs per thread along the blurred axis (a multiple of 4)

// One kernel for the correlation and its adjoint.  A thread owns one channel quad of one line (all positions of
// the non-blurred axis x channel quads) and kBlurP consecutive positions along the blurred axis: it walks the
// kBlurP + ntaps - 1 inputs that reach them ONCE (one 16-byte load each) and feeds kBlurP accumulators --
// 9 loads per output instead of 65.  The weight of input q for output u is block-uniform, so the block builds the
// small table Wt[q][u] in LDS first (the adjoint folds the replicate-padding terms of the two border texels
// into it) and the inner loop reads it as two broadcast 16-byte LDS loads.
//   forward : out[u] = sum_t k[t] in[clamp(u + t - r)]
//   adjoint : g_in[u] = sum_x g_out[x] ( k[u - x + r] + [u == 0] sum_{t < r - x} k[t] + [u == n-1] sum_{t >= n - x + r} k[t] )
template <bool ADJ>
__device__ inline void blur_axis_block(const float* __restrict__ in, float* __restrict__ out, int H, int W, int C,
                                       int along_h, const float* __restrict__ taps, int ntaps, int block_x,
                                       int block_y, float* s_mem) {
  const int nq = kBlurP + ntaps - 1;
  float* s_wt = s_mem;                    // [nq][kBlurP]
  float* s_taps = s_mem + nq * kBlurP;    // [ntaps]
  float* s_cum = s_taps + ntaps;          // [ntaps + 1]
  const int r = ntaps / 2;
  const int C4 = C / 4;
  const int n = along_h ? H : W;
  const int nlines = (along_h ? W : H) * C4;
  const int p0 = block_y * kBlurP;
  for (int t = threadIdx.x; t < ntaps; t += blockDim.x) s_taps[t] = taps[t];
  __syncthreads();
  if (ADJ && threadIdx.x == 0) {
    float c = 0.f;
    for (int t = 0; t < ntaps; ++t) {
      s_cum[t] = c;
      c += s_taps[t];
    }
    s_cum[ntaps] = c;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < nq * kBlurP; idx += blockDim.x) {
    const int qi = idx / kBlurP, j = idx - qi * kBlurP;
    const int q = p0 - r + qi, u = p0 + j;
    float w = 0.f;
    if (!ADJ) {
      const int t = qi - j;
      if (t >= 0 && t < ntaps) w = s_taps[t];
    } else if (q >= 0 && q < n && u < n) {
      const int t = u - q + r;
      if (t >= 0 && t < ntaps) w = s_taps[t];
      if (u == 0) w += s_cum[min(max(r - q, 0), ntaps)];                          // taps that land left of 0
      if (u == n - 1) w += s_cum[ntaps] - s_cum[min(max(n - q + r, 0), ntaps)];   // taps that land right of n-1
    }
    s_wt[idx] = w;
  }
  __syncthreads();
  const int line = block_x * blockDim.x + threadIdx.x;
  if (line >= nlines) return;
  long base, stride;
  if (along_h) {
    base = (long)line * 4;  // (x, channel quad) is contiguous in a row
    stride = (long)W * C;
  } else {
    const int y = line / C4, c4 = line - y * C4;
    base = (long)y * W * C + c4 * 4;
    stride = C;
  }
  float4 acc[kBlurP];
#pragma unroll
  for (int j = 0; j < kBlurP; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int q_lo = ADJ ? max(p0 - r, 0) : p0 - r;
  const int q_hi = ADJ ? min(p0 - r + nq, n) : p0 - r + nq;  // the adjoint takes zeros outside, skip them
#pragma unroll 8
  for (int q = q_lo; q < q_hi; ++q) {
    const int qc = ADJ ? q : min(max(q, 0), n - 1);
    const float4 v = ld4(in + base + (long)qc * stride);
    const float* wrow = s_wt + (q - (p0 - r)) * kBlurP;
    float w[kBlurP];
#pragma unroll
    for (int j = 0; j < kBlurP; j += 4) {
      const float4 wq = *reinterpret_cast<const float4*>(wrow + j);
      w[j] = wq.x, w[j + 1] = wq.y, w[j + 2] = wq.z, w[j + 3] = wq.w;
    }
#pragma unroll
    for (int j = 0; j < kBlurP; ++j) {
      acc[j].x += w[j] * v.x;
      acc[j].y += w[j] * v.y;
      acc[j].z += w[j] * v.z;
      acc[j].w += w[j] * v.w;
    }
  }
#pragma unroll
  for (int j = 0; j < kBlurP; ++j)
    if (p0 + j < n) *reinterpret_cast<float4*>(out + base + (long)(p0 + j) * stride) = acc[j];
}

template <bool ADJ>
__global__ __launch_bounds__(256) void k_blur_axis(const float* __restrict__ in, float* __restrict__ out, int H, int W,
                                                   int C, int along_h, const float* __restrict__ taps, int ntaps) {
  extern __shared__ __align__(16) float s_dyn[];
  blur_axis_block<ADJ>(in, out, H, W, C, along_h, taps, ntaps, blockIdx.x, blockIdx.y, s_dyn);
}

// ---- all factors of a scene in one launch per pass ----------------------------------------------------------
struct BlurPass {
  const float* in;
  float* out;
  const float* taps;
  int H, W, C, along_h, ntaps;
  int gx, block0;  // blocks along the lines, first flat block index of this item
};

constexpr int kBlurMaxItems = 12;

struct BlurBatch {
  BlurPass p[kBlurMaxItems];
  int n;
};

template <bool ADJ>
__global__ __launch_bounds__(128) void k_blur_batch(BlurBatch B) {
  extern __shared__ __align__(16) float s_dyn[];
  int it = 0;
#pragma unroll 1
  for (int i = 1; i < B.n; ++i)
    if ((int)blockIdx.x >= B.p[i].block0) it = i;
  const BlurPass& P = B.p[it];
  const int local = blockIdx.x - P.block0;
  blur_axis_block<ADJ>(P.in, P.out, P.H, P.W, P.C, P.along_h, P.taps, P.ntaps, local % P.gx, local / P.gx, s_dyn);
}

// ---- round 4: the same pass with the line staged in LDS -------------------------------------------------------------------
// The kernels above let every thread fetch its own kBlurP + ntaps - 1 inputs: 9 x the bytes through the vector-memory path,
// in 16-byte pieces 192 bytes apart (along W) or a whole row apart (along H) -- 171 us per pass over the 123 MB of a 400^3 scene's
// factors, 23 % of what streaming them once costs (profiles/round4_blur_trace.txt).  Here a workgroup owns LINE CHUNKS: all n
// positions along the blurred axis of one line (a row for the pass along W, a column for the pass along H) x kLineQ channel
// quads (64 contiguous bytes per position).  The chunk is read ONCE into LDS as [quad][position] (replicate halo filled in for
// the forward pass, zero-extended for the adjoint), every thread then slides over its kBlurP + ntaps - 1 inputs out of LDS with
// the same block-uniform weight table, and writes its outputs.  Both passes of a plane are this one kernel with a different
// position stride; the adjoint's two border sums (the taps that the forward's replicate padding sent to texel 0 / n - 1) are
// two short extra sums per quad.
constexpr int kLineQ = 4;     // channel quads per line chunk
constexpr int kLineP = 8;     // outputs per thread
constexpr int kLineThreads = 256;

struct LinePass {
  const float* in;
  float* out;
  const float* taps;
  int n, nlines, C, ntaps;      // positions along the blurred axis, lines across it, channels
  long pos_stride, line_stride; // floats between positions / between lines (a "line" = the other axis index)
  int chunk0;                   // first flat chunk index of this item; chunks = nlines * (C / 4 / kLineQ)
};
struct LineBatch {
  LinePass p[kBlurMaxItems];
  int n;
  int total;  // chunks over all items
};

template <bool ADJ>
__global__ __launch_bounds__(kLineThreads) void k_blur_line(LineBatch B, int npad) {
  extern __shared__ __align__(16) float s_dyn[];
  float4* s_in = reinterpret_cast<float4*>(s_dyn);                 // [kLineQ][npad]
  float* s_wt = s_dyn + (size_t)kLineQ * npad * 4;                 // [kLineP + ntaps - 1][kLineP]
  // persistent: a workgroup takes a CONTIGUOUS range of chunks -- the chunks of one line (64-byte pieces of the same 128-byte
  // lines) one after the other, so that the second and third find their lines in the cache; the weight table is rebuilt only
  // when the tap vector changes (density / colour items)
  const int per = (B.total + (int)gridDim.x - 1) / (int)gridDim.x;
  const int c_begin = (int)blockIdx.x * per, c_end = min(c_begin + per, B.total);
  const float* cur_taps = nullptr;
  int cur_ntaps = 0;
  struct Desc {               // (by value: a pointer into the by-value launch argument would put the whole batch on the stack)
    const float* src;
    float* dst;
    const float* taps;
    long pos_stride;
    int n, r, ntaps, nquad, ngroups, elems;
  };
  auto describe = [&](int chunk) {
    int it = 0;
#pragma unroll 1
    for (int i = 1; i < B.n; ++i)
      if (chunk >= B.p[i].chunk0) it = i;
    Desc d;
    const LinePass& P = B.p[it];
    d.taps = P.taps, d.pos_stride = P.pos_stride;
    d.ntaps = P.ntaps, d.r = d.ntaps / 2, d.n = P.n;
    const int C4 = P.C / 4, cpl = (C4 + kLineQ - 1) / kLineQ;   // chunks per line (the last may be partial: VM-20 has 5 quads)
    const int local = chunk - P.chunk0;
    const int line = local / cpl, cq = local - line * cpl;
    d.nquad = min(kLineQ, C4 - cq * kLineQ);
    d.src = P.in + (long)line * P.line_stride + cq * (kLineQ * 4);
    d.dst = P.out + (long)line * P.line_stride + cq * (kLineQ * 4);
    d.ngroups = (d.n + kLineP - 1) / kLineP;
    // positions -r .. (n rounded up to whole groups) + r - 1: forward = replicate padding, adjoint = zeros outside (the window
    // of the last group reaches past n + r - 1; its weights there are zero, the values must still be finite)
    d.elems = (d.ngroups * kLineP + 2 * d.r) * kLineQ;
    return d;
  };
  auto element = [&](const Desc& d, int idx) {
    const int q = idx % kLineQ, pos = idx / kLineQ - d.r;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#if JT_BLUR_ABL & 2   // profiling knob: no global loads (what the fetch of the chunks costs)
    return make_float4((float)idx, 1.f, 2.f, 3.f);
#endif
    if (q < d.nquad && (!ADJ || (pos >= 0 && pos < d.n)))
      v = ld4(d.src + (long)min(max(pos, 0), d.n - 1) * d.pos_stride + q * 4);
    return v;
  };
  // (fetching the next chunk into registers while this one is computed was measured: 92 / 109 us per pass against 98 / 101 --
  //  the pass is not waiting for its loads, profiles/round4_blur_trace.txt)
  for (int chunk = c_begin; chunk < c_end; ++chunk) {
    const Desc d = describe(chunk);
    const Desc& P = d;
    const int ntaps = d.ntaps, r = d.r, n = d.n, nq = kLineP + ntaps - 1, nquad = d.nquad, ngroups = d.ngroups;
    float* dst = d.dst;
    float* s_cum = s_wt + nq * kLineP;                // [ntaps + 1] (adjoint: prefix sums of the taps)
    __syncthreads();                                  // the previous chunk's readers are done with the LDS
    // weight table Wt[qi][j]: input p0 - r + qi feeds output p0 + j with taps[t], t = qi - j (forward) / 2 r - (qi - j) (adjoint)
    if (P.taps != cur_taps || ntaps != cur_ntaps) {
      cur_taps = P.taps, cur_ntaps = ntaps;
      for (int idx = threadIdx.x; idx < nq * kLineP; idx += kLineThreads) {
        const int qi = idx / kLineP, j = idx - qi * kLineP;
        const int t = ADJ ? 2 * r - (qi - j) : qi - j;
        s_wt[idx] = (t >= 0 && t < ntaps) ? P.taps[t] : 0.f;
      }
      if (ADJ && threadIdx.x < 64) {   // inclusive prefix of the taps by one wave: s_cum[t] = sum of taps[0 .. t-1]
        float carry = 0.f;
        for (int t0 = 0; t0 < ntaps; t0 += 64) {
          const int t = t0 + (int)threadIdx.x;
          float v = t < ntaps ? P.taps[t] : 0.f;
#pragma unroll
          for (int o = 1; o < 64; o <<= 1) {
            const float u = __shfl_up(v, o);
            if ((int)threadIdx.x >= o) v += u;
          }
          if (t < ntaps) s_cum[t + 1] = carry + v;
          carry += __shfl(v, 63);
        }
        if (threadIdx.x == 0) s_cum[0] = 0.f;
      }
    }
    for (int idx = threadIdx.x; idx < d.elems; idx += kLineThreads) s_in[(idx % kLineQ) * npad + idx / kLineQ] = element(d, idx);
    __syncthreads();
    for (int item = threadIdx.x; item < ngroups * kLineQ; item += kLineThreads) {
      const int q = item % kLineQ, p0 = (item / kLineQ) * kLineP;
      if (q >= nquad) continue;
      const float4* row = s_in + q * npad + p0;       // input p0 - r + qi sits at row[qi]
      float4 acc[kLineP];
#pragma unroll
      for (int j = 0; j < kLineP; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
#if JT_BLUR_ABL & 1   // profiling knob: one input per item instead of kLineP + ntaps - 1 (what the arithmetic costs)
      const int nq_run = 1;
#else
      const int nq_run = nq;
#endif
#pragma unroll 4
      for (int qi = 0; qi < nq_run; ++qi) {
        const float4 v = row[qi];
        const float4 w0 = *reinterpret_cast<const float4*>(s_wt + qi * kLineP);
        const float4 w1 = *reinterpret_cast<const float4*>(s_wt + qi * kLineP + 4);
        const float w[kLineP] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
        for (int j = 0; j < kLineP; ++j) {
          acc[j].x += w[j] * v.x;
          acc[j].y += w[j] * v.y;
          acc[j].z += w[j] * v.z;
          acc[j].w += w[j] * v.w;
        }
      }
      if (ADJ) {
        // what the forward's replicate padding read from texel 0 (positions < 0) and texel n - 1 (positions >= n) comes back to
        // them: g_in[0] += sum_x g[x] sum_{t < r - x} k[t],  g_in[n-1] += sum_x g[x] sum_{t >= n - x + r} k[t]
        if (p0 == 0) {
          float4 e = make_float4(0.f, 0.f, 0.f, 0.f);
          for (int x = 0; x < min(r, n); ++x) {
            const float c = s_cum[r - x];
            const float4 g = s_in[q * npad + x + r];
            e.x += c * g.x, e.y += c * g.y, e.z += c * g.z, e.w += c * g.w;
          }
          acc[0].x += e.x, acc[0].y += e.y, acc[0].z += e.z, acc[0].w += e.w;
        }
        if (p0 <= n - 1 && n - 1 < p0 + kLineP) {
          float4 e = make_float4(0.f, 0.f, 0.f, 0.f);
          for (int x = max(n - r, 0); x < n; ++x) {
            const float c = s_cum[ntaps] - s_cum[min(max(n - x + r, 0), ntaps)];
            const float4 g = s_in[q * npad + x + r];
            e.x += c * g.x, e.y += c * g.y, e.z += c * g.z, e.w += c * g.w;
          }
          const int j = n - 1 - p0;
#pragma unroll
          for (int jj = 0; jj < kLineP; ++jj)
            if (jj == j) acc[jj].x += e.x, acc[jj].y += e.y, acc[jj].z += e.z, acc[jj].w += e.w;
        }
      }
#pragma unroll
      for (int j = 0; j < kLineP; ++j)
        if (p0 + j < n) *reinterpret_cast<float4*>(dst + (long)(p0 + j) * P.pos_stride + q * 4) = acc[j];
    }
  }
}


// ---- the pass on the matrix cores (round 5) ----------------------------------------------------------------------------------
// A 65-tap correlation of a line is a banded Toeplitz product: out[p] = sum_q T[p][q] in[q], T[p][q] = taps[q - p + r].  For a
// block of 16 outputs the band is 16 + 64 = 80 inputs wide: twenty v_mfma_f32_16x16x4_f32 with A = the block's slice of T
// (THE SAME for every block of every line: twenty registers per lane, built once per tap vector), B = 16 channels of four
// consecutive input positions straight from the LDS chunk, D = 16 outputs x 16 channels.  k_blur_line does the same sums as
// 16 packed FMAs per input on the vector pipe (98 us per pass at 400^3: vector-issue bound, 2.5 TB/s of the 6.3 TB/s streaming
// roof); here the vector pipe only moves data.  fp32 inputs, fp32 products, fp32 accumulation: the same precision.
// A workgroup of four waves stages one line chunk (all positions + halo x 16 channels: 30 KB at n = 400) and its waves take
// pairs of output blocks (two independent accumulator chains per wave).  Taps <= kMfmaTaps (the factor blur's 65; the 201-tap
// 2-D blur of the supervising images keeps the vector kernel).
constexpr int kMfmaTaps = 65;
constexpr int kMfmaK = (16 + kMfmaTaps - 1 + 3) / 4;   // K steps of a block: 20
typedef float blur_f32x4 __attribute__((ext_vector_type(4)));

template <bool ADJ>
__global__ __launch_bounds__(256) void k_blur_mfma(LineBatch B, int npos) {
  extern __shared__ __align__(16) float s_dyn[];
  float* s_in = s_dyn;                       // [npos][16]: position -r + idx, channels of the chunk
  float* s_cum = s_dyn + (size_t)npos * 16;  // [ntaps + 1] prefix sums of the taps (adjoint borders)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, nn = lane & 15;
  const int per = (B.total + (int)gridDim.x - 1) / (int)gridDim.x;
  const int c_begin = (int)blockIdx.x * per, c_end = min(c_begin + per, B.total);
  const float* cur_taps = nullptr;
  int cur_ntaps = 0;
  float a[kMfmaK];
#pragma unroll
  for (int kk = 0; kk < kMfmaK; ++kk) a[kk] = 0.f;
  for (int chunk = c_begin; chunk < c_end; ++chunk) {
    int it = 0;
#pragma unroll 1
    for (int i = 1; i < B.n; ++i)
      if (chunk >= B.p[i].chunk0) it = i;
    const LinePass& P = B.p[it];
    const float* taps = P.taps;
    const long pos_stride = P.pos_stride;
    const int ntaps = P.ntaps, r = ntaps / 2, n = P.n;
    const int C4 = P.C / 4, cpl = (C4 + kLineQ - 1) / kLineQ;
    const int local = chunk - P.chunk0;
    const int line = local / cpl, cq = local - line * cpl;
    const int nch = 4 * min(kLineQ, C4 - cq * kLineQ);          // live channels of the chunk (16, or a partial last chunk)
    const float* src = P.in + (long)line * P.line_stride + cq * (kLineQ * 4);
    float* dst = P.out + (long)line * P.line_stride + cq * (kLineQ * 4);
    const int nblocks = (n + 15) >> 4;
    const int npos_c = nblocks * 16 + 4 * kMfmaK - 16;          // positions -r .. : what the last block's band reaches
    __syncthreads();                                            // the previous chunk's readers are done with the LDS
    if (taps != cur_taps || ntaps != cur_ntaps) {
      cur_taps = taps, cur_ntaps = ntaps;
      // A: row = output j of the block (lane & 15), K step kk, slice g  <->  input qi = 4 kk + g of the band (position p0 - r + qi)
#pragma unroll
      for (int kk = 0; kk < kMfmaK; ++kk) {
        const int qi = 4 * kk + g;
        const int t = ADJ ? 2 * r - (qi - nn) : qi - nn;
        a[kk] = (t >= 0 && t < ntaps) ? taps[t] : 0.f;
      }
      if (ADJ && threadIdx.x < 64) {   // inclusive prefix of the taps by one wave: s_cum[t] = sum of taps[0 .. t-1]
        float carry = 0.f;
        for (int t0 = 0; t0 < ntaps; t0 += 64) {
          const int t = t0 + (int)threadIdx.x;
          float v = t < ntaps ? taps[t] : 0.f;
#pragma unroll
          for (int o = 1; o < 64; o <<= 1) {
            const float u = __shfl_up(v, o);
            if ((int)threadIdx.x >= o) v += u;
          }
          if (t < ntaps) s_cum[t + 1] = carry + v;
          carry += __shfl(v, 63);
        }
        if (threadIdx.x == 0) s_cum[0] = 0.f;
      }
    }
    // stage: forward = replicate padding, adjoint = zeros outside; idle channels of a partial chunk are zero.  Eight loads in
    // flight per thread before the first LDS write (as a plain loop every load waited for the one before it)
    for (int base = 0; base < npos_c * 4; base += 256 * 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = base + u * 256 + (int)threadIdx.x;
        const int q = idx & 3, pos = (idx >> 2) - r;
        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (idx < npos_c * 4 && 4 * q < nch && (!ADJ || (pos >= 0 && pos < n)))
          v[u] = ld4(src + (long)min(max(pos, 0), n - 1) * pos_stride + q * 4);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = base + u * 256 + (int)threadIdx.x;
        if (idx < npos_c * 4) *reinterpret_cast<float4*>(s_in + (size_t)(idx >> 2) * 16 + (idx & 3) * 4) = v[u];
      }
    }
    __syncthreads();
    // adjoint: what the forward's replicate padding read from texel 0 / n - 1 comes back to them (k_blur_line's two sums),
    // per channel: the four lane groups take every fourth x
    float e0 = 0.f, e1 = 0.f;
    if (ADJ) {
      for (int x = g; x < min(r, n); x += 4) e0 += s_cum[r - x] * s_in[(size_t)(x + r) * 16 + nn];
      for (int x = max(n - r, 0) + g; x < n; x += 4)
        e1 += (s_cum[ntaps] - s_cum[min(max(n - x + r, 0), ntaps)]) * s_in[(size_t)(x + r) * 16 + nn];
      e0 += __shfl_xor(e0, 16), e0 += __shfl_xor(e0, 32);
      e1 += __shfl_xor(e1, 16), e1 += __shfl_xor(e1, 32);
    }
    for (int b0 = 2 * wv; b0 < nblocks; b0 += 8) {
      const bool two = b0 + 1 < nblocks;
      blur_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
      const float* bp = s_in + (size_t)(16 * b0 + g) * 16 + nn;   // B: K slice g = position p0 - r + 4 kk + g, column = channel
#if JT_BLUR_ABL & 4   // profiling knob: one K step per block pair (what the matrix products and their LDS reads cost)
#pragma unroll
      for (int kk = 0; kk < 1; ++kk) {
#else
#pragma unroll
      for (int kk = 0; kk < kMfmaK; ++kk) {
#endif
        const float v0 = bp[kk * 64];
        const float v1 = two ? bp[kk * 64 + 256] : 0.f;
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kk], v0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kk], v1, acc1, 0, 0, 0);
      }
      // D: register i of lane (g, channel) is output p0 + 4 g + i
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int p0 = 16 * b0 + 4 * g + i, p1 = p0 + 16;
        float o0 = acc0[i], o1 = acc1[i];
        if (ADJ) {
          if (p0 == 0) o0 += e0;
          if (p0 == n - 1) o0 += e1;
          if (p1 == n - 1) o1 += e1;
        }
        if (nn < nch) {
          if (p0 < n) dst[(long)p0 * pos_stride + nn] = o0;
          if (two && p1 < n) dst[(long)p1 * pos_stride + nn] = o1;
        }
      }
    }
  }
}

}  // namespace jt

using namespace jt;

static int blur_args(const void* a, const void* b, const void* tmp, int H, int W, int C, const float* taps,
                     int n_taps) {
  if (!a || !b || !taps || H < 1 || W < 1 || C < 4 || n_taps < 1) return JT_ERR_ARG;
  if ((C % 4) != 0 || (n_taps % 2) != 1) return JT_ERR_UNSUPPORTED;
  if (W > 1 && H > 1 && !tmp) return JT_ERR_ARG;
  return JT_OK;
}

template <bool ADJ>
static bool launch_line_batch(const BlurPass* passes, int n, hipStream_t st);

template <bool ADJ>
static void launch_blur_axis(const float* in, float* out, int H, int W, int C, int along_h, const float* taps,
                             int n_taps, hipStream_t st) {
  BlurPass one = {in, out, taps, H, W, C, along_h, n_taps, 0, 0};
  if (launch_line_batch<ADJ>(&one, 1, st)) return;
  const int n = along_h ? H : W;
  const int nlines = (along_h ? W : H) * (C / 4);
  const int threads = nlines >= 4096 ? 256 : (nlines >= 128 ? 128 : 64);
  dim3 grid((nlines + threads - 1) / threads, (n + kBlurP - 1) / kBlurP);
  const size_t lds = ((size_t)(kBlurP + n_taps - 1) * kBlurP + 2 * n_taps + 1) * sizeof(float);
  hipLaunchKernelGGL(k_blur_axis<ADJ>, grid, dim3(threads), lds, st, in, out, H, W, C, along_h, taps, n_taps);
}

extern "C" int jt_blur_forward(const float* in, float* out, float* tmp, int H, int W, int C, const float* taps,
                               int n_taps, void* stream) {
  int rc = blur_args(in, out, tmp, H, W, C, taps, n_taps);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (W > 1 && H > 1) {
    // along W (last logical axis) first, then along H -- the order of bateRF.py:28-36
    launch_blur_axis<false>(in, tmp, H, W, C, 0, taps, n_taps, st);
    JT_LAUNCH_CHECK();
    launch_blur_axis<false>(tmp, out, H, W, C, 1, taps, n_taps, st);
  } else {
    launch_blur_axis<false>(in, out, H, W, C, H > 1 ? 1 : 0, taps, n_taps, st);
  }
  JT_LAUNCH_CHECK();
  return JT_OK;
}

extern "C" int jt_blur_backward(const float* g_out, float* g_in, float* tmp, int H, int W, int C,
                                const float* taps, int n_taps, void* stream) {
  int rc = blur_args(g_out, g_in, tmp, H, W, C, taps, n_taps);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (W > 1 && H > 1) {
    launch_blur_axis<true>(g_out, tmp, H, W, C, 1, taps, n_taps, st);
    JT_LAUNCH_CHECK();
    launch_blur_axis<true>(tmp, g_in, H, W, C, 0, taps, n_taps, st);
  } else {
    launch_blur_axis<true>(g_out, g_in, H, W, C, H > 1 ? 1 : 0, taps, n_taps, st);
  }
  JT_LAUNCH_CHECK();
  return JT_OK;
}

// the LDS-staged line kernel for one pass of a batch; false = a shape it does not take (the caller falls back)
template <bool ADJ>
static bool launch_line_batch(const BlurPass* passes, int n, hipStream_t st) {
  static const bool enabled = [] { const char* e = getenv("JT_BLUR_LDS"); return !e || atoi(e) != 0; }();
  if (!enabled) return false;
  LineBatch L;
  L.n = n;
  int total = 0, max_n = 1, max_taps = 1;
  for (int i = 0; i < n; ++i) {
    const BlurPass& P = passes[i];
    LinePass& Q = L.p[i];
    Q.in = P.in, Q.out = P.out, Q.taps = P.taps, Q.C = P.C, Q.ntaps = P.ntaps;
    if (P.along_h) {
      Q.n = P.H, Q.nlines = P.W, Q.pos_stride = (long)P.W * P.C, Q.line_stride = P.C;
    } else {
      Q.n = P.W, Q.nlines = P.H, Q.pos_stride = P.C, Q.line_stride = (long)P.W * P.C;
    }
    Q.chunk0 = total;
    total += Q.nlines * ((P.C / 4 + kLineQ - 1) / kLineQ);
    max_n = std::max(max_n, Q.n);
    max_taps = std::max(max_taps, P.ntaps);
  }
  L.total = total;
  // the matrix-core pass where its tap count and LDS chunk fit (JT_BLUR_MFMA=0, read once: the vector kernel below)
  static const bool mfma_on = [] { const char* e = getenv("JT_BLUR_MFMA"); return !e || atoi(e) != 0; }();
  if (mfma_on && max_taps <= kMfmaTaps) {
    const int npos = (max_n + 15) / 16 * 16 + 4 * kMfmaK - 16;
    const size_t lds_m = ((size_t)npos * 16 + kMfmaTaps + 1) * sizeof(float);
    if (lds_m <= 64 * 1024) {
      static bool mattr[2] = {false, false};
      if (!mattr[ADJ]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_blur_mfma<ADJ>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  64 * 1024);
        mattr[ADJ] = true;
      }
      const int per_cu_m = std::max(1, std::min(5, (int)(160 * 1024 / (lds_m + 512))));
      const int blocks_m = std::min(total, 256 * per_cu_m);
      hipLaunchKernelGGL(k_blur_mfma<ADJ>, dim3(blocks_m), dim3(256), lds_m, st, L, npos);
      return true;
    }
  }
  // [quad][npad] with npad = 2 mod 16 float4s: the four quads of a 16-lane LDS group land on different bank quarters
  int npad = (max_n + kLineP - 1) / kLineP * kLineP + max_taps;
  npad += (18 - (npad % 16)) % 16;
  const size_t lds = ((size_t)kLineQ * npad * 4 + (size_t)(kLineP + max_taps - 1) * kLineP + max_taps + 1) * sizeof(float);
  if (lds > 64 * 1024) return false;  // (two workgroups per CU at least)
  static bool attr[2] = {false, false};
  if (!attr[ADJ]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_blur_line<ADJ>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              64 * 1024);
    attr[ADJ] = true;
  }
  const int per_cu = std::max(1, std::min(8, (int)(160 * 1024 / (lds + 512))));
  const int blocks = std::min(total, 256 * per_cu);
  hipLaunchKernelGGL(k_blur_line<ADJ>, dim3(blocks), dim3(kLineThreads), lds, st, L, npad);
  return true;
}

// pass 0 / 1 of the batch: planes run (W, then H) forward and (H, then W) backward; lines have one pass
template <bool ADJ>
static int launch_blur_batch(const JtBlurItem* items, int n_items, int pass, hipStream_t st) {
  BlurBatch B;
  B.n = 0;
  int blocks = 0, max_taps = 1;
  for (int i = 0; i < n_items; ++i) {
    const JtBlurItem& it = items[i];
    const bool plane = it.H > 1 && it.W > 1;
    if (!plane && pass == 1) continue;
    BlurPass& P = B.p[B.n];
    P.taps = it.taps;
    P.ntaps = it.n_taps;
    P.H = it.H;
    P.W = it.W;
    P.C = it.C;
    if (plane) {
      const int first_axis = ADJ ? 1 : 0;  // forward: along W then H (bateRF.py:28-36); adjoint: the reverse
      P.along_h = (pass == 0) ? first_axis : 1 - first_axis;
      P.in = (pass == 0) ? it.in : it.tmp;
      P.out = (pass == 0) ? it.tmp : it.out;
    } else {
      P.along_h = it.H > 1 ? 1 : 0;
      P.in = it.in;
      P.out = it.out;
    }
    const int n = P.along_h ? P.H : P.W;
    const int nlines = (P.along_h ? P.W : P.H) * (P.C / 4);
    P.gx = (nlines + 127) / 128;
    P.block0 = blocks;
    blocks += P.gx * ((n + kBlurP - 1) / kBlurP);
    max_taps = std::max(max_taps, P.ntaps);
    ++B.n;
  }
  if (B.n == 0) return JT_OK;
  if (launch_line_batch<ADJ>(B.p, B.n, st)) {
    JT_LAUNCH_CHECK();
    return JT_OK;
  }
  const size_t lds = ((size_t)(kBlurP + max_taps - 1) * kBlurP + 2 * max_taps + 1) * sizeof(float);
  hipLaunchKernelGGL(k_blur_batch<ADJ>, dim3(blocks), dim3(128), lds, st, B);
  JT_LAUNCH_CHECK();
  return JT_OK;
}

static int blur_batch_args(const JtBlurItem* items, int n_items) {
  if (!items || n_items < 1) return JT_ERR_ARG;
  if (n_items > kBlurMaxItems) return JT_ERR_UNSUPPORTED;
  for (int i = 0; i < n_items; ++i) {
    int rc = blur_args(items[i].in, items[i].out, items[i].tmp, items[i].H, items[i].W, items[i].C, items[i].taps,
                       items[i].n_taps);
    if (rc) return rc;
  }
  return JT_OK;
}

extern "C" int jt_blur_batch_forward(const JtBlurItem* items, int n_items, void* stream) {
  int rc = blur_batch_args(items, n_items);
  if (rc) return rc;
  if ((rc = launch_blur_batch<false>(items, n_items, 0, (hipStream_t)stream))) return rc;
  return launch_blur_batch<false>(items, n_items, 1, (hipStream_t)stream);
}

extern "C" int jt_blur_batch_backward(const JtBlurItem* items, int n_items, void* stream) {
  int rc = blur_batch_args(items, n_items);
  if (rc) return rc;
  if ((rc = launch_blur_batch<true>(items, n_items, 0, (hipStream_t)stream))) return rc;
  return launch_blur_batch<true>(items, n_items, 1, (hipStream_t)stream);
}
