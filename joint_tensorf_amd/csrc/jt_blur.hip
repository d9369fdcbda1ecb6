// Separable 1-D blur of channel-last VM factors with replicate padding (cross-correlation), and its
// exact adjoint.  Replaces BAT_VMSplit.convolute_plane / convolute_line (bateRF.py:8-39): pad by
// K//2 on both sides (replicate), correlate along W, then along H, same taps for every channel.
//
// Layout [H][W][C]: channels are the fastest axis, so a work-item owns 4 consecutive channels (16-byte
// accesses, coalesced across the channel lanes) of kBlurP consecutive positions along the blurred axis and
// slides over the inputs that reach them; all factors of a scene go through one launch per pass.
#include <algorithm>

#include "jt_common.h"

namespace jt {

constexpr int kBlurP = 8;  // outputs per thread along the blurred axis

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
    const float4 w0 = *reinterpret_cast<const float4*>(wrow), w1 = *reinterpret_cast<const float4*>(wrow + 4);
    const float w[kBlurP] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
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
static void launch_blur_axis(const float* in, float* out, int H, int W, int C, int along_h, const float* taps,
                             int n_taps, hipStream_t st) {
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
