// Separable 1-D blur of channel-last VM factors with replicate padding (cross-correlation), and its
// exact adjoint.  Replaces BAT_VMSplit.convolute_plane / convolute_line (bateRF.py:8-39): pad by
// K//2 on both sides (replicate), correlate along W, then along H, same taps for every channel.
//
// Layout [H][W][C]: channels are the fastest axis, so a work-item owns 4 consecutive channels of
// one texel (16-byte accesses, fully coalesced across the channel lanes) and walks the taps along
// the blurred axis with stride W*C or C.
#include "jt_common.h"

namespace jt {

// forward along one axis: out[p] = sum_t k[t] * in[clamp(p + t - r, 0, n-1)]
// axis_len = n, axis_stride = element stride between neighbours along the blurred axis,
// `outer` enumerates all (other-axis, channel-quad) positions.
__global__ __launch_bounds__(256) void k_blur_axis_fwd(const float* __restrict__ in, float* __restrict__ out,
                                                       int H, int W, int C, int along_h,
                                                       const float* __restrict__ taps, int ntaps) {
  extern __shared__ float s_taps[];
  for (int t = threadIdx.x; t < ntaps; t += blockDim.x) s_taps[t] = taps[t];
  __syncthreads();
  const int r = ntaps / 2;
  const int C4 = C / 4;
  const long total = (long)H * W * C4;
  const int n = along_h ? H : W;
  const long stride = along_h ? (long)W * C : C;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(idx % C4);
    const long tex = idx / C4;
    const int x = (int)(tex % W), y = (int)(tex / W);
    const int p = along_h ? y : x;
    const float* base = in + ((long)y * W + x) * C + c4 * 4 - (long)p * stride;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = 0; t < ntaps; ++t) {
      int q = min(max(p + t - r, 0), n - 1);
      float4 v = ld4(base + (long)q * stride);
      float k = s_taps[t];
      acc.x += k * v.x;
      acc.y += k * v.y;
      acc.z += k * v.z;
      acc.w += k * v.w;
    }
    *reinterpret_cast<float4*>(out + ((long)y * W + x) * C + c4 * 4) = acc;
  }
}

// adjoint along one axis: g_in[u] = sum_x g_out[x] * sum_t k[t] [clamp(x + t - r) == u]
//   interior part : t = u - x + r  (0 <= t < ntaps)
//   u == 0        : additionally all t with x + t - r < 0   -> prefix  sum_{t < r - x} k[t]
//   u == n-1      : additionally all t with x + t - r > n-1 -> suffix  sum_{t > n-1-x+r} k[t]
// cum[t] = sum_{j < t} k[j] (cum[0] = 0, cum[ntaps] = total) is passed in shared memory.
__global__ __launch_bounds__(256) void k_blur_axis_bwd(const float* __restrict__ g_out, float* __restrict__ g_in,
                                                       int H, int W, int C, int along_h,
                                                       const float* __restrict__ taps, int ntaps) {
  extern __shared__ float s_mem[];
  float* s_taps = s_mem;
  float* s_cum = s_mem + ntaps;
  for (int t = threadIdx.x; t < ntaps; t += blockDim.x) s_taps[t] = taps[t];
  __syncthreads();
  if (threadIdx.x == 0) {
    float c = 0.f;
    for (int t = 0; t < ntaps; ++t) {
      s_cum[t] = c;
      c += s_taps[t];
    }
    s_cum[ntaps] = c;
  }
  __syncthreads();
  const int r = ntaps / 2;
  const int C4 = C / 4;
  const long total = (long)H * W * C4;
  const int n = along_h ? H : W;
  const long stride = along_h ? (long)W * C : C;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(idx % C4);
    const long tex = idx / C4;
    const int x = (int)(tex % W), y = (int)(tex / W);
    const int u = along_h ? y : x;
    const float* base = g_out + ((long)y * W + x) * C + c4 * 4 - (long)u * stride;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int xlo = (u == 0 || u == n - 1) ? 0 : max(u - r, 0);
    const int xhi = (u == 0 || u == n - 1) ? n - 1 : min(u + r, n - 1);
    for (int xx = xlo; xx <= xhi; ++xx) {
      int t = u - xx + r;
      float k = (t >= 0 && t < ntaps) ? s_taps[t] : 0.f;
      if (u == 0) {
        int m = min(max(r - xx, 0), ntaps);  // taps t < r - xx land left of 0
        k += s_cum[m];
      }
      if (u == n - 1) {
        int first = min(max(n - xx + r, 0), ntaps);  // taps t >= n - xx + r land right of n-1
        k += s_cum[ntaps] - s_cum[first];
      }
      float4 v = ld4(base + (long)xx * stride);
      acc.x += k * v.x;
      acc.y += k * v.y;
      acc.z += k * v.z;
      acc.w += k * v.w;
    }
    *reinterpret_cast<float4*>(g_in + ((long)y * W + x) * C + c4 * 4) = acc;
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

static int blur_grid(int H, int W, int C) {
  long total = (long)H * W * (C / 4);
  return (int)min((total + 255) / 256, 4096L);
}

extern "C" int jt_blur_forward(const float* in, float* out, float* tmp, int H, int W, int C, const float* taps,
                               int n_taps, void* stream) {
  int rc = blur_args(in, out, tmp, H, W, C, taps, n_taps);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const int g = blur_grid(H, W, C);
  const size_t lds = n_taps * sizeof(float);
  if (W > 1 && H > 1) {
    // along W (last logical axis) first, then along H -- the order of bateRF.py:28-36
    hipLaunchKernelGGL(k_blur_axis_fwd, dim3(g), dim3(256), lds, st, in, tmp, H, W, C, 0, taps, n_taps);
    JT_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_blur_axis_fwd, dim3(g), dim3(256), lds, st, (const float*)tmp, out, H, W, C, 1, taps,
                       n_taps);
  } else {
    hipLaunchKernelGGL(k_blur_axis_fwd, dim3(g), dim3(256), lds, st, in, out, H, W, C, H > 1 ? 1 : 0, taps,
                       n_taps);
  }
  JT_LAUNCH_CHECK();
  return JT_OK;
}

extern "C" int jt_blur_backward(const float* g_out, float* g_in, float* tmp, int H, int W, int C,
                                const float* taps, int n_taps, void* stream) {
  int rc = blur_args(g_out, g_in, tmp, H, W, C, taps, n_taps);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const int g = blur_grid(H, W, C);
  const size_t lds = (2 * n_taps + 1) * sizeof(float);
  if (W > 1 && H > 1) {
    hipLaunchKernelGGL(k_blur_axis_bwd, dim3(g), dim3(256), lds, st, g_out, tmp, H, W, C, 1, taps, n_taps);
    JT_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_blur_axis_bwd, dim3(g), dim3(256), lds, st, (const float*)tmp, g_in, H, W, C, 0, taps,
                       n_taps);
  } else {
    hipLaunchKernelGGL(k_blur_axis_bwd, dim3(g), dim3(256), lds, st, g_out, g_in, H, W, C, H > 1 ? 1 : 0, taps,
                       n_taps);
  }
  JT_LAUNCH_CHECK();
  return JT_OK;
}
