// Dense Adam over all parameter tensors of an optimizer in ONE launch.  Replaces the step of
// the torch.optim.Adam the reference builds in tensorf.NeRF._get_optimizer (model/tensorf.py:463-478) with the
// same arithmetic:
//   m = b1 m + (1 - b1) g ;  v = b2 v + (1 - b2) g^2 ;  p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// Seven streams of 123 MB at the final grid: the kernel is pure HBM traffic (16-byte accesses, every tensor in the
// layout it is stored in -- parameter, gradient and both moments share it).
#include <algorithm>

#include "jt_common.h"

namespace jt {

#ifndef JT_ADAM_NT
#define JT_ADAM_NT 1
#endif
typedef float v4f_nt __attribute__((ext_vector_type(4)));
__device__ inline float4 nt_ld4(const float* p) {
  const v4f_nt v = __builtin_nontemporal_load(reinterpret_cast<const v4f_nt*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ inline void nt_st4(float* p, float4 a) {
  v4f_nt v = {a.x, a.y, a.z, a.w};
  __builtin_nontemporal_store(v, reinterpret_cast<v4f_nt*>(p));
}

constexpr int kAdamMaxItems = 32;
constexpr int kAdamElemsPerBlock = 256 * 4 * 4;  // 256 threads x 4 float4

struct AdamItemDev {
  float* p;
  const float* g;
  float* m;
  float* v;
  long n;
  float step_size;      // lr / bias_correction1
  float inv_bc2_sqrt;   // 1 / sqrt(bias_correction2)
  int block0;
};

struct AdamBatch {
  AdamItemDev t[kAdamMaxItems];
  int n;
  float b1, b2, eps;
  // not null: (step_size, inv_bc2_sqrt) of item i are read from dyn[2 i], dyn[2 i + 1] in device memory instead of
  // the launch arguments, so that a hipGraph that captured the launch can be replayed with this iteration's values
  const float* dyn;
};

constexpr int kPokeMaxWords = 256;
struct PokeArgs {
  uint32_t w[kPokeMaxWords];
};

__global__ __launch_bounds__(256) void k_poke(uint32_t* dst, PokeArgs a, int n) {
  if ((int)threadIdx.x < n) dst[threadIdx.x] = a.w[threadIdx.x];
}

__device__ inline float adam1(float& p, float g, float& m, float& v, float b1, float b2, float eps, float step_size,
                              float inv_bc2_sqrt) {
  m = b1 * m + (1.f - b1) * g;
  v = b2 * v + (1.f - b2) * g * g;
  const float denom = sqrtf(v) * inv_bc2_sqrt + eps;
  p -= step_size * (m / denom);
  return p;
}

__global__ __launch_bounds__(256) void k_adam_batch(AdamBatch B) {
  int it = 0;
#pragma unroll 1
  for (int i = 1; i < B.n; ++i)
    if ((int)blockIdx.x >= B.t[i].block0) it = i;
  const AdamItemDev& T = B.t[it];
  const float step_size = B.dyn ? B.dyn[2 * it] : T.step_size;
  const float inv_bc2_sqrt = B.dyn ? B.dyn[2 * it + 1] : T.inv_bc2_sqrt;
  const long base = (long)(blockIdx.x - T.block0) * kAdamElemsPerBlock;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long i = base + ((long)k * 256 + threadIdx.x) * 4;
    if (i + 3 < T.n) {
#if JT_ADAM_NT
      // the gradient and both moments stream through once per iteration; only the parameters are read again soon
      float4 p = *reinterpret_cast<float4*>(T.p + i), m = nt_ld4(T.m + i), v = nt_ld4(T.v + i);
      const float4 g = nt_ld4(T.g + i);
#else
      float4 p = *reinterpret_cast<float4*>(T.p + i), m = *reinterpret_cast<float4*>(T.m + i),
             v = *reinterpret_cast<float4*>(T.v + i);
      const float4 g = ld4(T.g + i);
#endif
      adam1(p.x, g.x, m.x, v.x, B.b1, B.b2, B.eps, step_size, inv_bc2_sqrt);
      adam1(p.y, g.y, m.y, v.y, B.b1, B.b2, B.eps, step_size, inv_bc2_sqrt);
      adam1(p.z, g.z, m.z, v.z, B.b1, B.b2, B.eps, step_size, inv_bc2_sqrt);
      adam1(p.w, g.w, m.w, v.w, B.b1, B.b2, B.eps, step_size, inv_bc2_sqrt);
      *reinterpret_cast<float4*>(T.p + i) = p;
#if JT_ADAM_NT
      nt_st4(T.m + i, m);
      nt_st4(T.v + i, v);
#else
      *reinterpret_cast<float4*>(T.m + i) = m;
      *reinterpret_cast<float4*>(T.v + i) = v;
#endif
    } else {
      for (long j = i; j < T.n && j < i + 4; ++j)
        adam1(T.p[j], T.g[j], T.m[j], T.v[j], B.b1, B.b2, B.eps, step_size, inv_bc2_sqrt);
    }
  }
}

}  // namespace jt

using namespace jt;

static int adam_launch(const JtAdamItem* items, int n_items, float beta1, float beta2, float eps, const float* dyn,
                       void* stream, const float* coefs_host = nullptr) {
  if (!items || n_items < 1) return JT_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  for (int first = 0; first < n_items; first += kAdamMaxItems) {
    AdamBatch B;
    B.dyn = dyn ? dyn + 2 * first : nullptr;
    B.n = std::min(kAdamMaxItems, n_items - first);
    B.b1 = beta1;
    B.b2 = beta2;
    B.eps = eps;
    int blocks = 0;
    for (int i = 0; i < B.n; ++i) {
      const JtAdamItem& s = items[first + i];
      if (!s.p || !s.g || !s.m || !s.v || s.n < 1) return JT_ERR_ARG;
      if (!dyn && !coefs_host && (!(s.bias_correction1 > 0.f) || !(s.bias_correction2 > 0.f))) return JT_ERR_ARG;
      if ((((uintptr_t)s.p | (uintptr_t)s.g | (uintptr_t)s.m | (uintptr_t)s.v) & 15) != 0) return JT_ERR_UNSUPPORTED;
      AdamItemDev& d = B.t[i];
      d.p = s.p;
      d.g = s.g;
      d.m = s.m;
      d.v = s.v;
      d.n = s.n;
      d.step_size = dyn ? 0.f : coefs_host ? coefs_host[2 * (first + i)] : s.lr / s.bias_correction1;
      d.inv_bc2_sqrt = dyn ? 0.f : coefs_host ? coefs_host[2 * (first + i) + 1] : 1.f / sqrtf(s.bias_correction2);
      d.block0 = blocks;
      blocks += (int)((s.n + kAdamElemsPerBlock - 1) / kAdamElemsPerBlock);
    }
    hipLaunchKernelGGL(k_adam_batch, dim3(blocks), dim3(256), 0, st, B);
    JT_LAUNCH_CHECK();
  }
  return JT_OK;
}

extern "C" int jt_adam_step(const JtAdamItem* items, int n_items, float beta1, float beta2, float eps, void* stream) {
  return adam_launch(items, n_items, beta1, beta2, eps, nullptr, stream);
}

extern "C" int jt_adam_step_dyn(const JtAdamItem* items, int n_items, float beta1, float beta2, float eps,
                                const float* dyn, void* stream) {
  if (!dyn) return JT_ERR_ARG;
  return adam_launch(items, n_items, beta1, beta2, eps, dyn, stream);
}

extern "C" int jt_adam_step_coefs(const JtAdamItem* items, int n_items, float beta1, float beta2, float eps,
                                  const float* coefs_host, void* stream) {
  if (!coefs_host) return JT_ERR_ARG;
  return adam_launch(items, n_items, beta1, beta2, eps, nullptr, stream, coefs_host);
}

extern "C" int jt_poke(void* dst, const uint32_t* words, int n_words, void* stream) {
  if (!dst || !words || n_words < 1 || n_words > kPokeMaxWords) return JT_ERR_ARG;
  if (((uintptr_t)dst & 3) != 0) return JT_ERR_UNSUPPORTED;
  PokeArgs a;
  for (int i = 0; i < n_words; ++i) a.w[i] = words[i];
  hipLaunchKernelGGL(k_poke, dim3(1), dim3(256), 0, (hipStream_t)stream, (uint32_t*)dst, a, n_words);
  JT_LAUNCH_CHECK();
  return JT_OK;
}
