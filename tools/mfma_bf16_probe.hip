// Probe of v_mfma_f32_32x32x16_bf16 on gfx950: (1) the lane -> (row, k) mapping assumed by the split-bf16 forward chain
// (A: lane l holds A[l % 32][8 (l / 32) + e], B: lane l holds B[8 (l / 32) + e][l % 32], D: register r of lane l holds
// D[(r & 3) + 8 (r >> 2) + 4 (l / 32)][l % 32]); (2) the error of an fp32 product sum emulated with three bf16 pieces per
// operand and six MFMAs (a0 b0 + a0 b1 + a1 b0 + a1 b1 + a0 b2 + a2 b0; without a1 b1 the error is 2^-18 relative: bf16
// carries 8 significant bits, so the second pieces are 2^-9 of the values) against double precision, next to plain fp32 FMA.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_bf16_probe.hip -o tools/bin/mfma_bf16_probe && tools/bin/mfma_bf16_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__device__ inline void split3(float x, __bf16& p0, __bf16& p1, __bf16& p2) {
  p0 = (__bf16)x;
  float r = x - (float)p0;
  p1 = (__bf16)r;
  r -= (float)p1;
  p2 = (__bf16)r;
}

// A [32][K], B [K][32] row-major, K a multiple of 16; D [32][32]
__global__ void k_probe(const float* A, const float* B, int K, float* D) {
  const int l = threadIdx.x, i = l & 31, h = l >> 5;
  f16v acc = {};
  for (int k0 = 0; k0 < K; k0 += 16) {
    bf8 a0, a1, a2, b0, b1, b2;
    for (int e = 0; e < 8; ++e) {
      __bf16 p0, p1, p2;
      split3(A[i * K + k0 + 8 * h + e], p0, p1, p2);
      a0[e] = p0; a1[e] = p1; a2[e] = p2;
      split3(B[(k0 + 8 * h + e) * 32 + i], p0, p1, p2);
      b0[e] = p0; b1[e] = p1; b2[e] = p2;
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = acc[r];
}

int main() {
  const int K = 160;
  std::vector<float> A(32 * K), B(K * 32), D(32 * 32);
  srand(1);
  for (auto& v : A) v = (rand() / (float)RAND_MAX - 0.5f) * 2.f;
  for (auto& v : B) v = (rand() / (float)RAND_MAX - 0.5f) * 4.f;
  float *dA, *dB, *dD;
  (void)hipMalloc(&dA, A.size() * 4); (void)hipMalloc(&dB, B.size() * 4); (void)hipMalloc(&dD, D.size() * 4);
  (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, dA, dB, K, dD);
  (void)hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
  double e_split = 0, e_f32 = 0, mx = 0;
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      double ref = 0;
      float f = 0.f;
      for (int k = 0; k < K; ++k) {
        ref += (double)A[i * K + k] * (double)B[k * 32 + j];
        f = fmaf(A[i * K + k], B[k * 32 + j], f);
      }
      e_split = fmax(e_split, fabs(D[i * 32 + j] - ref));
      e_f32 = fmax(e_f32, fabs((double)f - ref));
      mx = fmax(mx, fabs(ref));
    }
  printf("K = %d, max |sum| %.3f: max abs error of the 3 x bf16 / 6-MFMA product sum %.3e, of an fp32 FMA chain %.3e\n", K, mx,
         e_split, e_f32);
  return e_split < 50 * e_f32 + 1e-5 ? 0 : 1;
}
