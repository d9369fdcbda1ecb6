// Issue rate of the vector instructions the split-bf16 forward leans on, per SIMD, in cycles per wave instruction
// (s_memtime / clock64 around an unrolled dependent-free run), and the shader clock under that load (wall clock vs cycles).
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o tools/bin/valu_rate && tools/bin/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define REP 64
#define ITER 256

template <int OP>
__global__ __launch_bounds__(256) void k_rate(float* out, long long* cyc, int waves_note) {
  float a[8];
  for (int i = 0; i < 8; ++i) a[i] = (float)(threadIdx.x + i) * 1.0001f;
  unsigned u[8];
  for (int i = 0; i < 8; ++i) u[i] = threadIdx.x * 2654435761u + i;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  f32x2 p[4];
  for (int i = 0; i < 4; ++i) p[i].x = a[i], p[i].y = a[i + 4];
  const unsigned sel = 0x07060302u;
  bf8 pa, pb;
  for (int i = 0; i < 8; ++i) pa[i] = (__bf16)(float)i, pb[i] = (__bf16)(float)(i + 1);
  const long long t0 = clock64();
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int r = 0; r < REP / 8; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {   // eight independent destinations: nothing waits for a previous result
        if (OP == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
        if (OP == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]));
        if (OP == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i & 3]) : "v"(p[(i + 1) & 3]), "v"(p[(i + 2) & 3]));
        if (OP == 3) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]), "v"(sel));
        if (OP == 4) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(u[i]) : "v"(u[(i + 1) & 7]));
        if (OP == 5) asm volatile("v_and_b32 %0, %1, %2" : "=v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(sel));
        if (OP == 7) asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(p[i & 3]) : "v"(p[(i + 1) & 3]), "v"(p[(i + 2) & 3]));
        if (OP == 8) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p[i & 3]) : "v"(p[(i + 1) & 3]), "v"(p[(i + 2) & 3]));
      }
    }
    if (OP == 6) {
#pragma unroll
      for (int r = 0; r < REP / 8; ++r) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, pb, acc, 0, 0, 0);
    }
  }
  const long long t1 = clock64();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += a[i] + (float)u[i] + p[i & 3].x + p[i & 3].y;
  for (int r = 0; r < 16; ++r) s += acc[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int OP>
static void run(const char* name, int per_iter, int threads) {
  float* out;
  long long* cyc;
  hipMalloc(&out, 256 * 1024 * sizeof(float));
  hipMalloc(&cyc, 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  k_rate<OP><<<256 * 4, threads>>>(out, cyc, 0);
  hipEventRecord(e0);
  k_rate<OP><<<256 * 4, threads>>>(out, cyc, 0);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  long long c;
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  // blocks of `threads` lanes: threads/64 waves per block, 4 blocks per CU resident at once (1024 blocks over 256 CUs)
  printf("%-28s %3d thr/block: %8.2f clock64 ticks per instruction per wave (x %d instr), kernel %.3f ms\n", name, threads,
         (double)c / (ITER * (double)per_iter), per_iter, ms);
  hipFree(out), hipFree(cyc);
}

int main() {
  for (int threads : {64, 256}) {
    run<0>("v_fma_f32", REP, threads);
    run<1>("v_cvt_pk_bf16_f32", REP, threads);
    run<2>("v_pk_fma_f32", REP, threads);
    run<8>("v_pk_mul_f32", REP, threads);
    run<7>("v_pk_add_f32 (neg)", REP, threads);
    run<3>("v_perm_b32", REP, threads);
    run<4>("v_lshlrev_b32", REP, threads);
    run<5>("v_and_b32", REP, threads);
    run<6>("v_mfma_f32_32x32x16_bf16", REP / 8, threads);
  }
  return 0;
}
