// How fast are LDS atomics on gfx950?  One 1024-thread workgroup per CU hammers a 32 KB LDS array with
//   mode 0: ds_add_f32 (no return)     1: ds_add_u32      2: ds_add_u64      3: plain read-add-write (racy, rate only)
//   4: ds_add_rtn_f32                  5: ds_max_i32       7: ds_add_f64      8 / 9: ds_add_f32 / rtn with MODE.fp_denorm(single) = flush
// pattern 0: lane l -> float l of a 64-float row (row chosen per wave and step)
// pattern 1: 16-lane group g -> 16 consecutive floats at a random 16-float-aligned offset (the tile scatter's shape)
// pattern 2: all four groups on the SAME 16 floats (same-address conflicts)
// build: hipcc --offload-arch=gfx950 -O3 tools/lds_atomic_rate.hip -o tools/bin/lds_atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int MODE, int PAT>
__global__ __launch_bounds__(1024) void k(float* out, int iters) {
  __shared__ __align__(16) double sm64[4096];
  float* sm = reinterpret_cast<float*>(sm64);
  for (int i = threadIdx.x; i < 8192; i += 1024) sm[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, cl = lane & 15;
  unsigned s = 1234567u * (wv + 1) + blockIdx.x;
  // modes 8 / 9: ds_add_f32 with the wave's fp32 denormal mode set to flush (MODE register bits 5:4 = 0)
  if (MODE == 8 || MODE == 9) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 4, 2), 0");
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      s = s * 1664525u + 1013904223u;
      int idx;
      if (PAT == 0) idx = ((s >> 8) & 63) * 64 + lane;
      else if (PAT == 1) idx = (((s >> (8 + 6 * g)) & 0xff) * 16 + cl) & 4095;
      else idx = ((s >> 8) & 0xff) * 16 + cl;
      const float v = 1.0f + lane;
      if (MODE == 0 || MODE == 8) atomicAdd(&sm[idx], v);
      else if (MODE == 9) acc += atomicAdd(&sm[idx], v);
      else if (MODE == 1) atomicAdd(reinterpret_cast<unsigned*>(&sm[idx]), (unsigned)lane);
      else if (MODE == 2) atomicAdd(reinterpret_cast<unsigned long long*>(&sm64[idx & 4095]), (unsigned long long)lane);
      else if (MODE == 3) sm[idx] += v;
      else if (MODE == 4) acc += atomicAdd(&sm[idx], v);
      else if (MODE == 5) atomicMax(reinterpret_cast<int*>(&sm[idx]), lane);
      else if (MODE == 7) atomicAdd(&sm64[idx & 4095], (double)v);
    }
  }
  __syncthreads();
  if (threadIdx.x < 64) out[blockIdx.x * 64 + threadIdx.x] = sm[threadIdx.x] + acc;
}

template <int MODE, int PAT>
void run(const char* name) {
  float* out;
  hipMalloc(&out, 256 * 64 * 4);
  const int iters = 2000;
  hipEvent_t a, b;
  hipEventCreate(&a), hipEventCreate(&b);
  hipLaunchKernelGGL((k<MODE, PAT>), dim3(256), dim3(1024), 0, 0, out, 10);
  hipEventRecord(a);
  hipLaunchKernelGGL((k<MODE, PAT>), dim3(256), dim3(1024), 0, 0, out, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double winstr = 256.0 * 16 * iters * 8;  // wave instructions chip-wide
  printf("%-28s pattern %d: %8.3f ms  %7.2f G wave-instr/s chip  = %6.1f cycles per wave-instr per CU (2.4 GHz)\n", name, PAT, ms,
         winstr / ms * 1e-6, ms * 1e-3 * 2.4e9 / (16.0 * iters * 8));
  hipFree(out);
}

int main() {
  run<0, 0>("ds_add_f32");
  run<0, 1>("ds_add_f32");
  run<0, 2>("ds_add_f32");
  run<1, 0>("ds_add_u32");
  run<1, 1>("ds_add_u32");
  run<1, 2>("ds_add_u32");
  run<2, 0>("ds_add_u64");
  run<2, 1>("ds_add_u64");
  run<3, 0>("read-add-write");
  run<3, 1>("read-add-write");
  run<4, 0>("ds_add_rtn_f32");
  run<4, 1>("ds_add_rtn_f32");
  run<5, 0>("ds_max_i32");
  run<7, 0>("ds_add_f64");
  run<7, 1>("ds_add_f64");
  run<8, 0>("ds_add_f32 denorm-flush");
  run<8, 1>("ds_add_f32 denorm-flush");
  run<8, 2>("ds_add_f32 denorm-flush");
  run<9, 1>("ds_add_rtn_f32 denorm-flush");
  return 0;
}
