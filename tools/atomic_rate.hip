// Microbenchmark: chip-wide throughput of float atomics shaped like the gradient scatter of jt_walk.h
// (channel-last factor gradients: a texel is C consecutive floats; a flush adds 16 / 32 / 48 / 64 consecutive floats).
// Answers: is the rate bound per REQUEST (wave instruction / 64-byte segment) or per BYTE, and how much do a texel-local
// access sequence (a ray's walk) and a cache-resident target help?
//   hipcc --offload-arch=gfx950 -O3 tools/atomic_rate.hip -o /tmp/atomic_rate && /tmp/atomic_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                 \
  do {                                                                        \
    hipError_t e_ = (x);                                                      \
    if (e_ != hipSuccess) {                                                   \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                \
    }                                                                         \
  } while (0)

__device__ inline unsigned hash32(unsigned x) {
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}

// MODE 0: four 16-lane groups, each adds 16 consecutive floats at its own random texel          (4 x 64 B / instr)
// MODE 1: the whole wave adds 64 consecutive floats at one random texel                        (1 x 256 B / instr)
// MODE 2: 48 lanes add 48 consecutive floats at one random texel, 16 lanes idle                (1 x 192 B / instr)
// MODE 3: only group 0 is active (16 lanes, 64 B), as in a divergent flush                      (1 x 64 B / instr)
// MODE 4: like 0, but the texel advances by one along a row every second iteration (a walk)    (4 x 64 B / instr)
// MODE 5: like 0 with a plain store instead of the atomic (reference for the memory path)
// MODE 6: like 0 but a returning atomic (the wave waits for the old value)
// MODE 7: like 0 with only 4 of the 16 lanes of every group active                              (4 x 16 B / instr)
// MODE 8: like 0, every group adds to the SAME 16 floats iteration after iteration (one hot line per group)
// MODE 9 / 10 / 11: like 0 with workgroup / system / wavefront memory scope instead of the default agent scope
template <int MODE>
__global__ __launch_bounds__(256) void k_atomic(float* __restrict__ buf, unsigned n_texels, int C, int iters,
                                                unsigned seed) {
  const int lane = threadIdx.x & 63;
  const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int grp = lane >> 4, cl = lane & 15;
  float acc = 0.f;
  unsigned walk = hash32(wave * 4u + grp + seed) % n_texels;
  for (int it = 0; it < iters; ++it) {
    unsigned key = (MODE == 1 || MODE == 2) ? wave : wave * 4u + grp;
    unsigned t = hash32(key * 0x9e3779b9u + it * 0x85ebca6bu + seed) % n_texels;
    if (MODE == 4) {
      if (it & 1) walk = (walk + 1) % n_texels;
      t = walk;
    }
    float* p = buf + (size_t)t * C;
    const float v = 1.0f + lane;
    if (MODE == 0 || MODE == 4) atomicAdd(p + cl, v);
    if (MODE == 1) atomicAdd(p + (lane % C), v);
    if (MODE == 2) {
      if (lane < 48) atomicAdd(p + lane, v);
    }
    if (MODE == 3) {
      if (grp == 0) atomicAdd(p + cl, v);
    }
    if (MODE == 5) p[cl] = v;
    if (MODE == 7) {
      if (cl < 4) atomicAdd(p + cl, v);
    }
    if (MODE == 8) atomicAdd(buf + (size_t)(walk % n_texels) * C + cl, v);
    if (MODE == 6) acc += atomicAdd(p + cl, v);
    if (MODE == 9) __hip_atomic_fetch_add(p + cl, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (MODE == 10) __hip_atomic_fetch_add(p + cl, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (MODE == 11) __hip_atomic_fetch_add(p + cl, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  }
  if (MODE == 6 && acc == 123.456f) buf[0] = acc;
}

template <int MODE>
static void run(const char* name, float* buf, unsigned n_texels, int C, int segs_per_instr, int bytes_per_instr,
                int blocks = 256 * 8) {
  const int iters = 2000;
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  hipLaunchKernelGGL(k_atomic<MODE>, dim3(blocks), dim3(256), 0, 0, buf, n_texels, C, iters, 1u);
  CK(hipEventRecord(a, 0));
  const int reps = 3;
  for (int r = 0; r < reps; ++r)
    hipLaunchKernelGGL(k_atomic<MODE>, dim3(blocks), dim3(256), 0, 0, buf, n_texels, C, iters, 7u + r);
  CK(hipEventRecord(b, 0));
  CK(hipEventSynchronize(b));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, a, b));
  ms /= reps;
  const double instr = (double)blocks * 4 * iters;
  printf("%-44s blocks %5d texels %8u  %7.3f ms  %7.2f G instr/s  %7.2f G 64B-seg/s  %7.1f GB/s\n", name, blocks, n_texels, ms,
         instr / ms * 1e-6, instr * segs_per_instr / ms * 1e-6, instr * bytes_per_instr / ms * 1e-6);
}

int main() {
  const int C = 64;
  const size_t max_texels = 3u * 400u * 400u;  // ~ the three appearance planes of the 400^3 grid
  float* buf;
  CK(hipMalloc(&buf, max_texels * C * sizeof(float)));
  CK(hipMemset(buf, 0, max_texels * C * sizeof(float)));
  for (unsigned n : {(unsigned)max_texels, 16384u, 1024u}) {
    run<0>("4 groups x 16 lanes, random texels", buf, n, C, 4, 256);
    run<1>("64 lanes, one random texel (256 B)", buf, n, C, 4, 256);
    run<2>("48 lanes, one random texel (192 B)", buf, n, C, 3, 192);
    run<3>("1 group of 16 lanes active (64 B)", buf, n, C, 1, 64);
    run<4>("4 groups, walking texel by texel", buf, n, C, 4, 256);
    run<5>("4 groups, plain stores", buf, n, C, 4, 256);
    run<6>("4 groups, returning atomics", buf, n, C, 4, 256);
    run<7>("4 groups, 4 lanes each", buf, n, C, 4, 64);
    run<8>("4 groups, one hot line per group", buf, n, C, 4, 256);
    run<9>("4 groups, workgroup scope", buf, n, C, 4, 256);
    run<10>("4 groups, system scope", buf, n, C, 4, 256);
    run<11>("4 groups, wavefront scope", buf, n, C, 4, 256);
  }
  // where is the limit: fewer workgroups (1 / 2 / 4 / 8 per XCD ... all CUs once) at the full texel range
  for (int blocks : {8, 16, 32, 64, 128, 256, 512, 1024})
    run<0>("4 groups x 16 lanes, random texels", buf, (unsigned)max_texels, C, 4, 256, blocks);
  CK(hipFree(buf));
  return 0;
}
