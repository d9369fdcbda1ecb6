// What a 16-byte-per-lane gather costs on MI355X as a function of how the 64 lanes of a wave spread over cache lines:
// every lane reads 16 bytes of a pseudo-random 64-byte texel of a table (L2 / Infinity-Cache resident sizes), with
//   Q = 1: 64 lanes -> 64 different texels (one quad of each; the other quads by later instructions)  [k_march_fwd today]
//   Q = 2: 32 texels x 2 adjacent quads                                                              [the shade gathers]
//   Q = 4: 16 texels x 4 quads = whole lines
// Same bytes per instruction in all three.  Prints GB/s per mapping and table size.
//   hipcc --offload-arch=gfx950 -O3 tools/gather_rate.hip -o tools/bin/gather_rate && tools/bin/gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int Q>
__global__ __launch_bounds__(256) void k_gather(const float4* __restrict__ table, unsigned texels_mask, int iters,
                                                float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const unsigned wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int sub = lane / Q, quad = lane % Q;          // `sub`-th texel of the instruction, quad within the texel
  unsigned state = wave * 2654435761u + 12345u;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      state = state * 1664525u + 1013904223u;
      // consecutive `sub`s sit a few texels apart (samples along a ray), each wave instruction somewhere else
      const unsigned texel = ((state >> 8) + (unsigned)sub * 3u) & texels_mask;
#pragma unroll
      for (int q0 = 0; q0 < 4; q0 += Q) {             // 4 / Q instructions fetch all four quads of the texels
        const float4 v = table[(size_t)texel * 4 + q0 + quad];
        acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

// the walkers' shape (jt_walk.h): lane = channel, a 16-lane group reads the 64 bytes of ONE texel as 16 dwords, a wave
// instruction covers four texels -- the same bytes per texel visit as the 16-byte forms, a quarter of the bytes per instruction
__global__ __launch_bounds__(256) void k_gather_dword(const float* __restrict__ table, unsigned texels_mask, int iters,
                                                      float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const unsigned wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int grp = lane >> 4, ch = lane & 15;
  unsigned state = wave * 2654435761u + 12345u;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      state = state * 1664525u + 1013904223u;
      const unsigned texel = ((state >> 8) + (unsigned)grp * 3u) & texels_mask;
      acc += table[(size_t)texel * 16 + ch];
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

static void run_dword(const float4* table, unsigned mask, float* out) {
  const int iters = 256, blocks = 256 * 8;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
  k_gather_dword<<<blocks, 256>>>((const float*)table, mask, iters, out);
  (void)hipEventRecord(e0);
  k_gather_dword<<<blocks, 256>>>((const float*)table, mask, iters, out);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)blocks * 256 * iters * 8 * 4;
  const double instr = (double)blocks * 4 * iters * 8;
  printf("  %-34s %8.3f ms  %8.1f GB/s   %.1f G wave loads / s\n", "4 texels x 16 dwords (walker shape)", ms, bytes / ms * 1e-6,
         instr / ms * 1e-6);
}

template <int Q>
static void run(const float4* table, unsigned mask, float* out, const char* what) {
  const int iters = 256, blocks = 256 * 8;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
  k_gather<Q><<<blocks, 256>>>(table, mask, iters, out);
  (void)hipEventRecord(e0);
  k_gather<Q><<<blocks, 256>>>(table, mask, iters, out);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  // per lane: iters * 8 texel visits; Q lanes share a texel and fetch 64 B of it over 4 / Q instructions
  const double bytes = (double)blocks * 256 * iters * 8 * (4 / Q) * 16;
  printf("  %-34s %8.3f ms  %8.1f GB/s   %.1f G wave loads / s\n", what, ms, bytes / ms * 1e-6,
         (double)blocks * 4 * iters * 8 * (4 / Q) / ms * 1e-6);
}

int main() {
  for (int log2_texels : {14, 17, 20, 22}) {   // 1 MB, 8 MB, 64 MB, 256 MB of 64-byte texels
    const size_t texels = (size_t)1 << log2_texels;
    float4* table;
    float* out;
    (void)hipMalloc(&table, texels * 64);
    (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    (void)hipMemset(table, 0, texels * 64);
    printf("table %zu MB\n", texels * 64 >> 20);
    run<1>(table, (unsigned)texels - 1, out, "64 texels x 1 quad per instruction");
    run<2>(table, (unsigned)texels - 1, out, "32 texels x 2 quads");
    run<4>(table, (unsigned)texels - 1, out, "16 texels x 4 quads (whole lines)");
    run_dword(table, (unsigned)texels - 1, out);
    (void)hipFree(table), (void)hipFree(out);
  }
  return 0;
}
