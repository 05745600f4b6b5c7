// store_pattern.hip — how fast does one MI355X write / read 96-byte per-lane records as six 16-byte accesses:
// AoS (record stride 96 B: each wave-level store touches 64 different 96-B records) against SoA (six planes of float4:
// each wave-level store writes 1 KB contiguous).  The question behind the path-state layout of k_wf_shade.
// build: hipcc --offload-arch=gfx950 -O3 -o store_pattern tools/exp/store_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

template <bool SOA, bool READ>
__global__ __launch_bounds__(256) void k_rw(f4* buf, size_t n, float x, f4* sink) {
  f4 acc = {0, 0, 0, 0};
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256ull) {
#pragma unroll
    for (int k = 0; k < 6; k++) {
      const size_t at = SOA ? (size_t)k * n + i : 6 * i + k;
      if (READ) acc += buf[at];
      else { f4 v = {x + k, x, (float)i, x}; buf[at] = v; }
    }
  }
  if (READ && acc.x == 12345.678f) sink[0] = acc;
}

int main() {
  CHECK(hipSetDevice(0));
  const size_t n = 66355200;   // 1920 x 1080 x 32
  f4 *buf, *sink;
  CHECK(hipMalloc(&buf, n * 96));
  CHECK(hipMalloc(&sink, 64));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int grid = 256 * 16;
  for (int mode = 0; mode < 4; mode++) {
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
      CHECK(hipEventRecord(e0));
      if (mode == 0) hipLaunchKernelGGL((k_rw<false, false>), dim3(grid), dim3(256), 0, 0, buf, n, 1.0f, sink);
      if (mode == 1) hipLaunchKernelGGL((k_rw<true, false>), dim3(grid), dim3(256), 0, 0, buf, n, 1.0f, sink);
      if (mode == 2) hipLaunchKernelGGL((k_rw<false, true>), dim3(grid), dim3(256), 0, 0, buf, n, 1.0f, sink);
      if (mode == 3) hipLaunchKernelGGL((k_rw<true, true>), dim3(grid), dim3(256), 0, 0, buf, n, 1.0f, sink);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    const char* names[4] = {"write AoS 96 B", "write SoA 6 planes", "read AoS 96 B", "read SoA 6 planes"};
    printf("%-20s %8.3f ms  %7.1f GB/s\n", names[mode], ms, (double)n * 96 / (ms * 1e-3) * 1e-9);
  }
  return 0;
}
