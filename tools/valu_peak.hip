// valu_peak.hip — measured f32 VALU issue rate of one MI355X and the clock it holds while doing it.
//
// What the path-trace kernel is priced against (bench.py "roofline": bound = valu).  Non-packed f32 instructions
// (v_add_f32 / v_mul_f32 / v_fma_f32) from 16 independent accumulators per lane, 256-thread workgroups (one wave per
// SIMD each), 1 / 2 / 4 / 8 workgroups per CU: lane-instructions per second chip-wide, and the in-kernel clock
// (delta s_memtime / delta s_memrealtime x 100 MHz, MI355X_MICROARCH.md "DVFS give-back" item 6).
//
// build: hipcc --offload-arch=gfx950 -O3 -o valu_peak tools/valu_peak.hip      run: ./valu_peak
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                     \
      return 1;                                                                   \
    }                                                                             \
  } while (0)

#define OPS16(OP)                                                                                                  \
  asm volatile(OP " %0, %0, %16\n" OP " %1, %1, %16\n" OP " %2, %2, %16\n" OP " %3, %3, %16\n" OP " %4, %4, %16\n"   \
               OP " %5, %5, %16\n" OP " %6, %6, %16\n" OP " %7, %7, %16\n" OP " %8, %8, %16\n" OP " %9, %9, %16\n"   \
               OP " %10, %10, %16\n" OP " %11, %11, %16\n" OP " %12, %12, %16\n" OP " %13, %13, %16\n"               \
               OP " %14, %14, %16\n" OP " %15, %15, %16\n"                                                          \
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),  \
                 "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]),        \
                 "+v"(a[15])                                                                                      \
               : "v"(x))

template <int KIND>
__global__ __launch_bounds__(256) void k_valu(float* out, unsigned long long* stamps, int iters, float x) {
  float a[16];
  for (int i = 0; i < 16; i++) a[i] = (float)(threadIdx.x + i) * 1e-3f;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it++) {
    if (KIND == 0) {
      OPS16("v_add_f32");
    } else if (KIND == 1) {
      OPS16("v_mul_f32");
    } else {
      asm volatile("v_fma_f32 %0, %0, %16, %16\nv_fma_f32 %1, %1, %16, %16\nv_fma_f32 %2, %2, %16, %16\nv_fma_f32 %3, %3, %16, %16\n"
                   "v_fma_f32 %4, %4, %16, %16\nv_fma_f32 %5, %5, %16, %16\nv_fma_f32 %6, %6, %16, %16\nv_fma_f32 %7, %7, %16, %16\n"
                   "v_fma_f32 %8, %8, %16, %16\nv_fma_f32 %9, %9, %16, %16\nv_fma_f32 %10, %10, %16, %16\nv_fma_f32 %11, %11, %16, %16\n"
                   "v_fma_f32 %12, %12, %16, %16\nv_fma_f32 %13, %13, %16, %16\nv_fma_f32 %14, %14, %16, %16\nv_fma_f32 %15, %15, %16, %16\n"
                   : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                     "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])
                   : "v"(x));
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.0f;
  for (int i = 0; i < 16; i++) s += a[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = c1 - c0;
    stamps[2 * blockIdx.x + 1] = r1 - r0;
  }
}

int main() {
  int dev = 0;
  CHECK(hipSetDevice(dev));
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, dev));
  const int cus = prop.multiProcessorCount;
  const int max_blocks = cus * 8;
  float* out;
  unsigned long long* stamps;
  CHECK(hipMalloc(&out, (size_t)max_blocks * 256 * 4));
  CHECK(hipMalloc(&stamps, (size_t)max_blocks * 16));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int iters = 200000;  // 16 x 200 000 = 3.2 M instructions per wave: tens of ms per launch
  const char* names[3] = {"v_add_f32", "v_mul_f32", "v_fma_f32"};
  printf("device: %s, %d CUs\n", prop.gcnArchName, cus);
  // two seconds of back-to-back launches first, so the clock is the one held under load
  for (int w = 0; w < 40; w++) hipLaunchKernelGGL(k_valu<2>, dim3(cus * 4), dim3(256), 0, 0, out, stamps, iters, 1.0001f);
  CHECK(hipDeviceSynchronize());
  for (int kind = 0; kind < 3; kind++) {
    for (int per_cu : {1, 2, 4, 8}) {
      const int blocks = cus * per_cu;
      float best_ms = 1e30f;
      for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(e0, 0));
        if (kind == 0) hipLaunchKernelGGL(k_valu<0>, dim3(blocks), dim3(256), 0, 0, out, stamps, iters, 1.0001f);
        if (kind == 1) hipLaunchKernelGGL(k_valu<1>, dim3(blocks), dim3(256), 0, 0, out, stamps, iters, 1.0001f);
        if (kind == 2) hipLaunchKernelGGL(k_valu<2>, dim3(blocks), dim3(256), 0, 0, out, stamps, iters, 1.0001f);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        best_ms = std::min(best_ms, ms);
      }
      std::vector<unsigned long long> h((size_t)blocks * 2);
      CHECK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
      std::vector<double> ghz, cyc;
      for (int b = 0; b < blocks; b++) {
        if (h[2 * b + 1]) ghz.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 0.1);
        cyc.push_back((double)h[2 * b]);
      }
      std::sort(ghz.begin(), ghz.end());
      std::sort(cyc.begin(), cyc.end());
      const double lane_instr = (double)blocks * 256.0 * 16.0 * iters;
      const double wave_instr_per_simd = (double)per_cu * 16.0 * iters;  // one wave of every workgroup per SIMD
      printf("%s  %d waves/SIMD: %.2f ms, %.2f T lane-instr/s, in-kernel clock %.3f GHz (median), "
             "%.2f cycles per wave-instruction per SIMD\n",
             names[kind], per_cu, best_ms, lane_instr / (best_ms * 1e-3) / 1e12, ghz.empty() ? 0.0 : ghz[ghz.size() / 2],
             cyc[cyc.size() / 2] / wave_instr_per_simd);
    }
  }
  return 0;
}
