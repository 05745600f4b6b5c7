// issue_peak.hip — how many instructions per cycle one SIMD of an MI355X issues from W resident waves, by instruction kind:
// independent v_add_f32 (VALU), independent s_add_u32 (SALU), the two interleaved 1 : 1 and 3 : 1 — the mixes the traversal kernels
// run (PMC: 0.4-0.8 scalar instructions per vector instruction; lane masks, ballots, branches).  The walks of this renderer
// are neither bandwidth- nor FLOP-bound: they are bound by how fast a SIMD ISSUES their serial, mask-heavy instruction streams
// (DESIGN.md 4.1c); this tool gives that bound a measured denominator.
//
// Each kernel runs `iters` x 32 instructions per wave, 256-thread workgroups (one wave per SIMD each), W workgroups per CU.
// Printed: wave-instructions per cycle per SIMD, TWICE: from the per-workgroup s_memtime spans (round 3's only figure) and
// from the WALL CLOCK of the launch (hipEvent; launches of >= 10 ms) at the clock measured in the kernel.  Round 3's
// launches lasted 0.5-1.5 ms and the workgroups of a launch do not run all at once when W x CUs exceeds what is resident
// together with ramp-up, so the span-derived figure over-states the rate; the wall-clock figure is the one bench.py uses
// and the one tools/valu_peak.hip agrees with.
//
// build: hipcc --offload-arch=gfx950 -O3 -o issue_peak tools/issue_peak.hip      run: ./issue_peak
#include <hip/hip_runtime.h>

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

// operands: %0-%7 = eight VGPR accumulators, %8-%15 = eight SGPR counters (all read-write), %16 = the VGPR increment
#define V8 "v_add_f32 %0, %0, %16\n v_add_f32 %1, %1, %16\n v_add_f32 %2, %2, %16\n v_add_f32 %3, %3, %16\n v_add_f32 %4, %4, %16\n v_add_f32 %5, %5, %16\n v_add_f32 %6, %6, %16\n v_add_f32 %7, %7, %16\n"
#define S8 "s_add_u32 %8, %8, 1\n s_add_u32 %9, %9, 1\n s_add_u32 %10, %10, 1\n s_add_u32 %11, %11, 1\n s_add_u32 %12, %12, 1\n s_add_u32 %13, %13, 1\n s_add_u32 %14, %14, 1\n s_add_u32 %15, %15, 1\n"
#define VS8 "v_add_f32 %0, %0, %16\n s_add_u32 %8, %8, 1\n v_add_f32 %1, %1, %16\n s_add_u32 %9, %9, 1\n v_add_f32 %2, %2, %16\n s_add_u32 %10, %10, 1\n v_add_f32 %3, %3, %16\n s_add_u32 %11, %11, 1\n" \
            "v_add_f32 %4, %4, %16\n s_add_u32 %12, %12, 1\n v_add_f32 %5, %5, %16\n s_add_u32 %13, %13, 1\n v_add_f32 %6, %6, %16\n s_add_u32 %14, %14, 1\n v_add_f32 %7, %7, %16\n s_add_u32 %15, %15, 1\n"

#define VVVS8 "v_add_f32 %0, %0, %16\n v_add_f32 %1, %1, %16\n v_add_f32 %2, %2, %16\n s_add_u32 %8, %8, 1\n v_add_f32 %3, %3, %16\n v_add_f32 %4, %4, %16\n v_add_f32 %5, %5, %16\n s_add_u32 %9, %9, 1\n"

template <int KIND>
__global__ __launch_bounds__(256) void k_issue(int iters, float* out, unsigned long long* stamps) {
  float a[8];
  unsigned s[8];
  for (int i = 0; i < 8; i++) {
    a[i] = (float)(threadIdx.x + i);
    s[i] = (unsigned)__builtin_amdgcn_readfirstlane((int)(blockIdx.x + i));   // uniform: lives in an SGPR
  }
  const float inc = 1.0f;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it++) {
#define OPERANDS                                                                                                     \
  : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),                  \
    "+s"(s[0]), "+s"(s[1]), "+s"(s[2]), "+s"(s[3]), "+s"(s[4]), "+s"(s[5]), "+s"(s[6]), "+s"(s[7])                   \
  : "v"(inc)                                                                                                         \
  : "scc"
    if (KIND == 0) asm volatile(V8 V8 V8 V8 OPERANDS);                       // 32 VALU
    if (KIND == 1) asm volatile(S8 S8 S8 S8 OPERANDS);                       // 32 SALU
    if (KIND == 2) asm volatile(VS8 VS8 OPERANDS);                           // 16 VALU + 16 SALU, alternating
    if (KIND == 3) asm volatile(V8 S8 V8 V8 OPERANDS);                       // 24 VALU + 8 SALU (3 : 1, the path-trace kernel's mix), in runs of 8
    if (KIND == 4) asm volatile(VVVS8 VVVS8 VVVS8 VVVS8 OPERANDS);           // 24 VALU + 8 SALU, a scalar instruction after every third vector one
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float acc = 0.0f;
  for (int i = 0; i < 8; i++) acc += a[i] + (float)s[i];
  out[blockIdx.x * 256 + threadIdx.x] = acc;
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = c1 - c0;
    stamps[2 * blockIdx.x + 1] = r1 - r0;
  }
}

int main() {
  CHECK(hipSetDevice(0));
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("device: %s, %d CUs\n", prop.gcnArchName, cus);
  float* out;
  unsigned long long* stamps;
  CHECK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
  CHECK(hipMalloc(&stamps, (size_t)cus * 8 * 16));
  std::vector<unsigned long long> hs((size_t)cus * 8 * 2);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  printf("%-28s %-10s %20s %20s %12s %10s %16s\n", "stream", "waves/SIMD", "instr/cyc/SIMD (span)", "instr/cyc/SIMD (wall)", "clock GHz", "ms", "T lane-instr/s");
  const char* names[5] = {"v_add_f32", "s_add_u32", "v_add_f32 : s_add_u32", "3 v_add : 1 s_add (runs)", "3 v_add : 1 s_add (mixed)"};
  for (int kind = 0; kind < 5; kind++) {
    for (int waves : {1, 2, 4, 5, 6, 8}) {
      const int iters = 400000, grid = cus * waves;   // 12.8 M instructions per wave: >= 10 ms per launch
      float ms = 0.f;
      for (int rep = 0; rep < 2; rep++) {
        CHECK(hipEventRecord(e0));
        if (kind == 0) hipLaunchKernelGGL(k_issue<0>, dim3(grid), dim3(256), 0, 0, iters, out, stamps);
        if (kind == 1) hipLaunchKernelGGL(k_issue<1>, dim3(grid), dim3(256), 0, 0, iters, out, stamps);
        if (kind == 2) hipLaunchKernelGGL(k_issue<2>, dim3(grid), dim3(256), 0, 0, iters, out, stamps);
        if (kind == 3) hipLaunchKernelGGL(k_issue<3>, dim3(grid), dim3(256), 0, 0, iters, out, stamps);
        if (kind == 4) hipLaunchKernelGGL(k_issue<4>, dim3(grid), dim3(256), 0, 0, iters, out, stamps);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
      }
      CHECK(hipMemcpy(hs.data(), stamps, (size_t)grid * 16, hipMemcpyDeviceToHost));
      double clk = 0, cyc = 0;
      for (int b = 0; b < grid; b++) {
        clk += (double)hs[2 * b] / (double)hs[2 * b + 1] * 0.1;
        cyc += (double)hs[2 * b];
      }
      clk /= grid;
      cyc /= grid;   // cycles one workgroup (= one wave per SIMD) needed for its iters x 32 instructions
      // wall clock: every SIMD of the chip holds `waves` waves, each issuing iters x 32 instructions, in `ms` at `clk` GHz
      const double wall_cycles = (double)ms * 1e-3 * clk * 1e9;
      const double wall_rate = (double)waves * iters * 32.0 / wall_cycles;
      const double vec_share = kind == 0 ? 1.0 : (kind == 1 ? 0.0 : (kind == 2 ? 0.5 : 0.75));
      const double lane_rate = (double)grid * 256.0 * iters * 32.0 * vec_share / ((double)ms * 1e-3) / 1e12;
      printf("%-28s %-10d %20.3f %20.3f %12.2f %10.2f %16.2f\n", names[kind], waves, (double)waves * iters * 32.0 / cyc, wall_rate, clk, ms,
             lane_rate);
      fflush(stdout);
    }
  }
  return 0;
}
