// ieee_ref.cpp — the HOST CPU's IEEE-754 results over the input sets of the IEEE-sequence check
// (webgpu-raytracer_amd/csrc/k_ieee_inputs.h), as checksums: tests/test_gpu_ieee.py compares them with the checksums
// rt_debug_ieee_check computes on the GPU from the compiler's IEEE expansions.  Test infrastructure; plain C++ with
// std::thread.  Compiled with -O2 -ffp-contract=off -fno-fast-math: every `/` and sqrtf here is the x86 IEEE instruction.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "../../webgpu-raytracer_amd/csrc/k_ieee_inputs.h"

static inline float u2f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline uint32_t f2u(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

static uint64_t one(int op, uint64_t i) {
  switch (op) {
    case RT_IEEE_OP_RCP: return rt_ieee_mix(f2u(1.0f / u2f((uint32_t)i)), i);
    case RT_IEEE_OP_SQRT: return rt_ieee_mix(f2u(sqrtf(u2f((uint32_t)i))), i);
    case RT_IEEE_OP_RSQRT: return rt_ieee_mix(f2u(1.0f / sqrtf(u2f((uint32_t)i))), i);
    case RT_IEEE_OP_DIV_PI: return rt_ieee_mix(f2u(u2f((uint32_t)i) / 3.14159274101257324219f), i);
    case RT_IEEE_OP_UNORM8: return rt_ieee_mix(f2u((float)((uint32_t)i & 255u) / 255.0f), i);
    case RT_IEEE_OP_DIV: {
      uint32_t a, b;
      rt_ieee_div_operands(i, &a, &b);
      return rt_ieee_mix(f2u(u2f(a) / u2f(b)), i);
    }
    case RT_IEEE_OP_DIV3:
    case RT_IEEE_OP_DIV3Z: {
      uint32_t a[3], b;
      rt_ieee_div3_operands(i, op == RT_IEEE_OP_DIV3Z, a, &b);
      uint64_t s = 0;
      for (int c = 0; c < 3; c++) s += rt_ieee_mix(f2u(u2f(a[c]) / u2f(b)), 3ull * i + (uint64_t)c);
      return s;
    }
  }
  return 0;
}

extern "C" uint64_t ieee_ref_checksum(int op, uint64_t first, uint64_t count, int threads) {
  if (threads < 1) threads = 1;
  std::vector<uint64_t> part((size_t)threads, 0);
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; t++)
    pool.emplace_back([&, t]() {
      const uint64_t lo = first + count * (uint64_t)t / (uint64_t)threads, hi = first + count * (uint64_t)(t + 1) / (uint64_t)threads;
      uint64_t s = 0;
      for (uint64_t i = lo; i < hi; i++) s += one(op, i);
      part[(size_t)t] = s;
    });
  for (auto& th : pool) th.join();
  uint64_t s = 0;
  for (uint64_t p : part) s += p;
  return s;
}
// operands of one division sample (for the test's own spot checks)
extern "C" void ieee_ref_div_operands(uint64_t i, int zeros, uint32_t* a3, uint32_t* b) { rt_ieee_div3_operands(i, zeros, a3, b); }
