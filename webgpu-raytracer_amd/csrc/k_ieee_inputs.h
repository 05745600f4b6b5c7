// k_ieee_inputs.h — the input sets of the IEEE-sequence check (rt_debug_ieee_check, csrc/k_ieee_check.hip.h), as plain
// C++ for BOTH sides: the GPU kernels enumerate them, and tests/model/ieee_ref.cpp enumerates the same inputs on the host
// to compute the reference checksums with the CPU's own IEEE division / square root.
//
//   one operand (rcp, sqrt, rsqrt, x / pi): input i = the f32 with bit pattern i, i in [0, 2^32)
//   division: index i in [0, 2^32): exponent class e = i >> 24 (2^8 classes = 16 numerator x 16 denominator exponents,
//             the whole range: zero / denormal, 2^-126 .. 2^127, inf / NaN), mantissa sample m = i & 0xffffff
//             (2^24 hashed mantissa pairs per class, the first 4096 with mantissas at / near all-zeros and all-ones)
//   three numerators (div3 / div3z): the same classes; the second and third numerator take neighbouring exponents, and
//             for div3z every 8th sample has a zero numerator (+0 or -0)
// Consecutive indices share an exponent class, so the 64 lanes of a wave do: the wave-uniform guards of k_ieee.hip.h
// see whole waves in range (fast path) or out of range (plain operator), as in the renderer.
#ifndef MI355RT_K_IEEE_INPUTS_H
#define MI355RT_K_IEEE_INPUTS_H
#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define RT_IEEE_HD __host__ __device__ inline
#else
#define RT_IEEE_HD static inline
#endif

enum { RT_IEEE_OP_RCP = 0, RT_IEEE_OP_SQRT = 1, RT_IEEE_OP_RSQRT = 2, RT_IEEE_OP_DIV = 3, RT_IEEE_OP_DIV3 = 4, RT_IEEE_OP_DIV3Z = 5,
       RT_IEEE_OP_DIV_PI = 6, RT_IEEE_OP_UNORM8 = 7, RT_IEEE_OP_COUNT = 8 };

RT_IEEE_HD uint32_t rt_ieee_hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
RT_IEEE_HD uint32_t rt_ieee_exp_tab(uint32_t k) {   // 16 biased exponents over the whole range
  const uint32_t t[16] = {0u, 1u, 2u, 26u, 27u, 28u, 64u, 120u, 126u, 127u, 128u, 140u, 200u, 227u, 254u, 255u};
  return t[k & 15u];
}
RT_IEEE_HD uint32_t rt_ieee_mantissa(uint32_t m, uint32_t salt) {
  const uint32_t h = rt_ieee_hash32(m ^ salt);
  if (m < 4096u) {                                   // the corners: mantissas at / near 0 and 0x7fffff
    const uint32_t d = (h >> 8) & 7u;
    return (h & 1u) ? 0x7fffffu - d : d;
  }
  return h & 0x7fffffu;
}
// operands of division sample i: numerator bits a, denominator bits b
RT_IEEE_HD void rt_ieee_div_operands(uint64_t i, uint32_t* a, uint32_t* b) {
  const uint32_t e = (uint32_t)(i >> 24) & 255u, m = (uint32_t)i & 0xffffffu;
  const uint32_t h = rt_ieee_hash32(m + 0x85ebca6bu);
  *a = ((h & 1u) << 31) | (rt_ieee_exp_tab(e >> 4) << 23) | rt_ieee_mantissa(m, 0x9e3779b9u);
  *b = ((h & 2u) << 30) | (rt_ieee_exp_tab(e & 15u) << 23) | rt_ieee_mantissa(m, 0x7f4a7c15u);
}
// three numerators and one denominator of sample i; `zeros`: every 8th sample has a zero among its numerators
RT_IEEE_HD void rt_ieee_div3_operands(uint64_t i, int zeros, uint32_t a[3], uint32_t* b) {
  rt_ieee_div_operands(i, &a[0], b);
  const uint32_t m = (uint32_t)i & 0xffffffu;
  const uint32_t h = rt_ieee_hash32(m + 0x27d4eb2fu);
  const uint32_t e0 = (a[0] >> 23) & 255u;
  const uint32_t e1 = e0 == 0u || e0 == 255u ? e0 : (e0 + 1u > 254u ? 254u : e0 + 1u);
  const uint32_t e2 = e0 == 0u || e0 == 255u ? e0 : (e0 < 3u ? 1u : e0 - 2u);
  a[1] = ((h & 1u) << 31) | (e1 << 23) | rt_ieee_mantissa(m, 0x165667b1u);
  a[2] = ((h & 2u) << 30) | (e2 << 23) | rt_ieee_mantissa(m, 0xd3a2646cu);
  if (zeros && (m & 7u) == 3u) a[(h >> 4) % 3u] = (h & 4u) << 29;   // +0 or -0
}
RT_IEEE_HD uint32_t rt_ieee_canon(uint32_t bits) {   // NaN payloads and signs differ between machines: one NaN for the checksum
  return (bits & 0x7fffffffu) > 0x7f800000u ? 0x7fc00000u : bits;
}
RT_IEEE_HD uint64_t rt_ieee_mix(uint32_t result_bits, uint64_t i) { return (uint64_t)rt_ieee_canon(result_bits) * (2u * i + 1u); }

#endif
