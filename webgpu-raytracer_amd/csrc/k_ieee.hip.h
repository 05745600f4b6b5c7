// k_ieee.hip.h — correctly rounded f32 reciprocal / division / square root in fewer instructions than the compiler's
// expansions, for gfx950.  Included by device_scene.h BEFORE include/mi355rt_math.h, which routes rt_rcp / rt_div /
// rt_sqrt / rt_rsqrt / operator/(rt3, float) / rt_div3z / rt_div_pi / rt_from_unorm8 here in a device compilation
// (MI355RT_DEVICE_IEEE).
//
// The arithmetic contract of the kernels (mi355rt_math.h) is single IEEE-754 binary32 operations, and that does not
// change: every function below returns, for EVERY input, the bits of the IEEE operation.  What changes is how.  hipcc
// expands `a / b` into 11 vector instructions (v_div_scale x 2, v_rcp_f32, six fma / mul, v_div_fmas, v_div_fixup) and
// sqrtf into 16 (range scaling, v_sqrt_f32, a +-1 ulp choice by two residuals, un-scaling, class fix-up): 152 + 28
// such expansions were 35 % of the static vector instructions of the path-trace kernel.  Most of that is range
// handling (denormals, overflow, zeros, infinities) that the renderer's operands almost never need.  So each function
// runs a short Newton sequence on the hardware estimate (`*_seq`) and checks a GUARD (`*_guard`) — one or two compares
// that are true only where the short sequence was shown to be exact; when any active lane of the wave fails its guard
// the whole wave takes the plain operator instead (wave-uniform branch; the cold path is the compiler's expansion).
//
//   function            sequence                                         guard                               shown exact on
//   rcp(x)              y0 = rcp; e = fma(-x,y0,1); y = fma(e,y0,y0)     y0 is a normal number               all 2^32 x
//   sqrt(x)             y = rsq; g = x y; h = y/2; d = fma(-g,g,x);      2^-100 < x < inf  (or x = +-0:      all 2^32 x
//                       s = fma(d,h,g)                                   the result is x)
//   rsqrt(x)            sqrt, then rcp of it (two roundings: 1/sqrt(x)   2^-100 < x < inf (the root is       all 2^32 x
//                       as the shaders write it)                         then always in rcp's range)
//   div(a,b)            y as in rcp; q = a y; r = fma(-b,q,a);           q normal and |a| > 2^-100           2^24 mantissa pairs x 2^8 exponent
//                       q = fma(r,y,q)                                                                       pairs over the whole range
//   div3(a0,a1,a2,b)    one y, three quotients as in div                 all three as in div                 the same grid
//   div3z(a0,a1,a2,b)   the same + v_div_fixup_f32 per quotient          each a = +-0 or 2^-100 <= |a| <     the same grid with zero numerators
//                       (numerators that may be zero)                    2^100; |y0| in [2^-26, 2^26]
//   div_const(x,C)      q = x RN(1/C); r = fma(-C,q,x);                  x = +-0 or |x| >= 2^-100            all 2^32 x, C = pi
//                       q = fma(r,RN(1/C),q); v_div_fixup(q, C, x)
//   unorm8(n)           n RN(1/255) and one residual step                none: n = 0..255                    all 256 n
//
// "Shown exact": tools/ieee_probe.hip (the experiment behind the choices, profiles/r04_ieee_probe.txt: e.g. a
// reciprocal seeded from the root's own half-estimate fails for two mantissas per exponent, the rsq sequence fails below
// 2^-102) and tests/test_gpu_ieee.py — the proof that ships: through the C ABI (rt_debug_ieee_check, csrc/k_ieee_check.hip.h)
// every sequence is compared lane by lane, behind its guard, with the compiler's IEEE expansion on the GPU over the
// input sets of the table, the composed functions over all inputs, and the GPU's IEEE results with the host CPU's by
// checksum.  Theory: Markstein, "Computation of elementary functions on the IBM RISC System/6000 processor" (1990): one
// fma-residual correction of a faithful estimate rounds correctly when no intermediate under- or overflows — which is
// what the guards say.
//
// RT_IEEE_PLAIN (build flag): every function is the plain operator again (A/B timing, tools/SWEEPS.md).
#ifndef MI355RT_K_IEEE_HIP_H
#define MI355RT_K_IEEE_HIP_H

#if defined(__HIP_DEVICE_COMPILE__) && !defined(RT_IEEE_PLAIN)
#define MI355RT_DEVICE_IEEE 1

namespace rt_ieee {

#define RT_CLS_NORMAL 0x108                  // v_cmp_class_f32 mask: -normal | +normal
#define RT_CLS_NOT_NORMAL (0x3ff & ~RT_CLS_NORMAL)
#define RT_IEEE_LO 7.888609052210118e-31f    // 2^-100: below it a residual fma(-b, q, a) may underflow
#define RT_IEEE_HI 1.2676506002282294e30f    // 2^100
#define RT_IEEE_LO_BITS 0x0d800000u          // bits(2^-100)
__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ uint32_t bits_(float x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ bool normal_(float x) { return __builtin_amdgcn_classf(x, RT_CLS_NORMAL); }
// ((bits << 1) - 1) as an unsigned number is 0xffffffff for +-0 and 2 |bits| - 1 otherwise: two instructions say
// "x is +-0 or |x| >= 2^-100" (infinities and NaN pass: v_div_fixup_f32 or another term of the guard deals with them)
__device__ __forceinline__ uint32_t zero_or_big_key(float x) { return (bits_(x) << 1) - 1u; }
#define RT_IEEE_KEY_LO ((RT_IEEE_LO_BITS << 1) - 1u)
// A guard is evaluated as the MASK of the active lanes that fail it, one ballot per compare (a v_cmp writes the mask
// straight into scalar registers; a ballot of a combined bool goes through v_cndmask / v_cmp_ne first); the short result
// is used when the mask is empty.
typedef unsigned long long lanemask;
__device__ __forceinline__ lanemask lanes(bool c) { return __builtin_amdgcn_ballot_w64(c); }
__device__ __forceinline__ bool not_normal_(float x) { return __builtin_amdgcn_classf(x, RT_CLS_NOT_NORMAL); }
// A guard is ONE ballot of the OR of its per-lane conditions, used only as `!= 0`: the compiler then branches on the OR'ed
// compare masks themselves (v_cmp ... ; s_or ; s_cbranch_vccnz).  A ballot per condition OR'ed as 64-bit numbers costs a
// v_cndmask + v_cmp_ne round trip per v_cmp_class (the ballot of an fp-class test is not folded into the compare).
__device__ __forceinline__ lanemask bad_not_normal(float x) { return lanes(not_normal_(x)); }
// The fallback branch: the hint has to sit on the branch itself (inside a helper it is dropped before inlining), and the
// first statement of the branch is a volatile asm.  Division and square root have no side effects, so without the asm the
// optimiser "if-converts" the branch: it computes the short sequence AND the full expansion for every wave and selects
// (seen in the ISA of the first build: + 5 % on the path-trace kernel instead of a gain).
#define RT_IEEE_IF_ANY(bad) if (__builtin_expect((bad) != 0ull, 0))
#define RT_IEEE_COLD() asm volatile("; plain IEEE operator: an operand outside the guard of the short sequence")

// ------------------------------------------------------------------------------------------------ 1 / x
__device__ __forceinline__ float rcp_seq(float x, float& y0) {
  y0 = __builtin_amdgcn_rcpf(x);
  const float e = fma_(-x, y0, 1.0f);
  return fma_(e, y0, y0);
}
__device__ __forceinline__ bool rcp_guard(float y0) { return normal_(y0); }
__device__ __forceinline__ float rcp(float x) {
  float y0;
  float y = rcp_seq(x, y0);
  RT_IEEE_IF_ANY(bad_not_normal(y0)) {
    RT_IEEE_COLD();
    y = 1.0f / x;
  }
  return y;
}
// three reciprocals behind ONE branch (a ray's 1 / direction)
__device__ __forceinline__ void rcp3(float x0, float x1, float x2, float& r0, float& r1, float& r2) {
  float a0, a1, a2;
  r0 = rcp_seq(x0, a0);
  r1 = rcp_seq(x1, a1);
  r2 = rcp_seq(x2, a2);
  RT_IEEE_IF_ANY(lanes(not_normal_(a0) | not_normal_(a1) | not_normal_(a2))) {
    RT_IEEE_COLD();
    r0 = 1.0f / x0;
    r1 = 1.0f / x1;
    r2 = 1.0f / x2;
  }
}

// ------------------------------------------------------------------------------------- sqrt(x), 1 / sqrt(x)
__device__ __forceinline__ bool sqrt_guard(float x) { return x > RT_IEEE_LO && x < __builtin_inff(); }
__device__ __forceinline__ bool sqrt_bad_(float x) { return !(x > RT_IEEE_LO) | !(x < __builtin_inff()); }
__device__ __forceinline__ float sqrt_seq(float x) {
  const float y = __builtin_amdgcn_rsqf(x);
  const float g = x * y;
  const float h = 0.5f * y;
  const float d = fma_(-g, g, x);
  return fma_(d, h, g);
}
// sqrt(+-0) = +-0 is let through as well (one more compare and a select): a non-emissive surface asks for the length of
// a zero vector on every bounce (Raytracer.wgsl:681), and a wave whose guard fails pays both paths
__device__ __forceinline__ float sqrt(float x) {
  const float s = sqrt_seq(x);
  const bool zero = x == 0.0f;
  float r = zero ? x : s;
  RT_IEEE_IF_ANY(lanes(sqrt_bad_(x) & !zero)) {
    RT_IEEE_COLD();
    r = __builtin_sqrtf(x);
  }
  return r;
}
// 1 / sqrt(x) with both roundings of the source expression; behind the guard the root lies in [2^-50, 2^64], where the
// reciprocal estimate is always a normal number
__device__ __forceinline__ float rsqrt_seq(float x) {
  float y0;
  return rcp_seq(sqrt_seq(x), y0);
}
__device__ __forceinline__ float rsqrt(float x) {
  float y = rsqrt_seq(x);
  RT_IEEE_IF_ANY(lanes(sqrt_bad_(x))) {
    RT_IEEE_COLD();
    y = 1.0f / __builtin_sqrtf(x);
  }
  return y;
}

// ------------------------------------------------------------------------------------------------ a / b
__device__ __forceinline__ float div_step(float a, float b, float y) {   // y = the refined reciprocal of b
  const float q = a * y;
  const float r = fma_(-b, q, a);
  return fma_(r, y, q);
}
__device__ __forceinline__ float div_seq(float a, float b) {
  float y0;
  const float y = rcp_seq(b, y0);
  return div_step(a, b, y);
}
__device__ __forceinline__ bool div_guard(float q, float a) { return normal_(q) && __builtin_fabsf(a) > RT_IEEE_LO; }
__device__ __forceinline__ float div(float a, float b) {
  float q = div_seq(a, b);
  RT_IEEE_IF_ANY(lanes(not_normal_(q) | !(__builtin_fabsf(a) > RT_IEEE_LO))) {
    RT_IEEE_COLD();
    q = a / b;
  }
  return q;
}
// (a0, a1, a2) / b: one reciprocal for the three quotients, one branch
__device__ __forceinline__ void div3(float a0, float a1, float a2, float b, float& q0, float& q1, float& q2) {
  float y0;
  const float y = rcp_seq(b, y0);
  q0 = div_step(a0, b, y);
  q1 = div_step(a1, b, y);
  q2 = div_step(a2, b, y);
  const float amin = __builtin_fminf(__builtin_fminf(__builtin_fabsf(a0), __builtin_fabsf(a1)), __builtin_fabsf(a2));
  RT_IEEE_IF_ANY(lanes(not_normal_(q0) | not_normal_(q1) | not_normal_(q2) | !(amin > RT_IEEE_LO))) {
    RT_IEEE_COLD();
    q0 = a0 / b;
    q1 = a1 / b;
    q2 = a2 / b;
  }
}
// The same for numerators that may be ZERO (a black texel in a throughput, a radiance sum that found no light): every
// quotient goes through v_div_fixup_f32 (signed zeros; infinities / NaN on either side), and the guard bounds the
// operands instead of looking at the quotients — each numerator +-0 or 2^-100 <= |a| < 2^100, the reciprocal estimate in
// [2^-26, 2^26] — so that every non-zero quotient is a normal number and no residual underflows.
__device__ __forceinline__ float divz_step(float a, float b, float y) { return __builtin_amdgcn_div_fixupf(div_step(a, b, y), b, a); }
__device__ __forceinline__ uint32_t min3u_(uint32_t a, uint32_t b, uint32_t c) { return a < b ? (a < c ? a : c) : (b < c ? b : c); }
__device__ __forceinline__ float max3abs_(float a, float b, float c) {
  return __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(a), __builtin_fabsf(b)), __builtin_fabsf(c));
}
__device__ __forceinline__ bool div3z_guard(float a0, float a1, float a2, float y0) {
  const float ya = __builtin_fabsf(y0);
  return min3u_(zero_or_big_key(a0), zero_or_big_key(a1), zero_or_big_key(a2)) >= RT_IEEE_KEY_LO && max3abs_(a0, a1, a2) < RT_IEEE_HI &&
         ya >= 1.4901161193847656e-8f && ya <= 67108864.0f;
}
__device__ __forceinline__ void div3z(float a0, float a1, float a2, float b, float& q0, float& q1, float& q2) {
  float y0;
  const float y = rcp_seq(b, y0);
  q0 = divz_step(a0, b, y);
  q1 = divz_step(a1, b, y);
  q2 = divz_step(a2, b, y);
  const float ya = __builtin_fabsf(y0);
  RT_IEEE_IF_ANY(lanes((min3u_(zero_or_big_key(a0), zero_or_big_key(a1), zero_or_big_key(a2)) < RT_IEEE_KEY_LO) |
                       !(max3abs_(a0, a1, a2) < RT_IEEE_HI) | !(ya >= 1.4901161193847656e-8f) | !(ya <= 67108864.0f))) {
    RT_IEEE_COLD();
    q0 = a0 / b;
    q1 = a1 / b;
    q2 = a2 / b;
  }
}

// ------------------------------------------------------------------------------------ x / C, C a constant
// rc = RN(1 / C).  Zeros, infinities and NaN go through v_div_fixup_f32, so the guard only has to keep tiny non-zero
// numerators out.
__device__ __forceinline__ bool divc_guard(float x) { return zero_or_big_key(x) >= RT_IEEE_KEY_LO; }
__device__ __forceinline__ float divc_seq(float x, float c, float rc) {
  float q = x * rc;
  const float r = fma_(-c, q, x);
  q = fma_(r, rc, q);
  return __builtin_amdgcn_div_fixupf(q, c, x);
}
__device__ __forceinline__ float div_const(float x, float c, float rc) {
  float q = divc_seq(x, c, rc);
  RT_IEEE_IF_ANY(lanes(zero_or_big_key(x) < RT_IEEE_KEY_LO)) {
    RT_IEEE_COLD();
    q = x / c;
  }
  return q;
}
__device__ __forceinline__ void div_const3(float x0, float x1, float x2, float c, float rc, float& q0, float& q1, float& q2) {
  q0 = divc_seq(x0, c, rc);
  q1 = divc_seq(x1, c, rc);
  q2 = divc_seq(x2, c, rc);
  RT_IEEE_IF_ANY(lanes(min3u_(zero_or_big_key(x0), zero_or_big_key(x1), zero_or_big_key(x2)) < RT_IEEE_KEY_LO)) {
    RT_IEEE_COLD();
    q0 = x0 / c;
    q1 = x1 / c;
    q2 = x2 / c;
  }
}
// n / 255 for an integer 0..255 (unorm8 texel / G-buffer albedo): every one of the 256 inputs is exact, no guard
__device__ __forceinline__ float unorm8(uint32_t q8) {
  const float x = (float)q8;
  const float rc = 1.0f / 255.0f;
  const float q = x * rc;
  const float r = fma_(-255.0f, q, x);
  return fma_(r, rc, q);
}

}  // namespace rt_ieee
#endif  // device compilation

#endif
