/* mi355rt_math.h — numeric specification of the shading-language builtins.
 *
 * The reference kernels are WGSL (src/shaders/Raytracer.wgsl, Rasterizer.wgsl,
 * PostProcess.wgsl).  WGSL leaves the precision of sin/cos/exp/pow, the NaN
 * behaviour of min/max, the association order of dot()/matrix products and
 * FMA contraction implementation-defined.  This header pins ONE definition of
 * each builtin the three shaders use so that the CPU oracle (oracle/) and the
 * HIP kernels (webgpu-raytracer_amd/csrc/) evaluate bit-identical f32
 * arithmetic.  It contains no rendering algorithm: only the "standard
 * library" (scalar builtins, vec3, mat4 products, unorm/f16 conversion).
 *
 * Rules:
 *   - every operation is a single IEEE-754 binary32 op (+,-,*,/,sqrt), rounded
 *     to nearest-even; both compilers run with -ffp-contract=off, no fast-math;
 *   - min/max return the non-NaN operand, and order -0 < +0 (gfx950
 *     v_min_f32 / v_max_f32 semantics);
 *   - dot/cross/matrix products associate left to right;
 *   - normalize(v) = v * (1 / sqrt(dot(v,v)));
 *   - transcendental functions are the fixed polynomial kernels below.
 *
 * Division, reciprocal and square root are named functions here (rt_div, rt_rcp, rt_sqrt, rt_rsqrt, ...): they ARE the
 * IEEE operations — on the host the language operators; a gfx950 device compilation computes the same bits with shorter
 * instruction sequences (webgpu-raytracer_amd/csrc/k_ieee.hip.h defines MI355RT_DEVICE_IEEE; every sequence is checked
 * against the plain operator over its whole input space by tests/test_gpu_ieee.py).  The oracle never sees that header.
 */
#ifndef MI355RT_MATH_H
#define MI355RT_MATH_H

#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define RT_HD __host__ __device__ __forceinline__
#else
#define RT_HD static inline
#endif

#define RT_PI 3.14159274101257324219f      /* f32(3.141592653589793)            */
#define RT_TWO_PI 6.28318548202514648438f  /* f32(2.0 * 3.141592653589793)      */

/* ------------------------------------------------------------------ bits */
RT_HD uint32_t rt_f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
RT_HD float rt_u2f(uint32_t u) { return __builtin_bit_cast(float, u); }

/* ------------------------------------------------------------ scalar ops */
RT_HD float rt_min(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_fminf(a, b);
#else
  if (a != a) return b;
  if (b != b) return a;
  if (a == b) return (rt_f2u(a) >> 31) ? a : b; /* -0 < +0 */
  return a < b ? a : b;
#endif
}
RT_HD float rt_max(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_fmaxf(a, b);
#else
  if (a != a) return b;
  if (b != b) return a;
  if (a == b) return (rt_f2u(a) >> 31) ? b : a; /* +0 > -0 */
  return a > b ? a : b;
#endif
}
RT_HD float rt_clamp(float x, float lo, float hi) { return rt_min(rt_max(x, lo), hi); }
RT_HD float rt_saturate(float x) { return rt_clamp(x, 0.0f, 1.0f); }
RT_HD float rt_abs(float x) { return rt_u2f(rt_f2u(x) & 0x7fffffffu); }
/* ---- the IEEE operations with a name (see the header comment) */
#if defined(MI355RT_DEVICE_IEEE)
RT_HD float rt_sqrt(float x) { return rt_ieee::sqrt(x); }
RT_HD float rt_rcp(float x) { return rt_ieee::rcp(x); }                      /* 1 / x */
RT_HD float rt_div(float a, float b) { return rt_ieee::div(a, b); }          /* a / b */
RT_HD float rt_rsqrt(float x) { return rt_ieee::rsqrt(x); }                  /* 1 / sqrt(x): two roundings */
RT_HD float rt_div_pi(float x) { return rt_ieee::div_const(x, 3.14159274101257324219f, 1.0f / 3.14159274101257324219f); }
RT_HD float rt_from_unorm8(uint32_t q) { return rt_ieee::unorm8(q); }
#else
RT_HD float rt_sqrt(float x) { return __builtin_sqrtf(x); }
RT_HD float rt_rcp(float x) { return 1.0f / x; }
RT_HD float rt_div(float a, float b) { return a / b; }
RT_HD float rt_rsqrt(float x) { return 1.0f / __builtin_sqrtf(x); }
RT_HD float rt_div_pi(float x) { return x / 3.14159274101257324219f; }
RT_HD float rt_from_unorm8(uint32_t q) { return (float)q / 255.0f; }
#endif
RT_HD float rt_floor(float x) { return __builtin_floorf(x); }
RT_HD float rt_mix(float a, float b, float t) { return a * (1.0f - t) + b * t; }

/* u32(f): truncate toward zero, saturating, NaN -> 0 (WGSL conversion). */
RT_HD uint32_t rt_f2u32_sat(float f) {
  if (!(f > 0.0f)) return 0u;
  if (f >= 4294967296.0f) return 0xffffffffu;
  return (uint32_t)f;
}
RT_HD int32_t rt_f2i32_sat(float f) {
  if (f != f) return 0;
  if (f >= 2147483648.0f) return 2147483647;
  if (f <= -2147483648.0f) return (int32_t)0x80000000;
  return (int32_t)f;
}

/* 2^n for integer n in [-126, 127]. */
RT_HD float rt_pow2i(int n) { return rt_u2f((uint32_t)(n + 127) << 23); }

/* ---------------------------------------------------------- sin and cos
 * Cody-Waite reduction to |r| <= pi/4 by multiples of pi/2 (three-part
 * constant, exact products for |k| < 2^8), then the Cephes single-precision
 * minimax kernels.  Intended domain |x| <= ~800; the shaders only evaluate
 * angles in [0, 2*pi].
 */
RT_HD void rt_sincos(float x, float* s_out, float* c_out) {
  const float two_over_pi = 0.636619746685028076171875f;
  float kf = rt_floor(x * two_over_pi + 0.5f);
  int k = (int)kf;
  float r = x - kf * 1.5703125f;
  r = r - kf * 4.837512969970703125e-4f;
  r = r - kf * 7.54978995489188216e-8f;
  float z = r * r;
  float sp = -1.9515295891e-4f;
  sp = sp * z + 8.3321608736e-3f;
  sp = sp * z - 1.6666654611e-1f;
  float s = sp * z * r + r;
  float cp = 2.443315711809948e-5f;
  cp = cp * z - 1.388731625493765e-3f;
  cp = cp * z + 4.166664568298827e-2f;
  float c = cp * z * z - 0.5f * z + 1.0f;
  int q = k & 3;
  float ss = (q & 1) ? c : s;
  float cc = (q & 1) ? s : c;
  if (q == 1 || q == 2) cc = -cc;
  if (q >= 2) ss = -ss;
  *s_out = ss;
  *c_out = cc;
}
RT_HD float rt_sin(float x) { float s, c; rt_sincos(x, &s, &c); return s; }
RT_HD float rt_cos(float x) { float s, c; rt_sincos(x, &s, &c); return c; }

/* ------------------------------------------------------------------ exp
 * Cephes expf: n = round(x*log2(e)), two-part ln2, degree-5 polynomial.
 */
RT_HD float rt_exp(float x) {
  if (x != x) return x;
  if (x > 88.72283905206835f) return rt_u2f(0x7f800000u);
  if (x < -87.33654475055310898657f) return 0.0f;
  float nf = rt_floor(x * 1.44269504088896341f + 0.5f);
  int n = (int)nf;
  float r = x - nf * 0.693359375f;
  r = r - nf * -2.12194440e-4f;
  float z = r * r;
  float p = 1.9875691500e-4f;
  p = p * r + 1.3981999507e-3f;
  p = p * r + 8.3334519073e-3f;
  p = p * r + 4.1665795894e-2f;
  p = p * r + 1.6666665459e-1f;
  p = p * r + 5.0000001201e-1f;
  float y = p * z + r + 1.0f;
  /* scale by 2^n in two steps so n = 128 / n = -126 stay in range */
  int n1 = n / 2;
  int n2 = n - n1;
  return y * rt_pow2i(n1) * rt_pow2i(n2);
}

/* ------------------------------------------------------------------ log
 * Cephes logf for x > 0 (normal or subnormal); log(0) = -inf, log(<0) = NaN.
 */
RT_HD float rt_log(float x) {
  if (x != x) return x;
  if (x < 0.0f) return rt_u2f(0x7fc00000u);
  if (x == 0.0f) return rt_u2f(0xff800000u);
  if (rt_f2u(x) == 0x7f800000u) return x;
  int e = 0;
  uint32_t u = rt_f2u(x);
  if (u < 0x00800000u) { /* subnormal: scale by 2^23 */
    x = x * 8388608.0f;
    u = rt_f2u(x);
    e = -23;
  }
  e += (int)(u >> 23) - 126;
  float m = rt_u2f((u & 0x007fffffu) | 0x3f000000u); /* [0.5, 1) */
  if (m < 0.707106781186547524f) {
    e = e - 1;
    m = m + m - 1.0f;
  } else {
    m = m - 1.0f;
  }
  float z = m * m;
  float p = 7.0376836292e-2f;
  p = p * m - 1.1514610310e-1f;
  p = p * m + 1.1676998740e-1f;
  p = p * m - 1.2420140846e-1f;
  p = p * m + 1.4249322787e-1f;
  p = p * m - 1.6668057665e-1f;
  p = p * m + 2.0000714765e-1f;
  p = p * m - 2.4999993993e-1f;
  p = p * m + 3.3333331174e-1f;
  float y = m * z * p;
  float fe = (float)e;
  y = y + fe * -2.12194440e-4f;
  y = y - 0.5f * z;
  float r = m + y;
  r = r + fe * 0.693359375f;
  return r;
}

/* pow(x, y) for the post pass (x in [0,1], y = 1/2.2): exp(y*log(x)). */
RT_HD float rt_pow(float x, float y) {
  if (x == 0.0f) return (y > 0.0f) ? 0.0f : ((y == 0.0f) ? 1.0f : rt_u2f(0x7f800000u));
  if (x == 1.0f) return 1.0f;
  return rt_exp(y * rt_log(x));
}

/* ----------------------------------------------------------------- vec3 */
struct rt3 {
  float x, y, z;
};
struct rt2 {
  float x, y;
};
struct rt4 {
  float x, y, z, w;
};

RT_HD rt3 rt3_make(float x, float y, float z) { rt3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_HD rt3 rt3_splat(float s) { return rt3_make(s, s, s); }
RT_HD rt2 rt2_make(float x, float y) { rt2 r; r.x = x; r.y = y; return r; }
RT_HD rt3 operator+(rt3 a, rt3 b) { return rt3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_HD rt3 operator-(rt3 a, rt3 b) { return rt3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_HD rt3 operator*(rt3 a, rt3 b) { return rt3_make(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_HD rt3 operator/(rt3 a, rt3 b) { return rt3_make(rt_div(a.x, b.x), rt_div(a.y, b.y), rt_div(a.z, b.z)); }
RT_HD rt3 operator*(rt3 a, float s) { return rt3_make(a.x * s, a.y * s, a.z * s); }
RT_HD rt3 operator*(float s, rt3 a) { return rt3_make(s * a.x, s * a.y, s * a.z); }
RT_HD rt3 operator/(rt3 a, float s) {   /* three quotients by one divisor */
#if defined(MI355RT_DEVICE_IEEE)
  rt3 q;
  rt_ieee::div3(a.x, a.y, a.z, s, q.x, q.y, q.z);
  return q;
#else
  return rt3_make(a.x / s, a.y / s, a.z / s);
#endif
}
RT_HD rt3 rt_div3z(rt3 a, float s) {    /* a / s where components of a may be zero (device: the zero-tolerant sequence) */
#if defined(MI355RT_DEVICE_IEEE)
  rt3 q;
  rt_ieee::div3z(a.x, a.y, a.z, s, q.x, q.y, q.z);
  return q;
#else
  return rt3_make(a.x / s, a.y / s, a.z / s);
#endif
}
RT_HD rt3 rt_div3_plain(rt3 a, float s) { return rt3_make(a.x / s, a.y / s, a.z / s); }   /* the language operator on both sides */
RT_HD rt3 rt_rcp3(rt3 d) {              /* vec3(1) / d */
#if defined(MI355RT_DEVICE_IEEE)
  rt3 r;
  rt_ieee::rcp3(d.x, d.y, d.z, r.x, r.y, r.z);
  return r;
#else
  return rt3_make(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
#endif
}
RT_HD rt3 rt_div_pi3(rt3 a) {           /* a / RT_PI */
#if defined(MI355RT_DEVICE_IEEE)
  rt3 q;
  rt_ieee::div_const3(a.x, a.y, a.z, 3.14159274101257324219f, 1.0f / 3.14159274101257324219f, q.x, q.y, q.z);
  return q;
#else
  return rt3_make(a.x / 3.14159274101257324219f, a.y / 3.14159274101257324219f, a.z / 3.14159274101257324219f);
#endif
}
RT_HD rt3 operator-(rt3 a) { return rt3_make(-a.x, -a.y, -a.z); }
RT_HD rt2 operator+(rt2 a, rt2 b) { return rt2_make(a.x + b.x, a.y + b.y); }
RT_HD rt2 operator*(rt2 a, float s) { return rt2_make(a.x * s, a.y * s); }

RT_HD float rt_dot(rt3 a, rt3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RT_HD rt3 rt_cross(rt3 a, rt3 b) {
  return rt3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
RT_HD float rt_length(rt3 a) { return rt_sqrt(rt_dot(a, a)); }
RT_HD rt3 rt_normalize(rt3 a) {
  float inv = rt_rsqrt(rt_dot(a, a));
  return a * inv;
}
RT_HD rt3 rt_min3(rt3 a, rt3 b) { return rt3_make(rt_min(a.x, b.x), rt_min(a.y, b.y), rt_min(a.z, b.z)); }
RT_HD rt3 rt_max3(rt3 a, rt3 b) { return rt3_make(rt_max(a.x, b.x), rt_max(a.y, b.y), rt_max(a.z, b.z)); }
RT_HD rt3 rt_clamp3(rt3 v, rt3 lo, rt3 hi) { return rt_min3(rt_max3(v, lo), hi); }
RT_HD rt3 rt_mix3(rt3 a, rt3 b, float t) { return a * (1.0f - t) + b * t; }
/* reflect(i, n) = i - 2*dot(n,i)*n */
RT_HD rt3 rt_reflect(rt3 i, rt3 n) { return i - n * (2.0f * rt_dot(n, i)); }
/* refract(i, n, eta) per the WGSL spec */
RT_HD rt3 rt_refract(rt3 i, rt3 n, float eta) {
  float ni = rt_dot(n, i);
  float k = 1.0f - eta * eta * (1.0f - ni * ni);
  if (k < 0.0f) return rt3_splat(0.0f);
  return i * eta - n * (eta * ni + rt_sqrt(k));
}

/* ----------------------------------------------------------------- mat4
 * Column-major (WGSL mat4x4): m[0..3]=c0, m[4..7]=c1, m[8..11]=c2, m[12..15]=c3.
 * (M * vec4(p, w)).xyz = ((c0*p.x + c1*p.y) + c2*p.z) + c3*w
 * (vec4(n, 0) * M).xyz = (dot(n, c0.xyz), dot(n, c1.xyz), dot(n, c2.xyz))  [+ 0*c.w]
 */
RT_HD rt3 rt_mat_mul_point(const float* m, rt3 p) {
  return rt3_make(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12] * 1.0f,
                  m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13] * 1.0f,
                  m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14] * 1.0f);
}
RT_HD rt3 rt_mat_mul_dir(const float* m, rt3 d) {
  return rt3_make(m[0] * d.x + m[4] * d.y + m[8] * d.z + m[12] * 0.0f,
                  m[1] * d.x + m[5] * d.y + m[9] * d.z + m[13] * 0.0f,
                  m[2] * d.x + m[6] * d.y + m[10] * d.z + m[14] * 0.0f);
}
/* row-vector product vec4(n,0) * M: component j = dot(vec4(n,0), column j) */
RT_HD rt3 rt_vec_mul_mat_dir(rt3 n, const float* m) {
  return rt3_make(n.x * m[0] + n.y * m[1] + n.z * m[2] + 0.0f * m[3],
                  n.x * m[4] + n.y * m[5] + n.z * m[6] + 0.0f * m[7],
                  n.x * m[8] + n.y * m[9] + n.z * m[10] + 0.0f * m[11]);
}

/* ------------------------------------------------ unorm8 / f16 conversion */
RT_HD uint32_t rt_unorm8(float x) {
  float c = rt_clamp(x, 0.0f, 1.0f);
  if (c != c) c = 0.0f;
  return (uint32_t)rt_floor(c * 255.0f + 0.5f);
}

/* f32 -> f16 bits, round to nearest even, overflow -> inf, NaN stays NaN. */
RT_HD uint16_t rt_f32_to_f16(float f) {
  uint32_t u = rt_f2u(f);
  uint32_t sign = (u >> 16) & 0x8000u;
  uint32_t a = u & 0x7fffffffu;
  if (a > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);
  if (a >= 0x47800000u) return (uint16_t)(sign | 0x7c00u); /* >= 65536 -> inf (65520 handled by rounding) */
  if (a >= 0x38800000u) { /* normal half */
    uint32_t m = a - 0x38000000u; /* rebias exponent: 127-15 = 112 -> 112<<23 */
    uint32_t r = m + 0x00000fffu + ((m >> 13) & 1u);
    return (uint16_t)(sign | (r >> 13));
  }
  if (a < 0x33000000u) return (uint16_t)sign; /* < 2^-25 -> 0 */
  /* subnormal half: value = mant * 2^-24 */
  uint32_t e = a >> 23;
  uint32_t mant = (a & 0x007fffffu) | 0x00800000u;
  uint32_t shift = 126u - e; /* e in [102,112] -> shift in [14,24] */
  uint32_t half = mant >> shift;
  uint32_t rem = mant & ((1u << shift) - 1u);
  uint32_t halfway = 1u << (shift - 1u);
  if (rem > halfway || (rem == halfway && (half & 1u))) half += 1u;
  return (uint16_t)(sign | half);
}
RT_HD float rt_f16_to_f32(uint16_t h) {
  uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
  uint32_t e = (h >> 10) & 0x1fu;
  uint32_t m = h & 0x3ffu;
  if (e == 0) {
    if (m == 0) return rt_u2f(sign);
    /* subnormal: m * 2^-24 */
    float v = (float)m * 5.9604644775390625e-8f;
    return rt_u2f(rt_f2u(v) | sign);
  }
  if (e == 31) return rt_u2f(sign | 0x7f800000u | (m << 13));
  return rt_u2f(sign | ((e + 112u) << 23) | (m << 13));
}

/* ------------------------------------------------ texture ingest: resize to the 1024 x 1024 layer
 * createImageBitmap(blob, {resizeWidth: 1024, resizeHeight: 1024}) with the default resizeQuality "low"
 * (ResourceManager.ts:164-168): bilinear, pixel centres aligned, clamp to edge, per channel on straight RGBA8.
 * rt_resize_coord: source coordinate of destination texel d (0..dst-1) for a source of `src` texels: returns the
 * left/top texel index in *i0, its neighbour in *i1 and the weight of the neighbour. src == dst gives weight 0. */
RT_HD float rt_resize_coord(uint32_t d, uint32_t src, uint32_t dst, uint32_t* i0, uint32_t* i1) {
  float s = ((float)d + 0.5f) * ((float)src / (float)dst) - 0.5f;
  s = rt_clamp(s, 0.0f, (float)(src - 1u));
  float fl = rt_floor(s);
  uint32_t a = (uint32_t)fl;
  *i0 = a;
  *i1 = a + 1u < src ? a + 1u : src - 1u;
  return s - fl;
}
RT_HD uint32_t rt_bilinear_u8(uint32_t c00, uint32_t c10, uint32_t c01, uint32_t c11, float fx, float fy) {
  float top = (float)c00 + ((float)c10 - (float)c00) * fx;
  float bot = (float)c01 + ((float)c11 - (float)c01) * fx;
  float v = top + (bot - top) * fy;
  return (uint32_t)rt_floor(rt_clamp(v, 0.0f, 255.0f) + 0.5f);
}

#endif /* MI355RT_MATH_H */
