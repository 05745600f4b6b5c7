// ieee_probe.hip — which short sequences around v_rcp_f32 / v_rsq_f32 / v_sqrt_f32 are CORRECTLY ROUNDED on this chip.
//
// The kernels' arithmetic contract is single IEEE binary32 operations (include/mi355rt_math.h).  The compiler's
// expansion of a division is 11 vector instructions (v_div_scale x 2, v_rcp, 6 fma/mul, v_div_fmas, v_div_fixup), of a
// square root 16.  This probe measures, exhaustively where the domain allows it, for which inputs shorter sequences
// return the same bits as those expansions, so that csrc/k_ieee.hip.h can take them behind a guard.  It is an
// experiment tool: the shipped guard + sequences are proven again by tests/test_gpu_ieee.py through the C ABI.
//
//   rcp:   all 2^32 inputs                       sqrt: all 2^32 inputs
//   div:   2^32 pairs (hashed mantissas x an exponent grid)      x / const: all 2^32 inputs per constant
//   rsqrt-then-rcp (normalize): all 2^32 inputs
//
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o ieee_probe tools/ieee_probe.hip     run: ./ieee_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                     \
      return 1;                                                                   \
    }                                                                             \
  } while (0)

__device__ __forceinline__ uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
__device__ __forceinline__ float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
#define CLS_NORMAL 0x108   // -normal | +normal
__device__ __forceinline__ bool is_normal(float x) { return __builtin_amdgcn_classf(x, CLS_NORMAL); }   // the f32 form: the generic builtin widens to f64, where an f32 denormal is normal

struct Report {
  unsigned long long pass;       // inputs the guard lets through
  unsigned long long bad;        // of those: fast != IEEE
  unsigned long long bad_all;    // fast != IEEE over ALL inputs (what the guard has to catch)
  unsigned int n_list;
  unsigned int list[64 * 4];     // a, b, fast, ref
  unsigned int hist_pass[256];   // wrong behind the guard, by the biased exponent of the first operand
  unsigned int hist_all[256];    // wrong over all inputs, by the biased exponent of the first operand
};

__device__ void note(Report* r, bool pass, bool same, uint32_t a, uint32_t b, uint32_t fast, uint32_t ref) {
  if (pass) atomicAdd(&r->pass, 1ull);
  if (!same) {
    atomicAdd(&r->bad_all, 1ull);
    atomicAdd(&r->hist_all[(a >> 23) & 255u], 1u);
  }
  if (pass && !same) {
    atomicAdd(&r->bad, 1ull);
    atomicAdd(&r->hist_pass[(a >> 23) & 255u], 1u);
    unsigned int k = atomicAdd(&r->n_list, 1u);
    if (k < 64u) {
      r->list[4 * k] = a; r->list[4 * k + 1] = b; r->list[4 * k + 2] = fast; r->list[4 * k + 3] = ref;
    }
  }
}

// ----------------------------------------------------------------------------------------------- reciprocal
// guard: the hardware estimate is a normal number (|x| in [2^-126, 2^126], no zero / inf / NaN / denormal either side)
template <int V>
__device__ __forceinline__ float rcp_fast(float x, bool& ok) {
  float y0 = __builtin_amdgcn_rcpf(x);
  ok = is_normal(y0);
  float e = fma_(-x, y0, 1.0f);
  float y1 = fma_(e, y0, y0);
  if (V == 0) return y1;                       // 1 trans + 2
  float e1 = fma_(-x, y1, 1.0f);
  if (V == 1) return fma_(e1, y1, y1);         // 1 trans + 4
  if (V == 2) return fma_(e1, y0, y1);         // same, the correction scaled by the raw estimate
  return y1;
}
template <int V>
__global__ void k_rcp(Report* r) {
  for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < (1ull << 32);
       i += (unsigned long long)gridDim.x * blockDim.x) {
    float x = u2f((uint32_t)i);
    bool ok;
    float f = rcp_fast<V>(x, ok);
    float ref = 1.0f / x;
    note(r, ok, f2u(f) == f2u(ref), (uint32_t)i, 0u, f2u(f), f2u(ref));
  }
}

// ------------------------------------------------------------------------------------------------ square root
template <int V>
__device__ __forceinline__ float sqrt_fast(float x, bool& ok, float* half_rsq = nullptr) {
  if (V == 0) {   // v_sqrt_f32 and the +-1 ulp choice of the compiler's expansion, without its range scaling: 1 trans + 8
    float s = __builtin_amdgcn_sqrtf(x);
    ok = is_normal(s) && is_normal(x);
    float dn = u2f(f2u(s) - 1u), up = u2f(f2u(s) + 1u);
    float rdn = fma_(-dn, s, x), rup = fma_(-up, s, x);
    s = (rdn <= 0.0f) ? dn : s;
    s = (rup > 0.0f) ? up : s;
    return s;
  }
  if (V == 3) {   // v_sqrt_f32 corrected once by its residual times rsq / 2: 2 trans + 3
    float s = __builtin_amdgcn_sqrtf(x);
    float h = 0.5f * __builtin_amdgcn_rsqf(x);
    ok = is_normal(s) && is_normal(x);
    float d = fma_(-s, s, x);
    if (half_rsq) *half_rsq = h;
    return fma_(d, h, s);
  }
  // rsq-based (Markstein): 1 trans + 7
  float y = __builtin_amdgcn_rsqf(x);
  ok = is_normal(y) && is_normal(x);
  float g = x * y;
  float h = 0.5f * y;
  if (V == 2) {   // one correction only: 1 trans + 4
    float d = fma_(-g, g, x);
    if (half_rsq) *half_rsq = h;
    return fma_(d, h, g);
  }
  float e = fma_(-h, g, 0.5f);
  h = fma_(h, e, h);
  g = fma_(g, e, g);
  float d = fma_(-g, g, x);
  if (half_rsq) *half_rsq = h;
  return fma_(d, h, g);
}
template <int V>
__global__ void k_sqrt(Report* r) {
  for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < (1ull << 32);
       i += (unsigned long long)gridDim.x * blockDim.x) {
    float x = u2f((uint32_t)i);
    bool ok;
    float f = sqrt_fast<V>(x, ok);
    float ref = __builtin_sqrtf(x);
    note(r, ok, f2u(f) == f2u(ref), (uint32_t)i, 0u, f2u(f), f2u(ref));
  }
}

// ----------------------------------------------------------------------- 1 / sqrt(x) as TWO roundings (normalize)
// V = 0: sqrt variant 1, then the reciprocal seeded by 2h (no second transcendental), one refinement
// V = 1: the same with two refinements      V = 2: sqrt variant 1 + rcp variant 1 (v_rcp_f32 seed)
template <int V>
__global__ void k_rsqrt2(Report* r) {
  for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < (1ull << 32);
       i += (unsigned long long)gridDim.x * blockDim.x) {
    float x = u2f((uint32_t)i);
    bool ok, ok2 = true;
    float h;
    float s = sqrt_fast<1>(x, ok, &h);
    float f;
    if (V == 2) {
      f = rcp_fast<1>(s, ok2);
    } else {
      float y0 = h + h;
      ok2 = is_normal(y0);
      float e = fma_(-s, y0, 1.0f);
      float y1 = fma_(e, y0, y0);
      if (V == 0) {
        f = y1;
      } else {
        float e1 = fma_(-s, y1, 1.0f);
        f = fma_(e1, y1, y1);
      }
    }
    float ref = 1.0f / __builtin_sqrtf(x);
    note(r, ok && ok2, f2u(f) == f2u(ref), (uint32_t)i, 0u, f2u(f), f2u(ref));
  }
}

// --------------------------------------------------------------------------------------------------- division
__device__ __forceinline__ uint32_t hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
// guard: estimate normal, quotient normal, |a| > 2^-100 (the residual a - b q must not underflow)
template <int V, int G>
__device__ __forceinline__ float div_fast(float a, float b, bool& ok) {
  float y0 = __builtin_amdgcn_rcpf(b);
  float e = fma_(-b, y0, 1.0f);
  float y = fma_(e, y0, y0);
  if (V >= 2) {   // reciprocal refined twice first
    float e1 = fma_(-b, y, 1.0f);
    y = fma_(e1, y, y);
  }
  float q = a * y;
  float rr = fma_(-b, q, a);
  q = fma_(rr, y, q);
  if (V == 1 || V == 3) {   // second residual step (the compiler's sequence without scaling is V = 1)
    rr = fma_(-b, q, a);
    q = fma_(rr, y, q);
  }
  if (G == 0) ok = is_normal(q);                                                          // one compare
  if (G == 1) ok = is_normal(q) && (__builtin_fabsf(a) > 7.888609052210118e-31f);         // + |a| > 2^-100
  if (G == 2) ok = is_normal(y0) && is_normal(q) && (__builtin_fabsf(a) > 7.888609052210118e-31f);
  return q;
}
template <int V, int G>
__global__ void k_div(Report* r, int mode) {
  for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < (1ull << 32);
       i += (unsigned long long)gridDim.x * blockDim.x) {
    uint32_t h0 = hash32((uint32_t)i), h1 = hash32((uint32_t)i ^ 0x9e3779b9u), h2 = hash32(h0 + 0x85ebca6bu);
    uint32_t ea, eb;
    if (mode == 0) {   // exponents near 1: what the renderer mostly divides
      ea = 127u + (h2 & 15u) - 8u;
      eb = 127u + ((h2 >> 4) & 15u) - 8u;
    } else {           // the whole exponent range incl. zero / denormal / inf / NaN classes
      ea = (h2 >> 8) & 255u;
      eb = (h2 >> 16) & 255u;
    }
    uint32_t ma = h0 & 0x7fffffu, mb = h1 & 0x7fffffu;
    if ((h2 & 0x3000000u) == 0u) mb = (h1 & 1u) ? 0x7fffffu - (h1 >> 28) : (h1 >> 28);   // mantissas near all-ones / zero
    if ((h2 & 0xc000000u) == 0u) ma = (h0 & 1u) ? 0x7fffffu - (h0 >> 28) : (h0 >> 28);
    float a = u2f((h0 & 0x80000000u) | (ea << 23) | ma), b = u2f((h1 & 0x80000000u) | (eb << 23) | mb);
    bool ok;
    float f = div_fast<V, G>(a, b, ok);
    float ref = a / b;
    note(r, ok, f2u(f) == f2u(ref), f2u(a), f2u(b), f2u(f), f2u(ref));
  }
}

// ------------------------------------------------------------------------------------- x / c, c a constant
// q = x * RN(1/c); r = fma(-c, q, x); q' = fma(r, RN(1/c), q); guard: |x| in (2^-100, 2^100)
__global__ void k_divc(Report* r, float c, float rc, int steps, int guard) {
  for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < (1ull << 32);
       i += (unsigned long long)gridDim.x * blockDim.x) {
    float x = u2f((uint32_t)i);
    float q = x * rc;
    float rr = fma_(-c, q, x);
    q = fma_(rr, rc, q);
    if (steps == 2) {
      rr = fma_(-c, q, x);
      q = fma_(rr, rc, q);
    }
    bool ok = true;
    if (guard == 0) ok = __builtin_fabsf(x) > 7.888609052210118e-31f && __builtin_fabsf(x) < 1.2676506002282294e30f;
    if (guard == 1) {   // v_div_fixup for zeros / infinities / NaN, then only tiny non-zero numerators are left
      q = __builtin_amdgcn_div_fixupf(q, c, x);
      ok = __builtin_fabsf(x) > 7.888609052210118e-31f || x == 0.0f;
    }
    if (guard == 2) {   // one class compare: x is normal or zero; v_div_fixup for the sign of zero
      q = __builtin_amdgcn_div_fixupf(q, c, x);
      ok = __builtin_amdgcn_classf(x, 0x108 | 0x60);
    }
    float ref = x / c;
    note(r, ok, f2u(q) == f2u(ref), (uint32_t)i, f2u(c), f2u(q), f2u(ref));
  }
}

// what v_rcp_f32 / v_rsq_f32 / v_sqrt_f32 do with denormal inputs and how far they are from the rounded result
__global__ void k_raw(unsigned long long* hist) {   // hist[kind * 8 + bucket]: |ulp error| 0, 1, 2, 3, >3
  for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < (1ull << 31);
       i += (unsigned long long)gridDim.x * blockDim.x) {
    float x = u2f((uint32_t)i);
    if (!is_normal(x)) continue;
    float a[3] = {__builtin_amdgcn_rcpf(x), __builtin_amdgcn_rsqf(x), __builtin_amdgcn_sqrtf(x)};
    float b[3] = {1.0f / x, 1.0f / __builtin_sqrtf(x), __builtin_sqrtf(x)};
    for (int k = 0; k < 3; k++) {
      if (!is_normal(b[k])) continue;
      long long d = (long long)f2u(a[k]) - (long long)f2u(b[k]);
      if (d < 0) d = -d;
      atomicAdd(&hist[k * 8 + (d > 3 ? 4 : (int)d)], 1ull);
    }
  }
}

static int show(const char* name, Report* d) {
  Report h;
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost));
  printf("%-44s guard passes %11llu  wrong behind the guard %9llu  wrong over all inputs %11llu\n", name, h.pass, h.bad, h.bad_all);
  for (int which = 0; which < 2; which++) {
    const unsigned* hs = which ? h.hist_all : h.hist_pass;
    bool any = false;
    for (int e = 0; e < 256; e++) any |= hs[e] != 0;
    if (!any) continue;
    printf("      %s by exponent of the first operand:", which ? "wrong over all inputs" : "wrong behind the guard");
    for (int e = 0; e < 256;) {
      if (!hs[e]) { e++; continue; }
      int f = e;
      while (f + 1 < 256 && hs[f + 1] == hs[e]) f++;
      if (f > e) printf(" [%d..%d]: %u each", e, f, hs[e]); else printf(" [%d]: %u", e, hs[e]);
      e = f + 1;
    }
    printf("\n");
  }
  unsigned n = h.n_list < 6u ? h.n_list : 6u;
  for (unsigned k = 0; k < n; k++)
    printf("      a %08x b %08x fast %08x ieee %08x\n", h.list[4 * k], h.list[4 * k + 1], h.list[4 * k + 2], h.list[4 * k + 3]);
  fflush(stdout);
  CHECK(hipMemset(d, 0, sizeof(Report)));
  return 0;
}

int main() {
  CHECK(hipSetDevice(0));
  Report* d;
  CHECK(hipMalloc(&d, sizeof(Report)));
  CHECK(hipMemset(d, 0, sizeof(Report)));
  const dim3 grid(256 * 16), block(256);
  unsigned long long* hist;
  CHECK(hipMalloc(&hist, 24 * 8));
  CHECK(hipMemset(hist, 0, 24 * 8));
  hipLaunchKernelGGL(k_raw, grid, block, 0, 0, hist);
  unsigned long long hh[24];
  CHECK(hipMemcpy(hh, hist, sizeof(hh), hipMemcpyDeviceToHost));
  const char* rn[3] = {"v_rcp_f32", "v_rsq_f32", "v_sqrt_f32"};
  for (int k = 0; k < 3; k++)
    printf("%-12s ulp distance from the rounded result, normal inputs: 0: %llu  1: %llu  2: %llu  3: %llu  more: %llu\n", rn[k],
           hh[k * 8], hh[k * 8 + 1], hh[k * 8 + 2], hh[k * 8 + 3], hh[k * 8 + 4]);

#define RUN(NAME, K, ...)                                        \
  hipLaunchKernelGGL(K, grid, block, 0, 0, d, ##__VA_ARGS__);    \
  if (show(NAME, d)) return 1;
  RUN("rcp  y1 (rcp + 2 fma)", k_rcp<0>);
  RUN("rcp  y2 (rcp + 4 fma)", k_rcp<1>);
  RUN("rcp  y2' (second correction x y0)", k_rcp<2>);
  RUN("sqrt v_sqrt + ulp choice (1 + 8)", k_sqrt<0>);
  RUN("sqrt rsq Markstein (1 + 7)", k_sqrt<1>);
  RUN("sqrt rsq one correction (1 + 4)", k_sqrt<2>);
  RUN("sqrt v_sqrt + residual x rsq/2 (2 + 3)", k_sqrt<3>);
  RUN("1/sqrt  sqrt(1+7), seed 2h, 1 refinement", k_rsqrt2<0>);
  RUN("1/sqrt  sqrt(1+7), seed 2h, 2 refinements", k_rsqrt2<1>);
  RUN("1/sqrt  sqrt(1+7) + rcp y2", k_rsqrt2<2>);
  for (int mode = 0; mode < 2; mode++) {
    printf("division, %s\n", mode ? "exponents over the whole range" : "exponents within 2^+-8");
    RUN("div  (1 + 5), guard: q normal", (k_div<0, 0>), mode);
    RUN("div  (1 + 5), guard: q normal, |a| > 2^-100", (k_div<0, 1>), mode);
    RUN("div  (1 + 5), guard: y0 normal, q normal, |a| > 2^-100", (k_div<0, 2>), mode);
    RUN("div  (1 + 7, two residual steps), guard: q normal, |a| > 2^-100", (k_div<1, 1>), mode);
  }
  const float consts[4] = {3.14159274101257324219f, 255.0f, 6.28318548202514648438f, 3.0f};
  for (int k = 0; k < 4; k++) {
    char nm[64];
    for (int g = 0; g < 3; g++) {
      snprintf(nm, sizeof nm, "x / %.9g, one step, guard %d", consts[k], g);
      RUN(nm, k_divc, consts[k], 1.0f / consts[k], 1, g);
    }
  }
  return 0;
}
