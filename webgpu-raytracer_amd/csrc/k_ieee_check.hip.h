// k_ieee_check.hip.h — rt_debug_ieee_check: the device sequences of k_ieee.hip.h against the compiler's IEEE expansions,
// on the GPU, over the input sets of k_ieee_inputs.h.  Test infrastructure inside the library (the sequences are only
// reachable from device code); no renderer state is touched.  tests/test_gpu_ieee.py drives it.
//
// Per input: ref = the plain operator (hipcc's correctly rounded expansion: v_div_scale / v_div_fmas / v_div_fixup,
// scaled v_sqrt_f32 with the +-1 ulp choice).  Three things are recorded:
//   wrong_fn    the COMPOSED function (what the kernels call: guard + wave-uniform branch + cold path) differs from ref
//   guard_pass  the input passes the guard; wrong_fast: of those, the bare sequence differs from ref — evaluated lane by
//               lane, whatever the other lanes of the wave hold, so every guarded input is really put through the sequence
//   checksum    sum of mix(ref, index): compared with the host CPU's IEEE results over the same inputs
#ifndef MI355RT_K_IEEE_CHECK_HIP_H
#define MI355RT_K_IEEE_CHECK_HIP_H

#include "k_ieee_inputs.h"

namespace rtk {

struct IeeeReport {
  unsigned long long n, guard_pass, wrong_fast, wrong_fn, checksum;
  unsigned int n_bad;
  unsigned int bad[8 * 4];   // operand a, operand b, got, ref of the first mismatches
};

__device__ __forceinline__ void ieee_note(IeeeReport* r, bool wrong, uint32_t a, uint32_t b, uint32_t got, uint32_t ref) {
  if (wrong) {
    const unsigned int k = atomicAdd(&r->n_bad, 1u);
    if (k < 8u) {
      r->bad[4 * k] = a; r->bad[4 * k + 1] = b; r->bad[4 * k + 2] = got; r->bad[4 * k + 3] = ref;
    }
  }
}
__device__ __forceinline__ unsigned long long ieee_wave_sum(unsigned long long v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

template <int OP>
__global__ __launch_bounds__(256) void k_ieee_check(unsigned long long first, unsigned long long count, IeeeReport* rep) {
  unsigned long long n_pass = 0, n_wrong_fast = 0, n_wrong_fn = 0, sum = 0;
  const unsigned long long stride = (unsigned long long)gridDim.x * 256ull;
  // every lane runs the same number of trips (the composed functions hold wave-uniform branches); lanes past the end idle
  const unsigned long long trips = (count + stride - 1ull) / stride;
  for (unsigned long long t = 0; t < trips; t++) {
    const unsigned long long k = t * stride + (unsigned long long)blockIdx.x * 256ull + threadIdx.x;
    const bool live = k < count;
    const unsigned long long i = first + (live ? k : 0ull);
    (void)i;
    if (!live) continue;
#if defined(MI355RT_DEVICE_IEEE)
    if (OP == RT_IEEE_OP_RCP || OP == RT_IEEE_OP_SQRT || OP == RT_IEEE_OP_RSQRT || OP == RT_IEEE_OP_DIV_PI) {
      const float x = rt_u2f((uint32_t)i);
      float ref, fn, fast, y0 = 0.0f;
      bool pass;
      if (OP == RT_IEEE_OP_RCP) {
        ref = 1.0f / x; fn = rt_ieee::rcp(x); fast = rt_ieee::rcp_seq(x, y0); pass = rt_ieee::rcp_guard(y0);
      } else if (OP == RT_IEEE_OP_SQRT) {
        ref = __builtin_sqrtf(x); fn = rt_ieee::sqrt(x); pass = rt_ieee::sqrt_guard(x) || x == 0.0f;
        fast = x == 0.0f ? x : rt_ieee::sqrt_seq(x);
      } else if (OP == RT_IEEE_OP_RSQRT) {
        ref = 1.0f / __builtin_sqrtf(x); fn = rt_ieee::rsqrt(x); fast = rt_ieee::rsqrt_seq(x); pass = rt_ieee::sqrt_guard(x);
      } else {
        const float c = 3.14159274101257324219f;
        ref = x / c; fn = rt_ieee::div_const(x, c, 1.0f / c); fast = rt_ieee::divc_seq(x, c, 1.0f / c); pass = rt_ieee::divc_guard(x);
      }
      const uint32_t rb = rt_f2u(ref);
      const bool nan_ref = (rb & 0x7fffffffu) > 0x7f800000u;
      // NaN results: any NaN is the IEEE answer (payloads of the two routes may differ; no kernel compares NaN bits)
      const bool bad_fn = nan_ref ? !(fn != fn) : rt_f2u(fn) != rb;
      const bool bad_fast = pass && (nan_ref ? !(fast != fast) : rt_f2u(fast) != rb);
      n_pass += pass;
      n_wrong_fast += bad_fast;
      n_wrong_fn += bad_fn;
      sum += rt_ieee_mix(rb, i);
      ieee_note(rep, bad_fn || bad_fast, (uint32_t)i, 0u, rt_f2u(bad_fast ? fast : fn), rb);
    } else if (OP == RT_IEEE_OP_DIV) {
      uint32_t ab, bb;
      rt_ieee_div_operands(i, &ab, &bb);
      const float a = rt_u2f(ab), b = rt_u2f(bb);
      const float ref = a / b, fn = rt_ieee::div(a, b), fast = rt_ieee::div_seq(a, b);
      const bool pass = rt_ieee::div_guard(fast, a);
      const uint32_t rb = rt_f2u(ref);
      const bool nan_ref = (rb & 0x7fffffffu) > 0x7f800000u;
      const bool bad_fn = nan_ref ? !(fn != fn) : rt_f2u(fn) != rb;
      const bool bad_fast = pass && (nan_ref ? !(fast != fast) : rt_f2u(fast) != rb);
      n_pass += pass;
      n_wrong_fast += bad_fast;
      n_wrong_fn += bad_fn;
      sum += rt_ieee_mix(rb, i);
      ieee_note(rep, bad_fn || bad_fast, ab, bb, rt_f2u(bad_fast ? fast : fn), rb);
    } else if (OP == RT_IEEE_OP_DIV3 || OP == RT_IEEE_OP_DIV3Z) {
      uint32_t ab[3], bb;
      rt_ieee_div3_operands(i, OP == RT_IEEE_OP_DIV3Z, ab, &bb);
      const float a0 = rt_u2f(ab[0]), a1 = rt_u2f(ab[1]), a2 = rt_u2f(ab[2]), b = rt_u2f(bb);
      const float ref[3] = {a0 / b, a1 / b, a2 / b};
      float fn[3], fast[3], y0;
      bool pass;
      const float y = rt_ieee::rcp_seq(b, y0);
      if (OP == RT_IEEE_OP_DIV3) {
        rt_ieee::div3(a0, a1, a2, b, fn[0], fn[1], fn[2]);
        fast[0] = rt_ieee::div_step(a0, b, y); fast[1] = rt_ieee::div_step(a1, b, y); fast[2] = rt_ieee::div_step(a2, b, y);
        pass = rt_ieee::div_guard(fast[0], a0) && rt_ieee::div_guard(fast[1], a1) && rt_ieee::div_guard(fast[2], a2);
      } else {
        rt_ieee::div3z(a0, a1, a2, b, fn[0], fn[1], fn[2]);
        fast[0] = rt_ieee::divz_step(a0, b, y); fast[1] = rt_ieee::divz_step(a1, b, y); fast[2] = rt_ieee::divz_step(a2, b, y);
        pass = rt_ieee::div3z_guard(a0, a1, a2, y0);
      }
      n_pass += pass;
      for (int c = 0; c < 3; c++) {
        const uint32_t rb = rt_f2u(ref[c]);
        const bool nan_ref = (rb & 0x7fffffffu) > 0x7f800000u;
        const bool bad_fn = nan_ref ? !(fn[c] != fn[c]) : rt_f2u(fn[c]) != rb;
        const bool bad_fast = pass && (nan_ref ? !(fast[c] != fast[c]) : rt_f2u(fast[c]) != rb);
        n_wrong_fast += bad_fast;
        n_wrong_fn += bad_fn;
        sum += rt_ieee_mix(rb, 3ull * i + (unsigned long long)c);
        ieee_note(rep, bad_fn || bad_fast, ab[c], bb, rt_f2u(bad_fast ? fast[c] : fn[c]), rb);
      }
    } else if (OP == RT_IEEE_OP_UNORM8) {
      const uint32_t q = (uint32_t)i & 255u;
      const float ref = (float)q / 255.0f, fn = rt_ieee::unorm8(q);
      n_pass += 1;
      n_wrong_fast += rt_f2u(fn) != rt_f2u(ref);
      n_wrong_fn += rt_f2u(fn) != rt_f2u(ref);
      sum += rt_ieee_mix(rt_f2u(ref), i);
      ieee_note(rep, rt_f2u(fn) != rt_f2u(ref), q, 0u, rt_f2u(fn), rt_f2u(ref));
    }
#endif
  }
  n_pass = ieee_wave_sum(n_pass);
  n_wrong_fast = ieee_wave_sum(n_wrong_fast);
  n_wrong_fn = ieee_wave_sum(n_wrong_fn);
  sum = ieee_wave_sum(sum);
  if ((threadIdx.x & 63u) == 0u) {
    atomicAdd(&rep->guard_pass, n_pass);
    atomicAdd(&rep->wrong_fast, n_wrong_fast);
    atomicAdd(&rep->wrong_fn, n_wrong_fn);
    atomicAdd(&rep->checksum, sum);
  }
}

}  // namespace rtk
#endif
