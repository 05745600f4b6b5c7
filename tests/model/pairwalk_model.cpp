// pairwalk_model.cpp — TEST INFRASTRUCTURE: single rays through the child-pair walk of the HIP kernels, on the host.
// It compiles the very lane functions the GPU runs (webgpu-raytracer_amd/csrc/k_pairwalk.hip.h) and drives them one ray
// at a time; tests/test_pairwalk_model.py compares every ray's hit, occlusion bit, nodes_visited and tris_tested with
// the oracle's literal loop (oracle_trace_rays).  What it does NOT model is the wave-level scheduling of the kernels
// (which lane steps when) — that only changes WHEN a ray takes its next step, never what the step does.
// build: g++ -O2 -ffp-contract=off -std=c++17 -shared -fPIC (tests/test_pairwalk_model.py does it)
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/mi355rt_math.h"
#include "../../webgpu-raytracer_amd/csrc/k_pairwalk.hip.h"

using namespace rtk;

namespace {
struct HostStack {
  std::vector<uint32_t> w;
  std::vector<float> a;
  explicit HostStack(uint32_t k) : w(k), a(k) {}
  void push(uint32_t slot, uint32_t word, float av) { w[slot] = word; a[slot] = av; }
  void pop(uint32_t slot, uint32_t& word, float& av) { word = w[slot]; av = a[slot]; }
};

uint32_t fbits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

// hit_tri_nb of k_traverse.hip.h: branch-free Möller–Trumbore on {v0, e1, e2}
bool hit_tri(const float* g, const PwRay& r, float t_min, float t_max, float& t_out) {
  rt3 v0 = rt3_make(g[0], g[1], g[2]), e1 = rt3_make(g[4], g[5], g[6]), e2 = rt3_make(g[8], g[9], g[10]);
  rt3 h = rt_cross(r.d, e2);
  float a = rt_dot(e1, h);
  float f = 1.0f / a;
  rt3 s = r.o - v0;
  float u = f * rt_dot(s, h);
  rt3 q = rt_cross(s, e1);
  float v = f * rt_dot(r.d, q);
  float t = f * rt_dot(e2, q);
  t_out = t;
  bool reject = (rt_abs(a) < 1e-6f) | (u < 0.0f) | (u > 1.0f) | (v < 0.0f) | (u + v > 1.0f);
  return !reject & (t > t_min) & (t < t_max);
}

template <bool COUNT, uint32_t K>
void trace_one(const float* pairs, const float* troot, const float* inst_trav, const float* inst_root, const float* tri_geom,
               const float* q, bool any, float* out, uint64_t* counts, uint64_t* stats) {
  PairLane s;
  HostStack stk(K);
  uint32_t n_nodes = 0, n_tris = 0;
  const float t_min = q[3];
  pw_begin<COUNT>(s, true, any, rt3_make(q[0], q[1], q[2]), rt3_make(q[4], q[5], q[6]), t_min, q[7], troot[0], troot[1],
                  troot[2], fbits(troot[3]), troot[4], troot[5], troot[6], n_nodes);
  uint64_t steps = 0;
  bool went_stackless = false;
  while (s.state != PW_DONE) {
    steps++;
    went_stackless |= (s.flags & (PW_F_SL_TLAS | PW_F_SL_BLAS)) != 0u;
    switch (s.state) {
      case PW_FETCH:
      case PW_FETCHR: {
        if (s.state == PW_FETCHR) stats[1]++;
        const float* p = pairs + (size_t)s.curr * 16;
        pw_pair<COUNT, K>(s, s.curr, p[0], p[1], p[2], fbits(p[3]), p[4], p[5], p[6], p[8], p[9], p[10], fbits(p[11]), p[12],
                          p[13], p[14], fbits(p[15]), t_min, stk, n_nodes);
        stats[0]++;
        break;
      }
      case PW_POP:
        pw_pop<COUNT>(s, stk, n_nodes);
        break;
      case PW_ENTER: {
        const float* m = inst_trav + (size_t)s.cur_inst * 16;
        const float* rr = inst_root + (size_t)s.cur_inst * 8;
        pw_enter<COUNT, K>(s, m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7], m[8], m[9], m[10], m[11], rr[0], rr[1], rr[2],
                           fbits(rr[3]), rr[4], rr[5], rr[6], t_min, stk, n_nodes);
        break;
      }
      case PW_WAIT: {
        // the reference's leaf loop (Raytracer.wgsl:474-482 / :548-556); the kernels' triangle flush computes the same
        // minimum over (t, position)
        const uint32_t first = s.leaf >> 3, cnt = s.leaf & 7u;
        bool found = false;
        float best_t = s.closest;
        uint32_t best_tri = 0;
        for (uint32_t i = 0; i < cnt; i++) {
          n_tris++;
          float t;
          if (hit_tri(tri_geom + (size_t)(first + i) * 12, s.r, t_min, best_t, t)) {
            found = true;
            best_t = t;
            best_tri = first + i;
            if (any) break;
          }
        }
        pw_after_leaf(s, found, best_t, best_tri);
        break;
      }
      default:
        s.state = PW_DONE;
    }
  }
  if (went_stackless) stats[2]++;
  out[0] = s.closest;
  out[1] = (float)s.best_tri;
  out[2] = (float)s.best_inst;
  out[3] = pw_flag(s, PW_F_FOUND) ? 1.0f : 0.0f;
  counts[0] = n_nodes;
  counts[1] = n_tris;
  stats[3] += steps;
}
}  // namespace

extern "C" {
// stats (4 u64): pair fetches, stackless arrivals, rays that had a stackless level at some point, state-machine steps
void pwm_trace(const float* pairs, const float* troot, const float* inst_trav, const float* inst_root, const float* tri_geom,
               const float* rays, uint32_t n, int any, uint32_t k, int count, float* out, uint64_t* counts, uint64_t* stats) {
  for (int i = 0; i < 4; i++) stats[i] = 0;
  for (uint32_t i = 0; i < n; i++) {
    const float* q = rays + (size_t)i * 8;
    float* o = out + (size_t)i * 4;
    uint64_t* c = counts + (size_t)i * 2;
#define GO(KK)                                                                                              \
  if (count) trace_one<true, KK>(pairs, troot, inst_trav, inst_root, tri_geom, q, any != 0, o, c, stats); \
  else trace_one<false, KK>(pairs, troot, inst_trav, inst_root, tri_geom, q, any != 0, o, c, stats);
    switch (k) {
      case 1: GO(1) break;
      case 2: GO(2) break;
      case 4: GO(4) break;
      case 6: GO(6) break;
      case 8: GO(8) break;
      default: GO(64) break;
    }
#undef GO
  }
}
}
