// k_pairtrav.hip.h — the wave-level side of the child-pair walk: how the 64 lanes of a wave fetch their pair records,
// when they pop / enter instances / flush their triangle queue.  The per-ray decisions are k_pairwalk.hip.h's lane
// functions (checked on the host, tests/test_pairwalk_model.py); nothing here changes WHAT a ray does at a step, only
// WHEN a lane takes it.
// Part of the kernel set of csrc/kernels.hip.h (included from there, in order; not a stand-alone header).
#ifndef MI355RT_K_PAIRTRAV_HIP_H
#define MI355RT_K_PAIRTRAV_HIP_H

namespace rtk {

#ifndef RT_PW_STACK_K
#define RT_PW_STACK_K 8   // deferred right children a lane can hold (8 bytes each in LDS).  Fall-back rate measured on the
                          // host model (tests/test_pairwalk_model.py, random rays): K = 8: 0.3 % of the rays of the 263 k-
                          // triangle scene leave the stack, +0.5 % record fetches; K = 6: 2.7 %, +3 %; K = 4: 14 %, +15 %
#endif
#ifndef RT_PW_STEPS_PER_TRIP
#define RT_PW_STEPS_PER_TRIP 2   // record fetches (= two node tests each) between two looks at the triangle queue
#endif
#define RT_PW_STACK_BYTES_PER_WAVE (RT_PW_STACK_K * 64 * 8)
#define RT_PW_BYTES_PER_WAVE (RT_WORK_BYTES_PER_WAVE + RT_PW_STACK_BYTES_PER_WAVE)

typedef unsigned long long __attribute__((address_space(3))) * rt_lptr64;

// Where the walk's records live.  Pair records: all of them in LDS (l_pairs set: small scenes) or all in global memory;
// triangle records, instance rows and root records likewise, each on its own.  Slots are 16-byte units of the
// workgroup's dynamic LDS array.
struct PairMem {
  const f4* gpairs;     // 4 per pair record
  const f4* gtri;       // tri_geom, RT_TRI_STRIDE per triangle
  const f4* ginst;      // inst_trav, 4 per instance
  const f4* groot;      // root_rec, 2 per instance; the TLAS root is record n_inst
  uint32_t n_inst;
  uint32_t l_pairs, l_tri, l_inst, l_root;
  float t_min;          // Raytracer.wgsl T_MIN for path rays; the primary pass has its own (z_near / focal length)
};

struct LdsStack {       // slot k of this lane: base[k * 64] (consecutive lanes are consecutive 8-byte words: no bank conflicts)
  rt_lptr64 base;
  __device__ __forceinline__ void push(uint32_t slot, uint32_t word, float a) {
    base[slot * 64u] = ((unsigned long long)rt_f2u(a) << 32) | (unsigned long long)word;
  }
  __device__ __forceinline__ void pop(uint32_t slot, uint32_t& word, float& a) {
    const unsigned long long v = base[slot * 64u];
    word = (uint32_t)v;
    a = rt_u2f((uint32_t)(v >> 32));
  }
};
__device__ __forceinline__ LdsStack pw_stack_at(char* wave_base) {   // wave_base: this wave's RT_PW_BYTES_PER_WAVE block
  LdsStack s;
  s.base = (rt_lptr64)(reinterpret_cast<unsigned long long*>(wave_base + RT_WORK_BYTES_PER_WAVE) + (threadIdx.x & 63u));
  return s;
}

// What a workgroup stages in LDS behind its wave blocks (decided on the host, rt_api.hip plan_pairs): each array whole or
// not at all.
struct PairPlan {
  uint32_t stage_pairs, stage_inst, stage_tri, pad;
};
// Fill PairMem and stage what the plan names (all threads of the workgroup; the caller synchronises).
__device__ __forceinline__ void pw_stage(PairMem& M, f4* lds, uint32_t slot0, const DevScene& Sg, const PairPlan& P, uint32_t n_pairs,
                                         uint32_t n_tris, uint32_t n_inst, float t_min) {
  uint32_t slot = slot0;
  M.gpairs = reinterpret_cast<const f4*>(Sg.pairs);
  M.gtri = reinterpret_cast<const f4*>(Sg.tri_geom);
  M.ginst = reinterpret_cast<const f4*>(Sg.inst_trav);
  M.groot = reinterpret_cast<const f4*>(Sg.root_rec);
  M.n_inst = n_inst;
  M.t_min = t_min;
  M.l_pairs = M.l_tri = M.l_inst = M.l_root = RT_LDS_NONE;
  if (P.stage_pairs) {
    M.l_pairs = slot;
    lds_stage(lds + slot, Sg.pairs, (size_t)4 * n_pairs);
    slot += 4u * n_pairs;
  }
  if (P.stage_inst) {
    M.l_inst = slot;
    lds_stage(lds + slot, Sg.inst_trav, (size_t)4 * n_inst);
    slot += 4u * n_inst;
    M.l_root = slot;
    lds_stage(lds + slot, Sg.root_rec, (size_t)2 * n_inst);
    slot += 2u * n_inst;
  }
  if (P.stage_tri) {
    M.l_tri = slot;
    lds_stage(lds + slot, Sg.tri_geom, (size_t)RT_TRI_STRIDE * n_tris);
    slot += (uint32_t)RT_TRI_STRIDE * n_tris;
  }
}

// ---- DPP helpers (quad = 4 consecutive lanes).  Called from wave-uniform control flow only: a DPP read of a lane that
// is switched off returns 0 (bound_ctrl), not that lane's register.
template <int QP>
__device__ __forceinline__ uint32_t quad_bcast(uint32_t v) {   // lane QP of the quad, to all four
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, QP * 0x55, 0xf, 0xf, true);
}
template <int ROT>
__device__ __forceinline__ float quad_rot(float v) {           // lane r receives the value of lane (r + ROT) & 3
  constexpr int ctrl = ((0 + ROT) & 3) | (((1 + ROT) & 3) << 2) | (((2 + ROT) & 3) << 4) | (((3 + ROT) & 3) << 6);
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), ctrl, 0xf, 0xf, true));
}
template <int ROT>
__device__ __forceinline__ f4 quad_rot4(f4 v) {
  f4 r;
  r.x = quad_rot<ROT>(v.x); r.y = quad_rot<ROT>(v.y); r.z = quad_rot<ROT>(v.z); r.w = quad_rot<ROT>(v.w);
  return r;
}
__device__ __forceinline__ f4 sel4(bool c, f4 a, f4 b) {   // c ? a : b, per component (v_cndmask)
  f4 r;
  r.x = c ? a.x : b.x; r.y = c ? a.y : b.y; r.z = c ? a.z : b.z; r.w = c ? a.w : b.w;
  return r;
}

// QUAD-COOPERATIVE fetch of 64-byte pair records from global memory.  Instruction k (k = 0..3) serves the rays of the
// lanes 4q + k: the four lanes of quad q read the four 16-byte chunks of THAT ray's record — one 64-byte line per quad
// and instruction instead of one line per lane — lane 4q + c taking chunk (c - k) & 3.  Measured (tools/gather_peak.hip
// rows "4q"): a step costs the texture-address path what ONE 16-byte load per lane cost, L1-resident or not.  Then every
// lane rotates its four registers by its own position (two rounds of selects) and three quad rotations (DPP) hand each
// owner the chunks 1..3 of its record; chunk 0 is its own.
__device__ __forceinline__ void pw_fetch_quad(const f4* gpairs, bool need, uint32_t idx, f4& q0, f4& q1, f4& q2, f4& q3) {
  const uint32_t c = threadIdx.x & 3u;
  const uint32_t i0 = quad_bcast<0>(idx), i1 = quad_bcast<1>(idx), i2 = quad_bcast<2>(idx), i3 = quad_bcast<3>(idx);
  const uint32_t nd = need ? 1u : 0u;
  const bool n0 = quad_bcast<0>(nd) != 0u, n1 = quad_bcast<1>(nd) != 0u, n2 = quad_bcast<2>(nd) != 0u, n3 = quad_bcast<3>(nd) != 0u;
  f4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0, v2 = v0, v3 = v0;
  if (n0) v0 = ld_g(gpairs, 4 * (size_t)i0 + c);
  if (n1) v1 = ld_g(gpairs, 4 * (size_t)i1 + ((c + 3u) & 3u));
  if (n2) v2 = ld_g(gpairs, 4 * (size_t)i2 + ((c + 2u) & 3u));
  if (n3) v3 = ld_g(gpairs, 4 * (size_t)i3 + ((c + 1u) & 3u));
  // u[i] = v[(i + c) & 3]: rotate the register file of the lane left by its position in the quad
  const bool b0 = (c & 1u) != 0u, b1 = (c & 2u) != 0u;
  const f4 a0 = sel4(b0, v1, v0), a1 = sel4(b0, v2, v1), a2 = sel4(b0, v3, v2), a3 = sel4(b0, v0, v3);
  const f4 u0 = sel4(b1, a2, a0), u1 = sel4(b1, a3, a1), u2 = sel4(b1, a0, a2), u3 = sel4(b1, a1, a3);
  // chunk j of the owner's record sits in lane (owner + j) & 3, in that lane's u[(4 - j) & 3]
  q0 = u0;
  q1 = quad_rot4<1>(u3);
  q2 = quad_rot4<2>(u2);
  q3 = quad_rot4<3>(u1);
}

// can this lane still move without the wave's help?
__device__ __forceinline__ bool pw_can_step(const PairLane& s) {
  return s.state == PW_FETCH || s.state == PW_FETCHR || s.state == PW_POP || s.state == PW_LEVEL_END;
}
__device__ __forceinline__ bool pw_busy(const PairLane& s) { return s.state != PW_DONE; }

#ifndef RT_PW_ENTER_BATCH
#define RT_PW_ENTER_BATCH 16u   // lanes that wait for an instance entry before the wave does it (deferred entry, as before)
#endif

// STEPS record fetches for every lane that wants one, with the cheap transitions (pop, leave an instance, enter one) in
// between.  LDS = every record is in LDS (plain ds_read, entry on the spot).
template <bool COUNT, bool LDS, int STEPS>
__device__ __forceinline__ void pw_trip(const PairMem& M, const f4* lds, LdsStack& stk, PairLane& s, uint32_t& n_nodes) {
  const bool inst_lds = LDS || M.l_inst != RT_LDS_NONE;   // wave-uniform
#pragma unroll
  for (int k = 0; k < STEPS; k++) {
    // ---- transitions that need no record: twice, so that "level ended -> pop the TLAS entry" costs no extra round
#pragma unroll
    for (int rep = 0; rep < 2; rep++) {
      if (__ballot(s.state == PW_LEVEL_END) != 0ull) {
        if (s.state == PW_LEVEL_END) pw_level_end(s);
      }
      if (__ballot(s.state == PW_POP) != 0ull) {
        if (s.state == PW_POP) pw_pop<COUNT>(s, stk, n_nodes);
      }
    }
    // ---- instance entry: on the spot when the rows are in LDS; else when enough lanes wait, or nobody else can move
    {
      const unsigned long long em = __ballot(s.state == PW_ENTER);
      if (em != 0ull && (inst_lds || (uint32_t)__builtin_popcountll(em) >= RT_PW_ENTER_BATCH || __ballot(pw_can_step(s)) == 0ull)) {
        if (s.state == PW_ENTER) {
          f4 r0, r1, r2, b0, b1;
          if (inst_lds) {
            r0 = ld_l(lds, M.l_inst + 4u * s.cur_inst + 0u);
            r1 = ld_l(lds, M.l_inst + 4u * s.cur_inst + 1u);
            r2 = ld_l(lds, M.l_inst + 4u * s.cur_inst + 2u);
            b0 = ld_l(lds, M.l_root + 2u * s.cur_inst + 0u);
            b1 = ld_l(lds, M.l_root + 2u * s.cur_inst + 1u);
          } else {
            r0 = ld_g(M.ginst, 4 * (size_t)s.cur_inst + 0);
            r1 = ld_g(M.ginst, 4 * (size_t)s.cur_inst + 1);
            r2 = ld_g(M.ginst, 4 * (size_t)s.cur_inst + 2);
            b0 = ld_g(M.groot, 2 * (size_t)s.cur_inst + 0);
            b1 = ld_g(M.groot, 2 * (size_t)s.cur_inst + 1);
          }
          pw_enter<COUNT>(s, r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w, b0.x, b0.y, b0.z,
                          rt_f2u(b0.w), b1.x, b1.y, b1.z, M.t_min, n_nodes);
        }
      }
    }
    // ---- one pair record for every lane that wants one
    const bool need = s.state == PW_FETCH || s.state == PW_FETCHR;
    if (__ballot(need) != 0ull) {
      f4 q0, q1, q2, q3;
      if (LDS) {
        if (need) {
          q0 = ld_l(lds, M.l_pairs + 4u * s.curr + 0u);
          q1 = ld_l(lds, M.l_pairs + 4u * s.curr + 1u);
          q2 = ld_l(lds, M.l_pairs + 4u * s.curr + 2u);
          q3 = ld_l(lds, M.l_pairs + 4u * s.curr + 3u);
        }
      } else {
        pw_fetch_quad(M.gpairs, need, s.curr, q0, q1, q2, q3);
      }
      if (need)
        pw_pair<COUNT, (uint32_t)RT_PW_STACK_K>(s, s.curr, q0.x, q0.y, q0.z, rt_f2u(q0.w), q1.x, q1.y, q1.z, q2.x, q2.y, q2.z,
                                                rt_f2u(q2.w), q3.x, q3.y, q3.z, rt_f2u(q3.w), M.t_min, stk, n_nodes);
    }
  }
}

// Flush the wave's triangle queue when it is due (the LDS work queue of k_traverse.hip.h, per-lane ray kind).  Returns
// false when no lane has anything left to do.
template <bool COUNT, bool LDS>
__device__ __forceinline__ bool pw_flush(const PairMem& M, const f4* lds, const WaveWork& W, PairLane& s, uint32_t& n_tris) {
  const uint32_t lane = threadIdx.x & 63u;
  const bool waiting = s.state == PW_WAIT;
  const unsigned long long smask = __ballot(pw_busy(s) && !waiting);
  const unsigned long long wmask = __ballot(waiting);
  if ((smask | wmask) == 0ull) return false;
  if (wmask == 0ull) return true;
  const uint32_t cnt = waiting ? (s.leaf & 7u) : 0u;
  const unsigned long long b0 = __ballot((cnt & 1u) != 0u), b1 = __ballot((cnt & 2u) != 0u), b2 = __ballot((cnt & 4u) != 0u);
  const uint32_t total = (uint32_t)__builtin_popcountll(b0) + 2u * (uint32_t)__builtin_popcountll(b1) +
                         4u * (uint32_t)__builtin_popcountll(b2);
  // wait for more items only while somebody can still produce them without a flush
  if (total < RT_FLUSH_ITEMS && __ballot(pw_can_step(s) || s.state == PW_ENTER) != 0ull) return true;
  const uint32_t excl =
      __builtin_amdgcn_mbcnt_hi((uint32_t)(b0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b0, 0u)) +
      2u * __builtin_amdgcn_mbcnt_hi((uint32_t)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b1, 0u)) +
      4u * __builtin_amdgcn_mbcnt_hi((uint32_t)(b2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b2, 0u));
  const uint32_t first = s.leaf >> 3;
  if (waiting) {
    f4 ra, rb;
    ra.x = rt_opaque(s.r.o.x); ra.y = rt_opaque(s.r.o.y); ra.z = rt_opaque(s.r.o.z); ra.w = rt_u2f(s.any ? 1u : 0u);
    rb.x = rt_opaque(s.r.d.x); rb.y = rt_opaque(s.r.d.y); rb.z = rt_opaque(s.r.d.z); rb.w = s.closest;
    W.rays[2 * lane] = ra;
    W.rays[2 * lane + 1] = rb;
    W.res[lane] = ~0ull;
    const uint32_t tag = lane << 26;
#pragma unroll
    for (uint32_t i = 0; i < 4u; i++)
      if (i < cnt) W.items[excl + i] = tag | (first + i);
  }
  // leaves of the reference's builder hold <= 4 triangles (blas.rs:99); only its fallback leaves hold 5-7
  if ((b2 & (b0 | b1)) != 0ull) {
    if (waiting) {
      const uint32_t tag = lane << 26;
#pragma unroll
      for (uint32_t i = 4; i < 7u; i++)
        if (i < cnt) W.items[excl + i] = tag | (first + i);
    }
  }
  __builtin_amdgcn_wave_barrier();
  const bool tri_lds = LDS || M.l_tri != RT_LDS_NONE;   // wave-uniform
  for (uint32_t c = 0; c < total; c += 64u) {
    const uint32_t j = c + lane;
    if (j < total) {
      const uint32_t it = W.items[j];
      const uint32_t owner = it >> 26, tri = it & 0x03ffffffu;
      f4 ra = W.rays[2 * owner], rb = W.rays[2 * owner + 1];
      LocalRay q;
      q.o = rt3_make(ra.x, ra.y, ra.z);
      q.d = rt3_make(rb.x, rb.y, rb.z);
      f4 g0, g1, g2;
      if (tri_lds) {
        g0 = ld_l(lds, M.l_tri + (uint32_t)RT_TRI_STRIDE * tri);
        g1 = ld_l(lds, M.l_tri + (uint32_t)RT_TRI_STRIDE * tri + 1u);
        g2 = ld_l(lds, M.l_tri + (uint32_t)RT_TRI_STRIDE * tri + 2u);
      } else {
        g0 = ld_g(M.gtri, RT_TRI_STRIDE * (size_t)tri);
        g1 = ld_g(M.gtri, RT_TRI_STRIDE * (size_t)tri + 1);
        g2 = ld_g(M.gtri, RT_TRI_STRIDE * (size_t)tri + 2);
      }
      float t;
      const bool ok = hit_tri_nb(g0, g1, g2, q, M.t_min, rb.w, t);
      // the reference's leaf loop ends with the minimum over (t, position) of the tests that pass against the bound at
      // leaf entry (k_traverse.hip.h); a shadow ray needs only the first accepted position
      const bool any_ray = rt_f2u(ra.w) != 0u;
      if (ok) atomicMin(&W.res[owner], any_ray ? (unsigned long long)tri : (((unsigned long long)rt_f2u(t) << 32) | tri));
    }
  }
  __builtin_amdgcn_wave_barrier();
  if (waiting) {
    const unsigned long long best = W.res[lane];
    const bool found = best != ~0ull;
    if (COUNT) n_tris += (s.any && found) ? ((uint32_t)best - first + 1u) : cnt;   // the any-hit loop stops at its first hit
    pw_after_leaf(s, found, rt_u2f((uint32_t)(best >> 32)), (uint32_t)best);
  }
  __builtin_amdgcn_wave_barrier();
  return true;
}

// start a ray: the TLAS root comes from its own record (wave-uniform address: scalar loads)
template <bool COUNT>
__device__ __forceinline__ void pw_start(const PairMem& M, PairLane& s, bool active, bool any, uint32_t blas_base, rt3 o, rt3 d,
                                         float t_max, uint32_t& n_nodes) {
  const f4 t0 = ld_g(M.groot, 2 * (size_t)M.n_inst), t1 = ld_g(M.groot, 2 * (size_t)M.n_inst + 1);
  pw_begin<COUNT>(s, active && blas_base != 0u, any, o, d, M.t_min, t_max, t0.x, t0.y, t0.z, rt_f2u(t0.w), t1.x, t1.y, t1.z,
                  n_nodes);
}

// the whole walk of one wave's rays (persistent kernel): every lane brings one ray
template <bool COUNT, bool LDS>
__device__ __forceinline__ void pw_traverse(const PairMem& M, const f4* lds, const WaveWork& W, LdsStack& stk, PairLane& s,
                                            uint32_t& n_nodes, uint32_t& n_tris) {
  for (;;) {
    pw_trip<COUNT, LDS, RT_PW_STEPS_PER_TRIP>(M, lds, stk, s, n_nodes);
    if (!pw_flush<COUNT, LDS>(M, lds, W, s, n_tris)) break;
  }
}

}  // namespace rtk
#endif
