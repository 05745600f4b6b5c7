// k_pairtrav.hip.h — the wave-level side of the child-pair walk: how the 64 lanes of a wave fetch their pair records,
// when they pop / enter instances / flush their triangle queue.  The per-ray decisions are k_pairwalk.hip.h's lane
// functions (checked on the host, tests/test_pairwalk_model.py); nothing here changes WHAT a ray does at a step, only
// WHEN a lane takes it.
// Part of the kernel set of csrc/kernels.hip.h (included from there, in order; not a stand-alone header).
#ifndef MI355RT_K_PAIRTRAV_HIP_H
#define MI355RT_K_PAIRTRAV_HIP_H

namespace rtk {

#ifndef RT_PW_STACK_K
#define RT_PW_STACK_K 7   // deferred right children a lane can hold (8 bytes each in LDS).  Fall-back rate measured on the
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
struct TlasRoot {       // the TLAS root's box and word, by value in the kernel arguments (rt_api.hip fills it at upload)
  float lo[3];
  uint32_t word;
  float hi[3];
  uint32_t pad;
};
struct PairMem {
  TlasRoot troot;
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
  TlasRoot troot;
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
  M.troot = P.troot;
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

// ---- DPP helper (quad = 4 consecutive lanes).  Called from wave-uniform control flow only: a DPP read of a lane that is
// switched off returns 0 (bound_ctrl), not that lane's register.
template <int QP>
__device__ __forceinline__ uint32_t quad_bcast(uint32_t v) {   // lane QP of the quad, to all four
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, QP * 0x55, 0xf, 0xf, true);
}

typedef __attribute__((address_space(1))) const void* rt_gvptr;
typedef __attribute__((address_space(3))) void* rt_lvptr;

// QUAD-COOPERATIVE fetch of 64-byte pair records from global memory, landing in LDS (global_load_lds_dwordx4).
// Instruction k (k = 0..3) serves the rays of the lanes 4q + k: the four lanes of quad q read the four 16-byte chunks of THAT
// ray's record — one 64-byte line per quad and instruction instead of one line per lane — and the instruction's 64 x 16
// bytes land as one contiguous KB in LDS region k, i.e. as 16 whole records; the owner then reads its record back with
// four ds_read_b128.  Measured (tools/gather_peak.hip, rows "4d"): a step costs the texture-address path what ONE 16-byte
// load per lane cost (L1-resident 67-72 CU cycles at 40-64 lanes, L2-resident 92-97 / 149), and no register is shuffled
// (the same fetch through registers, rows "4q", needs 44 selects + DPP moves per step: the first version of this walk was
// VALU-bound on them; the same four loads as plain global_load_dwordx4 + four ds_write_b128 once they are back: 183.2
// against 176.7 ms per 32-frame batch of the 263 k-triangle hall — an LDS-DMA load costs the wave ~175 cycles to issue, but
// sixteen more live registers cost more).  The four regions (1 KB + 16 B of padding each, so that the read-back spreads over all banks) lie
// over the wave's triangle work queue, which is only live inside pw_flush.
#define RT_PW_REGION_SLOTS 65u   // 16-byte slots per region
__device__ __forceinline__ void pw_fetch_dma(const f4* gpairs, f4* wave_lds, unsigned long long need_mask, uint32_t idx) {
  const uint32_t c16 = (threadIdx.x & 3u) * 16u;
  const uint32_t i0 = quad_bcast<0>(idx), i1 = quad_bcast<1>(idx), i2 = quad_bcast<2>(idx), i3 = quad_bcast<3>(idx);
  // lanes of instruction k: every quad whose lane k wants a record (scalar arithmetic on the ballot: bit 4q + k -> bits 4q..4q+3)
  const unsigned long long every4 = 0x1111111111111111ull;
  const unsigned long long m0 = (need_mask & every4) * 15ull, m1 = ((need_mask >> 1) & every4) * 15ull,
                           m2 = ((need_mask >> 2) & every4) * 15ull, m3 = ((need_mask >> 3) & every4) * 15ull;
  const char* base = reinterpret_cast<const char*>(gpairs);
  if (__builtin_amdgcn_inverse_ballot_w64(m0))
    __builtin_amdgcn_global_load_lds((rt_gvptr)(base + (size_t)(i0 * 64u + c16)), (rt_lvptr)(wave_lds + 0u * RT_PW_REGION_SLOTS), 16, 0, 0);
  if (__builtin_amdgcn_inverse_ballot_w64(m1))
    __builtin_amdgcn_global_load_lds((rt_gvptr)(base + (size_t)(i1 * 64u + c16)), (rt_lvptr)(wave_lds + 1u * RT_PW_REGION_SLOTS), 16, 0, 0);
  if (__builtin_amdgcn_inverse_ballot_w64(m2))
    __builtin_amdgcn_global_load_lds((rt_gvptr)(base + (size_t)(i2 * 64u + c16)), (rt_lvptr)(wave_lds + 2u * RT_PW_REGION_SLOTS), 16, 0, 0);
  if (__builtin_amdgcn_inverse_ballot_w64(m3))
    __builtin_amdgcn_global_load_lds((rt_gvptr)(base + (size_t)(i3 * 64u + c16)), (rt_lvptr)(wave_lds + 3u * RT_PW_REGION_SLOTS), 16, 0, 0);
}
__device__ __forceinline__ void pw_fetch_wait() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the records are in LDS (an LDS-DMA load counts in vmcnt)
}

// can this lane still move without the wave's help?
__device__ __forceinline__ bool pw_can_step(const PairLane& s) {
  return s.state == PW_FETCH || s.state == PW_FETCHR || s.state == PW_POP;
}
__device__ __forceinline__ bool pw_busy(const PairLane& s) { return s.state != PW_DONE; }

#ifndef RT_PW_ENTER_BATCH
#define RT_PW_ENTER_BATCH 16u   // lanes that wait for an instance entry before the wave does it (deferred entry, as before)
#endif
#ifndef RT_PW_POP_REPS
#define RT_PW_POP_REPS 1        // pops a lane may take between two record fetches (2: a popped child that fails its re-test, or
                                // the way out of an instance, is followed by the next entry at once — swept 1 / 2 / 3 on MI355X,
                                // ms per 32-frame batch of the 263 k-triangle hall: 184.7 / 187.0 / 189.6)
#endif
#ifndef RT_PW_POP_FIRST
#define RT_PW_POP_FIRST 1       // 1: pops before the round's record request (pw_trip), 0: behind it
#endif

// STEPS record fetches for every lane that wants one, with the cheap transitions (pop, enter an instance) in between.
// LDS = every record is in LDS (plain ds_read, entry on the spot).  wave_lds: this wave's LDS block (the fetch regions lie
// over its triangle work queue).
// Diagnostic build only (-DRT_PW_STAMPS, tools/exp/pw_sections.py): s_memtime cycles a wave spends in the parts of one round
// of pw_trip, each closed by s_waitcnt 0: [0] asking for records, [1] pops, [2] instance entry, [3] waiting for + reading the
// records, [4] slab tests + decision; [5] rounds.  Nothing else reads it; the product build executes no stamp.
#ifdef RT_PW_STAMPS
#define PW_STAMP(k)                                                \
  {                                                                \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");    \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();    \
    pw_cyc[k] += t_ - pw_t;                                        \
    pw_t = t_;                                                     \
  }
#else
#define PW_STAMP(k)
#endif
template <bool COUNT, bool LDS, int STEPS>
__device__ __forceinline__ void pw_trip(const PairMem& M, const f4* lds, f4* wave_lds, LdsStack& stk, PairLane& s, uint32_t& n_nodes
#ifdef RT_PW_STAMPS
                                        , unsigned long long* pw_cyc
#endif
) {
#ifdef RT_PW_STAMPS
  unsigned long long pw_t = __builtin_amdgcn_s_memtime();
#endif
  const bool inst_lds = LDS || M.l_inst != RT_LDS_NONE;   // wave-uniform
  // this lane's record in the fetch regions: region (lane & 3), record (lane >> 2)
  const uint32_t lane = threadIdx.x & 63u;
  const f4* mine = wave_lds + (lane & 3u) * RT_PW_REGION_SLOTS + (lane >> 2) * 4u;
#pragma unroll
  for (int k = 0; k < STEPS; k++) {
    // ---- pops first (RT_PW_POP_FIRST, default): a lane whose two children both missed in the last round takes its next
    //      record off its stack BEFORE the records are asked for, so it tests a pair in EVERY round; with the pops behind the
    //      request (0: they overlap the loads' flight) such a lane tested a pair every second round — half the lanes of
    //      the slab tests idle (rocprof: 32 of 64 lanes per vector instruction)
#if RT_PW_POP_FIRST
    // ---- transitions that need no record
#pragma unroll
    for (int rep = 0; rep < RT_PW_POP_REPS; rep++) {
      if (__ballot(s.state == PW_POP) != 0ull) {
        if (s.state == PW_POP) pw_pop<COUNT>(s, stk, n_nodes);
      }
    }
#ifdef RT_PW_STAMPS
    { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); pw_cyc[1] += t_ - pw_t; pw_t = t_; }
#endif
#endif
    // ---- the lanes that know their next record ask for it first (global memory: the LDS-DMA loads are in flight while
    //      the other lanes pop / enter below; those lanes fetch in the next round)
    bool need = s.state == PW_FETCH || s.state == PW_FETCHR;
    unsigned long long need_mask = __ballot(need);
    if (!LDS && need_mask != 0ull) pw_fetch_dma(M.gpairs, wave_lds, need_mask, s.curr);
#ifdef RT_PW_STAMPS
    { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); pw_cyc[0] += t_ - pw_t; pw_t = t_; pw_cyc[5]++; }   // no wait here: the loads stay in flight
#endif
#if !RT_PW_POP_FIRST
    // ---- transitions that need no record
#pragma unroll
    for (int rep = 0; rep < RT_PW_POP_REPS; rep++) {
      if (__ballot(s.state == PW_POP) != 0ull) {
        if (s.state == PW_POP) pw_pop<COUNT>(s, stk, n_nodes);
      }
    }
#ifdef RT_PW_STAMPS
    { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); pw_cyc[1] += t_ - pw_t; pw_t = t_; }
#endif
#endif
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
          pw_enter<COUNT, (uint32_t)RT_PW_STACK_K>(s, r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w, b0.x,
                                                   b0.y, b0.z, rt_f2u(b0.w), b1.x, b1.y, b1.z, M.t_min, stk, n_nodes);
        }
      }
    }
    PW_STAMP(2)
    // ---- the records are there: test both children, decide (records in LDS: the lanes that have just popped come along)
    if (LDS) {
      need = s.state == PW_FETCH || s.state == PW_FETCHR;
      need_mask = __ballot(need);
    }
    if (need_mask != 0ull) {
      f4 q0, q1, q2, q3;
      if (LDS) {
        if (need) {
          q0 = ld_l(lds, M.l_pairs + 4u * s.curr + 0u);
          q1 = ld_l(lds, M.l_pairs + 4u * s.curr + 1u);
          q2 = ld_l(lds, M.l_pairs + 4u * s.curr + 2u);
          q3 = ld_l(lds, M.l_pairs + 4u * s.curr + 3u);
        }
      } else {
        pw_fetch_wait();
        if (need) {
          q0 = ld_l(mine, 0u);
          q1 = ld_l(mine, 1u);
          q2 = ld_l(mine, 2u);
          q3 = ld_l(mine, 3u);
        }
      }
      PW_STAMP(3)
      if (need)
        pw_pair<COUNT, (uint32_t)RT_PW_STACK_K>(s, s.curr, q0.x, q0.y, q0.z, rt_f2u(q0.w), q1.x, q1.y, q1.z, q2.x, q2.y, q2.z,
                                                rt_f2u(q2.w), q3.x, q3.y, q3.z, rt_f2u(q3.w), M.t_min, stk, n_nodes);
    }
    PW_STAMP(4)
  }
}

// Flush the wave's triangle queue when it is due (the LDS work queue of k_traverse.hip.h, per-lane ray kind).  Returns
// false when no lane has anything left to do.
template <bool COUNT, bool LDS>
__device__ __forceinline__ bool pw_flush(const PairMem& M, const f4* lds, const WaveWork& W, PairLane& s, uint32_t& n_tris) {
  const uint32_t lane = threadIdx.x & 63u;
  const bool waiting = s.state == PW_WAIT;
  const unsigned long long wmask = __builtin_amdgcn_ballot_w64(waiting);
  if (wmask == 0ull) return __builtin_amdgcn_ballot_w64(pw_busy(s)) != 0ull;
  // due when RT_FLUSH_LANES lanes wait (k_traverse.hip.h), or nobody can produce more items without a flush
  const unsigned long long pmask = __builtin_amdgcn_ballot_w64(pw_can_step(s) || s.state == PW_ENTER);
  const uint32_t n_wait = pmask != 0ull ? (uint32_t)__builtin_popcountll(wmask) : 64u;
  if (n_wait < RT_FLUSH_LANES) return true;
  const uint32_t cnt = waiting ? (s.leaf & 7u) : 0u;
  const unsigned long long b0 = __builtin_amdgcn_ballot_w64((cnt & 1u) != 0u), b1 = __builtin_amdgcn_ballot_w64((cnt & 2u) != 0u),
                           b2 = __builtin_amdgcn_ballot_w64((cnt & 4u) != 0u);
  const uint32_t total = (uint32_t)__builtin_popcountll(b0) + 2u * (uint32_t)__builtin_popcountll(b1) +
                         4u * (uint32_t)__builtin_popcountll(b2);
  const uint32_t excl =
      __builtin_amdgcn_mbcnt_hi((uint32_t)(b0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b0, 0u)) +
      2u * __builtin_amdgcn_mbcnt_hi((uint32_t)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b1, 0u)) +
      4u * __builtin_amdgcn_mbcnt_hi((uint32_t)(b2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b2, 0u));
  const uint32_t first = s.leaf >> 3;
  if (waiting) {
    f4 ra, rb;
    ra.x = rt_opaque(s.r.o.x); ra.y = rt_opaque(s.r.o.y); ra.z = rt_opaque(s.r.o.z); ra.w = rt_u2f(s.flags & PW_F_ANY);
    rb.x = rt_opaque(s.r.d.x); rb.y = rt_opaque(s.r.d.y); rb.z = rt_opaque(s.r.d.z); rb.w = s.closest;
    W.rays[2 * lane] = ra;
    W.rays[2 * lane + 1] = rb;
    W.res[lane] = ~0ull;
    // four unconditional ordered stores, the highest slot first (trav_flush, k_traverse.hip.h, has the argument)
    const rt_lptr32_ordered it = (rt_lptr32_ordered)(W.items + excl);
    const uint32_t word = (lane << 26) | first;
    it[3] = word + 3u;
    it[2] = word + 2u;
    it[1] = word + 1u;
    it[0] = word;
  }
  // leaves of the reference's builder hold <= 4 triangles (blas.rs:99); only its fallback leaves hold 5-7
  if ((b2 & (b0 | b1)) != 0ull) {
    if (waiting) {
      const rt_lptr32_ordered it = (rt_lptr32_ordered)(W.items + excl);
      const uint32_t word = (lane << 26) | first;
#pragma unroll
      for (uint32_t i = 4; i < 7u; i++)
        if (i < cnt) it[i] = word + i;
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
    if (COUNT) n_tris += (pw_flag(s, PW_F_ANY) && found) ? ((uint32_t)best - first + 1u) : cnt;   // the any-hit loop stops at its first hit
    pw_after_leaf(s, found, rt_u2f((uint32_t)(best >> 32)), (uint32_t)best);
  }
  __builtin_amdgcn_wave_barrier();
  return true;
}

// start a ray: the TLAS root comes from the kernel arguments (scalar registers)
template <bool COUNT>
__device__ __forceinline__ void pw_start(const PairMem& M, PairLane& s, bool active, bool any, uint32_t blas_base, rt3 o, rt3 d,
                                         float t_max, uint32_t& n_nodes) {
  pw_begin<COUNT>(s, active && blas_base != 0u, any, o, d, M.t_min, t_max, M.troot.lo[0], M.troot.lo[1], M.troot.lo[2],
                  M.troot.word, M.troot.hi[0], M.troot.hi[1], M.troot.hi[2], n_nodes);
}

}  // namespace rtk
#endif
