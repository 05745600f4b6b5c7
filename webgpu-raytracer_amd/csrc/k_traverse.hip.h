// k_traverse.hip.h — the wave-level TLAS / BLAS walk shared by the persistent kernel (traverse()) and the wavefront
// trace kernel (k_wf_trace): ONE copy of the node step and of the LDS triangle queue (Raytracer.wgsl:455-600).
// Part of the kernel set of csrc/kernels.hip.h (included from there, in order; not a stand-alone header).
#ifndef MI355RT_K_TRAVERSE_HIP_H
#define MI355RT_K_TRAVERSE_HIP_H

namespace rtk {

typedef float f4 __attribute__((ext_vector_type(4)));

// Loads with the address space spelled out.  HIP pointers are generic; when the same value can come from LDS or from
// global memory the compiler otherwise merges the two loads into ONE flat_load behind a pointer select — a flat access
// pays the address-space check per lane and waits on both vmcnt and lgkmcnt.
typedef const f4 __attribute__((address_space(1))) * rt_gptr;
typedef const f4 __attribute__((address_space(3))) * rt_lptr;
typedef const uint32_t __attribute__((address_space(1))) * rt_gptr32;
typedef const uint32_t __attribute__((address_space(3))) * rt_lptr32;
__device__ __forceinline__ f4 ld_g(const f4* p, size_t i) { return ((rt_gptr)p)[i]; }
__device__ __forceinline__ f4 ld_l(const f4* p, uint32_t i) { return ((rt_lptr)p)[i]; }
__device__ __forceinline__ uint32_t ld_g32(const uint32_t* p, size_t i) { return ((rt_gptr32)p)[i]; }
__device__ __forceinline__ uint32_t ld_l32(const void* p, uint32_t i) { return ((rt_lptr32)p)[i]; }
typedef float rt_f3_16 __attribute__((ext_vector_type(3)));   // 12 bytes, 16-byte aligned: one ds_read_b96
typedef volatile uint32_t __attribute__((address_space(3))) * rt_lptr32_ordered;   // LDS stores that keep their program order
// opaque copy of a register value: keeps the optimiser from re-reading adjacent struct fields as one vector load from
// a stack slot (which forced the ray of every lane through scratch memory)
__device__ __forceinline__ float rt_opaque(float v) {
  asm volatile("" : "+v"(v));
  return v;
}

// stage `slots` 16-byte records from global memory into LDS at `dst` (all threads of the workgroup; no barrier)
__device__ __forceinline__ void lds_stage(f4* dst, const void* src, size_t slots) {
  const f4* g = reinterpret_cast<const f4*>(src);
  for (uint32_t i = threadIdx.x; i < slots; i += blockDim.x) dst[i] = g[i];
}

// Where the traversal records live.  `tnodes` (k_treelet.hip.h) has its first k_lds nodes staged in LDS at slot l_nodes
// of the workgroup's dynamic LDS array (by default all of them or none, rt_api.hip plan_lds); the triangle records, the
// instance rows and the instance BLAS roots are staged as a whole when they fit (l_* != RT_LDS_NONE), else read through
// L1 / L2.  Slots are 16-byte units.
#define RT_LDS_NONE 0xffffffffu
struct TravMem {
  const f4* gnodes;           // tnodes, 2 per node
  const f4* gtri;             // tri_geom, 3 per triangle
  const f4* ginst;            // inst_trav, 4 per instance
  const uint32_t* groot;      // inst_root, 1 per instance: index in tnodes of the instance's BLAS root
  uint32_t k_lds;             // nodes [0, k_lds) are read from LDS
  uint32_t l_nodes, l_tri, l_inst, l_root;
};

// MODE_LDS: every record is in LDS (the whole scene fits: k_lds >= n_nodes and all l_* set) — the compiler sees plain
// ds_read.  MODE_MIXED: per-access choice (a lane-level compare for nodes, wave-uniform flags for the rest).
// RT_TRAV_MIXED_RAYREG: mixed mode that keeps the instance-space origin / direction in registers and posts them with every
// flush (trav_post_at_entry below).
enum { RT_TRAV_LDS = 1, RT_TRAV_MIXED = 2, RT_TRAV_MIXED_RAYREG = 3 };

template <int MODE>
__device__ __forceinline__ void trav_fetch_node(const TravMem& M, const f4* lds, uint32_t idx, f4& lo, f4& hi) {
  if (MODE == RT_TRAV_LDS || idx < M.k_lds) {
    lo = ld_l(lds, M.l_nodes + 2u * idx);
    hi = ld_l(lds, M.l_nodes + 2u * idx + 1u);
  } else {
    lo = ld_g(M.gnodes, 2 * (size_t)idx);
    hi = ld_g(M.gnodes, 2 * (size_t)idx + 1);
  }
}

__device__ __forceinline__ bool hit_box4(f4 lo, f4 hi, rt3 inv_d, rt3 o_inv_d, float t_min, float t_max) {
  float t1x = lo.x * inv_d.x - o_inv_d.x, t2x = hi.x * inv_d.x - o_inv_d.x;
  float t1y = lo.y * inv_d.y - o_inv_d.y, t2y = hi.y * inv_d.y - o_inv_d.y;
  float t1z = lo.z * inv_d.z - o_inv_d.z, t2z = hi.z * inv_d.z - o_inv_d.z;
  float nx = rt_min(t1x, t2x), ny = rt_min(t1y, t2y), nz = rt_min(t1z, t2z);
  float fx = rt_max(t1x, t2x), fy = rt_max(t1y, t2y), fz = rt_max(t1z, t2z);
  float tm_near = rt_max(t_min, rt_max(nx, rt_max(ny, nz)));
  float tm_far = rt_min(t_max, rt_min(fx, rt_min(fy, fz)));
  return tm_near <= tm_far;
}

// instance entry: object-space ray + the BLAS root (Raytracer.wgsl:507-512)
template <int MODE>
__device__ __forceinline__ LocalRay to_instance(const TravMem& M, const f4* lds, uint32_t inst, rt3 o, rt3 d,
                                                uint32_t& blas_root) {
  f4 r0, r1, r2;
  if (MODE == RT_TRAV_LDS || M.l_inst != RT_LDS_NONE) {
    r0 = ld_l(lds, M.l_inst + 4u * inst + 0u);
    r1 = ld_l(lds, M.l_inst + 4u * inst + 1u);
    r2 = ld_l(lds, M.l_inst + 4u * inst + 2u);
    blas_root = ld_l32(lds + M.l_root, inst);
  } else {
    r0 = ld_g(M.ginst, 4 * (size_t)inst + 0);
    r1 = ld_g(M.ginst, 4 * (size_t)inst + 1);
    r2 = ld_g(M.ginst, 4 * (size_t)inst + 2);
    blas_root = ld_g32(M.groot, inst);
  }
  rt3 lo = rt3_make(r0.x * o.x + r0.y * o.y + r0.z * o.z + r0.w * 1.0f, r1.x * o.x + r1.y * o.y + r1.z * o.z + r1.w * 1.0f,
                    r2.x * o.x + r2.y * o.y + r2.z * o.z + r2.w * 1.0f);
  rt3 ld = rt3_make(r0.x * d.x + r0.y * d.y + r0.z * d.z + r0.w * 0.0f, r1.x * d.x + r1.y * d.y + r1.z * d.z + r1.w * 0.0f,
                    r2.x * d.x + r2.y * d.y + r2.z * d.z + r2.w * 0.0f);
  return make_ray(lo, ld);
}

// Branch-free Möller–Trumbore: same operations and the same accept/reject truth table as
// hit_triangle_raw (Raytracer.wgsl:443-453), evaluated without early exits so that a wave testing
// 64 different triangles stays converged.
__device__ __forceinline__ bool hit_tri_nb(f4 g0, f4 g1, f4 g2, const LocalRay& r, float t_min, float t_max, float& t_out) {
  rt3 v0 = rt3_make(g0.x, g0.y, g0.z), e1 = rt3_make(g1.x, g1.y, g1.z), e2 = rt3_make(g2.x, g2.y, g2.z);
  rt3 h = rt_cross(r.d, e2);
  float a = rt_dot(e1, h);
  float f = rt_rcp(a);
  rt3 s = r.o - v0;
  float u = f * rt_dot(s, h);
  rt3 q = rt_cross(s, e1);
  float v = f * rt_dot(r.d, q);
  float t = f * rt_dot(e2, q);
  t_out = t;
  bool reject = (rt_abs(a) < 1e-6f) | (u < 0.0f) | (u > 1.0f) | (v < 0.0f) | (u + v > 1.0f);
  return !reject & (t > t_min) & (t < t_max);
}

// ---------------------------------------------------------------------------------------------
// One walk over TLAS and BLAS nodes for the 64 rays of a wave.
//
// Divergence control.  A lane is SEARCHING (walking nodes: slab tests, instance entry/exit) or
// WAITING (it reached a BLAS leaf whose box it hits and has queued that leaf's triangles).  Every
// trip lets all searching lanes take ONE node step (trav_step).  When enough triangle tests are queued
// (RT_FLUSH_ITEMS) or nobody is searching any more, the wave flushes the queue (trav_flush):
//   * (lane, triangle) work items are compacted into LDS with a ballot/mbcnt prefix sum over the
//     3-bit leaf counts, each owner also posts its instance-space ray;
//   * the items are tested 64 at a time, one item per lane, whatever lane they came from — a leaf
//     with 6 triangles no longer holds 63 other lanes hostage — each against its owner's bound at leaf entry;
//   * every accepted test does one LDS atomicMin of (bits(t) << 32 | triangle) into its owner's slot, the owner
//     reads the winner and goes back to searching.
// Equivalence with the reference's sequential leaf loop (Raytracer.wgsl:474-482): a test is accepted
// there iff geometry passes, t > t_min and t < the running closest; the running closest never exceeds
// the closest at leaf entry and only ever takes the value of an accepted t, so the loop ends with the SMALLEST
// accepted t and, among equal ones, the one met first (strict <) — the minimum of (t, position in the leaf) over the
// tests that pass against the leaf-entry bound, which is order-independent.  t > 0, so its bits order like its value.
// Per lane the sequence of visited nodes, tested triangles and tie-breaks is the reference's: the walk follows the
// explicit successors of tnodes, which name the same nodes as `curr + 1` / `node_start + skip` do in the bridge array.
// ANY = shadow ray (first accepted hit ends the ray), else closest hit.
struct WaveWork {
  f4* rays;                  // 64 x 2: {o.xyz, bound at leaf entry} {d.xyz, -} of the lane's instance-space ray
  uint32_t* items;           // up to 64*7: (owner lane << 26) | triangle id
  unsigned long long* res;   // 64: per owner lane, the smallest (bits(t) << 32 | triangle id) among its accepted tests
};
#define RT_WORK_BYTES_PER_WAVE (64 * 32 + 64 * 7 * 4 + 64 * 8)
__device__ __forceinline__ void wave_work_at(WaveWork& W, char* wbase) {
  W.rays = reinterpret_cast<f4*>(wbase);
  W.items = reinterpret_cast<uint32_t*>(wbase + 64 * 32);
  W.res = reinterpret_cast<unsigned long long*>(wbase + 64 * 32 + 64 * 7 * 4);
}
#ifndef RT_STEPS_PER_TRIP
#define RT_STEPS_PER_TRIP 4  // swept on MI355X (Cornell, ms per 32-frame launch): 1: 27.1, 2: 26.0, 3: 25.8, 4: 25.1, 6: 25.0, 8: 25.8, 12: 27.4
#endif
#ifndef RT_FLUSH_ITEMS
#define RT_FLUSH_ITEMS 24u  // queued triangle tests that trigger a flush; swept 1..128 on MI355X: flat optimum 16..32
                            // (fewer = partial 64-item chunks, more = lanes wait longer for their results)
#endif

// Diagnostic build only (-DRT_LANE_STATS, tools/lane_stats.py): how many lanes are active each time a wave executes one of
// the parts of a trip — g_lane_stats[2 k] = times part k ran, [2 k + 1] = active lanes summed.  Parts: 0 shade, 1 / 2 node step
// and triangle chunk of the shadow walk, 3 / 4 the same of the extension walk, 5 surface frame of the new hit, 6 start of a
// sample (camera ray + G-buffer surface), 7 end of a sample.  Nothing else reads the array; in the product build RT_LSTAT is empty.
__device__ unsigned long long g_lane_stats[32];
#ifdef RT_LANE_STATS
#define RT_LSTAT(k, cond)                                                                                  \
  do {                                                                                                     \
    const unsigned long long m_ = __ballot(cond);                                                          \
    if (m_ != 0ull && (threadIdx.x & 63u) == 0u) {                                                         \
      atomicAdd(&g_lane_stats[2 * (k)], 1ull);                                                             \
      atomicAdd(&g_lane_stats[2 * (k) + 1], (unsigned long long)__builtin_popcountll(m_));                 \
    }                                                                                                      \
  } while (0)
#else
#define RT_LSTAT(k, cond) ((void)0)
#endif

// One ray's traversal state.  Everything a lane needs to know about itself is a 32-bit word in a VGPR and every question the
// walk asks ("does this lane step?", "is it at the end of its array?", "does it wait for its leaf?") is ONE compare of such a
// word, which the hardware answers as a lane mask in SGPRs.  (Round 4: until then the state was a set of bools — searching,
// waiting, in_blas, entering, any.  The compiler keeps a bool that is live across divergent control flow as a lane mask and
// merges it at every join with three scalar instructions, or as a 0 / 1 VGPR that each use turns back into a mask with two
// more; a node step was 63 instructions of which 24 are the slab test.)
//   curr       the node to test next, or RT_CURR_END (the successor field of a node that ends its array), RT_CURR_ENTER (mixed
//              mode: the lane hit a TLAS leaf and waits for its instance transform), RT_CURR_IDLE (the lane does not search:
//              it waits for the triangles of a leaf, has finished, or never had a ray)
//   leaf       0, or the leaf word (first triangle << 3 | count, count >= 1) of the BLAS leaf whose triangles the lane waits for
//   resume     while waiting: the node to go on with
//   tlas_next  RT_TLAS_NONE while the lane walks the TLAS; inside an instance the successor of the TLAS leaf (a node or
//              RT_NODE_END)
//   best_tri   -1 until a hit is recorded; an any-hit ray records its first accepted triangle, which is all `occluded` needs
#define RT_CURR_END RT_NODE_END
#define RT_CURR_IDLE 0xfffffffeu
#define RT_CURR_ENTER 0xfffffffdu
#define RT_TLAS_NONE 0xfffffffeu
struct Trav {
  rt3 inv_d, o_inv_d;        // the slab-test form of the ray in the space it is currently walking (world or instance); the
                             // origin and direction of an instance-space ray live in LDS only (W.rays, posted at instance
                             // entry): the triangle tests read them there, the node steps never need them
  LocalRay rw;               // the world-space ray (rw.o, rw.d = the ray as given)
  rt3 io, id;                // trav_post_at_entry() == false only: origin and direction of the instance-space ray
  float closest;             // t_min is the constant RT_T_MIN for every ray of the reference (Raytracer.wgsl:6,688,732)
  int32_t best_tri, best_inst;
  uint32_t curr, tlas_next, cur_inst, leaf, resume;
#ifdef RT_LANE_STATS
  uint32_t stat_kind;        // 1 = shadow walk, 3 = extension walk (RT_LSTAT part of its node steps; + 1 = its triangle chunks)
#endif
};
__device__ __forceinline__ bool trav_stepping(const Trav& s) { return s.curr < RT_CURR_ENTER; }   // takes node steps
__device__ __forceinline__ bool trav_searching(const Trav& s) { return s.curr != RT_CURR_IDLE; }  // ... or is about to
__device__ __forceinline__ bool trav_waiting(const Trav& s) { return s.leaf != 0u; }
__device__ __forceinline__ bool trav_entering(const Trav& s) { return s.curr == RT_CURR_ENTER; }
__device__ __forceinline__ bool trav_busy(const Trav& s) { return (s.curr != RT_CURR_IDLE) | (s.leaf != 0u); }
__device__ __forceinline__ bool trav_any(const Trav& s) { return s.best_tri != -1; }

__device__ __forceinline__ void trav_begin(Trav& s, bool active, uint32_t blas_base, rt3 o, rt3 d, float t_max) {
  s.rw = make_ray(o, d);
  s.inv_d = s.rw.inv_d;
  s.o_inv_d = s.rw.o_inv_d;
  s.io = o;                  // (never read before an instance entry sets them)
  s.id = d;
  s.closest = t_max;
  s.best_tri = -1;
  s.best_inst = -1;
  s.curr = (active && blas_base != 0u) ? 0u : RT_CURR_IDLE;   // the TLAS root is node 0 of tnodes
  s.tlas_next = RT_TLAS_NONE;
  s.cur_inst = 0u;
  s.leaf = 0u;
  s.resume = 0u;
#ifdef RT_LANE_STATS
  s.stat_kind = 1u;
#endif
}

// the walk ran off its array: leave the instance (back to the world-space ray and the TLAS cursor), or finish.  Behind a
// wave-uniform test (one compare and a branch per step when nobody is at an end).
__device__ __forceinline__ void trav_leave(Trav& s) {
  const bool at_end = s.curr == RT_CURR_END;
  if (__builtin_amdgcn_ballot_w64(at_end) != 0ull) {
    const bool leave = at_end & (s.tlas_next < RT_TLAS_NONE);   // inside an instance whose TLAS leaf has a successor
    // the six selects that restore the ray sit behind a second wave-uniform test (as a per-lane branch the compiler
    // predicates them: nine moves whenever some lane is at an end, which in a one-instance scene is never a way back)
    if (__builtin_amdgcn_ballot_w64(leave) != 0ull) {
      s.inv_d.x = leave ? s.rw.inv_d.x : s.inv_d.x; s.inv_d.y = leave ? s.rw.inv_d.y : s.inv_d.y;
      s.inv_d.z = leave ? s.rw.inv_d.z : s.inv_d.z;
      s.o_inv_d.x = leave ? s.rw.o_inv_d.x : s.o_inv_d.x; s.o_inv_d.y = leave ? s.rw.o_inv_d.y : s.o_inv_d.y;
      s.o_inv_d.z = leave ? s.rw.o_inv_d.z : s.o_inv_d.z;
    }
    s.curr = leave ? s.tlas_next : (at_end ? RT_CURR_IDLE : s.curr);
    s.tlas_next = leave ? RT_TLAS_NONE : s.tlas_next;
  }
}

// Where the origin and direction of the instance-space ray go: to the lane's slot of W.rays once, at instance entry (the walk
// then carries six registers less and a flush writes 4 bytes per waiting lane instead of 32), or with every flush.
// Measured on MI355X (ms per 32 frames, at entry / with every flush): Cornell (LDS mode) 21.1 / 21.3; in the trace kernels
// instanced x1000 72.4 / 75.7 but glass blob 4K 306.2 / 292.1 — at 5, 6, 7 or 8 resident workgroups per CU alike, and with
// the flush itself 40 % cheaper under section stamps (tools/exp/ab_sections.sh); the cause is not established.  So both forms
// are compiled for the trace kernels and the host picks by scene (rt_api.hip: many nodes per instance = registers).
template <int MODE>
__device__ __forceinline__ constexpr bool trav_post_at_entry() { return MODE != RT_TRAV_MIXED_RAYREG; }
template <int MODE>
__device__ __forceinline__ uint32_t trav_into_instance(const TravMem& M, const f4* lds, const WaveWork& W, Trav& s) {
  uint32_t root;
  const LocalRay q = to_instance<MODE>(M, lds, s.cur_inst, s.rw.o, s.rw.d, root);
  s.inv_d = q.inv_d;
  s.o_inv_d = q.o_inv_d;
  if (trav_post_at_entry<MODE>()) {
    const uint32_t lane = threadIdx.x & 63u;
    f4 ra, rb;
    ra.x = q.o.x; ra.y = q.o.y; ra.z = q.o.z; ra.w = 0.0f;   // .w: the bound at leaf entry, written by every flush
    rb.x = q.d.x; rb.y = q.d.y; rb.z = q.d.z; rb.w = 0.0f;
    W.rays[2 * lane] = ra;
    W.rays[2 * lane + 1] = rb;
  } else {
    s.io = q.o;
    s.id = q.d;
  }
  return root;
}

// instance entry of the lanes that hit a TLAS leaf (deferred form)
template <int MODE>
__device__ __forceinline__ void trav_enter(const TravMem& M, const f4* lds, const WaveWork& W, Trav& s) {
  if (trav_entering(s)) s.curr = trav_into_instance<MODE>(M, lds, W, s);
}

// what a lane does with the node record it fetched (Raytracer.wgsl:462-473, 498-518): slab test against the current
// bound, successor, leaf bookkeeping.  DEFER: a TLAS-leaf hit only parks the lane as RT_CURR_ENTER (trav_enter does the
// transform later, for many lanes at once); otherwise the instance is entered on the spot.  Only stepping lanes come here,
// and a stepping lane has leaf == 0.
template <bool COUNT, int MODE, bool DEFER>
__device__ __forceinline__ void trav_node(const TravMem& M, const f4* lds, const WaveWork& W, Trav& s, f4 lo, f4 hi,
                                          uint32_t& n_nodes) {
  if (COUNT) n_nodes++;
  const uint32_t data = rt_f2u(hi.w);
#ifndef RT_NODE_BOOLS
  // The three facts about the lane (box hit, inner node, walking the TLAS) as lane masks, combined by scalar and / andn2 and
  // handed back as conditions (inverse ballot: no instruction).  Written as bools the compiler computes `!inner` with a
  // second vector compare and `!in_tlas` with a scalar xor: two instructions more per step.
  const unsigned long long hm = __builtin_amdgcn_ballot_w64(hit_box4(lo, hi, s.inv_d, s.o_inv_d, RT_T_MIN, s.closest));
  const unsigned long long im = __builtin_amdgcn_ballot_w64((int32_t)data < 0);           // RT_NODE_INNER is the sign bit
  const unsigned long long tm = __builtin_amdgcn_ballot_w64(s.tlas_next == RT_TLAS_NONE);
  const unsigned long long lm = hm & ~im;                                                  // a leaf whose box is hit
  const bool hit_inner = __builtin_amdgcn_inverse_ballot_w64(hm & im);
  const bool tlas_leaf = __builtin_amdgcn_inverse_ballot_w64(lm & tm);
  const bool got_leaf = __builtin_amdgcn_inverse_ballot_w64(lm & ~tm);
#else
  const bool hit = hit_box4(lo, hi, s.inv_d, s.o_inv_d, RT_T_MIN, s.closest);
  const bool inner = (data & RT_NODE_INNER) != 0u;
  const bool leafhit = hit & !inner;
  const bool hit_inner = hit & inner;
  const bool in_tlas = s.tlas_next == RT_TLAS_NONE;
  const bool got_leaf = leafhit & !in_tlas;
  const bool tlas_leaf = leafhit & in_tlas;
#endif
  uint32_t next = hit_inner ? (data & ~RT_NODE_INNER) : rt_f2u(lo.w);
  if (tlas_leaf) {  // TLAS leaf: enter the instance
    s.cur_inst = data >> 3;
    s.tlas_next = next;
    if (DEFER) {
      next = RT_CURR_ENTER;
    } else {
      next = trav_into_instance<MODE>(M, lds, W, s);
    }
  }
  s.leaf = got_leaf ? data : 0u;
  s.resume = got_leaf ? next : s.resume;
  s.curr = got_leaf ? RT_CURR_IDLE : next;
}

// LDS mode: lanes that ran off their array are dealt with once per trip (0) or before every step (1).  A lane at an end does
// not step anyway (RT_CURR_END is no node); looking for such lanes before every step cost 7 of the step's 49 instructions —
// in most steps of a walk some lane is just ending.  Once per trip a lane that leaves an instance mid-trip idles for up to
// STEPS - 1 steps; per ray the sequence of nodes and bounds is the same either way.
#ifndef RT_LEAVE_PER_STEP
#define RT_LEAVE_PER_STEP 0
#endif
// one node step for every stepping lane; select-based, two branches only
template <bool COUNT, int MODE>
__device__ __forceinline__ void trav_step(const TravMem& M, const f4* lds, const WaveWork& W, Trav& s, uint32_t& n_nodes) {
#if RT_LEAVE_PER_STEP
  trav_leave(s);
#endif
#ifdef RT_LANE_STATS
  RT_LSTAT(s.stat_kind, trav_stepping(s));
#endif
  if (trav_stepping(s)) {
    f4 lo, hi;
    trav_fetch_node<MODE>(M, lds, s.curr, lo, hi);
    trav_node<COUNT, MODE, false>(M, lds, W, s, lo, hi, n_nodes);
  }
}

// ---- mixed mode (part of the records in LDS, the rest behind the vector L1): phased trip.
// Measured on MI355X (tools/gather_peak.hip): a wave-level global_load_dwordx4 with lane-divergent addresses costs the
// CU's texture-address path 16 cycles + 0.35 per active lane whether it hits the L1 or not (30 at 40 lanes; a 32-byte
// node is two of them), a divergent ds_read_b128 6-9 — and the trace kernels spend 70 CU cycles per wave node step.  A
// step that lets every lane fetch "from wherever its node lives" therefore pays the full global instruction for the few
// lanes that need it.  The phased trip makes global instructions rare and full:
//   * the lanes whose node is global issue their two loads first;
//   * while those are in flight, the lanes whose node is in LDS take up to RT_LDS_SUBSTEPS steps of their own (each of
//     them has then either reached a global node, a leaf, an instance or its end);
//   * the global lanes finish their step;
//   * a lane that hit a TLAS leaf does not transform its ray on the spot (three more divergent loads for a handful of
//     lanes when the instance rows are not LDS-resident): it waits as `entering` until RT_ENTER_BATCH lanes do, or
//     nobody else can step.
// Per lane the sequence of nodes, tests and bounds is untouched — only WHEN a lane takes its next step changes.
#ifndef RT_LDS_SUBSTEPS
#define RT_LDS_SUBSTEPS 4
#endif
#ifndef RT_ENTER_BATCH
#define RT_ENTER_BATCH 16u
#endif
template <bool COUNT, int MODE, int ROUNDS>
__device__ __forceinline__ void trav_trip_mixed(const TravMem& M, const f4* lds, const WaveWork& W, Trav& s, uint32_t& n_nodes) {
  const bool inst_lds = M.l_inst != RT_LDS_NONE;   // wave-uniform
#pragma unroll
  for (int k = 0; k < ROUNDS; k++) {
    trav_leave(s);
    {
      const unsigned long long em = __builtin_amdgcn_ballot_w64(trav_entering(s));
      if (em != 0ull && (inst_lds || (uint32_t)__builtin_popcountll(em) >= RT_ENTER_BATCH ||
                         __builtin_amdgcn_ballot_w64(trav_stepping(s)) == 0ull))   // after trav_leave no lane is at an end
        trav_enter<MODE>(M, lds, W, s);
    }
    const bool g = trav_stepping(s) & (s.curr >= M.k_lds);
    f4 glo, ghi;
    if (g) {
      glo = ld_g(M.gnodes, 2 * (size_t)s.curr);
      ghi = ld_g(M.gnodes, 2 * (size_t)s.curr + 1);
    }
    if (M.k_lds != 0u) {
#pragma unroll 1
      for (int j = 0; j < RT_LDS_SUBSTEPS; j++) {
        if (inst_lds && __builtin_amdgcn_ballot_w64(trav_entering(s)) != 0ull) trav_enter<MODE>(M, lds, W, s);
        const bool l = s.curr < M.k_lds;   // no special value of curr is < k_lds, and a lane with a load in flight is at a global node
        if (__builtin_amdgcn_ballot_w64(l) == 0ull) break;
        if (l) {
          const f4 lo = ld_l(lds, M.l_nodes + 2u * s.curr), hi = ld_l(lds, M.l_nodes + 2u * s.curr + 1u);
          trav_node<COUNT, MODE, true>(M, lds, W, s, lo, hi, n_nodes);
        }
        trav_leave(s);   // a lane that ran off its array; the lanes with a load in flight are not at an end
      }
    }
    if (g) trav_node<COUNT, MODE, true>(M, lds, W, s, glo, ghi, n_nodes);
  }
}

// the node steps between two looks at the queues
template <bool COUNT, int MODE, int STEPS>
__device__ __forceinline__ void trav_trip(const TravMem& M, const f4* lds, const WaveWork& W, Trav& s, uint32_t& n_nodes) {
  if (MODE == RT_TRAV_LDS) {
#pragma unroll
    for (int k = 0; k < STEPS; k++) trav_step<COUNT, MODE>(M, lds, W, s, n_nodes);
#if !RT_LEAVE_PER_STEP
    trav_leave(s);   // before the look at the queues, which must see a finished lane as idle
#endif
  } else {
    trav_trip_mixed<COUNT, MODE, STEPS>(M, lds, W, s, n_nodes);
  }
}

// Flush the wave's triangle queue when it is due.  Returns false when no lane is searching or waiting any more
// (the walk of every ray of the wave is over) — `idle_ok` callers (k_wf_trace) ignore that and refill instead.
// Due: RT_FLUSH_LANES lanes wait at a leaf (a leaf of the reference's builder holds 1-4 triangles, 3.3 on average in Cornell:
// the round-2 threshold of 24 queued tests, for a third of the instructions of the look), or nobody can step any more.
#ifndef RT_FLUSH_LANES
#define RT_FLUSH_LANES 7u
#endif
template <bool ANY, bool COUNT, int MODE>
__device__ __forceinline__ bool trav_flush(const TravMem& M, const f4* lds, const WaveWork& W, Trav& s, uint32_t& n_tris) {
  const uint32_t lane = threadIdx.x & 63u;
  const bool waiting = trav_waiting(s);
  const unsigned long long wmask = __builtin_amdgcn_ballot_w64(waiting);
  const unsigned long long smask = __builtin_amdgcn_ballot_w64(trav_searching(s));
  if (wmask == 0ull) return smask != 0ull;
  // (one scalar select and one scalar compare: as `few waiting && some searching` the compiler builds lane masks for both)
  const uint32_t n_wait = smask != 0ull ? (uint32_t)__builtin_popcountll(wmask) : 64u;
#ifndef RT_EXP_ITEMS24
  if (n_wait < RT_FLUSH_LANES) return true;
#endif
  const uint32_t cnt = s.leaf & 7u;   // 0 for a lane that does not wait
  const unsigned long long b0 = __builtin_amdgcn_ballot_w64((cnt & 1u) != 0u), b1 = __builtin_amdgcn_ballot_w64((cnt & 2u) != 0u),
                           b2 = __builtin_amdgcn_ballot_w64((cnt & 4u) != 0u);
  const uint32_t total = (uint32_t)__builtin_popcountll(b0) + 2u * (uint32_t)__builtin_popcountll(b1) +
                         4u * (uint32_t)__builtin_popcountll(b2);
  const uint32_t excl =
      __builtin_amdgcn_mbcnt_hi((uint32_t)(b0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b0, 0u)) +
      2u * __builtin_amdgcn_mbcnt_hi((uint32_t)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b1, 0u)) +
      4u * __builtin_amdgcn_mbcnt_hi((uint32_t)(b2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b2, 0u));
#ifdef RT_EXP_ITEMS24
  if (total < 24u && smask != 0ull) return true;
#endif
  const uint32_t first = s.leaf >> 3;
  if (waiting) {
    if (trav_post_at_entry<MODE>()) {
      reinterpret_cast<float*>(W.rays)[8u * lane + 3u] = s.closest;   // the bound at leaf entry, beside the origin
    } else {
      f4 ra, rb;
      ra.x = rt_opaque(s.io.x); ra.y = rt_opaque(s.io.y); ra.z = rt_opaque(s.io.z); ra.w = s.closest;
      rb.x = rt_opaque(s.id.x); rb.y = rt_opaque(s.id.y); rb.z = rt_opaque(s.id.z); rb.w = 0.0f;
      W.rays[2 * lane] = ra;
      W.rays[2 * lane + 1] = rb;
    }
    W.res[lane] = ~0ull;
    // Items (owner lane << 26 | triangle) of the lane's leaf at [excl, excl + cnt).  Four UNCONDITIONAL stores, the highest
    // slot first: a store past the lane's count lands on slot j < i of a later lane (excl' >= excl + cnt), and that lane's own
    // store to its slot j is a LATER instruction — the LDS executes a wave's instructions in order, so the valid word is the
    // one that stays.  (volatile: the compiler may neither reorder nor merge them.)  The last slot written this way is
    // excl + 3 <= 63 * 7 + 3: inside the 448-entry queue.  Conditional stores cost a compare, an exec save and a branch each.
    const rt_lptr32_ordered it = (rt_lptr32_ordered)(W.items + excl);
    const uint32_t word = (lane << 26) | first;
#ifdef RT_EXP_COND_STORES
    for (uint32_t i = 0; i < 4u; i++) if (i < cnt) W.items[excl + i] = word + i;
#else
    it[3] = word + 3u;
    it[2] = word + 2u;
    it[1] = word + 1u;
    it[0] = word;
#endif
  }
  // leaves of the reference's builder hold <= 4 triangles (blas.rs:99); only its fallback leaves hold 5-7: those three
  // stores sit behind a wave-uniform test (count bit 2 set together with bit 0 or bit 1) and come after the four above
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
  const bool tri_lds = MODE == RT_TRAV_LDS || M.l_tri != RT_LDS_NONE;   // wave-uniform
  for (uint32_t c = 0; c < total; c += 64u) {
    const uint32_t j = c + lane;
#ifdef RT_LANE_STATS
    RT_LSTAT(s.stat_kind + 1u, j < total);
#endif
    if (j < total) {
      const uint32_t it = W.items[j];
      const uint32_t owner = it >> 26, tri = it & 0x03ffffffu;
      const f4 ra = W.rays[2 * owner];
      const rt_f3_16 rb = *(const rt_f3_16 __attribute__((address_space(3)))*)(W.rays + 2 * owner + 1);   // 12 of the 16 bytes
      LocalRay q;
      q.o = rt3_make(ra.x, ra.y, ra.z);
      q.d = rt3_make(rb.x, rb.y, rb.z);
      f4 g0, g1, g2;
      if (tri_lds) {
        const uint32_t slot = M.l_tri + __umul24(tri, (uint32_t)RT_TRI_STRIDE);   // tri < 2^26 / 3 slots: a 24-bit multiply
        g0 = ld_l(lds, slot);
        g1 = ld_l(lds, slot + 1u);
        g2 = ld_l(lds, slot + 2u);
      } else {
        g0 = ld_g(M.gtri, RT_TRI_STRIDE * (size_t)tri);
        g1 = ld_g(M.gtri, RT_TRI_STRIDE * (size_t)tri + 1);
        g2 = ld_g(M.gtri, RT_TRI_STRIDE * (size_t)tri + 2);
      }
      float t;
      const bool ok = hit_tri_nb(g0, g1, g2, q, RT_T_MIN, ra.w, t);
      // The reference's leaf loop (Raytracer.wgsl:474-482) accepts test i iff it passes and t_i < the running closest,
      // so it ends with the smallest accepted t and, among equal ones, the first in leaf order: a minimum over
      // (t, position), whatever the order of evaluation.  t > 0 here, so its bits order like the value; the triangle id
      // grows with the position in the leaf.  ANY (shadow rays) needs only the first accepted position.
      if (ok) atomicMin(&W.res[owner], ANY ? (unsigned long long)tri : (((unsigned long long)rt_f2u(t) << 32) | tri));
    }
  }
  __builtin_amdgcn_wave_barrier();
  if (waiting) {
    const unsigned long long best = W.res[lane];
    const bool found = best != ~0ull;
    if (COUNT) n_tris += (ANY && found) ? ((uint32_t)best - first + 1u) : cnt;   // the any-hit loop stops at its first hit
    if (found) {
      s.best_tri = (int32_t)(uint32_t)best;
      if (!ANY) {
        s.closest = rt_u2f((uint32_t)(best >> 32));
        s.best_inst = (int32_t)s.cur_inst;
      }
    }
    s.leaf = 0u;
    s.curr = (ANY && found) ? RT_CURR_IDLE : s.resume;
  }
  __builtin_amdgcn_wave_barrier();
  return true;
}

// traverse(): the whole walk of one wave's rays (persistent kernel, one traversal per bounce and ray kind)
template <bool ANY, bool COUNT, int MODE>
__device__ __forceinline__ void traverse(const TravMem& M, const f4* lds, const WaveWork& W, uint32_t blas_base, bool active,
                                         rt3 o, rt3 d, float t_max, float& out_t, int32_t& out_tri,
                                         int32_t& out_inst, bool& out_any, uint32_t& n_nodes, uint32_t& n_tris) {
  Trav s;
  trav_begin(s, active, blas_base, o, d, t_max);
#ifdef RT_LANE_STATS
  s.stat_kind = ANY ? 1u : 3u;
#endif
  for (;;) {
    // RT_STEPS_PER_TRIP node steps between two looks at the triangle queue: the look (ballots, population counts, the
    // branch) costs a third of a trip; a lane that reaches a leaf in an earlier step simply sits out the later ones
    trav_trip<COUNT, MODE, RT_STEPS_PER_TRIP>(M, lds, W, s, n_nodes);
    if (!trav_flush<ANY, COUNT, MODE>(M, lds, W, s, n_tris)) break;
  }
  out_t = s.closest;
  out_tri = s.best_tri;
  out_inst = s.best_inst;
  out_any = trav_any(s);
}

}  // namespace rtk
#endif
