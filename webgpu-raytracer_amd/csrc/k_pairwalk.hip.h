// k_pairwalk.hip.h — the per-ray state machine of the traversal (Raytracer.wgsl:455-600) over CHILD-PAIR records.
// Part of the kernel set of csrc/kernels.hip.h (included from there, in order); the lane functions below are plain
// C++ (`RT_HD`): tests/model/pairwalk_model.hip compiles the very same functions for the host and walks single rays
// with them against the oracle's literal loop (results and all counters), so the logic that runs on the GPU is the
// logic that was checked on the CPU.
//
// WHY.  The reference walks a pre-order node array with skip pointers: one 32-byte node per dependent step.  Measured
// on MI355X (tools/gather_peak.hip, profiles/r03_gather_peak.txt): a wave-level step of lane-divergent 16-byte loads
// costs the CU's texture-address path about 20 + 0.3 x lanes cycles per INSTRUCTION when the lines are L1-resident and
// 2.3 cycles per missing LINE when they come from L2 — and four lanes that read the four 16-byte chunks of ONE
// 64-byte-aligned record in one instruction cost as much as one lane reading 16 bytes.  So the traversal array is
// re-laid-out as one 64-byte record per INNER node holding BOTH children (k_pairs.hip.h), fetched quad-cooperatively:
// half the dependent rounds per ray, and a round costs what a 32-byte node cost before.
//
// SAME FUNCTION.  The reference tests a node against the closest hit of the moment it is REACHED in pre-order.  A pair
// record delivers the right child R early, so:
//   * the left child L is tested at once (it is reached now);
//   * R's slab interval is computed at once, but R is only REACHED after L's subtree.  If its test fails against the
//     bound of now it fails against every later (smaller) bound: it is dropped.  If it passes, (word, A = max(t_min,
//     near)) goes on a short per-lane stack in LDS and is re-tested when popped: `A <= closest of that moment`, which
//     is the reference's test (tm_near <= min(closest, far)) given that it held for the older bound (k_pairwalk notes
//     at pw_pair).  Per ray, the sequence of nodes REACHED with a passing test, every bound they are tested against,
//     the triangles tested and all tie-breaks are the reference's.
//   * Counters: a node counts when it is reached (L at the fetch; R when L missed, when it is popped, or on a stackless
//     arrival) — in the counting (DETAIL) build R is pushed even when it already failed (A = NaN: the pop counts it and
//     moves on), so nodes_visited equals the oracle's, also for shadow rays that end early.
//   * Entering an instance pushes a sentinel; popping it is how a lane leaves the instance again (world-space ray back).
//   * Stack overflow (K entries per lane): the level (TLAS or BLAS) drops its entries and goes on STACKLESS, with the
//     skip pointers every record still carries (q3.w: where the walk goes after this subtree); dropped entries are
//     later in pre-order than the walk, so it simply reaches them again.  A stackless arrival re-reads a pair record
//     for its right half.  Correct for any tree depth; rare with K = 6-8.
#ifndef MI355RT_K_PAIRWALK_HIP_H
#define MI355RT_K_PAIRWALK_HIP_H

namespace rtk {

#define RT_PAIR_INNER 0x80000000u   // child word: inner node -> low bits = index of the child's own pair record
#define RT_REF_END 0xffffffffu      // skip reference: the walk leaves the TLAS / the BLAS

#define RT_PW_SENTINEL 0x7ffffff0u  // stack word pushed at instance entry: popping it leaves the instance (no leaf word
                                    // looks like it: triangle ids stay below 2^26, k_traverse.hip.h items)

// lane states
#define PW_DONE 0u        // no ray, or the ray's walk is over
#define PW_FETCH 1u       // fetch pair record `curr`, test both children
#define PW_FETCHR 2u      // stackless arrival at the RIGHT child of pair `curr`: fetch the record, test R only
#define PW_POP 3u         // take the next pending entry from the stack (a right child, or the way out of the instance)
#define PW_WAIT 4u        // BLAS leaf queued, waiting for the wave's triangle flush
#define PW_ENTER 5u       // TLAS leaf hit: instance entry pending (batched)

#ifndef RT_HD
#define RT_HD __host__ __device__ __forceinline__
#endif

struct PwRay {  // same fields as LocalRay (k_intersect.hip.h), host-compilable
  rt3 o, d, inv_d, o_inv_d;
};
RT_HD PwRay pw_make_ray(rt3 o, rt3 d) {  // make_ray, Raytracer.wgsl:83-86
  PwRay r;
  r.o = o;
  r.d = d;
  r.inv_d = rt_rcp3(d);
  r.o_inv_d = o * r.inv_d;
  return r;
}

// lane flags (one register: as separate bools the compiler kept them as scalar lane masks and spent three or four scalar
// instructions on every conditional update)
#define PW_F_IN_BLAS 1u
#define PW_F_ANY 2u        // shadow ray: the first accepted hit ends the walk
#define PW_F_FOUND 4u      // ... and it was found
#define PW_F_SL_TLAS 8u    // the TLAS level has dropped its stack entries and walks stackless
#define PW_F_SL_BLAS 16u   // the same for the BLAS the lane is in

struct PairLane {
  PwRay r;                  // the ray in the space it is walking (world or instance)
  PwRay rw;                 // the world-space ray
  float closest;
  int32_t best_tri, best_inst;
  uint32_t curr;            // pair record to fetch (PW_FETCH / PW_FETCHR)
  uint32_t cur_inst, leaf;
  uint32_t sp, floor;       // stack entries in use; while in a BLAS, entries [0, floor) belong to the TLAS level (the last of
                            // them is the sentinel of the instance)
  uint32_t resume;          // stackless level: where the walk goes after the leaf / instance it is busy with
  uint32_t tlas_resume;     // stackless TLAS: where the TLAS walk goes after the instance it is in
  uint32_t state;
  uint32_t flags;
};
RT_HD bool pw_flag(const PairLane& s, uint32_t f) { return (s.flags & f) != 0u; }

// slab test of one child (intersect_aabb, Raytracer.wgsl:433-441): hit iff max(t_min, near) <= min(t_max, far);
// a_out = max(t_min, near), never NaN (t_min is not) — what a deferred right child is re-tested with
RT_HD bool pw_box(float lx, float ly, float lz, float hx, float hy, float hz, const PwRay& r, float t_min, float t_max,
                  float& a_out) {
  float t1x = lx * r.inv_d.x - r.o_inv_d.x, t2x = hx * r.inv_d.x - r.o_inv_d.x;
  float t1y = ly * r.inv_d.y - r.o_inv_d.y, t2y = hy * r.inv_d.y - r.o_inv_d.y;
  float t1z = lz * r.inv_d.z - r.o_inv_d.z, t2z = hz * r.inv_d.z - r.o_inv_d.z;
  float nx = rt_min(t1x, t2x), ny = rt_min(t1y, t2y), nz = rt_min(t1z, t2z);
  float fx = rt_max(t1x, t2x), fy = rt_max(t1y, t2y), fz = rt_max(t1z, t2z);
  float tm_near = rt_max(t_min, rt_max(nx, rt_max(ny, nz)));
  float tm_far = rt_min(t_max, rt_min(fx, rt_min(fy, fz)));
  a_out = tm_near;
  return tm_near <= tm_far;
}

// The walk is at a node whose test has just passed (`take`; nothing happens otherwise): descend, queue the leaf, or ask for
// the instance.  `after`: stackless successor of this node's subtree (only read on a stackless level).  Everything is a
// select on the values: straight-line code, no lane-mask juggling (and as branches that store into one field or another
// the compiler formed a pointer select and the lane state went to scratch memory).
RT_HD void pw_child(PairLane& s, uint32_t word, uint32_t after, bool take) {
  const bool inner = (word & RT_PAIR_INNER) != 0u;
  const bool in_blas = pw_flag(s, PW_F_IN_BLAS);
  const bool blas_leaf = take && !inner && in_blas, tlas_leaf = take && !inner && !in_blas;
  s.curr = (take && inner) ? (word & ~RT_PAIR_INNER) : s.curr;
  s.leaf = blas_leaf ? word : s.leaf;
  s.resume = blas_leaf ? after : s.resume;
  s.cur_inst = tlas_leaf ? (word >> 3) : s.cur_inst;
  s.tlas_resume = tlas_leaf ? after : s.tlas_resume;
  const uint32_t st = inner ? PW_FETCH : (in_blas ? PW_WAIT : PW_ENTER);
  s.state = take ? st : s.state;
}

// stackless jump: to the right child of pair `ref`; off the end of the level = whatever the stack holds next (in a BLAS:
// the way out of the instance; on the TLAS level a stackless level has nothing stacked: the ray is done)
RT_HD void pw_goto(PairLane& s, uint32_t ref) {
  s.curr = ref;
  s.state = ref == RT_REF_END ? PW_POP : PW_FETCHR;
}

// a new ray: the TLAS root is tested from its own record (two float4: {min, word} {max, -})
template <bool COUNT>
RT_HD void pw_begin(PairLane& s, bool active, bool any, rt3 o, rt3 d, float t_min, float t_max, float rlx, float rly,
                    float rlz, uint32_t rword, float rhx, float rhy, float rhz, uint32_t& n_nodes) {
  s.rw = pw_make_ray(o, d);
  s.r = s.rw;
  s.closest = t_max;
  s.best_tri = -1;
  s.best_inst = -1;
  s.curr = 0u;
  s.cur_inst = 0u;
  s.leaf = 0u;
  s.sp = 0u;
  s.floor = 0u;
  s.resume = RT_REF_END;
  s.tlas_resume = RT_REF_END;
  s.flags = any ? PW_F_ANY : 0u;
  s.state = PW_DONE;
  if (active) {
    if (COUNT) n_nodes++;
    float a;
    const bool hit = pw_box(rlx, rly, rlz, rhx, rhy, rhz, s.r, t_min, s.closest, a);
    pw_child(s, rword, RT_REF_END, hit);
  }
}

// What a lane does with the pair record it fetched.  q0 = {L.min, wordL}, q1 = {L.max, -}, q2 = {R.min, wordR},
// q3 = {R.max, skipX}: skipX = reference of the pair whose right child follows this subtree in pre-order (or END).
// STK: push(slot, word, a).  K: stack entries per lane.
template <bool COUNT, uint32_t K, class STK>
RT_HD void pw_pair(PairLane& s, uint32_t self, float l0x, float l0y, float l0z, uint32_t word_l, float l1x, float l1y,
                   float l1z, float r0x, float r0y, float r0z, uint32_t word_r, float r1x, float r1y, float r1z,
                   uint32_t skip_x, float t_min, STK& stk, uint32_t& n_nodes) {
  const bool only_r = s.state == PW_FETCHR;   // stackless arrival: L's subtree is done already
  float a_l, a_r;
  bool hit_l = pw_box(l0x, l0y, l0z, l1x, l1y, l1z, s.r, t_min, s.closest, a_l);
  const bool hit_r = pw_box(r0x, r0y, r0z, r1x, r1y, r1z, s.r, t_min, s.closest, a_r);
  hit_l = hit_l && !only_r;
  if (COUNT) n_nodes += (only_r ? 0u : 1u) + (hit_l ? 0u : 1u);   // L is reached now; R too when L is not entered
  const bool in_blas = pw_flag(s, PW_F_IN_BLAS);
  bool sl = pw_flag(s, in_blas ? PW_F_SL_BLAS : PW_F_SL_TLAS);
  // With L hit, R is reached after L's subtree.  It passes later iff it passes now AND a_r <= the closest of that moment:
  // tm_far = min(closest, far) only shrinks with closest, and a_r <= min(c_old, far) implies a_r <= far.
  const bool want_push = hit_l && !sl && (hit_r || COUNT);
  const bool overflow = want_push && s.sp >= K;
  const bool do_push = want_push && !overflow;
  if (do_push) stk.push(s.sp, word_r, hit_r ? a_r : rt_u2f(0x7fc00000u));   // NaN: counted when popped, never entered
  s.sp = do_push ? s.sp + 1u : s.sp;
  if (overflow) {               // rare: this level drops its entries and goes on stackless
    s.sp = in_blas ? s.floor : 0u;
    s.flags |= in_blas ? PW_F_SL_BLAS : PW_F_SL_TLAS;
    sl = true;
  }
  const bool take = hit_l || hit_r;
  // stackless successor of the child's subtree: after L the right child of this very pair, after R whatever follows the pair
  pw_child(s, hit_l ? word_l : word_r, hit_l ? self : skip_x, take);
  // nothing entered: the next pending entry, or (stackless) the node that follows the pair
  const uint32_t jump = skip_x == RT_REF_END ? PW_POP : PW_FETCHR;
  s.state = take ? s.state : (sl ? jump : PW_POP);
  s.curr = (!take && sl) ? skip_x : s.curr;
}

// next pending entry.  STK: pop(slot, word&, a&).  An empty stack ends the ray (a lane inside an instance always has at
// least the entry's sentinel below it).  The sentinel: the BLAS is finished, back to the world-space ray and the TLAS level.
template <bool COUNT, class STK>
RT_HD void pw_pop(PairLane& s, STK& stk, uint32_t& n_nodes) {
  const bool empty = s.sp == 0u;
  s.sp = empty ? 0u : s.sp - 1u;
  uint32_t word;
  float a;
  stk.pop(s.sp, word, a);       // (an empty lane reads slot 0 and ignores it)
  const bool sent = !empty && word == RT_PW_SENTINEL;
  const bool child = !empty && !sent;
  if (COUNT) n_nodes += child ? 1u : 0u;          // a right child, reached now
  const bool pass = child && a <= s.closest;
  const bool sl_tlas = pw_flag(s, PW_F_SL_TLAS);
  const bool more = sl_tlas ? (s.tlas_resume != RT_REF_END) : (s.sp != 0u);
  if (sent && more) s.r = s.rw;
  s.flags = sent ? (s.flags & ~(PW_F_IN_BLAS | PW_F_SL_BLAS)) : s.flags;
  const uint32_t out_state = more ? (sl_tlas ? PW_FETCHR : PW_POP) : PW_DONE;
  s.curr = (sent && sl_tlas) ? s.tlas_resume : s.curr;
  s.state = empty ? PW_DONE : (sent ? out_state : s.state);   // a child that fails its re-test: stays PW_POP
  pw_child(s, word, RT_REF_END, pass);            // `after` is never read: stack levels do not jump
}

// instance entry (Raytracer.wgsl:507-512): object-space ray, the way out on the stack, then the BLAS root from its record
template <bool COUNT, uint32_t K, class STK>
RT_HD void pw_enter(PairLane& s, float m00, float m01, float m02, float m03, float m10, float m11, float m12, float m13,
                    float m20, float m21, float m22, float m23, float rlx, float rly, float rlz, uint32_t rword, float rhx,
                    float rhy, float rhz, float t_min, STK& stk, uint32_t& n_nodes) {
  const rt3 o = s.rw.o, d = s.rw.d;
  rt3 lo = rt3_make(m00 * o.x + m01 * o.y + m02 * o.z + m03 * 1.0f, m10 * o.x + m11 * o.y + m12 * o.z + m13 * 1.0f,
                    m20 * o.x + m21 * o.y + m22 * o.z + m23 * 1.0f);
  rt3 ld = rt3_make(m00 * d.x + m01 * d.y + m02 * d.z + m03 * 0.0f, m10 * d.x + m11 * d.y + m12 * d.z + m13 * 0.0f,
                    m20 * d.x + m21 * d.y + m22 * d.z + m23 * 0.0f);
  s.r = pw_make_ray(lo, ld);
  const bool full = s.sp >= K;   // no room for the sentinel: the TLAS level drops its entries (only possible when the leaf came
  s.sp = full ? 0u : s.sp;       // from a pair test, whose `after` is in tlas_resume: a popped leaf has just freed a slot)
  s.flags = (s.flags | PW_F_IN_BLAS | (full ? PW_F_SL_TLAS : 0u)) & ~PW_F_SL_BLAS;
  stk.push(s.sp, RT_PW_SENTINEL, 0.0f);
  s.sp++;
  s.floor = s.sp;
  if (COUNT) n_nodes++;
  float a;
  const bool hit = pw_box(rlx, rly, rlz, rhx, rhy, rhz, s.r, t_min, s.closest, a);
  s.state = PW_POP;             // a missed root: straight out again
  pw_child(s, rword, RT_REF_END, hit);
}

// the leaf's triangles have been tested (trav flush): record the result, go on
RT_HD void pw_after_leaf(PairLane& s, bool found, float t, uint32_t tri) {
  const bool any = pw_flag(s, PW_F_ANY);
  const bool keep = found && !any;
  s.closest = keep ? t : s.closest;
  s.best_tri = keep ? (int32_t)tri : s.best_tri;
  s.best_inst = keep ? (int32_t)s.cur_inst : s.best_inst;
  s.flags |= (found && any) ? PW_F_FOUND : 0u;
  if (pw_flag(s, PW_F_SL_BLAS)) pw_goto(s, s.resume); else s.state = PW_POP;
  s.state = (found && any) ? PW_DONE : s.state;
}

}  // namespace rtk
#endif
