// k_wavefront.hip.h — wavefront form: k_wf_shade / k_wf_trace over device queues, k_accumulate_frames.
// Part of the kernel set of csrc/kernels.hip.h (included from there, in order; not a stand-alone header).
#ifndef MI355RT_K_WAVEFRONT_HIP_H
#define MI355RT_K_WAVEFRONT_HIP_H

namespace rtk {

// ================================================================== path tracer, wavefront form
// For scenes whose traversal records do not fit LDS (hundreds of thousands of triangles, a thousand instances) a ray
// visits 70+ nodes with a long tail, and the per-trip lockstep of the persistent kernel leaves 60 % of the lanes idle
// while they wait on L2 / Infinity-Cache latency.  The wavefront form splits a bounce into stages with the path state in
// HBM (one 64-B record per path, 288 GB to spare):
//   k_wf_shade   one lane per live path: picks up the results of the previous depth's rays by queue slot (pending NEE
//                term, hit of the extension ray), surface frame + shade_bounce(); appends the shadow ray and the
//                extension ray WITH their ray data to device queues (wave-aggregated atomics), finishes paths that end
//   k_wf_trace   persistent waves, RAY-level regeneration: a lane that finishes its ray writes the result to the ray's
//                queue slot and pulls the next ray of the queue (batched, >= RT_WF_REFILL lanes), so the slowest ray
//                no longer holds 63 lanes; same node step / LDS triangle queue as traverse().  Rays in, results out,
//                both streamed by slot: the path records are never touched here.
// Stages of one depth run as separate launches in stream order; the host enqueues all depths without reading anything
// back (queue sizes stay on the device).  Per path the arithmetic, RNG order and f32 addition order are unchanged, so
// the result is bit-identical to the other forms; frame colours go through frame_col + k_accumulate_frames.
// Restriction: SPP == 1 (the reference's default); other SPP values use the persistent kernel.
#ifndef RT_WF_REFILL
#define RT_WF_REFILL 24   // idle lanes that trigger a pull; re-swept with RT_WF_STEPS_PER_TRIP on the final kernels (below)
#endif

// Device queues are filled and drained in chunks of RT_WF_CHUNK entries: a wave reserves a chunk with ONE atomic
// and then appends with ballot/mbcnt ranks (a queue counter is a single address: ~88 atomics/us chip-wide, so one atomic
// per wave-append or per 16-ray pull caps a stage at a few Grays/s). Unused tail entries of a chunk hold RT_WF_INVALID.
#define RT_WF_CHUNK 256u
#define RT_WF_INVALID 0xffffffffu
struct WaveQueueWriter {
  uint32_t pos, end;  // wave-uniform cursor into the current chunk
};
// returns the slot for lanes with want == true (RT_WF_INVALID otherwise); call from wave-uniform control flow
__device__ __forceinline__ uint32_t wq_append(WaveQueueWriter& w, uint32_t* counter, uint32_t* ids, bool want) {
  const uint32_t lane = threadIdx.x & 63u;
  const unsigned long long mask = __ballot(want);
  if (mask == 0ull) return RT_WF_INVALID;
  const uint32_t n = (uint32_t)__builtin_popcountll(mask);
  if (w.pos + n > w.end) {
    for (uint32_t i = w.pos + lane; i < w.end; i += 64u) ids[i] = RT_WF_INVALID;
    uint32_t b = 0;
    if (lane == 0u) b = atomicAdd(counter, RT_WF_CHUNK);
    b = __shfl(b, 0, 64);
    w.pos = b;
    w.end = b + RT_WF_CHUNK;
  }
  const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
  const uint32_t slot = w.pos + rank;
  w.pos += n;
  return want ? slot : RT_WF_INVALID;
}
__device__ __forceinline__ void wq_finish(const WaveQueueWriter& w, uint32_t* ids) {
  const uint32_t lane = threadIdx.x & 63u;
  for (uint32_t i = w.pos + lane; i < w.end; i += 64u) ids[i] = RT_WF_INVALID;
}

__device__ __forceinline__ void wf_store_path(WfPath* rec, const PathState& p, uint32_t flags, rt3 nee, uint32_t sslot,
                                              uint32_t eslot) {
  rec->c = make_float4(p.throughput.x, p.throughput.y, p.throughput.z, rt_u2f(p.rng));
  rec->d = make_float4(p.radiance.x, p.radiance.y, p.radiance.z, rt_u2f(flags));
  rec->e = make_float4(nee.x, nee.y, nee.z, p.prev_pdf);
  rec->m = make_uint4(sslot, eslot, 0u, 0u);   // the extension ray (p.ro, p.rd) is in Q.ext_rays[depth & 1] at eslot
}

// One lane per path that is alive at `depth`.  FIRST: the paths start at the pixels of the batch (camera ray + G-buffer
// surface).  Otherwise a path comes from the active list the previous depth's shade kernel wrote, and FIRST picks up what
// the two trace kernels left for it BY QUEUE SLOT: the occlusion bit of its shadow ray (the pending NEE term is added
// now, before anything else touches the radiance — the order of f32 additions of the reference) and the hit of its
// extension ray (a miss ends the path).  A path that had ended but was waiting for its shadow ray is finished here; the
// host launches one more pass after the last depth for those.  The trace kernels therefore never touch the path state:
// they stream their queue in and their results out.
template <bool FIRST, bool DETAIL>
#ifndef RT_SHADE_WAVES
#define RT_SHADE_WAVES 4
#endif
__global__ __launch_bounds__(256, RT_SHADE_WAVES) void k_wf_shade(DevScene S, DevFrame F, rt_scene_uniforms U, WfState W, WfQueues Q,
                                                  const DevFrameSlot* __restrict__ slots, uint32_t n_slots,
                                                  uint32_t depth) {
  const uint32_t npx = U.width * U.height;
  uint32_t* cnt = Q.counters + 8u * depth;
  const uint32_t count = FIRST ? npx * n_slots : cnt[0];
  const uint32_t* active_in = Q.active[depth & 1u];
  uint32_t* next_active = Q.active[(depth + 1u) & 1u];
  uint32_t* next_count = Q.counters + 8u * (depth + 1u);
  uint32_t cnt_shaded = 0;
  WaveQueueWriter wq_shadow = {0u, 0u}, wq_ext = {0u, 0u}, wq_next = {0u, 0u};
  // wave-uniform loop (every lane of a wave takes part in the queue appends)
  for (uint32_t base_idx = (blockIdx.x * 256u + (threadIdx.x & ~63u)); base_idx < count; base_idx += gridDim.x * 256u) {
    const uint32_t idx = base_idx + (threadIdx.x & 63u);
    bool live = idx < count;
    uint32_t id = 0u;
    if (live) {
      id = FIRST ? idx : active_in[idx];
      live = id != RT_WF_INVALID;
    }
    BounceOut bo;
    bo.want_shadow = bo.want_extend = bo.nee_valid = bo.ended = false;
    bo.sh_o = bo.sh_d = bo.nee = rt3_splat(0.0f);
    bo.sh_tmax = 0.0f;
    PathState p;
    p.col = rt3_splat(0.0f);
    p.sample = 0u;
    p.pixel = id % npx;
    p.tri = p.inst = p.depth = p.rng = 0u;
    p.hit_t = p.prev_pdf = 0.0f;
    p.specular = true;
    p.ro = p.rd = p.throughput = p.radiance = p.normal = p.geom_n = p.albedo = rt3_splat(0.0f);
    p.tex_uv = rt2_make(0.0f, 0.0f);
    if (live) {
    if (FIRST) {
      const uint32_t x = p.pixel % U.width, y = p.pixel / U.width;
      if (!owns_row(F, y)) live = false;
      const DevFrameSlot slot = slots[id / npx];
      p.rng = init_rng(p.pixel, slot.frame_count);  // SPP == 1: frame_count * SPP + 0
      rt3 cam_o = rt3_make(U.camera.origin[0], U.camera.origin[1], U.camera.origin[2]);
      rt3 off = rt3_splat(0.0f);
      const float lens = U.camera.origin[3];
      if (lens > 0.0f) {
        float r = rt_sqrt(rand_pcg(p.rng));
        float theta = RT_TWO_PI * rand_pcg(p.rng);
        float st, ct;
        rt_sincos(theta, &st, &ct);
        rt3 rdk = lens * rt3_make(r * ct, r * st, 0.0f);
        rt3 cu = rt3_make(U.camera.u[0], U.camera.u[1], U.camera.u[2]);
        rt3 cv = rt3_make(U.camera.v[0], U.camera.v[1], U.camera.v[2]);
        off = cu * rdk.x + cv * rdk.y;
      }
      rt3 cam_ll = rt3_make(U.camera.lower_left[0], U.camera.lower_left[1], U.camera.lower_left[2]);
      rt3 cam_h = rt3_make(U.camera.horizontal[0], U.camera.horizontal[1], U.camera.horizontal[2]);
      rt3 cam_v = rt3_make(U.camera.vertical[0], U.camera.vertical[1], U.camera.vertical[2]);
      float u = ((float)x + 0.5f + slot.jitter_x * (float)U.width) / (float)U.width;
      float v = 1.0f - ((float)y + 0.5f + slot.jitter_y * (float)U.height) / (float)U.height;
      p.rd = cam_ll + u * cam_h + v * cam_v - cam_o - off;
      p.ro = cam_o + off;
      p.throughput = rt3_splat(1.0f);
      p.radiance = rt3_splat(0.0f);
      p.prev_pdf = 0.0f;
      p.specular = true;
      p.depth = 0u;
      if (live && (slot.depth[p.pixel] >= 1.0f || F.max_depth == 0u)) {  // background (or MAX_DEPTH = 0): black sample
        F.frame_col[id] = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
        live = false;
      }
      if (live) {
        float4 g = slot.normal_id[p.pixel];
        p.tri = rt_f2u(g.z);
        p.inst = rt_f2u(g.w);
        setup_surface(S, p, true, g.x, g.y, slot.albedo[p.pixel]);
      }
    } else {
      const WfPath* const rec = W.p[depth & 1u] + idx;   // the path's record sits at its position in this depth's list
      const float4 d = rec->d, e = rec->e;
      const uint4 m = rec->m;     // m.x: slot of the shadow ray, m.y: slot of the extension ray
      const uint32_t fl = rt_f2u(d.w);
      p.radiance = xyz(d);
      if (m.x != RT_WF_INVALID && (fl & WF_FLAG_NEE_VALID) != 0u && Q.occluded[m.x] == 0u)
        p.radiance = p.radiance + xyz(e);  // radiance += pending NEE term (nothing is added when bsdf_pdf <= 0)
      float4 h = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      bool over = (fl & WF_FLAG_ENDED) != 0u;
      if (!over) {
        h = Q.ext_hit[m.y];
        over = (int32_t)rt_f2u(h.z) < 0;  // miss: the path ends with what it has
      }
      if (over) {
        F.frame_col[id] = make_float4(p.radiance.x, p.radiance.y, p.radiance.z, 1.0f);  // SPP == 1: col / 1
        live = false;
      } else {
        const float4* went = Q.ext_rays[(depth + 1u) & 1u];   // the previous depth's extension rays: the ray the path went along
        const float4 a = went[2 * m.y], b = went[2 * m.y + 1], c = rec->c;
        p.ro = xyz(a);
        p.rd = xyz(b);
        p.prev_pdf = e.w;
        p.throughput = xyz(c);
        p.rng = rt_f2u(c.w);
        p.depth = (fl & 0xffu) + 1u;
        p.specular = (fl & WF_FLAG_SPECULAR) != 0u;
        p.hit_t = h.x;
        p.tri = rt_f2u(h.y);
        p.inst = rt_f2u(h.z);
        setup_surface(S, p, false, 0.0f, 0.0f, 0u);
      }
    }
    if (live) {
      if (DETAIL) cnt_shaded++;
      shade_bounce(S, U.light_count, F.max_depth, p, bo);
    }
    }  // if (live) — everything below runs for the whole wave
    const uint32_t sslot = wq_append(wq_shadow, &cnt[1], Q.shadow_ids, live && bo.want_shadow);
    if (sslot != RT_WF_INVALID) {
      Q.shadow_ids[sslot] = id;
      Q.shadow_rays[2 * sslot] = make_float4(bo.sh_o.x, bo.sh_o.y, bo.sh_o.z, bo.sh_tmax);
      Q.shadow_rays[2 * sslot + 1] = make_float4(bo.sh_d.x, bo.sh_d.y, bo.sh_d.z, 0.0f);
    }
    const uint32_t eslot = wq_append(wq_ext, &cnt[2], Q.ext_ids, live && bo.want_extend);
    if (eslot != RT_WF_INVALID) {
      Q.ext_ids[eslot] = id;
      float4* out = Q.ext_rays[depth & 1u];
      out[2 * eslot] = make_float4(p.ro.x, p.ro.y, p.ro.z, 0.0f);
      out[2 * eslot + 1] = make_float4(p.rd.x, p.rd.y, p.rd.z, 0.0f);
    }
    // the path goes on to the next depth's shade pass when something is pending for it
    const uint32_t nslot = wq_append(wq_next, &next_count[0], next_active, live && (bo.want_shadow || bo.want_extend));
    if (nslot != RT_WF_INVALID) next_active[nslot] = id;
    if (live) {
      if (bo.ended && !bo.want_shadow) {
        F.frame_col[id] = make_float4(p.radiance.x, p.radiance.y, p.radiance.z, 1.0f);  // SPP == 1: col / 1
      } else {
        const uint32_t flags = (p.depth & 0xffu) | (p.specular ? WF_FLAG_SPECULAR : 0u) |
                               (bo.ended ? WF_FLAG_ENDED : 0u) | (bo.nee_valid ? WF_FLAG_NEE_VALID : 0u);
        wf_store_path(W.p[(depth + 1u) & 1u] + nslot, p, flags, bo.nee, sslot, eslot);   // stored iff the path is on the next list
      }
    }
  }
  wq_finish(wq_shadow, Q.shadow_ids);
  wq_finish(wq_ext, Q.ext_ids);
  wq_finish(wq_next, next_active);
  if (DETAIL) {
    LaneCounters c = {0, 0, 0, 0, 0, cnt_shaded};
    flush_counters<true>(c, F.counters, blockIdx.x * 4u + (threadIdx.x >> 6));
  }
}

// Persistent ray tracer over a device queue: rays stream in by queue slot ({o, t_max} {d} written by k_wf_shade) and
// results stream out by queue slot — ANY (shadow rays): one occlusion word; else (extension rays): {t, triangle,
// instance} of the closest hit.  The path state is not touched here.  Ray-level regeneration: a lane whose ray is
// finished writes its result and, when >= RT_WF_REFILL lanes are idle, the wave pulls the next rays of its chunk.
// BLOCK threads per workgroup (256 / 512 / 1024; 256 x 6 per CU by default): a bigger workgroup shares one staged copy of
// the records among more waves at the price of fewer waves per SIMD (LdsPlan, rt_api.hip plan_lds; measured without gain,
// DESIGN.md 4.1b).  LDS = true: every traversal record fits (RT_TRAV_LDS).
#ifndef RT_WF_WAVES
#define RT_WF_WAVES 5   // waves per SIMD of the trace kernels: what the per-wave LDS block (work queue + stack, 8.3 KB at K = 8) leaves room for
#endif
#ifndef RT_WF_STEPS_PER_TRIP
#define RT_WF_STEPS_PER_TRIP 4   // RECORD fetches (two node tests each) between two looks at the ray queue and the triangle queue.
                                 // Round 2, single-node steps: first sweep on
                                 // sponza-like (ms per 32 frames): 1: 153.0, 2: 137.7, 3: 134.5, 4: 133.2, 6: 132.0, 8: 131.4.
                                 // Final kernels, (steps, refill) -> ms per frame sponza-like / instanced x1000 / glass blob 4K:
                                 // (4, 16) 4.39 / 2.93 / 10.32, (6, 16) 4.39 / 2.90 / 10.19, (6, 24) 4.31 / 2.95 / 10.11,
                                 // (8, 24) 4.31 / 2.95 / 10.20, (6..12, 32..40) worse on instanced (up to +14 %)
#endif
// Diagnostic build only (-DRT_TRACE_STAMPS, tools/trace_sections.py): per-wave s_memtime cycles spent in the three
// sections of the trace loop, summed over all waves of all launches: [ANY][0..2] = cycles in retire/pull, node step,
// triangle flush; [3..5] = how often each section did work; [6] = waves; [7] = loop trips.  Nothing else reads it.
__device__ unsigned long long g_trace_sections[2][8];
// ---- walk over single nodes (rounds 1-2): kept behind rt_set_walk / MI355RT_WALK=node for A/B measurements
#ifndef RT_WF_NODE_WAVES
#define RT_WF_NODE_WAVES 6
#endif
#ifndef RT_WF_NODE_STEPS_PER_TRIP
#define RT_WF_NODE_STEPS_PER_TRIP 6
#endif
// RAYREG (mixed mode only): the instance-space origin / direction stay in registers (RT_TRAV_MIXED_RAYREG, k_traverse.hip.h).
template <bool ANY, bool DETAIL, bool LDS, int BLOCK, bool RAYREG = false>
__global__ __launch_bounds__(BLOCK, BLOCK == 256 ? (LDS ? 4 : RT_WF_NODE_WAVES) : (BLOCK == 512 ? 2 : 4))
void k_wf_trace(DevScene Sg, DevFrame F, rt_scene_uniforms U, WfQueues Q, uint32_t depth, uint32_t n_nodes_total,
                uint32_t n_tris_total, uint32_t n_inst_total, LdsPlan plan) {
  extern __shared__ f4 s_scene[];
  WaveWork W;
  char* const wbase = reinterpret_cast<char*>(s_scene) + (threadIdx.x >> 6) * RT_WORK_BYTES_PER_WAVE;
  wave_work_at(W, wbase);
  const uint32_t rec0 = ((BLOCK / 64) * RT_WORK_BYTES_PER_WAVE) / 16;
  TravMem M;
  if (LDS) {
    LdsPlan all;
    all.k_nodes = n_nodes_total;
    all.stage_inst = all.stage_tri = 1u;
    all.pad = 0u;
    trav_stage_mixed(M, s_scene, rec0, Sg, all, n_tris_total, n_inst_total);
  } else {
    trav_stage_mixed(M, s_scene, rec0, Sg, plan, n_tris_total, n_inst_total);
  }
  __syncthreads();
  constexpr int MODE = LDS ? RT_TRAV_LDS : (RAYREG ? RT_TRAV_MIXED_RAYREG : RT_TRAV_MIXED);
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t blas_base = U.blas_base_idx;
  uint32_t* cnt = Q.counters + 8u * depth;
  const uint32_t n_rays = ANY ? cnt[1] : cnt[2];
  uint32_t* head = ANY ? &cnt[3] : &cnt[4];
  const uint32_t* ids = ANY ? Q.shadow_ids : Q.ext_ids;
  const float4* rays = ANY ? Q.shadow_rays : Q.ext_rays[depth & 1u];

  // per-lane ray + traversal state
  bool have_ray = false;
  uint32_t slot = 0u;
  Trav s;
  trav_begin(s, false, blas_base, rt3_splat(1.0f), rt3_splat(1.0f), 0.0f);
  bool queue_left = true;
  uint32_t chunk_pos = 0u, chunk_end = 0u;  // wave-uniform cursor into the chunk of the input queue this wave holds
  uint32_t n_nodes = 0, n_tris = 0, n_traced = 0;

#ifdef RT_TRACE_STAMPS
  unsigned long long st_cyc[3] = {0, 0, 0}, st_cnt[3] = {0, 0, 0}, st_trips = 0;
#endif
  for (;;) {
#ifdef RT_TRACE_STAMPS
    const unsigned long long st0 = __builtin_amdgcn_s_memtime();
    st_trips++;
#endif
    // ---- retire finished rays and pull new ones (batched: a block that runs for one lane costs as much as for 64)
    const bool done = have_ray && !trav_busy(s);
    const bool idle = !have_ray || done;
    const unsigned long long idle_m = __ballot(idle), done_m = __ballot(done);
    const unsigned long long busy_m = __ballot(trav_busy(s));
    if (idle_m != 0ull &&
        ((uint32_t)__builtin_popcountll(done_m) >= RT_WF_REFILL ||
         (queue_left && (uint32_t)__builtin_popcountll(idle_m) >= RT_WF_REFILL) || busy_m == 0ull)) {
#ifdef RT_TRACE_STAMPS
      st_cnt[0]++;
#endif
      if (done) {
        if (ANY)
          Q.occluded[slot] = trav_any(s) ? 1u : 0u;
        else
          Q.ext_hit[slot] = make_float4(s.closest, rt_u2f((uint32_t)s.best_tri), rt_u2f((uint32_t)s.best_inst), 0.0f);
        have_ray = false;
      }
      // pull: needy lanes take consecutive entries of the wave's chunk; a new chunk costs one atomic
      const bool need = !have_ray;
      const unsigned long long need_m = __ballot(need);
      if (queue_left && need_m != 0ull) {
        if (chunk_pos >= chunk_end) {
          uint32_t bq = 0;
          if (lane == 0u) bq = atomicAdd(head, RT_WF_CHUNK);
          bq = __shfl(bq, 0, 64);
          if (bq >= n_rays) {
            queue_left = false;
          } else {
            chunk_pos = bq;
            chunk_end = bq + RT_WF_CHUNK < n_rays ? bq + RT_WF_CHUNK : n_rays;
          }
        }
        if (queue_left) {
          const uint32_t rank =
              __builtin_amdgcn_mbcnt_hi((uint32_t)(need_m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need_m, 0u));
          const uint32_t qi = chunk_pos + rank;
          chunk_pos += (uint32_t)__builtin_popcountll(need_m);
          if (need && qi < chunk_end && ids[qi] != RT_WF_INVALID) {
            const float4 r0 = rays[2 * qi], r1 = rays[2 * qi + 1];
            slot = qi;
            n_traced++;
            have_ray = true;
            trav_begin(s, true, blas_base, xyz(r0), xyz(r1), ANY ? r0.w : RT_T_MAX);
          }
        }
      }
    }
    if (!queue_left && __ballot(have_ray) == 0ull) break;  // queue exhausted and every ray retired

#ifdef RT_TRACE_STAMPS
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long st1 = __builtin_amdgcn_s_memtime();
    st_cyc[0] += st1 - st0;
    if (__ballot(trav_searching(s)) != 0ull) st_cnt[1]++;
#endif
    trav_trip<DETAIL, MODE, RT_WF_NODE_STEPS_PER_TRIP>(M, s_scene, W, s, n_nodes);
#ifdef RT_TRACE_STAMPS
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long st2 = __builtin_amdgcn_s_memtime();
    st_cyc[1] += st2 - st1;
    const bool was_waiting = __ballot(trav_waiting(s)) != 0ull;
#endif
    trav_flush<ANY, DETAIL, MODE>(M, s_scene, W, s, n_tris);
#ifdef RT_TRACE_STAMPS
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    st_cyc[2] += __builtin_amdgcn_s_memtime() - st2;
    if (was_waiting && __ballot(trav_waiting(s)) == 0ull) st_cnt[2]++;
#endif
  }
#ifdef RT_TRACE_STAMPS
  if (lane == 0u) {
    for (int k = 0; k < 3; k++) {
      atomicAdd(&g_trace_sections[ANY ? 1 : 0][k], st_cyc[k]);
      atomicAdd(&g_trace_sections[ANY ? 1 : 0][3 + k], st_cnt[k]);
    }
    atomicAdd(&g_trace_sections[ANY ? 1 : 0][6], 1ull);
    atomicAdd(&g_trace_sections[ANY ? 1 : 0][7], st_trips);
  }
#endif
  LaneCounters c = {0, ANY ? 0u : n_traced, ANY ? n_traced : 0u, n_nodes, n_tris, 0};
  flush_counters<DETAIL>(c, F.counters, blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6));
}

// ---- walk over child-pair records (round 3, k_pairwalk.hip.h / k_pairtrav.hip.h)
#ifndef RT_WF_PAIR_REFILL
#define RT_WF_PAIR_REFILL 16   // idle lanes that trigger a pull (swept 8 / 12 / 16 / 24 / 32 on the 263 k-triangle hall, ms per
                               // 32-frame batch: 181.3 / 179.6 (one pop per round) / 181.6 / 187.0 / 226.9)
#endif
template <bool ANY, bool DETAIL, bool LDS, int BLOCK>
__global__ __launch_bounds__(BLOCK, BLOCK == 256 ? (LDS ? 4 : RT_WF_WAVES) : (BLOCK == 512 ? 2 : 4))
void k_wf_trace_pairs(DevScene Sg, DevFrame F, rt_scene_uniforms U, WfQueues Q, uint32_t depth, uint32_t n_pairs_total,
                uint32_t n_tris_total, uint32_t n_inst_total, PairPlan plan) {
  extern __shared__ f4 s_scene[];
  WaveWork W;
  char* const wbase = reinterpret_cast<char*>(s_scene) + (threadIdx.x >> 6) * RT_PW_BYTES_PER_WAVE;
  wave_work_at(W, wbase);
  LdsStack stk = pw_stack_at(wbase);
  const uint32_t rec0 = ((BLOCK / 64) * RT_PW_BYTES_PER_WAVE) / 16;
  PairMem M;
  if (LDS) {
    PairPlan all = plan;
    all.stage_pairs = all.stage_inst = all.stage_tri = 1u;
    pw_stage(M, s_scene, rec0, Sg, all, n_pairs_total, n_tris_total, n_inst_total, RT_T_MIN);
  } else {
    pw_stage(M, s_scene, rec0, Sg, plan, n_pairs_total, n_tris_total, n_inst_total, RT_T_MIN);
  }
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t blas_base = U.blas_base_idx;
  uint32_t* cnt = Q.counters + 8u * depth;
  const uint32_t n_rays = ANY ? cnt[1] : cnt[2];
  uint32_t* head = ANY ? &cnt[3] : &cnt[4];
  const uint32_t* ids = ANY ? Q.shadow_ids : Q.ext_ids;
  const float4* rays = ANY ? Q.shadow_rays : Q.ext_rays[depth & 1u];

  // per-lane ray + traversal state
  bool have_ray = false;
  uint32_t slot = 0u;
  PairLane s;
  uint32_t n_nodes = 0, n_tris = 0, n_traced = 0;
  pw_start<false>(M, s, false, ANY, blas_base, rt3_splat(1.0f), rt3_splat(1.0f), 0.0f, n_nodes);
  bool queue_left = true;
  uint32_t chunk_pos = 0u, chunk_end = 0u;  // wave-uniform cursor into the chunk of the input queue this wave holds
#ifdef RT_PW_STAMPS
  unsigned long long pw_cyc[6] = {0, 0, 0, 0, 0, 0};
#endif

#ifdef RT_TRACE_STAMPS
  unsigned long long st_cyc[3] = {0, 0, 0}, st_cnt[3] = {0, 0, 0}, st_trips = 0;
#endif
  for (;;) {
#ifdef RT_TRACE_STAMPS
    const unsigned long long st0 = __builtin_amdgcn_s_memtime();
    st_trips++;
#endif
    // ---- retire finished rays and pull new ones (batched: a block that runs for one lane costs as much as for 64)
    const bool done = have_ray && s.state == PW_DONE;
    const bool idle = !have_ray || done;
    const unsigned long long idle_m = __ballot(idle), done_m = __ballot(done);
    const unsigned long long busy_m = __ballot(s.state != PW_DONE);
    if (idle_m != 0ull &&
        ((uint32_t)__builtin_popcountll(done_m) >= RT_WF_PAIR_REFILL ||
         (queue_left && (uint32_t)__builtin_popcountll(idle_m) >= RT_WF_PAIR_REFILL) || busy_m == 0ull)) {
#ifdef RT_TRACE_STAMPS
      st_cnt[0]++;
#endif
      if (done) {
        if (ANY)
          Q.occluded[slot] = pw_flag(s, PW_F_FOUND) ? 1u : 0u;
        else
          Q.ext_hit[slot] = make_float4(s.closest, rt_u2f((uint32_t)s.best_tri), rt_u2f((uint32_t)s.best_inst), 0.0f);
        have_ray = false;
      }
      // pull: needy lanes take consecutive entries of the wave's chunk; a new chunk costs one atomic
      const bool need = !have_ray;
      const unsigned long long need_m = __ballot(need);
      if (queue_left && need_m != 0ull) {
        if (chunk_pos >= chunk_end) {
          uint32_t bq = 0;
          if (lane == 0u) bq = atomicAdd(head, RT_WF_CHUNK);
          bq = __shfl(bq, 0, 64);
          if (bq >= n_rays) {
            queue_left = false;
          } else {
            chunk_pos = bq;
            chunk_end = bq + RT_WF_CHUNK < n_rays ? bq + RT_WF_CHUNK : n_rays;
          }
        }
        if (queue_left) {
          const uint32_t rank =
              __builtin_amdgcn_mbcnt_hi((uint32_t)(need_m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need_m, 0u));
          const uint32_t qi = chunk_pos + rank;
          chunk_pos += (uint32_t)__builtin_popcountll(need_m);
          if (need && qi < chunk_end && ids[qi] != RT_WF_INVALID) {
            const float4 r0 = rays[2 * qi], r1 = rays[2 * qi + 1];
            slot = qi;
            n_traced++;
            have_ray = true;
            pw_start<DETAIL>(M, s, true, ANY, blas_base, xyz(r0), xyz(r1), ANY ? r0.w : RT_T_MAX, n_nodes);
          }
        }
      }
    }
    if (!queue_left && __ballot(have_ray) == 0ull) break;  // queue exhausted and every ray retired

#ifdef RT_TRACE_STAMPS
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long st1 = __builtin_amdgcn_s_memtime();
    st_cyc[0] += st1 - st0;
    if (__ballot(pw_can_step(s)) != 0ull) st_cnt[1]++;
#endif
#ifdef RT_PW_STAMPS
    pw_trip<DETAIL, LDS, RT_WF_STEPS_PER_TRIP>(M, s_scene, reinterpret_cast<f4*>(wbase), stk, s, n_nodes, pw_cyc);
#else
    pw_trip<DETAIL, LDS, RT_WF_STEPS_PER_TRIP>(M, s_scene, reinterpret_cast<f4*>(wbase), stk, s, n_nodes);
#endif
#ifdef RT_TRACE_STAMPS
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long st2 = __builtin_amdgcn_s_memtime();
    st_cyc[1] += st2 - st1;
    const bool was_waiting = __ballot(s.state == PW_WAIT) != 0ull;
#endif
    pw_flush<DETAIL, LDS>(M, s_scene, W, s, n_tris);
#ifdef RT_TRACE_STAMPS
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    st_cyc[2] += __builtin_amdgcn_s_memtime() - st2;
    if (was_waiting && __ballot(s.state == PW_WAIT) == 0ull) st_cnt[2]++;
#endif
  }
#ifdef RT_TRACE_STAMPS
  if (lane == 0u) {
    for (int k = 0; k < 3; k++) {
      atomicAdd(&g_trace_sections[ANY ? 1 : 0][k], st_cyc[k]);
      atomicAdd(&g_trace_sections[ANY ? 1 : 0][3 + k], st_cnt[k]);
    }
    atomicAdd(&g_trace_sections[ANY ? 1 : 0][6], 1ull);
    atomicAdd(&g_trace_sections[ANY ? 1 : 0][7], st_trips);
  }
#endif
#ifdef RT_PW_STAMPS
  if (lane == 0u) {
    for (int k = 0; k < 6; k++) atomicAdd(&g_trace_sections[ANY ? 1 : 0][k], pw_cyc[k]);
    atomicAdd(&g_trace_sections[ANY ? 1 : 0][6], 1ull);
  }
#endif
  LaneCounters c = {0, ANY ? 0u : n_traced, ANY ? n_traced : 0u, n_nodes, n_tris, 0};
  flush_counters<DETAIL>(c, F.counters, blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6));
}

// Ordered accumulation of a batched dispatch: acc = (frame_count > 1 ? acc : 0) + (col_f, 1) for f = 0..n-1, the
// exact sequence of f32 additions n separate dispatches perform (Raytracer.wgsl:813-818).
__global__ __launch_bounds__(256) void k_accumulate_frames(DevFrame F, const DevFrameSlot* __restrict__ slots,
                                                           uint32_t n_slots, uint32_t width, uint32_t height) {
  const uint32_t p = blockIdx.x * 256u + threadIdx.x;
  const uint32_t npx = width * height;
  if (p >= npx || !owns_row(F, p / width)) return;
  float4 acc = F.accum[p];
  for (uint32_t f = 0; f < n_slots; f++) {
    const float4 c = F.frame_col[(size_t)f * npx + p];
    if (slots[f].frame_count > 1u)
      acc = make_float4(acc.x + c.x, acc.y + c.y, acc.z + c.z, acc.w + 1.0f);
    else
      acc = make_float4(c.x, c.y, c.z, 1.0f);
  }
  F.accum[p] = acc;
}

}  // namespace rtk
#endif
