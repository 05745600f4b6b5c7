// rt_api.hip — C ABI of include/mi355rt.h: resource management + kernel launches.
//
// Host-side restatement of the reference's ResourceManager / pass objects:
//   buffer growth policy (1.5x, needsRebind)    src/renderer/ResourceManager.ts:209-228
//   uniform packing + Halton jitter             src/renderer/ResourceManager.ts:348-447
//   pass order compute(): raster -> raytrace    src/renderer/WebGPURenderer.ts:88-102
//   present(): post pass, history ping-pong     src/renderer/WebGPURenderer.ts:104-129
// There is no CPU fallback anywhere in this file: without a HIP device rt_create fails.

#include <hip/hip_runtime.h>

#include <execinfo.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <exception>
#include <string>
#include <vector>

#include "../../include/mi355rt.h"
#include "kernels.hip.h"
#include "bvh_build.hip.h"
#include "world_update.hip.h"
#include "k_ieee_check.hip.h"

namespace {

std::string g_create_error;

struct DeviceBuffer {
  void* ptr = nullptr;
  size_t capacity = 0;  // bytes allocated
  size_t size = 0;      // bytes in use
};

struct EventPair {
  hipEvent_t a, b;
};

}  // namespace

struct rt_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  // wavefront form: the any-hit trace of a depth runs on `side_stream` beside the closest-hit trace on `stream` (the two
  // read the same shade output and write different result arrays), so the drain tail of one — a few long rays stepping
  // alone — is filled by the other.  MI355RT_WF_OVERLAP=0 puts both on `stream` again.
  hipStream_t side_stream = nullptr;
  hipEvent_t side_fork = nullptr, side_join = nullptr;
  bool wf_overlap = true;
  int shade_per_cu = 8;          // workgroups per CU of the wavefront shade kernels (MI355RT_SHADE_BLOCKS_PER_CU); swept 4 / 8 /
                                 // 12 / 16 / 24, shade ms per image: sponza-like 64.6 / 63.9 / 63.9 / 64.3 / 64.7, instanced x1000
                                 // 35.9 / 36.4 / 36.7 / 37.4 / 38.7 (every wave may leave a partly used queue chunk per launch)
  bool no_lds_staging = false;   // MI355RT_NO_LDS_STAGING=1 (test hook): every record through the global-memory paths
  std::string error;

  // scene buffers (raw bridge layout)
  DeviceBuffer topology, instances, lights, draw_commands, pos, nrm, uv, nodes, textures, tex_staging;
  uint32_t bv_levels = 0, bv_big_levels = 0, bv_last_tris = 0, bv_gen = 0;   // depth of the last rt_build_blas tree: how many levels the next build launches
  void* bv_pinned = nullptr;  // 64 KB of pinned host memory for the read-back of the build's bookkeeping
  DeviceBuffer bv_in, bv_tri, bv_order, bv_nodes, bv_out, bv_counters, bv_big;  // BLAS build work space
  // device-resident World::update(t) (rt_world_update, csrc/world_update.hip.h)
  struct {
    bool valid = false;
    uint64_t epoch = 0;                   // rt_world_frame::static_epoch the static buffers were made from
    std::vector<wu::Geom> geoms;          // device pointers into `stat`
    std::vector<uint32_t> levels, big_levels;   // per geometry: tree levels / large-node levels the next build launches
    std::vector<float> inst_scale2;       // per instance (declaration order): squared linear scale (treelet weights)
    std::vector<uint32_t> inst_geom;
    uint32_t n_joints = 0, n_verts = 0, n_tris = 0, n_lights = 0, n_tlas = 0, n_inst = 0, n_skins = 0;
    DeviceBuffer stat, joints, raw_inst, geom_rows, node_base, em_flag, em_list, em_blk, tlas_scratch, stats;
    void* pinned = nullptr;               // joint-matrix staging and the end-of-update read-back
    size_t pinned_bytes = 0;
    wu::TlasArgs tlas;
    // static-geometry cache: a geometry without a skin has the same BLAS in every frame of a static description
    bool cache_enabled = true, static_cached = false, any_skinned = false;
    DeviceBuffer static_nodes;
    std::vector<uint32_t> static_count, static_off;   // per geometry: cached node count, first node in static_nodes
    double last_ms = 0;                   // stream time of the last update (rt_world_last_ms)
    double last_tlas_ms = 0;              // ... of its k_tlas launch alone (rt_world_last_tlas_ms)
    bool tlas_lds_set = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_t0 = nullptr, ev_t1 = nullptr;
  } world;
  // derived buffers (device_scene.h)
  DeviceBuffer tri_geom, tri_shade, inst_trav, light_rec;
  DeviceBuffer tnodes, node_key, node_newidx, inst_root, root_w, treelet_work;   // k_treelet.hip.h
  DeviceBuffer pairs, pair_of, pair_parent, root_rec;                             // k_pairs.hip.h
  uint32_t n_pairs = 0;             // inner nodes of the uploaded TLAS ++ BLAS arrays (counted on the host at upload)
  rtk::TlasRoot troot = {};         // box and word of the TLAS root (node 0), passed to the trace kernels by value
  bool roots_dirty = true;          // root records: rebuilt on every instance upload (cheap); the pair records only when the
                                    // node array or the SET of BLAS roots changed
  std::vector<float> root_w_host;                                   // per entry of blas_roots: sum of squared instance scales
  bool tris_dirty = true, inst_dirty = true, lights_dirty = true, nodes_dirty = true, pairs_dirty = true;
  bool pairs_wanted = false;        // rt_debug_read_pairs: build the pair records whatever walk is selected
  int wf_block = 0;              // threads per workgroup of the wavefront trace kernels (0 = default; MI355RT_WF_BLOCK)
  size_t lds_per_cu = 160 * 1024;
  bool validate_dirty = true, scene_valid = false;  // k_validate_scene: run once per upload
  std::string scene_problem;
  std::vector<uint32_t> blas_roots;                 // sorted unique BLAS-local root offsets of the instances
  DeviceBuffer val_roots, val_bad;
  uint32_t n_tris = 0, n_instances = 0, n_lights = 0, n_verts = 0, n_nodes = 0, tex_layers = 0;
  std::vector<uint32_t> draw_commands_host;  // kept like ResourceManager.drawCommandsArray

  // screen resources
  uint32_t width = 0, height = 0;
  DeviceBuffer accum, render_target, g_normal, g_depth, history[2], counters;
  void* external_accum = nullptr;
  void* present_source = nullptr;   // rt_bind_present_source: present() reads this instead of the accumulation buffer
  bool accum_stale = false;         // rt_bind_accum's buffer was sized for the screen before the last rt_resize ...
  bool present_stale = false;       // ... and so was rt_bind_present_source's: each is cleared only by its own bind call
  int history_index = 0;

  // uniforms + host state (ResourceManager fields)
  rt_scene_uniforms uniforms;
  float prev_camera[24];
  double acc_jx = 0, acc_jy = 0, jx = 0, jy = 0, avg_jx = 0, avg_jy = 0;
  uint32_t blas_offset = 0, vertex_count = 0, light_count = 0;
  uint32_t total_frames = 0;  // WebGPURenderer.totalFrames
  uint32_t max_depth = 10, spp = 1;
  bool pipeline_built = false;
  bool detailed_counters = false;
  uint32_t stripe_rows = 0, stripe_rank = 0, stripe_count = 1;
  int variant = 3;        // 3 = auto (default): persistent kernel for LDS-resident scenes, wavefront for larger ones;
                          // 0 = one-pixel-per-lane megakernel, 1 = persistent kernel, 2 = wavefront
  int num_cus = 256;      // multiProcessorCount of the device
  int occ_blocks[4] = {0, 0, 0, 0};   // cached occupancy query per persistent-kernel variant
  int wf_occ_blocks[2] = {0, 0};      // ... and for the two wavefront trace kernels
  size_t wf_occ_dyn = (size_t)-1;
  int wf_occ_detail = -1, wf_occ_block = 0, wf_occ_walk = -1;
  int walk = 2;                  // traversal of the wavefront trace kernels: 1 = child-pair records, 0 = single nodes, 2 = auto
                                 // (MI355RT_WALK): pairs for a scene of ONE instance (measured: the 263 k-triangle hall -9 % per
                                 // batch; glass blob, 2 instances and short walks: +7 %; 1 001 instances of 8 triangles: +20 %)
  int wf_blocks_per_cu = 0;      // 0 = default for the block size (MI355RT_WF_BLOCKS_PER_CU)
  int wf_rayreg = -1;            // node-walk trace kernels: -1 = by scene, 0 / 1 = MI355RT_WF_RAYREG
  long treelet_cap = -1;
  int treelet_order = 2;          // order of tnodes: 0 = by visit probability, 1 = the bridge's depth-first order, 2 = auto (MI355RT_TREELET_ORDER)
  bool nodes_from_device = false; // the node array was made by rt_world_update (an animated world), not uploaded
  size_t occ_dyn[4] = {0, 0, 0, 0};
  DeviceBuffer ticket;    // tile ticket counter of the persistent kernel
  DeviceBuffer slots;     // DevFrameSlot table of the current (batched) dispatch
  // pinned staging ring for the slot tables: the H2D copy of a dispatch's table is truly asynchronous and its source
  // outlives it (entry k is reused only after the event recorded behind its copy has completed)
  static constexpr int kSlotRing = 8;
  DevFrameSlot* slot_ring = nullptr;   // kSlotRing x 64 slots, hipHostMalloc
  hipEvent_t slot_ring_ev[kSlotRing] = {};
  bool slot_ring_used[kSlotRing] = {};
  int slot_ring_next = 0;
  // Speculative lookahead of the live loop (rt_set_lookahead): a compute(f) that continues a run f-1, f traces the frames
  // f .. f+L-1 as ONE batched dispatch (L doubles while the run goes on) and accumulates only frame f; the following
  // compute(f+1) ... find their frame colours ready and only add them.  Any call that changes what a frame looks like
  // bumps `epoch` and drops what was traced ahead.
  uint32_t lookahead_max = 0;       // 0 / 1: off
  uint64_t epoch = 0;
  struct {
    bool valid = false;
    uint64_t epoch = 0;
    uint32_t n = 0, k = 0;          // frames traced, frames consumed
    uint32_t fc0 = 0, tf0 = 0;      // frame_count / totalFrames of the first traced frame
    const DevFrameSlot* dslots = nullptr;
    std::vector<DevFrameSlot> slots;   // host copy (jitter check, G-buffer planes of each frame)
    DevFrame frame;
  } spec;
  uint32_t run_len = 0, run_last_fc = 0, run_last_tf = 0;   // the run of consecutive single-frame compute() calls
  uint64_t run_epoch = 0;
  uint32_t look = 1;                // frames the next dispatch of the run traces
  uint32_t look_limit = 0;          // rt_set_lookahead_limit: frames (this one included) until the caller will end the run; 0 = unknown
  int gbuf_slot = -1;               // >= 0: the last compute() consumed this frame of the traced batch (rt_read_gbuffer)
  uint32_t acc_frames = 0;          // frames k_accumulate_frames adds at the end of a dispatch (all of them unless traced ahead)
  DeviceBuffer gbuf_batch;  // G-buffer planes of frames 0..n-2 of a batch (the last frame uses the main planes)
  DeviceBuffer frame_col;   // per-frame colours of a batch, added in frame order by k_accumulate_frames
  DeviceBuffer wf_state, wf_queues, wf_counters;  // wavefront form: path state, ray / path queues, queue counters

  // kernel timing
  bool timing = false;
  std::deque<EventPair> ev_pool;   // deque: pointers handed out by next_events stay valid while later timers nest inside
  size_t ev_used = 0;
  std::vector<std::pair<size_t, int>> ev_tags;  // (pool index, RT_TIMER_* of mi355rt.h)

  rt_ctx() {
    std::memset(&uniforms, 0, sizeof(uniforms));
    std::memset(prev_camera, 0, sizeof(prev_camera));
  }
};

namespace {

int fail(rt_ctx* c, int code, const std::string& msg) {
  if (c) c->error = msg;
  return code;
}
int hip_fail(rt_ctx* c, hipError_t e, const char* what) {
  char buf[256];
  snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
  return fail(c, RT_ERR_HIP, buf);
}
#define HIP_TRY(ctx, expr)                                   \
  do {                                                       \
    hipError_t _e = (expr);                                  \
    if (_e != hipSuccess) return hip_fail((ctx), _e, #expr); \
  } while (0)

// ensureBuffer (ResourceManager.ts:209-228): keep when large enough, else reallocate at 1.5x.
// Returns 1 when the buffer was (re)allocated, 0 when kept, <0 on error.
int ensure_buffer(rt_ctx* c, DeviceBuffer& b, size_t bytes, bool grow_policy) {
  b.size = bytes;
  if (b.ptr && b.capacity >= bytes) return 0;
  if (b.ptr) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipFree(b.ptr));
    b.ptr = nullptr;
  }
  size_t cap = bytes;
  if (grow_policy) {
    cap = (size_t)std::ceil((double)bytes * 1.5);
    cap = (cap + 3) & ~(size_t)3;
  }
  if (cap < 16) cap = 16;
  // 16 bytes of slack beyond the reported capacity: kernels stage 8-byte arrays in 16-byte slots
  HIP_TRY(c, hipMalloc(&b.ptr, cap + 16));
  b.capacity = cap;
  return 1;
}
void free_buffer(DeviceBuffer& b) {
  if (b.ptr) (void)hipFree(b.ptr);
  b = DeviceBuffer();
}
int upload(rt_ctx* c, DeviceBuffer& b, const void* src, size_t bytes) {
  int r = ensure_buffer(c, b, bytes, true);
  if (r < 0) return r;
  if (bytes) {
    // queue.writeBuffer semantics: the source may be reused right after return
    HIP_TRY(c, hipMemcpyAsync(b.ptr, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
  }
  return r;
}

double halton(uint32_t index, uint32_t base) {  // ResourceManager.ts:348-357 (JS doubles)
  double f = 1, r = 0;
  while (index > 0) {
    f = f / (double)base;
    r = r + f * (double)(index % base);
    index = index / base;
  }
  return r;
}
// jitter + running average, shared by updateSceneUniforms (:366-371,385-394) and updateFrameUniforms (:408-424)
void step_jitter(rt_ctx* c, uint32_t halton_source, uint32_t frame_count) {
  c->jx = (halton((halton_source % 16u) + 1u, 2) - 0.5) / (double)c->width;
  c->jy = (halton((halton_source % 16u) + 1u, 3) - 0.5) / (double)c->height;
  if (frame_count == 1u) {
    c->acc_jx = c->jx;
    c->acc_jy = c->jy;
  } else {
    c->acc_jx += c->jx;
    c->acc_jy += c->jy;
  }
  c->avg_jx = c->acc_jx / (double)frame_count;
  c->avg_jy = c->acc_jy / (double)frame_count;
}
void write_mixed(rt_ctx* c, uint32_t frame_count) {  // the 48 bytes at offset 192
  rt_scene_uniforms& u = c->uniforms;
  u.frame_count = frame_count;
  u.blas_base_idx = c->blas_offset;
  u.vertex_count = c->vertex_count;
  u.rand_seed = 0;  // Math.random() in the reference; no shader reads it
  u.light_count = c->light_count;
  u.width = c->width;
  u.height = c->height;
  u.pad = 0;
  u.jitter[0] = (float)c->jx;
  u.jitter[1] = (float)c->jy;
  u.average_jitter[0] = (float)c->avg_jx;
  u.average_jitter[1] = (float)c->avg_jy;
}

float4* accum_ptr(rt_ctx* c) { return (float4*)(c->external_accum ? c->external_accum : c->accum.ptr); }

DevScene dev_scene(const rt_ctx* c);

// Every index the kernels follow, checked on the GPU once per upload (k_validate.hip.h): a malformed scene is refused
// here instead of faulting or hanging a kernel.
static int validate_scene(rt_ctx* c) {
  if (!c->validate_dirty) return c->scene_valid ? RT_OK : fail(c, RT_ERR_INVALID, c->scene_problem.c_str());
  int r = ensure_buffer(c, c->val_bad, 32, false);
  if (r < 0) return r;
  r = ensure_buffer(c, c->val_roots, std::max<size_t>(4, c->blas_roots.size() * 4), false);
  if (r < 0) return r;
  HIP_TRY(c, hipMemsetAsync(c->val_bad.ptr, 0, 32, c->stream));
  if (!c->blas_roots.empty())
    HIP_TRY(c, hipMemcpyAsync(c->val_roots.ptr, c->blas_roots.data(), c->blas_roots.size() * 4, hipMemcpyHostToDevice, c->stream));
  rtk::ValidateArgs A;
  A.topo = (const float4*)c->topology.ptr;
  A.nodes = (const float4*)c->nodes.ptr;
  A.inst = (const float4*)c->instances.ptr;
  A.lights = (const uint2*)c->lights.ptr;
  A.roots = (const uint32_t*)c->val_roots.ptr;
  A.n_tris = c->n_tris;
  A.n_verts = c->n_verts;
  A.n_nodes = c->n_nodes;
  A.n_tlas = c->blas_offset;
  A.n_inst = c->n_instances;
  A.n_lights = c->n_lights;
  A.n_roots = (uint32_t)c->blas_roots.size();
  A.bad = (uint32_t*)c->val_bad.ptr;
  const uint32_t n = std::max(std::max(c->n_tris, c->n_nodes), std::max(c->n_instances, c->n_lights));
  if (n) hipLaunchKernelGGL(rtk::k_validate_scene, dim3((n + 255) / 256), dim3(256), 0, c->stream, A);
  HIP_TRY(c, hipGetLastError());
  uint32_t bad[5] = {0, 0, 0, 0, 0};
  HIP_TRY(c, hipMemcpyAsync(bad, c->val_bad.ptr, sizeof(bad), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->validate_dirty = false;
  c->scene_valid = !(bad[0] | bad[1] | bad[2] | bad[3] | bad[4]);
  if (!c->scene_valid) {
    c->scene_problem = "scene arrays refer to elements out of range: " + std::to_string(bad[0]) + " triangles (vertex ids), " +
                       std::to_string(bad[1]) + " TLAS nodes, " + std::to_string(bad[2]) + " BLAS nodes, " +
                       std::to_string(bad[3]) + " instances (BLAS offset), " + std::to_string(bad[4]) + " light references";
    return fail(c, RT_ERR_INVALID, c->scene_problem.c_str());
  }
  return RT_OK;
}

int prepare_scene(rt_ctx* c) {
  {
    int r = validate_scene(c);
    if (r < 0) return r;
  }
  if (c->tris_dirty && c->n_tris && c->n_verts) {
    int r = ensure_buffer(c, c->tri_geom, (size_t)c->n_tris * 16 * RT_TRI_STRIDE, true);
    if (r < 0) return r;
    hipLaunchKernelGGL(rtk::k_prepare_tris, dim3((c->n_tris + 255) / 256), dim3(256), 0, c->stream,
                       (const float4*)c->topology.ptr, (const float4*)c->pos.ptr, (float4*)c->tri_geom.ptr,
                       c->n_tris, c->n_verts);
    HIP_TRY(c, hipGetLastError());
    r = ensure_buffer(c, c->tri_shade, (size_t)c->n_tris * 128, true);
    if (r < 0) return r;
    hipLaunchKernelGGL(rtk::k_prepare_tri_shade, dim3((c->n_tris + 255) / 256), dim3(256), 0, c->stream,
                       (const float4*)c->topology.ptr, (const float4*)c->nrm.ptr, (const float2*)c->uv.ptr,
                       (float4*)c->tri_shade.ptr, c->n_tris, c->n_verts);
    HIP_TRY(c, hipGetLastError());
    c->tris_dirty = false;
  }
  if (c->inst_dirty && c->n_instances) {
    int r = ensure_buffer(c, c->inst_trav, (size_t)c->n_instances * 64, true);
    if (r < 0) return r;
    hipLaunchKernelGGL(rtk::k_prepare_instances, dim3((c->n_instances + 255) / 256), dim3(256), 0, c->stream,
                       (const float4*)c->instances.ptr, (float4*)c->inst_trav.ptr, c->n_instances);
    HIP_TRY(c, hipGetLastError());
    c->inst_dirty = false;
  }
  if (c->nodes_dirty && c->n_nodes && c->n_instances) {
    // traversal copy of the node array: explicit successors, most-visited nodes first (k_treelet.hip.h); validate_scene
    // has just uploaded the sorted BLAS roots into val_roots and vouches for every pointer followed here
    int r = ensure_buffer(c, c->tnodes, (size_t)c->n_nodes * 32, true);
    if (r < 0) return r;
    if ((r = ensure_buffer(c, c->node_key, (size_t)c->n_nodes * 4, true)) < 0) return r;
    if ((r = ensure_buffer(c, c->node_newidx, (size_t)c->n_nodes * 4, true)) < 0) return r;
    if ((r = ensure_buffer(c, c->inst_root, (size_t)c->n_instances * 4, true)) < 0) return r;
    if ((r = ensure_buffer(c, c->root_w, std::max<size_t>(4, c->root_w_host.size() * 4), true)) < 0) return r;
    if (!c->root_w_host.empty())
      HIP_TRY(c, hipMemcpyAsync(c->root_w.ptr, c->root_w_host.data(), c->root_w_host.size() * 4, hipMemcpyHostToDevice, c->stream));
    rtk::TreeletArgs T;
    T.nodes = (const float4*)c->nodes.ptr;
    T.tnodes = (float4*)c->tnodes.ptr;
    T.key = (uint32_t*)c->node_key.ptr;
    T.new_index = (uint32_t*)c->node_newidx.ptr;
    T.roots = (const uint32_t*)c->val_roots.ptr;
    T.root_w = (const float*)c->root_w.ptr;
    T.n_nodes = c->n_nodes;
    T.n_tlas = c->blas_offset;
    T.n_roots = (uint32_t)c->blas_roots.size();
    T.k_max = (uint32_t)(c->lds_per_cu / 32);
    const dim3 grid((c->n_nodes + 255) / 256);
    const uint32_t n_blocks = (c->n_nodes + 1023u) / 1024u;
    if ((r = ensure_buffer(c, c->treelet_work, ((size_t)RT_TREELET_WORK_HEAD + n_blocks) * 4, true)) < 0) return r;
    uint32_t* work = (uint32_t*)c->treelet_work.ptr;
    HIP_TRY(c, hipMemsetAsync(work, 0, (size_t)RT_TREELET_WORK_HEAD * 4, c->stream));
    const dim3 hgrid(std::min<uint32_t>((c->n_nodes + 255) / 256, 256u));
    // A device-resident update(t) (rt_world_update) re-lays the nodes out on every displayed frame: unless a partial
    // treelet is asked for (MI355RT_TREELET_MAX) the visit-probability order buys nothing there, so it keeps the bridge's
    // depth-first order: 3 launches instead of 10 (MI355RT_TREELET_ORDER=weight / identity forces either, for measurements)
    const bool identity = c->treelet_order == 1 || (c->treelet_order == 2 && c->nodes_from_device && c->treelet_cap < 0);
    if (identity) {
      hipLaunchKernelGGL(rtk::k_treelet_iota, grid, dim3(256), 0, c->stream, T.new_index, c->n_nodes);
    } else {
      hipLaunchKernelGGL(rtk::k_treelet_weight, grid, dim3(256), 0, c->stream, T);
      hipLaunchKernelGGL(rtk::k_treelet_hist<0>, hgrid, dim3(256), 0, c->stream, T, work);
      hipLaunchKernelGGL(rtk::k_treelet_pick<0>, dim3(1), dim3(256), 0, c->stream, T, work);
      hipLaunchKernelGGL(rtk::k_treelet_hist<1>, hgrid, dim3(256), 0, c->stream, T, work);
      hipLaunchKernelGGL(rtk::k_treelet_pick<1>, dim3(1), dim3(256), 0, c->stream, T, work);
      hipLaunchKernelGGL(rtk::k_treelet_count, dim3(n_blocks), dim3(1024), 0, c->stream, T, work);
      hipLaunchKernelGGL(rtk::k_treelet_blockscan, dim3(1), dim3(1024), 0, c->stream, work, n_blocks);
      hipLaunchKernelGGL(rtk::k_treelet_number, dim3(n_blocks), dim3(1024), 0, c->stream, T, (const uint32_t*)work);
    }
    hipLaunchKernelGGL(rtk::k_treelet_remap, grid, dim3(256), 0, c->stream, T);
    hipLaunchKernelGGL(rtk::k_treelet_inst_roots, dim3((c->n_instances + 255) / 256), dim3(256), 0, c->stream,
                       (const float4*)c->instances.ptr, (const uint32_t*)c->node_newidx.ptr, (uint32_t*)c->inst_root.ptr,
                       c->n_instances, c->blas_offset, c->n_nodes);
    HIP_TRY(c, hipGetLastError());
    c->nodes_dirty = false;
  }
  // the pair records are only walked by the wavefront trace kernels under the pair walk (rt_set_walk): a scene that takes
  // the node walk does not pay for them on every update(t); they are made when a launch (or rt_debug_read_pairs) wants them
  const bool want_pairs = c->pairs_wanted || c->walk == 1 || (c->walk == 2 && c->n_instances == 1);
  if (want_pairs && (c->pairs_dirty || c->roots_dirty) && c->n_nodes && c->n_instances) {
    // child-pair records of the walk (k_pairs.hip.h); validate_scene has uploaded the sorted BLAS roots into val_roots and
    // vouches for every pointer followed here.  An instance upload that keeps the set of BLAS roots only redoes the root
    // records (one small launch) — what an animated scene pays per update(t).
    int r;
    if ((r = ensure_buffer(c, c->pairs, std::max<size_t>(64, (size_t)c->n_pairs * 64), true)) < 0) return r;
    if ((r = ensure_buffer(c, c->pair_of, (size_t)c->n_nodes * 4, true)) < 0) return r;
    if ((r = ensure_buffer(c, c->pair_parent, (size_t)c->n_nodes * 4, true)) < 0) return r;
    if ((r = ensure_buffer(c, c->root_rec, ((size_t)c->n_instances + 1) * 32, true)) < 0) return r;
    rtk::PairArgs P;
    P.nodes = (const float4*)c->nodes.ptr;
    P.pairs = (float4*)c->pairs.ptr;
    P.pair_of = (uint32_t*)c->pair_of.ptr;
    P.parent = (uint32_t*)c->pair_parent.ptr;
    P.roots = (const uint32_t*)c->val_roots.ptr;
    P.n_nodes = c->n_nodes;
    P.n_tlas = c->blas_offset;
    P.n_roots = (uint32_t)c->blas_roots.size();
    P.pad = 0;
    if (c->pairs_dirty) {
      const uint32_t n_blocks = (c->n_nodes + 1023u) / 1024u;
      if ((r = ensure_buffer(c, c->treelet_work, ((size_t)RT_TREELET_WORK_HEAD + n_blocks) * 4, true)) < 0) return r;
      uint32_t* work = (uint32_t*)c->treelet_work.ptr;
      const dim3 grid((c->n_nodes + 255) / 256);
      hipLaunchKernelGGL(rtk::k_pair_count, dim3(n_blocks), dim3(1024), 0, c->stream, P, work);
      hipLaunchKernelGGL(rtk::k_treelet_blockscan, dim3(1), dim3(1024), 0, c->stream, work, n_blocks);
      hipLaunchKernelGGL(rtk::k_pair_number, dim3(n_blocks), dim3(1024), 0, c->stream, P, (const uint32_t*)work);
      hipLaunchKernelGGL(rtk::k_pair_parent, grid, dim3(256), 0, c->stream, P);
      hipLaunchKernelGGL(rtk::k_pair_emit, grid, dim3(256), 0, c->stream, P);
      c->pairs_dirty = false;
    }
    hipLaunchKernelGGL(rtk::k_pair_roots, dim3((c->n_instances + 256) / 256), dim3(256), 0, c->stream, P,
                       (const float4*)c->instances.ptr, (float4*)c->root_rec.ptr, c->n_instances);
    HIP_TRY(c, hipGetLastError());
    c->roots_dirty = false;
  }
  if (c->lights_dirty && c->n_lights && c->n_tris && c->n_instances && c->n_verts) {
    int r = ensure_buffer(c, c->light_rec, (size_t)c->n_lights * 64, true);
    if (r < 0) return r;
    DevScene S = dev_scene(c);
    hipLaunchKernelGGL(rtk::k_prepare_lights, dim3((c->n_lights + 255) / 256), dim3(256), 0, c->stream, S,
                       (float4*)c->light_rec.ptr, c->n_lights, c->n_tris, c->n_instances);
    HIP_TRY(c, hipGetLastError());
    c->lights_dirty = false;
  }
  return RT_OK;
}

// Host-side validation of everything the kernels index, so a malformed upload is refused
// instead of faulting the GPU.
bool scene_ready(const rt_ctx* c) {
  return c->pipeline_built && c->width && c->height && c->n_tris && c->n_verts && c->n_instances && c->n_nodes &&
         c->accum.ptr && c->lights.ptr;  // RaytracePass.updateBindGroup requires lightsBuffer (:38-46)
}

DevScene dev_scene(const rt_ctx* c) {
  DevScene s;
  s.nodes = (const float4*)c->nodes.ptr;
  s.tnodes = (const float4*)c->tnodes.ptr;
  s.inst_root = (const uint32_t*)c->inst_root.ptr;
  s.pairs = (const float4*)c->pairs.ptr;
  s.root_rec = (const float4*)c->root_rec.ptr;
  s.tri_geom = (const float4*)c->tri_geom.ptr;
  s.tri_shade = (const float4*)c->tri_shade.ptr;
  s.inst_trav = (const float4*)c->inst_trav.ptr;
  s.topo = (const float4*)c->topology.ptr;
  s.pos = (const float4*)c->pos.ptr;
  s.nrm = (const float4*)c->nrm.ptr;
  s.uv = (const float2*)c->uv.ptr;
  s.inst = (const float4*)c->instances.ptr;
  s.lights = (const uint2*)c->lights.ptr;
  s.light_rec = (const float4*)c->light_rec.ptr;
  s.tex = c->tex_layers ? (const uint8_t*)c->textures.ptr : nullptr;
  s.tex_layers = c->tex_layers;
  s.n_lights = c->n_lights;
  return s;
}

EventPair* next_events(rt_ctx* c, int tag) {
  if (!c->timing) return nullptr;
  if (c->ev_used == c->ev_pool.size()) {
    EventPair p;
    if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return nullptr;
    c->ev_pool.push_back(p);
  }
  c->ev_tags.emplace_back(c->ev_used, tag);
  return &c->ev_pool[c->ev_used++];
}

}  // namespace

extern "C" {

int rt_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// MI355RT_DEBUG_TERMINATE=1: print a native backtrace when an exception escapes (the HIP runtime throws from inside
// some calls; without this only "terminate called" is seen).
static void debug_terminate() {
  void* frames[64];
  const int n = backtrace(frames, 64);
  fprintf(stderr, "mi355rt: std::terminate, native backtrace:\n");
  backtrace_symbols_fd(frames, n, 2);
  abort();
}

rt_ctx* rt_create(int device_ordinal) {
  if (const char* dbg = getenv("MI355RT_DEBUG_TERMINATE"))
    if (dbg[0] == '1') std::set_terminate(debug_terminate);
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    g_create_error = "no HIP device available (the MI355X renderer has no CPU fallback)";
    return nullptr;
  }
  if (device_ordinal < 0 || device_ordinal >= n) {
    g_create_error = "device ordinal out of range";
    return nullptr;
  }
  if (hipSetDevice(device_ordinal) != hipSuccess) {
    g_create_error = "hipSetDevice failed";
    return nullptr;
  }
  rt_ctx* c = new rt_ctx();
  c->device = device_ordinal;
  if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
    g_create_error = "hipStreamCreate failed";
    delete c;
    return nullptr;
  }
  c->stream = c->own_stream;
  if (hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&c->side_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->side_join, hipEventDisableTiming) != hipSuccess) {
    g_create_error = "hipStreamCreate / hipEventCreate (side stream) failed";
    if (c->side_fork) (void)hipEventDestroy(c->side_fork);
    if (c->side_stream) (void)hipStreamDestroy(c->side_stream);
    (void)hipStreamDestroy(c->own_stream);
    delete c;
    return nullptr;
  }
  if (const char* e = getenv("MI355RT_WF_OVERLAP")) c->wf_overlap = atoi(e) != 0;
  if (const char* e = getenv("MI355RT_NO_LDS_STAGING")) c->no_lds_staging = atoi(e) != 0;
  if (const char* e = getenv("MI355RT_SHADE_BLOCKS_PER_CU")) c->shade_per_cu = std::max(1, std::min(4096, atoi(e)));
  if (const char* e = getenv("MI355RT_WF_BLOCK")) {
    const int b = atoi(e);
    if (b == 256 || b == 512 || b == 1024) c->wf_block = b;
  }
  if (const char* e = getenv("MI355RT_TREELET_MAX")) c->treelet_cap = atol(e);
  if (const char* e = getenv("MI355RT_TREELET_ORDER")) c->treelet_order = e[0] == 'w' ? 0 : (e[0] == 'i' ? 1 : 2);
  if (const char* e = getenv("MI355RT_WALK")) c->walk = (e[0] == 'n' || e[0] == '0') ? 0 : ((e[0] == 'a' || e[0] == '2') ? 2 : 1);   // node / pairs / auto
  if (const char* e = getenv("MI355RT_WF_RAYREG")) c->wf_rayreg = atoi(e) != 0 ? 1 : 0;
  if (const char* e = getenv("MI355RT_WF_BLOCKS_PER_CU")) {
    const int b = atoi(e);
    if (b >= 1 && b <= 8) c->wf_blocks_per_cu = b;
  }
  // counters: 2 banks (primary kernel, path-trace kernel) x RT_COUNTER_SHARDS x 6 u64
  if (hipMalloc(&c->counters.ptr, 2 * RT_COUNTER_SHARDS * 6 * sizeof(uint64_t)) != hipSuccess) {
    g_create_error = "hipMalloc(counters) failed";
    (void)hipStreamDestroy(c->own_stream);
    delete c;
    return nullptr;
  }
  c->counters.capacity = c->counters.size = 2 * RT_COUNTER_SHARDS * 6 * sizeof(uint64_t);
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_ordinal) == hipSuccess && prop.multiProcessorCount > 0)
      c->num_cus = prop.multiProcessorCount;
    (void)hipMalloc(&c->ticket.ptr, 256);
    c->ticket.capacity = c->ticket.size = 256;
  }
  (void)hipMemsetAsync(c->counters.ptr, 0, c->counters.size, c->stream);
  // lights buffer exists from the start with one dummy entry so that light_count == 0 scenes run
  (void)hipMalloc(&c->lights.ptr, 16);
  c->lights.capacity = 16;
  (void)hipMemsetAsync(c->lights.ptr, 0, 16, c->stream);
  (void)hipStreamSynchronize(c->stream);
  return c;
}

void rt_destroy(rt_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  DeviceBuffer* all[] = {&c->topology, &c->instances, &c->lights, &c->draw_commands, &c->pos, &c->nrm, &c->uv,
                         &c->nodes, &c->textures, &c->tri_geom, &c->tri_shade, &c->inst_trav, &c->light_rec, &c->accum, &c->render_target,
                         &c->g_normal, &c->g_depth, &c->history[0], &c->history[1], &c->counters, &c->ticket,
                         &c->slots, &c->gbuf_batch, &c->frame_col, &c->wf_state, &c->wf_queues, &c->wf_counters,
                         &c->tex_staging, &c->bv_in, &c->bv_tri, &c->bv_order, &c->bv_nodes, &c->bv_out,
                         &c->bv_counters, &c->bv_big, &c->val_roots, &c->val_bad, &c->tnodes, &c->node_key, &c->node_newidx,
                         &c->inst_root, &c->root_w, &c->treelet_work, &c->pairs, &c->pair_of, &c->pair_parent, &c->root_rec,
                         &c->world.stat, &c->world.joints, &c->world.raw_inst, &c->world.geom_rows, &c->world.node_base,
                         &c->world.em_flag, &c->world.em_list, &c->world.em_blk, &c->world.tlas_scratch, &c->world.stats,
                         &c->world.static_nodes};
  for (DeviceBuffer* b : all) free_buffer(*b);
  if (c->world.pinned) (void)hipHostFree(c->world.pinned);
  if (c->world.ev0) (void)hipEventDestroy(c->world.ev0);
  if (c->world.ev1) (void)hipEventDestroy(c->world.ev1);
  if (c->world.ev_t0) (void)hipEventDestroy(c->world.ev_t0);
  if (c->world.ev_t1) (void)hipEventDestroy(c->world.ev_t1);
  for (EventPair& p : c->ev_pool) {
    (void)hipEventDestroy(p.a);
    (void)hipEventDestroy(p.b);
  }
  if (c->bv_pinned) (void)hipHostFree(c->bv_pinned);
  if (c->slot_ring) (void)hipHostFree(c->slot_ring);
  for (int k = 0; k < rt_ctx::kSlotRing; k++)
    if (c->slot_ring_ev[k]) (void)hipEventDestroy(c->slot_ring_ev[k]);
  if (c->side_join) (void)hipEventDestroy(c->side_join);
  if (c->side_fork) (void)hipEventDestroy(c->side_fork);
  if (c->side_stream) (void)hipStreamDestroy(c->side_stream);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
}

const char* rt_last_error(const rt_ctx* c) { return c ? c->error.c_str() : g_create_error.c_str(); }

int rt_set_pipeline(rt_ctx* c, uint32_t max_depth, uint32_t spp) {
  if (!c) return RT_ERR_INVALID;
  c->epoch++;   // drops frames traced ahead (rt_set_lookahead)
  if (spp == 0) return fail(c, RT_ERR_INVALID, "SPP must be >= 1");
  c->max_depth = max_depth;
  c->spp = spp;
  c->pipeline_built = true;
  return RT_OK;
}

int rt_resize(rt_ctx* c, uint32_t width, uint32_t height) {
  if (!c) return RT_ERR_INVALID;
  c->epoch++;   // drops frames traced ahead (rt_set_lookahead) ...
  // ... and what rt_read_gbuffer remembers of them: their planes are freed / sized for the old screen below
  c->gbuf_slot = -1;
  c->spec.valid = false;
  c->spec.slots.clear();
  // 65535: the persistent kernel packs a pixel's x and y into one word (WebGPU's maxTextureDimension2D is 8192)
  if (width == 0 || height == 0 || width > 65535u || height > 65535u || (uint64_t)width * height > (1ull << 28))
    return fail(c, RT_ERR_INVALID, "invalid screen size");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  const size_t n = (size_t)width * height;
  // updateScreenSize destroys and recreates every screen resource (ResourceManager.ts:97-142);
  // new WebGPU resources are zero-initialised.
  struct {
    DeviceBuffer* b;
    size_t bytes;
  } items[] = {{&c->accum, n * 16}, {&c->render_target, n * 4}, {&c->g_normal, n * 16},
               {&c->g_depth, n * 4}, {&c->history[0], n * 8},   {&c->history[1], n * 8}};
  for (auto& it : items) {
    free_buffer(*it.b);
    int r = ensure_buffer(c, *it.b, it.bytes, false);
    if (r < 0) return r;
    HIP_TRY(c, hipMemsetAsync(it.b->ptr, 0, it.bytes, c->stream));
  }
  c->width = width;
  c->height = height;
  // A bound accumulation / present buffer was sized for the old screen: it is dropped, and until the caller binds
  // again (rt_bind_accum / rt_bind_present_source, NULL included) compute() and present() refuse to run instead of
  // silently rendering into the internal buffer while the caller keeps reducing its stale one.
  if (c->external_accum) c->accum_stale = true;
  if (c->present_source) c->present_stale = true;
  c->external_accum = nullptr;
  c->present_source = nullptr;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return RT_OK;
}

int rt_reset_accum(rt_ctx* c) {
  if (!c) return RT_ERR_INVALID;
  if (!c->accum.ptr) return RT_OK;  // `if (!this.accumulateBuffer) return;`
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipMemsetAsync(accum_ptr(c), 0, (size_t)c->width * c->height * 16, c->stream));
  return RT_OK;
}

int rt_upload_textures(rt_ctx* c, const uint8_t* rgba, uint32_t layers) {
  if (!c) return RT_ERR_INVALID;
  c->epoch++;   // drops frames traced ahead (rt_set_lookahead)
  HIP_TRY(c, hipSetDevice(c->device));
  if (layers == 0) {
    c->tex_layers = 0;
    return RT_OK;
  }
  if (!rgba) return fail(c, RT_ERR_INVALID, "null texture data");
  const size_t bytes = (size_t)layers * RT_TEX_SIZE * RT_TEX_SIZE * 4;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  free_buffer(c->textures);
  int r = ensure_buffer(c, c->textures, bytes, false);
  if (r < 0) return r;
  HIP_TRY(c, hipMemcpyAsync(c->textures.ptr, rgba, bytes, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->tex_layers = layers;
  return RT_OK;
}

int rt_alloc_texture_layers(rt_ctx* c, uint32_t layers) {
  if (!c) return RT_ERR_INVALID;
  c->epoch++;   // drops frames traced ahead (rt_set_lookahead)
  HIP_TRY(c, hipSetDevice(c->device));
  if (layers == 0) {
    c->tex_layers = 0;
    return RT_OK;
  }
  const size_t bytes = (size_t)layers * RT_TEX_SIZE * RT_TEX_SIZE * 4;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  free_buffer(c->textures);
  int r = ensure_buffer(c, c->textures, bytes, false);
  if (r < 0) return r;
  HIP_TRY(c, hipMemsetAsync(c->textures.ptr, 0xff, bytes, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->tex_layers = layers;
  return RT_OK;
}

int rt_upload_texture_image(rt_ctx* c, uint32_t layer, const uint8_t* rgba, uint32_t width, uint32_t height) {
  if (!c) return RT_ERR_INVALID;
  c->epoch++;   // drops frames traced ahead (rt_set_lookahead)
  if (layer >= c->tex_layers || !c->textures.ptr) return fail(c, RT_ERR_INVALID, "texture layer out of range");
  if (rgba && (width == 0 || height == 0 || width > 32768u || height > 32768u))
    return fail(c, RT_ERR_INVALID, "texture image size out of range");
  HIP_TRY(c, hipSetDevice(c->device));
  uint32_t* dst = (uint32_t*)c->textures.ptr + (size_t)layer * RT_TEX_SIZE * RT_TEX_SIZE;
  const uint32_t* src = nullptr;
  if (rgba) {
    const size_t bytes = (size_t)width * height * 4;
    int r = ensure_buffer(c, c->tex_staging, bytes, false);
    if (r < 0) return r;
    HIP_TRY(c, hipMemcpyAsync(c->tex_staging.ptr, rgba, bytes, hipMemcpyHostToDevice, c->stream));
    src = (const uint32_t*)c->tex_staging.ptr;
  }
  hipLaunchKernelGGL(rtk::k_resize_texture, dim3(RT_TEX_SIZE / 256, RT_TEX_SIZE), dim3(256), 0, c->stream, src, width, height,
                     dst);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipStreamSynchronize(c->stream));  // the caller may free `rgba` and the staging buffer is reused
  return RT_OK;
}

int rt_read_texture_layer(rt_ctx* c, uint32_t layer, uint8_t* out, size_t cap) {
  if (!c || !out) return RT_ERR_INVALID;
  const size_t bytes = (size_t)RT_TEX_SIZE * RT_TEX_SIZE * 4;
  if (layer >= c->tex_layers || !c->textures.ptr) return fail(c, RT_ERR_INVALID, "texture layer out of range");
  if (cap < bytes) return fail(c, RT_ERR_INVALID, "buffer too small for a texture layer");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipMemcpy(out, (const uint8_t*)c->textures.ptr + (size_t)layer * bytes, bytes, hipMemcpyDeviceToHost));
  return RT_OK;
}

// ---- binned-SAH BLAS build on the GPU (csrc/bvh_build.hip.h) ----
// Work space of one build of n triangles (shared by successive builds: they run one after the other on c->stream).
static int blas_workspace(rt_ctx* c, uint32_t n_tris, bvhb::Build& B, bvhb::BigNode*& d_big, bvhb::Chunk*& d_chunks, uint32_t*& d_cnt,
                          uint32_t*& d_base, uint32_t*& d_lb, uint32_t*& d_blk) {
  const size_t n = n_tris, max_nodes = 2 * n;  // a binary tree with >= 1 triangle per leaf has < 2n nodes
  const size_t n_blk = (n + 1023) / 1024 + 1;
  int r;
  if ((r = ensure_buffer(c, c->bv_tri, n * 48, false)) < 0) return r;
  // ord[0], ord[1], order_final, scratch_l, scratch_r, leaf_flag, small_ids, lb (n + 1), block counts; bin cache (bytes)
  if ((r = ensure_buffer(c, c->bv_order, (8 * n + 4 + n_blk) * 4 + n + 16, false)) < 0) return r;
  // node records: ids [0, 2n) of the level kernels, [2n, 4n) blocks of the in-wave subtrees; a fresh array is zeroed once
  // (records carry the stamp of the build that made them, stamps start at 1)
  if ((r = ensure_buffer(c, c->bv_nodes, 2 * max_nodes * sizeof(bvhb::BNode), false)) < 0) return r;
  if (r > 0) HIP_TRY(c, hipMemsetAsync(c->bv_nodes.ptr, 0, c->bv_nodes.capacity, c->stream));
  if ((r = ensure_buffer(c, c->bv_counters, sizeof(bvhb::Ctl), false)) < 0) return r;
  const uint32_t nb = bvhb::big_cap(n_tris), nc = bvhb::chunk_cap(n_tris);
  if ((r = ensure_buffer(c, c->bv_big, (size_t)nb * sizeof(bvhb::BigNode) + (size_t)nc * (sizeof(bvhb::Chunk) + 16), false)) < 0) return r;
  float4* tri = (float4*)c->bv_tri.ptr;
  uint32_t* ord = (uint32_t*)c->bv_order.ptr;
  B.tri_mn = tri;
  B.tri_mx = tri + n;
  B.tri_c = tri + 2 * n;
  B.ord[0] = ord;
  B.ord[1] = ord + n;
  B.order_final = ord + 2 * n;
  B.scratch_l = ord + 3 * n;
  B.scratch_r = ord + 4 * n;
  B.leaf_flag = ord + 5 * n;
  B.small_ids = ord + 6 * n;
  d_lb = ord + 7 * n;              // n + 1 entries
  d_blk = ord + 8 * n + 4;
  B.bin_cache = (uint8_t*)(d_blk + n_blk);
  B.nodes = (bvhb::BNode*)c->bv_nodes.ptr;
  B.ctl = (bvhb::Ctl*)c->bv_counters.ptr;
  B.n_tris = n_tris;
  B.gen = ++c->bv_gen;
  if (B.gen == 0u) {   // stamp wrap-around: forget every old record
    HIP_TRY(c, hipMemsetAsync(c->bv_nodes.ptr, 0, c->bv_nodes.capacity, c->stream));
    B.gen = c->bv_gen = 1u;
  }
  d_big = (bvhb::BigNode*)c->bv_big.ptr;
  d_chunks = (bvhb::Chunk*)(d_big + nb);
  d_cnt = (uint32_t*)(d_chunks + nc);
  d_base = d_cnt + 2 * (size_t)nc;
  return RT_OK;
}
static uint32_t ceil_log2(uint32_t v) {
  uint32_t l = 0;
  while (l < 32u && (1ull << l) < v) l++;
  return l;
}
// first guesses for a mesh never built before: tree depth and the levels that still hold a node above kBig triangles
static uint32_t blas_guess_levels(uint32_t n_tris) { return std::min<uint32_t>(bvhb::kMaxLevels, ceil_log2(std::max(n_tris / 4u, 1u)) + 8u); }
static uint32_t blas_guess_big_levels(uint32_t n_tris) { return n_tris > bvhb::kBig ? ceil_log2((n_tris + bvhb::kBig - 1) / bvhb::kBig) + 3u : 0u; }

// Enqueue one whole build on c->stream: `levels` tree levels (the first `big_levels` with the large-node kernels), the
// leaf prefix sum and the node array, written to out + 2 * node_base[0] (node_base, device: [1] = [0] + node count is
// written; NULL: at out).  Nothing is read back: the caller checks Ctl::cnt[levels] == 0 (the tree was not deeper)
// whenever it next synchronises.  d_pos / d_idx / out are device pointers.
static int blas_enqueue(rt_ctx* c, const float4* d_pos, const uint32_t* d_idx, uint32_t n_tris, uint32_t levels, uint32_t big_levels,
                        float4* out, uint32_t* node_base, uint32_t topo_start, bvhb::Build& B) {
  bvhb::BigNode* d_big;
  bvhb::Chunk* d_chunks;
  uint32_t *d_cnt, *d_base, *d_lb, *d_blk;
  int r = blas_workspace(c, n_tris, B, d_big, d_chunks, d_cnt, d_base, d_lb, d_blk);
  if (r < 0) return r;
  levels = std::min<uint32_t>(levels, bvhb::kMaxLevels);
  big_levels = std::min(big_levels, levels);
  hipLaunchKernelGGL(bvhb::k_tri_boxes, dim3((n_tris + 255) / 256), dim3(256), 0, c->stream, d_pos, d_idx, n_tris, B);
  const dim3 gn_thread((bvhb::big_cap(n_tris) + 63) / 64), gn_wave(bvhb::big_cap(n_tris)), gc(bvhb::chunk_cap(n_tris));
  for (uint32_t level = 0; level < levels; level++) {
    const uint32_t bound = level >= 31u ? n_tris : std::min<uint32_t>(1u << level, n_tris);   // most nodes the level can have
    if (level < big_levels) {
      hipLaunchKernelGGL(bvhb::k_big_plan, dim3(1), dim3(1024), 0, c->stream, B, level, d_big, d_chunks);
      if (level == 0)  // only the root reduces its box; children get theirs from the parent's sweep
        hipLaunchKernelGGL(bvhb::k_big_bounds, gc, dim3(256), 0, c->stream, B, level, d_big, (const bvhb::Chunk*)d_chunks);
      hipLaunchKernelGGL(bvhb::k_big_setup, gn_thread, dim3(64), 0, c->stream, B, d_big);
      hipLaunchKernelGGL(bvhb::k_big_bin, gc, dim3(256), 0, c->stream, B, level, d_big, (const bvhb::Chunk*)d_chunks);
      hipLaunchKernelGGL(bvhb::k_big_split, gn_wave, dim3(64), 0, c->stream, B, d_big);
      hipLaunchKernelGGL(bvhb::k_big_count, gc, dim3(256), 0, c->stream, B, (const bvhb::BigNode*)d_big, (const bvhb::Chunk*)d_chunks, d_cnt);
      hipLaunchKernelGGL(bvhb::k_big_scan, gn_thread, dim3(64), 0, c->stream, B, d_big, (const uint32_t*)d_cnt, d_base);
      hipLaunchKernelGGL(bvhb::k_big_scatter, gc, dim3(256), 0, c->stream, B, (const bvhb::BigNode*)d_big, (const bvhb::Chunk*)d_chunks,
                         (const uint32_t*)d_cnt, (const uint32_t*)d_base);
      hipLaunchKernelGGL(bvhb::k_big_swap, gc, dim3(256), 0, c->stream, B, level, (const bvhb::BigNode*)d_big, (const bvhb::Chunk*)d_chunks);
      hipLaunchKernelGGL(bvhb::k_big_copy, gc, dim3(256), 0, c->stream, B, level, (const bvhb::BigNode*)d_big, (const bvhb::Chunk*)d_chunks);
      hipLaunchKernelGGL((bvhb::k_level<256, true>), dim3(std::min<uint32_t>(bound, 1024u)), dim3(256), 0, c->stream, B, level);
    } else if (level < big_levels + 3u) {
      hipLaunchKernelGGL((bvhb::k_level<256, false>), dim3(std::min<uint32_t>(bound, 2048u)), dim3(256), 0, c->stream, B, level);
    } else {
      // deep levels: tens of thousands of nodes of a few dozen triangles — one wave per node keeps four times as many
      // nodes in flight per CU
      hipLaunchKernelGGL((bvhb::k_level<64, false>), dim3(std::min<uint32_t>(bound, 8192u)), dim3(64), 0, c->stream, B, level);
    }
  }
  const uint32_t n_blk = (n_tris + 1023u) / 1024u;
  hipLaunchKernelGGL(bvhb::k_scan_blocks, dim3(n_blk), dim3(1024), 0, c->stream, (const uint32_t*)B.leaf_flag, n_tris, d_blk);
  hipLaunchKernelGGL(bvhb::k_scan_top, dim3(1), dim3(1024), 0, c->stream, d_blk, n_blk, B.ctl, node_base);
  hipLaunchKernelGGL(bvhb::k_scan_apply, dim3(n_blk), dim3(1024), 0, c->stream, (const uint32_t*)B.leaf_flag, n_tris, (const uint32_t*)d_blk,
                     (const bvhb::Ctl*)B.ctl, d_lb);
  hipLaunchKernelGGL(bvhb::k_emit, dim3((4 * n_tris + 255) / 256), dim3(256), 0, c->stream, (const bvhb::BNode*)B.nodes, 4u * n_tris, B.gen,
                     (const bvhb::Ctl*)B.ctl, (const uint32_t*)d_lb, (const uint32_t*)node_base, topo_start, out);
  HIP_TRY(c, hipGetLastError());
  return RT_OK;
}

// The tree and the triangle order of the scene compiler's BlasBuilder, byte for byte.  Host arrays in, host arrays out
// (the scene compiler's ms_blas_builder hook; the device-resident update, rt_world_update, uses blas_enqueue directly).
int rt_build_blas(rt_ctx* c, const float* verts4, uint32_t n_verts, const uint32_t* indices, uint32_t n_tris, float* nodes_out,
                  uint32_t nodes_cap, uint32_t* n_nodes_out, uint32_t* order_out) {
  if (!c || !n_nodes_out) return RT_ERR_INVALID;
  *n_nodes_out = 0;
  if (n_tris == 0) return RT_OK;
  if (!verts4 || !indices || !nodes_out || !order_out || n_verts == 0) return fail(c, RT_ERR_INVALID, "rt_build_blas: null argument");
  if (n_tris > (1u << 28)) return fail(c, RT_ERR_INVALID, "rt_build_blas: too many triangles for the 29-bit leaf field");
  for (size_t i = 0; i < (size_t)n_tris * 3; i++)
    if (indices[i] >= n_verts) return fail(c, RT_ERR_INVALID, "rt_build_blas: vertex index out of range");
  HIP_TRY(c, hipSetDevice(c->device));
  const size_t n = n_tris;
  int r;
  if ((r = ensure_buffer(c, c->bv_in, (size_t)n_verts * 16 + n * 12, false)) < 0) return r;
  if ((r = ensure_buffer(c, c->bv_out, 2 * n * 32, false)) < 0) return r;
  float4* d_pos = (float4*)c->bv_in.ptr;
  uint32_t* d_idx = (uint32_t*)((char*)c->bv_in.ptr + (size_t)n_verts * 16);
  HIP_TRY(c, hipMemcpyAsync(d_pos, verts4, (size_t)n_verts * 16, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(d_idx, indices, n * 12, hipMemcpyHostToDevice, c->stream));
  if (!c->bv_pinned) HIP_TRY(c, hipHostMalloc(&c->bv_pinned, 1 << 16, hipHostMallocDefault));
  bvhb::Ctl* ctl = (bvhb::Ctl*)c->bv_pinned;
  // as many levels as the last build of a mesh of this size needed (+ 2), else a guess; a deeper tree is built again
  uint32_t levels = (c->bv_levels && c->bv_last_tris == n_tris) ? c->bv_levels + 2u : blas_guess_levels(n_tris);
  uint32_t big_levels = (c->bv_levels && c->bv_last_tris == n_tris) ? c->bv_big_levels : blas_guess_big_levels(n_tris);
  bvhb::Build B;
  for (;;) {
    levels = std::min<uint32_t>(levels, bvhb::kMaxLevels);
    if ((r = blas_enqueue(c, d_pos, d_idx, n_tris, levels, big_levels, (float4*)c->bv_out.ptr, nullptr, 0u, B)) < 0) return r;
    HIP_TRY(c, hipMemcpyAsync(ctl, B.ctl, sizeof(bvhb::Ctl), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (ctl->cnt[levels] == 0u) break;
    if (levels >= bvhb::kMaxLevels) return fail(c, RT_ERR_INVALID, "rt_build_blas: the tree is deeper than 1024 levels");
    levels = levels * 2u + 8u;
  }
  const uint32_t total = ctl->n_nodes;
  if (total > 2 * n) return fail(c, RT_ERR_INTERNAL, "rt_build_blas: node bookkeeping broke");
  if (total > nodes_cap) return fail(c, RT_ERR_INVALID, "rt_build_blas: node buffer too small");
  HIP_TRY(c, hipMemcpyAsync(nodes_out, c->bv_out.ptr, (size_t)total * 32, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipMemcpyAsync(order_out, B.order_final, n * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  *n_nodes_out = total;
  uint32_t depth = 0;
  while (depth < bvhb::kMaxLevels && ctl->cnt[depth]) depth++;
  c->bv_levels = depth;
  c->bv_big_levels = ctl->big_levels;
  c->bv_last_tris = n_tris;
  return RT_OK;
}

// ---- device-resident World::update(t): include/mi355rt.h rt_world_update, kernels in csrc/world_update.hip.h ----
// (Re)build the static side: the skinning input, index lists and attribute rows of every geometry and the instance list
// go to the device once per static_epoch; the scene buffers are sized for the whole world.
static int world_make_static(rt_ctx* c, const rt_world_frame* f) {
  auto& W = c->world;
  W.valid = false;
  W.static_cached = false;
  if (!f->n_instances || !f->instances || (f->n_geometries && !f->geometries))
    return fail(c, RT_ERR_INVALID, "rt_world_update: empty scene description");
  const uint32_t G = f->n_geometries, N = f->n_instances;
  if (N > RT_TLAS_MAX_INSTANCES)   // k_tlas sorts the ranges of a depth in LDS: 16 384 64-bit keys are 128 KB of the CU's 160
    return fail(c, RT_ERR_INVALID, "rt_world_update: more than 16 384 instances are not taken by the device path");
  // layout of the static buffer (256-byte aligned arrays) and the world's totals
  size_t bytes = 0;
  auto take = [&](size_t n) { size_t o = bytes; bytes += (n + 255) & ~(size_t)255; return o; };
  struct Off { size_t pos, nrm, uv, joints, weights, idx, attr; };
  std::vector<Off> off(G);
  W.geoms.assign(G, wu::Geom());
  uint64_t verts = 0, tris = 0, lights = 0;
  std::vector<uint32_t> em_count(G, 0u);
  for (uint32_t g = 0; g < G; g++) {
    const rt_world_geometry& d = f->geometries[g];
    if (d.n_verts && (!d.positions || !d.normals || !d.joints || !d.weights)) return fail(c, RT_ERR_INVALID, "rt_world_update: null vertex array");
    if (d.n_tris && (!d.indices || !d.attributes)) return fail(c, RT_ERR_INVALID, "rt_world_update: null triangle array");
    if (d.n_uvs > d.n_verts || (d.n_uvs && !d.uvs)) return fail(c, RT_ERR_INVALID, "rt_world_update: bad uv array");
    if (d.n_verts && !d.n_tris) return fail(c, RT_ERR_INVALID, "rt_world_update: a geometry with vertices and no triangles is not supported on the device");
    if (d.n_tris > (1u << 28)) return fail(c, RT_ERR_INVALID, "rt_world_update: too many triangles for the 29-bit leaf field");
    if (d.skin >= 0 && (uint32_t)d.skin >= f->n_skins) return fail(c, RT_ERR_INVALID, "rt_world_update: skin index out of range");
    for (size_t k = 0; k < (size_t)d.n_tris * 3; k++)
      if (d.indices[k] >= d.n_verts) return fail(c, RT_ERR_INVALID, "rt_world_update: vertex index out of range");
    for (uint32_t t = 0; t < d.n_tris; t++) em_count[g] += std::fabs(d.attributes[(size_t)t * 16 + 3] - 3.0f) < 1e-6f ? 1u : 0u;
    off[g].pos = take((size_t)d.n_verts * 12);
    off[g].nrm = take((size_t)d.n_verts * 12);
    off[g].uv = take((size_t)d.n_uvs * 8);
    off[g].joints = take((size_t)d.n_verts * 16);
    off[g].weights = take((size_t)d.n_verts * 16);
    off[g].idx = take((size_t)d.n_tris * 12);
    off[g].attr = take((size_t)d.n_tris * 64);
    wu::Geom& D = W.geoms[g];
    D.n_verts = d.n_verts;
    D.n_uvs = d.n_uvs;
    D.n_tris = d.n_tris;
    D.v_offset = (uint32_t)verts;
    D.topo_start = (uint32_t)tris;
    D.skinned = d.skin >= 0 ? 1u : 0u;
    D.joint_first = D.n_joints = 0u;   // per frame (skin_first)
    D.em_count = em_count[g];
    D.em_first = 0u;
    D.pad0 = (uint32_t)(d.skin >= 0 ? d.skin : 0);   // the skin's index, for the per-frame joint range
    D.pad1 = 0u;
    verts += d.n_verts;
    tris += d.n_tris;
  }
  if (!verts || !tris) return fail(c, RT_ERR_INVALID, "rt_world_update: the world has no triangles");
  if (verts > 0xfffffff0ull || tris > (1ull << 28)) return fail(c, RT_ERR_INVALID, "rt_world_update: world too large");
  W.inst_geom.assign(N, 0u);
  W.inst_scale2.assign(N, 1.0f);
  for (uint32_t i = 0; i < N; i++) {
    const rt_instance& I = f->instances[i];
    if (I.instance_id >= G || !f->geometries[I.instance_id].n_tris)
      return fail(c, RT_ERR_INVALID, "rt_world_update: an instance of a missing or empty geometry is not supported on the device");
    W.inst_geom[i] = I.instance_id;
    lights += em_count[I.instance_id];
    const float* m = I.transform;   // column-major 4x4: squared linear scale = |det(M3x3)|^(2/3), as in rt_upload
    const double det = (double)m[0] * ((double)m[5] * m[10] - (double)m[6] * m[9]) - (double)m[4] * ((double)m[1] * m[10] - (double)m[2] * m[9]) +
                       (double)m[8] * ((double)m[1] * m[6] - (double)m[2] * m[5]);
    double s2 = std::pow(std::fabs(det), 2.0 / 3.0);
    if (!(s2 > 0.0) || !std::isfinite(s2)) s2 = 1.0;
    W.inst_scale2[i] = (float)s2;
  }
  if (lights > 0x7fffffffull) return fail(c, RT_ERR_INVALID, "rt_world_update: too many lights");
  HIP_TRY(c, hipSetDevice(c->device));
  int r;
  if ((r = ensure_buffer(c, W.stat, std::max<size_t>(bytes, 256), false)) < 0) return r;
  char* base = (char*)W.stat.ptr;
  uint32_t em_first = 0, max_tris = 0;
  std::vector<wu::GeomRow> rows(G);
  for (uint32_t g = 0; g < G; g++) {
    const rt_world_geometry& d = f->geometries[g];
    wu::Geom& D = W.geoms[g];
    auto put = [&](size_t o, const void* src, size_t n) -> hipError_t {
      return n ? hipMemcpyAsync(base + o, src, n, hipMemcpyHostToDevice, c->stream) : hipSuccess;
    };
    HIP_TRY(c, put(off[g].pos, d.positions, (size_t)d.n_verts * 12));
    HIP_TRY(c, put(off[g].nrm, d.normals, (size_t)d.n_verts * 12));
    HIP_TRY(c, put(off[g].uv, d.uvs, (size_t)d.n_uvs * 8));
    HIP_TRY(c, put(off[g].joints, d.joints, (size_t)d.n_verts * 16));
    HIP_TRY(c, put(off[g].weights, d.weights, (size_t)d.n_verts * 16));
    HIP_TRY(c, put(off[g].idx, d.indices, (size_t)d.n_tris * 12));
    HIP_TRY(c, put(off[g].attr, d.attributes, (size_t)d.n_tris * 64));
    D.pos3 = (const float*)(base + off[g].pos);
    D.nrm3 = (const float*)(base + off[g].nrm);
    D.uv2 = (const float*)(base + off[g].uv);
    D.joints = (const uint32_t*)(base + off[g].joints);
    D.weights = (const float*)(base + off[g].weights);
    D.idx = (const uint32_t*)(base + off[g].idx);
    D.attr = (const float*)(base + off[g].attr);
    D.em_first = em_first;
    em_first += D.em_count;
    max_tris = std::max(max_tris, D.n_tris);
    rows[g] = wu::GeomRow{D.n_tris, D.topo_start, D.em_count, D.em_first};
  }
  const uint32_t n_tlas = 2u * N - 1u;
  if ((r = ensure_buffer(c, W.raw_inst, (size_t)N * sizeof(rt_instance), false)) < 0) return r;
  HIP_TRY(c, hipMemcpyAsync(W.raw_inst.ptr, f->instances, (size_t)N * sizeof(rt_instance), hipMemcpyHostToDevice, c->stream));
  if ((r = ensure_buffer(c, W.geom_rows, std::max<size_t>(16, (size_t)G * sizeof(wu::GeomRow)), false)) < 0) return r;
  HIP_TRY(c, hipMemcpyAsync(W.geom_rows.ptr, rows.data(), (size_t)G * sizeof(wu::GeomRow), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));   // the sources above belong to the caller
  if ((r = ensure_buffer(c, W.node_base, ((size_t)G + 1) * 4, false)) < 0) return r;
  if ((r = ensure_buffer(c, W.stats, std::max<size_t>(64, (size_t)G * 16 + 16), false)) < 0) return r;
  if ((r = ensure_buffer(c, W.em_flag, (size_t)max_tris * 4, false)) < 0) return r;
  if ((r = ensure_buffer(c, W.em_blk, ((size_t)max_tris / 1024 + 2) * 4, false)) < 0) return r;
  if ((r = ensure_buffer(c, W.em_list, std::max<size_t>(16, (size_t)em_first * 4), false)) < 0) return r;
  // k_tlas scratch: box 6, ctr 3, ord 1, ord2 1, seg 3, skey 18, sinfo 2, light_off 1 words per instance + 4 status words
  if ((r = ensure_buffer(c, W.tlas_scratch, ((size_t)N * 35 + 4) * 4, false)) < 0) return r;
  // the renderer's scene buffers, sized for the whole world (node array: TLAS + at most 2 n - 1 nodes per geometry)
  size_t max_nodes = n_tlas;
  for (uint32_t g = 0; g < G; g++) max_nodes += W.geoms[g].n_tris ? 2 * (size_t)W.geoms[g].n_tris - 1 : 0;
  bool grew = false;
  auto scene_buffer = [&](DeviceBuffer& b, size_t n) -> int {
    int rr = ensure_buffer(c, b, n, true);
    if (rr > 0) grew = true;
    return rr;
  };
  if ((r = scene_buffer(c->pos, (size_t)verts * 16)) < 0) return r;
  if ((r = scene_buffer(c->nrm, (size_t)verts * 16)) < 0) return r;
  if ((r = scene_buffer(c->uv, (size_t)verts * 8)) < 0) return r;
  if ((r = scene_buffer(c->topology, (size_t)tris * sizeof(rt_topology))) < 0) return r;
  if ((r = scene_buffer(c->nodes, max_nodes * sizeof(rt_node))) < 0) return r;
  if ((r = scene_buffer(c->instances, (size_t)N * sizeof(rt_instance))) < 0) return r;
  if ((r = scene_buffer(c->lights, std::max<size_t>(16, (size_t)lights * sizeof(rt_light_ref)))) < 0) return r;
  if ((r = ensure_buffer(c, c->draw_commands, (size_t)N * 16, false)) < 0) return r;
  if (!W.ev0) HIP_TRY(c, hipEventCreate(&W.ev0));
  if (!W.ev1) HIP_TRY(c, hipEventCreate(&W.ev1));
  if (!W.ev_t0) HIP_TRY(c, hipEventCreate(&W.ev_t0));
  if (!W.ev_t1) HIP_TRY(c, hipEventCreate(&W.ev_t1));
  W.levels.assign(G, 0u);
  W.big_levels.assign(G, 0u);
  for (uint32_t g = 0; g < G; g++) {
    W.levels[g] = blas_guess_levels(W.geoms[g].n_tris);
    W.big_levels[g] = blas_guess_big_levels(W.geoms[g].n_tris);
  }
  W.any_skinned = false;
  for (uint32_t g = 0; g < G; g++) W.any_skinned = W.any_skinned || (W.geoms[g].n_verts && W.geoms[g].skinned);
  W.n_verts = (uint32_t)verts;
  W.n_tris = (uint32_t)tris;
  W.n_lights = (uint32_t)lights;
  W.n_tlas = n_tlas;
  W.n_inst = N;
  uint32_t* ts = (uint32_t*)W.tlas_scratch.ptr;
  wu::TlasArgs& A = W.tlas;
  A.raw = (const float4*)W.raw_inst.ptr;
  A.geoms = (const wu::GeomRow*)W.geom_rows.ptr;
  A.node_base = (const uint32_t*)W.node_base.ptr;
  A.box = (float*)ts;
  A.ctr = (float*)(ts + (size_t)N * 6);
  A.ord = ts + (size_t)N * 9;
  A.ord2 = ts + (size_t)N * 10;
  A.seg = ts + (size_t)N * 11;
  A.skey = ts + (size_t)N * 14;
  A.sinfo = ts + (size_t)N * 32;
  A.light_off = ts + (size_t)N * 34;
  A.status = ts + (size_t)N * 35;
  A.n_inst = N;
  A.n_tlas = n_tlas;
  A.n_lights = (uint32_t)lights;
  A.pad = 0;
  W.epoch = f->static_epoch;
  W.n_skins = f->n_skins;
  W.valid = true;
  return grew ? RT_REALLOCATED : RT_OK;
}

// *touched is set before the first kernel that writes into the live scene buffers is enqueued
static int world_update_body(rt_ctx* c, const rt_world_frame* f, bool* touched) {
  auto& W = c->world;
  c->epoch++;   // drops frames traced ahead (rt_set_lookahead)
  HIP_TRY(c, hipSetDevice(c->device));
  int r, ret = RT_OK;
  if (!W.valid || W.epoch != f->static_epoch) {
    if ((r = world_make_static(c, f)) < 0) return r;
    ret = r;
  }
  const uint32_t G = (uint32_t)W.geoms.size(), N = W.n_inst;
  if (f->n_geometries != G || f->n_instances != N) return fail(c, RT_ERR_INVALID, "rt_world_update: the scene changed under an unchanged static_epoch");
  if (W.cache_enabled && W.static_cached && !W.any_skinned) {
    // nothing in this world moves: every array of this frame is the one the last update left in the scene buffers (and the
    // renderer's derived records are those of that scene) - World::update would rebuild all of it to the same bytes
    W.last_ms = 0.0;
    return ret;
  }
  // this frame's joint matrices: the per-frame arguments are checked against the static description on every call (a
  // direct C-ABI caller may hand anything): the skin table must be the one the geometries' skin indices were checked against
  if (f->n_skins != W.n_skins) return fail(c, RT_ERR_INVALID, "rt_world_update: the number of skins changed under an unchanged static_epoch");
  if (f->n_skins && !f->skin_first) return fail(c, RT_ERR_INVALID, "rt_world_update: null skin table");
  for (uint32_t k = 0; k < f->n_skins; k++)
    if (f->skin_first[k + 1] < f->skin_first[k]) return fail(c, RT_ERR_INVALID, "rt_world_update: the skin table is not non-decreasing");
  if (f->n_skins && f->skin_first[0] != 0u) return fail(c, RT_ERR_INVALID, "rt_world_update: the skin table does not start at joint 0");
  const uint32_t n_joints = f->n_skins ? f->skin_first[f->n_skins] : 0u;
  if (n_joints && !f->joint_mats) return fail(c, RT_ERR_INVALID, "rt_world_update: null joint matrices");
  const size_t joint_bytes = (size_t)n_joints * 64;
  if ((r = ensure_buffer(c, W.joints, std::max<size_t>(64, joint_bytes), false)) < 0) return r;
  // pinned staging: the end-of-update read-back (node_base, per-geometry stats, k_tlas status, TLAS root), then the joints
  const size_t rb_bytes = (((size_t)G + 1) * 4 + (size_t)G * 16 + 16 + 32 + 4095) & ~(size_t)4095;
  if (rb_bytes + joint_bytes > W.pinned_bytes) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (W.pinned) (void)hipHostFree(W.pinned);
    W.pinned = nullptr;
    W.pinned_bytes = 0;
    HIP_TRY(c, hipHostMalloc(&W.pinned, (rb_bytes + joint_bytes) * 2 + 65536, hipHostMallocDefault));
    W.pinned_bytes = (rb_bytes + joint_bytes) * 2 + 65536;
  }
  char* pinned_rb = (char*)W.pinned;
  char* pinned_joints = (char*)W.pinned + rb_bytes;
  for (uint32_t g = 0; g < G; g++) {
    wu::Geom& D = W.geoms[g];
    if (D.skinned) {
      const uint32_t si = D.pad0;
      D.joint_first = f->skin_first[si];
      D.n_joints = f->skin_first[si + 1] - f->skin_first[si];
    }
  }
  uint32_t* d_node_base = (uint32_t*)W.node_base.ptr;
  uint32_t* d_stats = (uint32_t*)W.stats.ptr;
  float4* nodes = (float4*)c->nodes.ptr;
  for (int attempt = 0;; attempt++) {
    *touched = true;   // from here on the live scene buffers are being rewritten
    HIP_TRY(c, hipEventRecord(W.ev0, c->stream));
    if (joint_bytes) {
      std::memcpy(pinned_joints, f->joint_mats, joint_bytes);
      HIP_TRY(c, hipMemcpyAsync(W.joints.ptr, pinned_joints, joint_bytes, hipMemcpyHostToDevice, c->stream));
    }
    HIP_TRY(c, hipMemsetAsync(d_node_base, 0, 4, c->stream));
    HIP_TRY(c, hipMemsetAsync(W.tlas.status, 0, 16, c->stream));
    for (uint32_t g = 0; g < G; g++) {
      const wu::Geom& D = W.geoms[g];
      if (!D.n_verts) {   // an empty geometry owns no nodes: the next one starts where this one would
        HIP_TRY(c, hipMemcpyAsync(d_node_base + g + 1, d_node_base + g, 4, hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(c, hipMemsetAsync(d_stats + 4 * (size_t)g, 0, 16, c->stream));
        continue;
      }
      if (W.cache_enabled && W.static_cached && !D.skinned) {   // same vertices as last frame: rows stay, nodes come from the cache
        const uint32_t cnt = W.static_count[g];
        hipLaunchKernelGGL(wu::k_static_nodes, dim3((2 * cnt + 255) / 256), dim3(256), 0, c->stream,
                           (const float4*)W.static_nodes.ptr + 2 * (size_t)W.static_off[g], cnt, d_node_base + g, nodes + 2 * (size_t)W.n_tlas);
        continue;
      }
      hipLaunchKernelGGL(wu::k_skin, dim3((D.n_verts + 255) / 256), dim3(256), 0, c->stream, D, (const float*)W.joints.ptr, (float4*)c->pos.ptr,
                         (float4*)c->nrm.ptr, (float2*)c->uv.ptr);
      bvhb::Build B;
      if ((r = blas_enqueue(c, (const float4*)c->pos.ptr + D.v_offset, D.idx, D.n_tris, W.levels[g], W.big_levels[g], nodes + 2 * (size_t)W.n_tlas,
                            d_node_base + g, D.topo_start, B)) < 0)
        return r;
      hipLaunchKernelGGL(wu::k_build_stats, dim3(1), dim3(1), 0, c->stream, (const bvhb::Ctl*)B.ctl, std::min<uint32_t>(W.levels[g], bvhb::kMaxLevels),
                         d_stats + 4 * (size_t)g);
      uint32_t* em_flag = D.em_count ? (uint32_t*)W.em_flag.ptr : nullptr;
      hipLaunchKernelGGL(wu::k_topology, dim3((D.n_tris + 255) / 256), dim3(256), 0, c->stream, D, g, (const uint32_t*)B.order_final,
                         (float4*)c->topology.ptr, em_flag);
      if (D.em_count) {
        const uint32_t n_blk = (D.n_tris + 1023u) / 1024u;
        hipLaunchKernelGGL(bvhb::k_scan_blocks, dim3(n_blk), dim3(1024), 0, c->stream, (const uint32_t*)em_flag, D.n_tris, (uint32_t*)W.em_blk.ptr);
        hipLaunchKernelGGL(bvhb::k_scan_top, dim3(1), dim3(1024), 0, c->stream, (uint32_t*)W.em_blk.ptr, n_blk, (bvhb::Ctl*)nullptr, (uint32_t*)nullptr);
        hipLaunchKernelGGL(wu::k_emissive_apply, dim3(n_blk), dim3(1024), 0, c->stream, D, (const uint32_t*)em_flag, (const uint32_t*)W.em_blk.ptr,
                           (uint32_t*)W.em_list.ptr);
      }
    }
    wu::TlasArgs A = W.tlas;
    A.nodes = nodes;
    A.inst_out = (float4*)c->instances.ptr;
    A.draw_out = (uint4*)c->draw_commands.ptr;
    {
      uint32_t m = 1024u;
      while (m < N) m <<= 1;
      if (!W.tlas_lds_set) {
        HIP_TRY(c, hipFuncSetAttribute((const void*)wu::k_tlas, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(RT_TLAS_MAX_INSTANCES * 8u)));
        W.tlas_lds_set = true;
      }
      HIP_TRY(c, hipEventRecord(W.ev_t0, c->stream));
      if (N <= 1024u && !getenv("MI355RT_TLAS_GENERIC"))   // one position per lane, everything in LDS / registers (config 3: 1 001)
        hipLaunchKernelGGL(wu::k_tlas_small, dim3(1), dim3(1024), 0, c->stream, A);
      else
        hipLaunchKernelGGL(wu::k_tlas, dim3(1), dim3(1024), (size_t)m * 8, c->stream, A);
      HIP_TRY(c, hipEventRecord(W.ev_t1, c->stream));
    }
    if (W.n_lights)
      hipLaunchKernelGGL(wu::k_lights, dim3(N), dim3(256), 0, c->stream, A, (const uint32_t*)W.em_list.ptr, (uint2*)c->lights.ptr);
    HIP_TRY(c, hipGetLastError());
    // one read-back: where every BLAS starts, whether every tree was finished, the TLAS root
    uint32_t* rb_base = (uint32_t*)pinned_rb;
    uint32_t* rb_stats = rb_base + G + 1;
    uint32_t* rb_status = rb_stats + 4 * (size_t)G;
    float* rb_root = (float*)(rb_status + 4);
    HIP_TRY(c, hipMemcpyAsync(rb_base, d_node_base, ((size_t)G + 1) * 4, hipMemcpyDeviceToHost, c->stream));
    if (G) HIP_TRY(c, hipMemcpyAsync(rb_stats, d_stats, (size_t)G * 16, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(rb_status, W.tlas.status, 16, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(rb_root, nodes, 32, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipEventRecord(W.ev1, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    bool again = false;
    const bool was_cached = W.cache_enabled && W.static_cached;
    for (uint32_t g = 0; g < G; g++) {
      if (!W.geoms[g].n_verts || (was_cached && !W.geoms[g].skinned)) continue;
      const uint32_t* st = rb_stats + 4 * (size_t)g;
      if (st[0]) {   // deeper than launched: build again with more levels
        if (W.levels[g] >= bvhb::kMaxLevels) return fail(c, RT_ERR_INVALID, "rt_world_update: a BLAS is deeper than 1024 levels");
        W.levels[g] = std::min<uint32_t>(bvhb::kMaxLevels, W.levels[g] * 2u + 8u);
        again = true;
      } else {
        W.levels[g] = std::min<uint32_t>(bvhb::kMaxLevels, st[1] + 2u);
        W.big_levels[g] = st[2];
      }
    }
    if (again) {
      if (attempt >= 8) return fail(c, RT_ERR_INTERNAL, "rt_world_update: BLAS depth did not settle");
      continue;
    }
    if (rb_status[0]) return fail(c, RT_ERR_INVALID, "rt_world_update: an instance box has a NaN centre; the host path defines this update");
    if (rb_status[1] != W.n_lights) return fail(c, RT_ERR_INTERNAL, "rt_world_update: light count mismatch");
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, W.ev0, W.ev1) == hipSuccess) W.last_ms = ms;
    if (hipEventElapsedTime(&ms, W.ev_t0, W.ev_t1) == hipSuccess) W.last_tlas_ms = ms;
    // ---- host bookkeeping, what rt_upload / rt_upload_geometry / rt_upload_bvh would have set
    const uint32_t n_blas = rb_base[G];
    c->n_verts = c->vertex_count = W.n_verts;
    c->n_tris = W.n_tris;
    c->n_instances = N;
    c->n_lights = W.n_lights;
    c->blas_offset = W.n_tlas;
    c->n_nodes = W.n_tlas + n_blas;
    c->pos.size = (size_t)W.n_verts * 16;
    c->nodes.size = (size_t)c->n_nodes * sizeof(rt_node);
    c->draw_commands_host.clear();   // made on the device (rt_world_read gives them)
    uint32_t inner = N - 1u;
    for (uint32_t g = 0; g < G; g++) {
      const uint32_t cnt = rb_base[g + 1] - rb_base[g];
      if (cnt) inner += (cnt - 1u) / 2u;
    }
    c->n_pairs = inner;
    std::memset(&c->troot, 0, sizeof(c->troot));
    for (int k = 0; k < 3; k++) {
      c->troot.lo[k] = rb_root[k];
      c->troot.hi[k] = rb_root[4 + k];
    }
    {
      uint32_t w7;
      std::memcpy(&w7, &rb_root[7], 4);
      c->troot.word = w7 == 0u ? RT_PAIR_INNER : w7;
    }
    const std::vector<uint32_t> old_roots = c->blas_roots;
    c->blas_roots.resize(N);
    for (uint32_t i = 0; i < N; i++) c->blas_roots[i] = rb_base[W.inst_geom[i]];
    std::sort(c->blas_roots.begin(), c->blas_roots.end());
    c->blas_roots.erase(std::unique(c->blas_roots.begin(), c->blas_roots.end()), c->blas_roots.end());
    c->root_w_host.assign(c->blas_roots.size(), 0.0f);
    for (uint32_t i = 0; i < N; i++) {
      const size_t k = std::lower_bound(c->blas_roots.begin(), c->blas_roots.end(), rb_base[W.inst_geom[i]]) - c->blas_roots.begin();
      c->root_w_host[k] += W.inst_scale2[i];
    }
    // the arrays were made by this library from an index-checked description: no validation pass; the derived passes
    // of prepare_scene read the sorted roots from val_roots
    if ((r = ensure_buffer(c, c->val_roots, std::max<size_t>(4, c->blas_roots.size() * 4), false)) < 0) return r;
    HIP_TRY(c, hipMemcpyAsync(c->val_roots.ptr, c->blas_roots.data(), c->blas_roots.size() * 4, hipMemcpyHostToDevice, c->stream));
    c->validate_dirty = false;
    c->scene_valid = true;
    c->nodes_from_device = true;
    c->tris_dirty = c->inst_dirty = c->lights_dirty = c->nodes_dirty = c->pairs_dirty = c->roots_dirty = true;
    if (W.cache_enabled && !W.static_cached) {   // first full update of this description: keep the static geometries' node blocks
      W.static_count.assign(G, 0u);
      W.static_off.assign(G, 0u);
      size_t total = 0;
      for (uint32_t g = 0; g < G; g++)
        if (W.geoms[g].n_verts && !W.geoms[g].skinned) {
          W.static_off[g] = (uint32_t)total;
          W.static_count[g] = rb_base[g + 1] - rb_base[g];
          total += W.static_count[g];
        }
      if ((r = ensure_buffer(c, W.static_nodes, std::max<size_t>(32, total * sizeof(rt_node)), false)) < 0) return r;
      for (uint32_t g = 0; g < G; g++)
        if (W.static_count[g])
          HIP_TRY(c, hipMemcpyAsync((char*)W.static_nodes.ptr + (size_t)W.static_off[g] * sizeof(rt_node),
                                    (const char*)c->nodes.ptr + ((size_t)W.n_tlas + rb_base[g]) * sizeof(rt_node),
                                    (size_t)W.static_count[g] * sizeof(rt_node), hipMemcpyDeviceToDevice, c->stream));
      W.static_cached = true;
    }
    return ret;
  }
}

int rt_world_update(rt_ctx* c, const rt_world_frame* f) {
  if (!c || !f) return RT_ERR_INVALID;
  bool touched = false;
  const int r = world_update_body(c, f, &touched);
  if (r < 0 && touched) {
    // The update writes straight into the live scene buffers (positions, topology, nodes, instances, lights).  A failure
    // after its first kernel leaves a mix of two scenes there, with the counts and the derived records of the old one:
    // nothing may be traced from it.  compute() refuses until the scene has been uploaded again (rt_upload* re-validates);
    // the Python / Node bridges do exactly that (the scene compiler then runs this update on the host).
    const std::string why = c->error;
    c->scene_valid = false;
    c->validate_dirty = false;
    c->scene_problem = "the device-resident world update failed part-way (" + why + "): the scene buffers hold a mix of two scenes; upload the scene again";
    c->tris_dirty = c->inst_dirty = c->lights_dirty = c->nodes_dirty = c->pairs_dirty = c->roots_dirty = true;
    c->world.static_cached = false;
    c->world.valid = false;
    c->error = why;
  }
  return r;
}

double rt_world_last_ms(const rt_ctx* c) { return c ? c->world.last_ms : 0.0; }
double rt_world_last_tlas_ms(const rt_ctx* c) { return c ? c->world.last_tlas_ms : 0.0; }

int rt_world_set_static_cache(rt_ctx* c, int enabled) {
  if (!c) return RT_ERR_INVALID;
  c->world.cache_enabled = enabled != 0;
  c->world.static_cached = false;   // the next update builds everything (and refills the cache when enabled)
  return RT_OK;
}

// the bridge arrays as the device update left them (tests; a host that wants them back)
int rt_world_read(rt_ctx* c, int which, void* out, size_t cap_bytes, size_t* bytes_out) {
  if (!c || !bytes_out) return RT_ERR_INVALID;
  *bytes_out = 0;
  if (!c->world.valid) return fail(c, RT_ERR_NOT_READY, "rt_world_read: no device-resident world");
  HIP_TRY(c, hipSetDevice(c->device));
  const void* src = nullptr;
  size_t n = 0;
  switch (which) {
    case RT_WORLD_VERTICES: src = c->pos.ptr; n = (size_t)c->n_verts * 16; break;
    case RT_WORLD_NORMALS: src = c->nrm.ptr; n = (size_t)c->n_verts * 16; break;
    case RT_WORLD_UVS: src = c->uv.ptr; n = (size_t)c->n_verts * 8; break;
    case RT_WORLD_TOPOLOGY: src = c->topology.ptr; n = (size_t)c->n_tris * sizeof(rt_topology); break;
    case RT_WORLD_TLAS: src = c->nodes.ptr; n = (size_t)c->blas_offset * sizeof(rt_node); break;
    case RT_WORLD_BLAS: src = (const char*)c->nodes.ptr + (size_t)c->blas_offset * sizeof(rt_node); n = (size_t)(c->n_nodes - c->blas_offset) * sizeof(rt_node); break;
    case RT_WORLD_INSTANCES: src = c->instances.ptr; n = (size_t)c->n_instances * sizeof(rt_instance); break;
    case RT_WORLD_LIGHTS: src = c->lights.ptr; n = (size_t)c->n_lights * sizeof(rt_light_ref); break;
    case RT_WORLD_DRAW_COMMANDS: src = c->draw_commands.ptr; n = (size_t)c->n_instances * 16; break;
    default: return fail(c, RT_ERR_INVALID, "rt_world_read: unknown array");
  }
  *bytes_out = n;
  if (!out) return RT_OK;
  if (cap_bytes < n) return fail(c, RT_ERR_INVALID, "rt_world_read: buffer too small");
  if (n) HIP_TRY(c, hipMemcpyAsync(out, src, n, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return RT_OK;
}

// depth of the last rt_build_blas tree (levels of the breadth-first build that held nodes) | large-node levels << 16
int rt_build_blas_levels(const rt_ctx* c) { return c ? (int)(c->bv_levels | (c->bv_big_levels << 16)) : 0; }

int rt_upload(rt_ctx* c, rt_kind kind, const void* data, size_t bytes) {
  if (!c) return RT_ERR_INVALID;
  c->epoch++;
  c->world.static_cached = false;   // host arrays replace what the device update left in the scene buffers   // drops frames traced ahead (rt_set_lookahead)
  if (bytes && !data) return fail(c, RT_ERR_INVALID, "null data");
  HIP_TRY(c, hipSetDevice(c->device));
  int r;
  switch (kind) {
    case RT_KIND_TOPOLOGY:
      if (bytes % sizeof(rt_topology)) return fail(c, RT_ERR_INVALID, "topology must be 80 bytes per triangle");
      r = upload(c, c->topology, data, bytes);
      if (r < 0) return r;
      c->n_tris = (uint32_t)(bytes / sizeof(rt_topology));
      c->validate_dirty = true;
      c->tris_dirty = true;
      c->lights_dirty = true;
      return r;
    case RT_KIND_INSTANCE:
      if (bytes % sizeof(rt_instance)) return fail(c, RT_ERR_INVALID, "instances must be 144 bytes each");
      r = upload(c, c->instances, data, bytes);
      if (r < 0) return r;
      c->n_instances = (uint32_t)(bytes / sizeof(rt_instance));
      {  // BLAS roots the instances refer to (host copy of one word per instance: the validation kernel needs them sorted)
        const rt_instance* hi = (const rt_instance*)data;
        const std::vector<uint32_t> old_roots = c->blas_roots;
        c->blas_roots.resize(c->n_instances);
        for (uint32_t k = 0; k < c->n_instances; k++) c->blas_roots[k] = hi[k].blas_node_offset;
        std::sort(c->blas_roots.begin(), c->blas_roots.end());
        c->blas_roots.erase(std::unique(c->blas_roots.begin(), c->blas_roots.end()), c->blas_roots.end());
        // visit-probability weight of each BLAS for the treelet order: sum over its instances of the squared linear
        // scale, |det(M3x3)|^(2/3) (object-space box areas times this approximate world-space areas)
        c->root_w_host.assign(c->blas_roots.size(), 0.0f);
        for (uint32_t k = 0; k < c->n_instances; k++) {
          const float* m = hi[k].transform;   // column-major 4x4
          const double det = (double)m[0] * ((double)m[5] * m[10] - (double)m[6] * m[9]) -
                             (double)m[4] * ((double)m[1] * m[10] - (double)m[2] * m[9]) +
                             (double)m[8] * ((double)m[1] * m[6] - (double)m[2] * m[5]);
          double s2 = std::pow(std::fabs(det), 2.0 / 3.0);
          if (!(s2 > 0.0) || !std::isfinite(s2)) s2 = 1.0;
          const size_t r = std::lower_bound(c->blas_roots.begin(), c->blas_roots.end(), hi[k].blas_node_offset) - c->blas_roots.begin();
          c->root_w_host[r] += (float)s2;
        }
        if (old_roots != c->blas_roots) c->pairs_dirty = true;   // BLAS-local skips are resolved against the set of roots
      }
      c->roots_dirty = true;
      c->nodes_dirty = true;
      c->validate_dirty = true;
      c->inst_dirty = true;
      c->lights_dirty = true;
      return r;
    case RT_KIND_LIGHTS:
      if (bytes % sizeof(rt_light_ref)) return fail(c, RT_ERR_INVALID, "lights must be 8 bytes each");
      r = upload(c, c->lights, data, bytes);
      if (r < 0) return r;
      c->n_lights = (uint32_t)(bytes / sizeof(rt_light_ref));
      c->validate_dirty = true;
      c->lights_dirty = true;
      return r;
    case RT_KIND_DRAW_COMMANDS: {
      // not on the 1.5x policy in the reference (ResourceManager.ts:264-278); kept on the host too
      bool grew = !c->draw_commands.ptr || c->draw_commands.capacity < bytes;
      if (grew) {
        free_buffer(c->draw_commands);
        r = ensure_buffer(c, c->draw_commands, bytes, false);
        if (r < 0) return r;
      }
      if (bytes) {
        HIP_TRY(c, hipMemcpyAsync(c->draw_commands.ptr, data, bytes, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
      }
      c->draw_commands.size = bytes;
      c->draw_commands_host.assign((const uint32_t*)data, (const uint32_t*)data + bytes / 4);
      return grew ? RT_REALLOCATED : RT_OK;
    }
  }
  return fail(c, RT_ERR_INVALID, "unknown buffer kind");
}

int rt_upload_geometry(rt_ctx* c, const float* pos4, const float* nrm4, const float* uv2, uint32_t vertex_count) {
  if (!c) return RT_ERR_INVALID;
  c->epoch++;
  c->world.static_cached = false;   // host arrays replace what the device update left in the scene buffers   // drops frames traced ahead (rt_set_lookahead)
  if (vertex_count && (!pos4 || !nrm4 || !uv2)) return fail(c, RT_ERR_INVALID, "null geometry array");
  HIP_TRY(c, hipSetDevice(c->device));
  // The reference packs the three arrays into one buffer at 256-byte aligned offsets because WebGPU
  // binds sub-ranges (ResourceManager.ts:286-323); HIP needs no such aliasing, so they stay separate.
  int r0 = upload(c, c->pos, pos4, (size_t)vertex_count * 16);
  if (r0 < 0) return r0;
  int r1 = upload(c, c->nrm, nrm4, (size_t)vertex_count * 16);
  if (r1 < 0) return r1;
  int r2 = upload(c, c->uv, uv2, (size_t)vertex_count * 8);
  if (r2 < 0) return r2;
  c->n_verts = vertex_count;
  c->vertex_count = vertex_count;
  c->validate_dirty = true;
  c->tris_dirty = true;
  c->lights_dirty = true;
  return (r0 | r1 | r2) ? RT_REALLOCATED : RT_OK;
}

int rt_upload_bvh(rt_ctx* c, const float* tlas, uint32_t n_tlas, const float* blas, uint32_t n_blas) {
  if (!c) return RT_ERR_INVALID;
  c->epoch++;
  c->world.static_cached = false;   // host arrays replace what the device update left in the scene buffers   // drops frames traced ahead (rt_set_lookahead)
  if ((n_tlas && !tlas) || (n_blas && !blas)) return fail(c, RT_ERR_INVALID, "null BVH array");
  HIP_TRY(c, hipSetDevice(c->device));
  const size_t total = ((size_t)n_tlas + n_blas) * sizeof(rt_node);
  int r = ensure_buffer(c, c->nodes, total, true);
  if (r < 0) return r;
  if (n_tlas)
    HIP_TRY(c, hipMemcpyAsync(c->nodes.ptr, tlas, (size_t)n_tlas * 32, hipMemcpyHostToDevice, c->stream));
  if (n_blas)
    HIP_TRY(c, hipMemcpyAsync((char*)c->nodes.ptr + (size_t)n_tlas * 32, blas, (size_t)n_blas * 32,
                              hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->blas_offset = n_tlas;  // this.blasOffset = tlas.length / 8
  c->n_nodes = n_tlas + n_blas;
  {  // inner nodes (data word 0): one child-pair record each
    uint32_t inner = 0;
    const uint32_t* tw = reinterpret_cast<const uint32_t*>(tlas);
    const uint32_t* bw = reinterpret_cast<const uint32_t*>(blas);
    for (uint32_t k = 0; k < n_tlas; k++) inner += tw[8 * (size_t)k + 7] == 0u;
    for (uint32_t k = 0; k < n_blas; k++) inner += bw[8 * (size_t)k + 7] == 0u;
    c->n_pairs = inner;
    std::memset(&c->troot, 0, sizeof(c->troot));
    if (n_tlas) {   // node 0; an inner root is the first inner node of the array: pair record 0
      for (int k = 0; k < 3; k++) {
        c->troot.lo[k] = tlas[k];
        c->troot.hi[k] = tlas[4 + k];
      }
      c->troot.word = tw[7] == 0u ? RT_PAIR_INNER : tw[7];
    }
  }
  c->roots_dirty = true;
  c->validate_dirty = true;
  c->nodes_dirty = true;
  c->pairs_dirty = true;
  c->nodes_from_device = false;
  return r ? RT_REALLOCATED : RT_OK;
}

int rt_set_scene(rt_ctx* c, const float camera[24], uint32_t frame_count, uint32_t light_count) {
  if (!c || !camera) return RT_ERR_INVALID;
  c->epoch++;   // drops frames traced ahead (rt_set_lookahead)
  c->light_count = light_count;
  step_jitter(c, frame_count, frame_count);
  std::memcpy(&c->uniforms.camera, camera, 96);
  std::memcpy(&c->uniforms.prev_camera, c->prev_camera, 96);
  write_mixed(c, frame_count);
  std::memcpy(c->prev_camera, camera, 96);
  return RT_OK;
}

int rt_recreate_bind_group(rt_ctx* c) { return c ? RT_OK : RT_ERR_INVALID; }

// What one workgroup stages in LDS behind its wave queues, given `budget` bytes of LDS per workgroup: the tnodes, the
// triangle records and the instance rows + BLAS roots, each if it fits whole.
// *dyn_bytes = dynamic LDS size of the launch.
static rtk::LdsPlan plan_lds(const rt_ctx* c, size_t budget, size_t queue_bytes, size_t* dyn_bytes) {
  rtk::LdsPlan P;
  P.k_nodes = P.stage_inst = P.stage_tri = P.pad = 0;
  if (c->no_lds_staging) {
    *dyn_bytes = queue_bytes;
    return P;
  }
  budget &= ~(size_t)2047;   // LDS is allocated in granules: leave room so that the intended number of workgroups fits a CU
  size_t avail = budget > queue_bytes ? budget - queue_bytes : 0;
  avail &= ~(size_t)15;
  // Nodes: all of them or none.  A partial treelet (the most visited nodes in LDS, the rest behind the L1) was measured
  // at 350 ... 3 200 nodes and never paid; MI355RT_TREELET_MAX = n stages min(n, what fits) for sweeps.
  size_t k = (size_t)c->n_nodes * 32 <= avail ? c->n_nodes : 0;
  if (c->treelet_cap >= 0) k = std::min<size_t>(std::min<size_t>(c->n_nodes, avail / 32), (size_t)c->treelet_cap);
  P.k_nodes = (uint32_t)k;
  avail -= k * 32;
  const size_t tri_bytes = (size_t)c->n_tris * 16 * RT_TRI_STRIDE;
  if (tri_bytes <= avail) {
    P.stage_tri = 1;
    avail -= tri_bytes;
  }
  const size_t inst_bytes = (size_t)c->n_instances * 64 + (((size_t)c->n_instances + 3) / 4) * 16;
  if (inst_bytes <= avail) {
    P.stage_inst = 1;
    avail -= inst_bytes;
  }
  *dyn_bytes = queue_bytes + (size_t)P.k_nodes * 32 + (P.stage_tri ? tri_bytes : 0) + (P.stage_inst ? inst_bytes : 0);
  return P;
}

// The same for the child-pair walk of the trace kernels: pair records, triangle records, instance rows + root records.
static rtk::PairPlan plan_pairs(const rt_ctx* c, size_t budget, size_t queue_bytes, size_t* dyn_bytes) {
  rtk::PairPlan P;
  P.stage_pairs = P.stage_inst = P.stage_tri = P.pad = 0;
  if (c->no_lds_staging) {
    *dyn_bytes = queue_bytes;
    return P;
  }
  budget &= ~(size_t)2047;
  size_t avail = budget > queue_bytes ? budget - queue_bytes : 0;
  avail &= ~(size_t)15;
  const size_t pair_bytes = (size_t)c->n_pairs * 64, tri_bytes = (size_t)c->n_tris * 16 * RT_TRI_STRIDE,
               inst_bytes = (size_t)c->n_instances * 96;
  if (pair_bytes <= avail) {
    P.stage_pairs = 1;
    avail -= pair_bytes;
  }
  if (tri_bytes <= avail) {
    P.stage_tri = 1;
    avail -= tri_bytes;
  }
  if (inst_bytes <= avail) {
    P.stage_inst = 1;
    avail -= inst_bytes;
  }
  *dyn_bytes = queue_bytes + (P.stage_pairs ? pair_bytes : 0) + (P.stage_tri ? tri_bytes : 0) + (P.stage_inst ? inst_bytes : 0);
  return P;
}

extern "C++" {
template <int BLOCK>
static const void* wf_trace_fn(bool any, bool detail, bool lds, bool rayreg) {
  if (BLOCK == 256 && rayreg && !lds) {   // instance-space ray in registers: compiled for the default workgroup size only
    if (any) return detail ? (const void*)rtk::k_wf_trace<true, true, false, 256, true> : (const void*)rtk::k_wf_trace<true, false, false, 256, true>;
    return detail ? (const void*)rtk::k_wf_trace<false, true, false, 256, true> : (const void*)rtk::k_wf_trace<false, false, false, 256, true>;
  }
  if (any) {
    if (detail) return lds ? (const void*)rtk::k_wf_trace<true, true, true, BLOCK> : (const void*)rtk::k_wf_trace<true, true, false, BLOCK>;
    return lds ? (const void*)rtk::k_wf_trace<true, false, true, BLOCK> : (const void*)rtk::k_wf_trace<true, false, false, BLOCK>;
  }
  if (detail) return lds ? (const void*)rtk::k_wf_trace<false, true, true, BLOCK> : (const void*)rtk::k_wf_trace<false, true, false, BLOCK>;
  return lds ? (const void*)rtk::k_wf_trace<false, false, true, BLOCK> : (const void*)rtk::k_wf_trace<false, false, false, BLOCK>;
}
}  // extern "C++"
extern "C++" {
template <int BLOCK>
static const void* wf_trace_pairs_fn(bool any, bool detail, bool lds) {
  if (any) {
    if (detail) return lds ? (const void*)rtk::k_wf_trace_pairs<true, true, true, BLOCK> : (const void*)rtk::k_wf_trace_pairs<true, true, false, BLOCK>;
    return lds ? (const void*)rtk::k_wf_trace_pairs<true, false, true, BLOCK> : (const void*)rtk::k_wf_trace_pairs<true, false, false, BLOCK>;
  }
  if (detail) return lds ? (const void*)rtk::k_wf_trace_pairs<false, true, true, BLOCK> : (const void*)rtk::k_wf_trace_pairs<false, true, false, BLOCK>;
  return lds ? (const void*)rtk::k_wf_trace_pairs<false, false, true, BLOCK> : (const void*)rtk::k_wf_trace_pairs<false, false, false, BLOCK>;
}
}  // extern "C++"

// Wavefront form: per depth one shade launch and two trace launches, all enqueued without host readback.
static int launch_wavefront(rt_ctx* c, const DevScene& S, const DevFrame& F, const DevFrameSlot* dslots, uint32_t n,
                            bool fits_lds) {
  const size_t npx = (size_t)c->width * c->height;
  const size_t items = npx * n;
  if (items >= (1ull << 31)) return fail(c, RT_ERR_INVALID, "batch too large for the wavefront queues");
  int r;
  // queues are reserved in chunks of RT_WF_CHUNK per wave: room for every item plus one partial chunk per wave
  // shade kernels: persistent waves that loop over the items; every wave may leave one partly used chunk per queue
  const uint32_t shade_blocks = (uint32_t)std::min<size_t>((items + 255) / 256, (size_t)c->num_cus * (size_t)c->shade_per_cu);
  const size_t qcap = items + (size_t)shade_blocks * 4 * 256 + 1024 * 1024;
  // per item: active[2] + shadow ids + ext ids + occlusion word (5 x 4 B) + shadow rays, extension rays of even and of odd
  // depths (3 x 32 B) + hits (16 B)
  r = ensure_buffer(c, c->wf_queues, qcap * 132, false);
  if (r < 0) return r;
  r = ensure_buffer(c, c->wf_state, 2 * qcap * sizeof(WfPath), false);   // path records in list order, double-buffered by depth parity
  if (r < 0) return r;
  const uint32_t depths = c->max_depth ? c->max_depth : 1u;
  r = ensure_buffer(c, c->wf_counters, (size_t)(depths + 2) * 32, false);
  if (r < 0) return r;
  HIP_TRY(c, hipMemsetAsync(c->wf_counters.ptr, 0, (size_t)(depths + 2) * 32, c->stream));
  WfState W;
  W.p[0] = (WfPath*)c->wf_state.ptr;
  W.p[1] = W.p[0] + qcap;
  WfQueues Q;
  char* qb = (char*)c->wf_queues.ptr;
  Q.shadow_rays = (float4*)qb;
  Q.ext_rays[0] = (float4*)(qb + qcap * 32);
  Q.ext_rays[1] = (float4*)(qb + qcap * 64);
  Q.ext_hit = (float4*)(qb + qcap * 96);
  Q.active[0] = (uint32_t*)(qb + qcap * 112);
  Q.active[1] = (uint32_t*)(qb + qcap * 116);
  Q.shadow_ids = (uint32_t*)(qb + qcap * 120);
  Q.ext_ids = (uint32_t*)(qb + qcap * 124);
  Q.occluded = (uint32_t*)(qb + qcap * 128);
  Q.counters = (uint32_t*)c->wf_counters.ptr;
  const bool detail = c->detailed_counters;
  const bool pairs = c->walk == 1 || (c->walk == 2 && c->n_instances == 1);   // rt_set_walk
  int block = 256, blocks_per_cu = 0;
  size_t dyn = 0;
  bool trace_lds = false;
  rtk::PairPlan plan;
  rtk::LdsPlan nplan;
  const void* trace_fn[2];
  if (pairs) {
    // Workgroup shape of the trace kernels.  Every wave owns RT_PW_BYTES_PER_WAVE of LDS (triangle work queue + the stack of
    // deferred right children).  Everything fits beside four wave blocks in 64 KB: 256-thread workgroups, all records in
    // LDS.  Otherwise 256-thread workgroups, as many per CU as the wave blocks allow (4 at K = 8), each staging what fits
    // whole in its share of the LDS (plan_pairs); MI355RT_WF_BLOCK / MI355RT_WF_BLOCKS_PER_CU override the shape for sweeps.
    const size_t lds_records = ((size_t)4 * c->n_pairs + (size_t)RT_TRI_STRIDE * c->n_tris + (size_t)6 * c->n_instances) * 16;
    trace_lds = !c->no_lds_staging && fits_lds && lds_records + (size_t)4 * RT_PW_BYTES_PER_WAVE <= 64 * 1024;
    if (!trace_lds) {
      block = c->wf_block ? c->wf_block : 256;
      const int fit = (int)(c->lds_per_cu / ((size_t)(block / 64) * RT_PW_BYTES_PER_WAVE));
      blocks_per_cu = c->wf_blocks_per_cu ? c->wf_blocks_per_cu : std::max(1, std::min(fit, (RT_WF_WAVES * 256) / block));
    }
    const size_t queue_bytes = (size_t)(block / 64) * RT_PW_BYTES_PER_WAVE;
    dyn = queue_bytes + lds_records;
    plan.stage_pairs = plan.stage_inst = plan.stage_tri = 1;
    plan.pad = 0;
    if (!trace_lds) plan = plan_pairs(c, c->lds_per_cu / (size_t)blocks_per_cu, queue_bytes, &dyn);
    plan.troot = c->troot;
    for (int k = 0; k < 2; k++)
      trace_fn[k] = block == 1024 ? wf_trace_pairs_fn<1024>(k == 0, detail, trace_lds)
                                  : (block == 512 ? wf_trace_pairs_fn<512>(k == 0, detail, trace_lds) : wf_trace_pairs_fn<256>(k == 0, detail, trace_lds));
    if (c->wf_occ_dyn != dyn || c->wf_occ_detail != (int)detail || c->wf_occ_block != block || c->wf_occ_walk != (int)pairs || c->wf_occ_blocks[0] == 0) {
      for (int k = 0; k < 2; k++) {
        HIP_TRY(c, hipFuncSetAttribute(trace_fn[k], hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
        int per_cu = 0;
        HIP_TRY(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, trace_fn[k], block, dyn));
        c->wf_occ_blocks[k] = per_cu < 1 ? 1 : per_cu;
      }
      c->wf_occ_dyn = dyn;
      c->wf_occ_detail = (int)detail;
      c->wf_occ_block = block;
      c->wf_occ_walk = (int)pairs;
    }
  } else {
    // Workgroup shape of the trace kernels.  Everything fits beside four wave queues in 64 KB: 256-thread workgroups, all
    // records in LDS.  Otherwise six 256-thread workgroups per CU (6 waves per SIMD), each staging what fits whole in its
    // sixth of the LDS (plan_lds); MI355RT_WF_BLOCK / MI355RT_WF_BLOCKS_PER_CU override the shape for sweeps.
    const size_t lds_records = ((size_t)2 * c->n_nodes + (size_t)RT_TRI_STRIDE * c->n_tris + (size_t)4 * c->n_instances + ((size_t)c->n_instances + 3) / 4) * 16;
    trace_lds = !c->no_lds_staging && fits_lds && lds_records + (size_t)4 * RT_WORK_BYTES_PER_WAVE <= 64 * 1024;
    if (!trace_lds) {
      block = c->wf_block ? c->wf_block : 256;
      blocks_per_cu = c->wf_blocks_per_cu ? c->wf_blocks_per_cu : (block == 1024 ? 1 : (block == 512 ? 2 : 6));
    }
    const size_t queue_bytes = (size_t)(block / 64) * RT_WORK_BYTES_PER_WAVE;
    dyn = queue_bytes + lds_records;
    nplan.k_nodes = c->n_nodes;
    nplan.stage_inst = nplan.stage_tri = 1;
    nplan.pad = 0;
    if (!trace_lds) nplan = plan_lds(c, c->lds_per_cu / (size_t)blocks_per_cu, queue_bytes, &dyn);
    // few instances with deep trees (glass blob: 3 instances, 400 k nodes): a ray enters an instance once and then waits at
    // many leaves; measured, the form that keeps its instance-space origin / direction in registers is the faster one there,
    // the other one where rays enter many small instances (k_traverse.hip.h, trav_post_at_entry; MI355RT_WF_RAYREG=0/1 overrides)
    const bool rayreg = c->wf_rayreg < 0 ? (size_t)c->n_nodes >= (size_t)1024 * std::max<size_t>(1, c->n_instances) : c->wf_rayreg != 0;
    for (int k = 0; k < 2; k++)
      trace_fn[k] = block == 1024 ? wf_trace_fn<1024>(k == 0, detail, trace_lds, rayreg)
                                  : (block == 512 ? wf_trace_fn<512>(k == 0, detail, trace_lds, rayreg) : wf_trace_fn<256>(k == 0, detail, trace_lds, rayreg));
    if (c->wf_occ_dyn != dyn || c->wf_occ_detail != (int)detail || c->wf_occ_block != block || c->wf_occ_walk != (rayreg ? 2 : 0) || c->wf_occ_blocks[0] == 0) {
      for (int k = 0; k < 2; k++) {
        HIP_TRY(c, hipFuncSetAttribute(trace_fn[k], hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
        int per_cu = 0;
        HIP_TRY(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, trace_fn[k], block, dyn));
        c->wf_occ_blocks[k] = per_cu < 1 ? 1 : per_cu;
      }
      c->wf_occ_dyn = dyn;
      c->wf_occ_detail = (int)detail;
      c->wf_occ_block = block;
      c->wf_occ_walk = rayreg ? 2 : 0;
    }
  }
  if (getenv("MI355RT_DEBUG_SHAPE"))
    fprintf(stderr, "[mi355rt] trace kernels: %s walk, %d threads per workgroup, %zu bytes of LDS, resident workgroups per CU: any-hit %d, closest-hit %d\n",
            pairs ? "pair" : "node", block, dyn, c->wf_occ_blocks[0], c->wf_occ_blocks[1]);
  uint32_t nn = pairs ? c->n_pairs : c->n_nodes, nt = c->n_tris, ni = c->n_instances;
  EventPair* ev = next_events(c, RT_TIMER_PATHTRACE);
  if (ev) HIP_TRY(c, hipEventRecord(ev->a, c->stream));
  for (uint32_t depth = 0; depth < depths; depth++) {
    EventPair* evs = next_events(c, RT_TIMER_WF_SHADE);
    if (evs) HIP_TRY(c, hipEventRecord(evs->a, c->stream));
    if (depth == 0) {
      if (detail)
        hipLaunchKernelGGL((rtk::k_wf_shade<true, true>), dim3(shade_blocks), dim3(256), 0, c->stream, S, F, c->uniforms, W, Q, dslots, n, depth);
      else
        hipLaunchKernelGGL((rtk::k_wf_shade<true, false>), dim3(shade_blocks), dim3(256), 0, c->stream, S, F, c->uniforms, W, Q, dslots, n, depth);
    } else {
      if (detail)
        hipLaunchKernelGGL((rtk::k_wf_shade<false, true>), dim3(shade_blocks), dim3(256), 0, c->stream, S, F, c->uniforms, W, Q, dslots, n, depth);
      else
        hipLaunchKernelGGL((rtk::k_wf_shade<false, false>), dim3(shade_blocks), dim3(256), 0, c->stream, S, F, c->uniforms, W, Q, dslots, n, depth);
    }
    if (evs) HIP_TRY(c, hipEventRecord(evs->b, c->stream));
    for (int k = 0; k < 2; k++) {
      // resident workgroups: what fits (registers, LDS; since round 4 the node-walk kernels leave room for a seventh wave per
      // SIMD), or MI355RT_WF_BLOCKS_PER_CU
      uint32_t per_cu = (uint32_t)c->wf_occ_blocks[k];
      if (c->wf_blocks_per_cu > 0 && per_cu > (uint32_t)c->wf_blocks_per_cu) per_cu = (uint32_t)c->wf_blocks_per_cu;
      uint32_t blocks = per_cu * (uint32_t)c->num_cus;
      const uint32_t max_useful = (uint32_t)std::min<size_t>((items + (size_t)block - 1) / (size_t)block, (size_t)0x7fffffff);
      if (blocks > max_useful) blocks = max_useful ? max_useful : 1;
      DevScene Sa = S;
      DevFrame Fa = F;
      rt_scene_uniforms Ua = c->uniforms;
      void* args[] = {&Sa, &Fa, &Ua, &Q, &depth, &nn, &nt, &ni, pairs ? (void*)&plan : (void*)&nplan};
      EventPair* evt = next_events(c, k == 0 ? RT_TIMER_WF_TRACE_SHADOW : RT_TIMER_WF_TRACE_EXT);
      hipStream_t st = c->stream;
      if (k == 0 && c->wf_overlap) {   // any-hit trace: fork to the side stream behind this depth's shade kernel
        st = c->side_stream;
        HIP_TRY(c, hipEventRecord(c->side_fork, c->stream));
        HIP_TRY(c, hipStreamWaitEvent(st, c->side_fork, 0));
      }
      if (evt) HIP_TRY(c, hipEventRecord(evt->a, st));
      HIP_TRY(c, hipLaunchKernel(trace_fn[k], dim3(blocks), dim3(block), args, dyn, st));
      if (evt) HIP_TRY(c, hipEventRecord(evt->b, st));
      if (k == 0 && c->wf_overlap) HIP_TRY(c, hipEventRecord(c->side_join, st));
    }
    if (c->wf_overlap) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->side_join, 0));   // the next shade needs both results
  }
  {
    // one more shade pass: the paths that ended at the last depth but were waiting for their shadow ray are finished
    // here (every path of this list carries the ENDED flag, so nothing is shaded or queued)
    EventPair* evs = next_events(c, RT_TIMER_WF_SHADE);
    if (evs) HIP_TRY(c, hipEventRecord(evs->a, c->stream));
    const uint32_t depth = depths;
    if (detail)
      hipLaunchKernelGGL((rtk::k_wf_shade<false, true>), dim3(shade_blocks), dim3(256), 0, c->stream, S, F, c->uniforms, W, Q, dslots, n, depth);
    else
      hipLaunchKernelGGL((rtk::k_wf_shade<false, false>), dim3(shade_blocks), dim3(256), 0, c->stream, S, F, c->uniforms, W, Q, dslots, n, depth);
    if (evs) HIP_TRY(c, hipEventRecord(evs->b, c->stream));
  }
  if (ev) HIP_TRY(c, hipEventRecord(ev->b, c->stream));
  hipLaunchKernelGGL(rtk::k_accumulate_frames, dim3((uint32_t)((npx + 255) / 256)), dim3(256), 0, c->stream, F, dslots, c->acc_frames,
                     c->width, c->height);
  HIP_TRY(c, hipGetLastError());
  return RT_OK;
}

// compute() for n consecutive frame counts in ONE dispatch of each kernel (n == 1: the plain compute()).
// n_commit < n: frames n_commit .. n-1 are traced AHEAD (speculative lookahead): their host state is not committed and their
// colours are not accumulated yet — consume_ahead() does both when the matching compute() call arrives.
static int compute_frames(rt_ctx* c, const uint32_t* frame_counts, uint32_t n, uint32_t n_commit) {
  if (!c || !frame_counts || n == 0 || n_commit == 0 || n_commit > n) return RT_ERR_INVALID;
  c->spec.valid = false;   // whatever was traced ahead is dropped: its buffers are about to be reused
  c->gbuf_slot = -1;
  // host state advances exactly as n successive compute() calls would (WebGPURenderer.ts:88-91)
  std::vector<DevFrameSlot> slots(n);
  const uint32_t tf_first = c->total_frames + 1;
  struct HostState {
    uint32_t total_frames;
    double jx, jy, acc_jx, acc_jy, avg_jx, avg_jy;
    rt_scene_uniforms uniforms;
  } committed = {};
  for (uint32_t i = 0; i < n; i++) {
    c->total_frames++;
    step_jitter(c, c->total_frames, frame_counts[i]);  // updateFrameUniforms(frameCount, totalFrames)
    write_mixed(c, frame_counts[i]);
    slots[i].frame_count = frame_counts[i];
    slots[i].jitter_x = c->uniforms.jitter[0];
    slots[i].jitter_y = c->uniforms.jitter[1];
    slots[i].pad = 0;
    slots[i].pad2 = 0;
    if (i + 1 == n_commit) committed = {c->total_frames, c->jx, c->jy, c->acc_jx, c->acc_jy, c->avg_jx, c->avg_jy, c->uniforms};
  }
  // the renderer's host state is that of the committed frames only
  c->total_frames = committed.total_frames;
  c->jx = committed.jx; c->jy = committed.jy;
  c->acc_jx = committed.acc_jx; c->acc_jy = committed.acc_jy;
  c->avg_jx = committed.avg_jx; c->avg_jy = committed.avg_jy;
  c->uniforms = committed.uniforms;
  c->acc_frames = n_commit;
  if (!scene_ready(c)) return RT_SKIPPED;
  if (c->accum_stale)
    return fail(c, RT_ERR_INVALID, "the bound accumulation buffer was dropped by rt_resize: call rt_bind_accum again");
  HIP_TRY(c, hipSetDevice(c->device));
  // Host-side shape checks before any kernel indexes these buffers.
  if (c->uniforms.light_count > c->n_lights)
    return fail(c, RT_ERR_INVALID, "light_count exceeds the uploaded lights buffer");
  if (c->blas_offset > c->n_nodes) return fail(c, RT_ERR_INVALID, "blas_base_idx exceeds the node buffer");
  if (c->variant == 0 && n > 1) return fail(c, RT_ERR_INVALID, "batched dispatch needs the persistent or wavefront kernel form");
  int r = prepare_scene(c);
  if (r < 0) return r;

  const size_t npx = (size_t)c->width * c->height;
  const size_t scene_lds = rtk::scene_lds_slots(c->n_nodes, c->n_tris, c->n_instances, c->n_verts, c->n_lights) * 16;
  const bool fits_lds = !c->no_lds_staging && scene_lds + (size_t)4 * RT_WORK_BYTES_PER_WAVE <= 64 * 1024;
  // auto: the wavefront form pays from 4 frames per dispatch (measured: 1 frame 11.9 vs 8.7 ms persistent, 2: 17.0 vs
  // 15.7, 4: 27.8 vs 29.5, 8: 47.9 vs 57.0 on sponza-like) — a single frame leaves its deeper stages too few rays
  const bool wavefront = c->spp == 1 && (c->variant == 2 || (c->variant == 3 && !fits_lds && n >= 4));
  if (n > 1) {
    r = ensure_buffer(c, c->gbuf_batch, (size_t)(n - 1) * npx * 24, false);
    if (r < 0) return r;
  }
  if (n > 1 || wavefront) {
    r = ensure_buffer(c, c->frame_col, (size_t)n * npx * 16, false);
    if (r < 0) return r;
  }
  for (uint32_t i = 0; i < n; i++) {
    if (i + 1 == n) {  // the last frame lands in the main G-buffer, like a plain compute()
      slots[i].albedo = (uint32_t*)c->render_target.ptr;
      slots[i].normal_id = (float4*)c->g_normal.ptr;
      slots[i].depth = (float*)c->g_depth.ptr;
    } else {
      char* base = (char*)c->gbuf_batch.ptr + (size_t)i * npx * 24;
      slots[i].normal_id = (float4*)base;
      slots[i].albedo = (uint32_t*)(base + npx * 16);
      slots[i].depth = (float*)(base + npx * 20);
    }
  }
  // The table goes through a pinned ring, one device table per ring entry: the copy is asynchronous, its source
  // outlives it, and a dispatch still in flight keeps reading its own table while the next one is being written.
  if (!c->slot_ring) {
    HIP_TRY(c, hipHostMalloc((void**)&c->slot_ring, (size_t)rt_ctx::kSlotRing * 64 * sizeof(DevFrameSlot), hipHostMallocDefault));
    for (int k = 0; k < rt_ctx::kSlotRing; k++) HIP_TRY(c, hipEventCreateWithFlags(&c->slot_ring_ev[k], hipEventDisableTiming));
  }
  r = ensure_buffer(c, c->slots, (size_t)rt_ctx::kSlotRing * 64 * sizeof(DevFrameSlot), false);
  if (r < 0) return r;
  const int ring = c->slot_ring_next;
  c->slot_ring_next = (ring + 1) % rt_ctx::kSlotRing;
  if (c->slot_ring_used[ring]) HIP_TRY(c, hipEventSynchronize(c->slot_ring_ev[ring]));  // the dispatch that used this entry is done
  DevFrameSlot* host_slots = c->slot_ring + (size_t)ring * 64;
  std::memcpy(host_slots, slots.data(), (size_t)n * sizeof(DevFrameSlot));
  DevFrameSlot* dev_slots = (DevFrameSlot*)c->slots.ptr + (size_t)ring * 64;
  HIP_TRY(c, hipMemcpyAsync(dev_slots, host_slots, (size_t)n * sizeof(DevFrameSlot), hipMemcpyHostToDevice, c->stream));
  const DevFrameSlot* dslots = dev_slots;

  DevScene S = dev_scene(c);
  DevFrame F;
  F.accum = accum_ptr(c);
  F.frame_col = (n > 1 || wavefront) ? (float4*)c->frame_col.ptr : nullptr;
  F.albedo = (uint32_t*)c->render_target.ptr;
  F.normal_id = (float4*)c->g_normal.ptr;
  F.depth = (float*)c->g_depth.ptr;
  F.counters = (uint64_t*)c->counters.ptr;
  F.max_depth = c->max_depth;
  F.spp = c->spp;
  F.stripe_rows = c->stripe_rows;
  F.stripe_rank = c->stripe_rank;
  F.stripe_count = c->stripe_count;
  F.own_run = F.own_period = F.own_first = F.own_tile_rows = 0;
  if (c->stripe_count > 1 && c->stripe_rows && c->stripe_rows % 8 == 0) {
    const uint32_t tile_rows = (c->height + 7) / 8;
    F.own_run = c->stripe_rows / 8;
    F.own_period = F.own_run * c->stripe_count;
    F.own_first = F.own_run * c->stripe_rank;
    uint32_t owned = 0;
    for (uint32_t ty = 0; ty < tile_rows; ty++)
      if ((ty / F.own_run) % c->stripe_count == c->stripe_rank) owned++;
    F.own_tile_rows = owned;
  }
  DevFrame Fp = F;  // the primary kernel counts into bank 0, the path tracer into bank 1
  F.counters = (uint64_t*)c->counters.ptr + (size_t)RT_COUNTER_SHARDS * 6;
  const uint32_t tiles = ((c->width + 7) / 8) * ((c->height + 7) / 8);
  const uint32_t ptiles = F.own_period ? ((c->width + 7) / 8) * F.own_tile_rows : tiles;  // primary kernel grid

  // 1. primary visibility (the reference clears + rasterises the G-buffer every compute()); frame = blockIdx.y
  EventPair* ev = next_events(c, RT_TIMER_PRIMARY);
  if (ev) HIP_TRY(c, hipEventRecord(ev->a, c->stream));
  {
    const size_t plds = rtk::primary_lds_slots(c->n_nodes, c->n_tris, c->n_instances, c->n_verts) * 16;
    const uint32_t nn = c->n_nodes, nt = c->n_tris, ni = c->n_instances, nv = c->n_verts;
    if (plds <= 32 * 1024 && !c->no_lds_staging) {  // small scene: records staged in LDS, four tiles per workgroup
      const dim3 grid((ptiles + 3) / 4, n);
      if (c->detailed_counters)
        hipLaunchKernelGGL((rtk::k_primary_visibility<true, true>), grid, dim3(256), plds, c->stream, S, Fp, c->uniforms, dslots, ptiles, nn, nt, ni, nv);
      else
        hipLaunchKernelGGL((rtk::k_primary_visibility<false, true>), grid, dim3(256), plds, c->stream, S, Fp, c->uniforms, dslots, ptiles, nn, nt, ni, nv);
    } else if (c->detailed_counters) {
      hipLaunchKernelGGL((rtk::k_primary_visibility<true, false>), dim3(ptiles, n), dim3(64), 0, c->stream, S, Fp, c->uniforms, dslots, ptiles, nn, nt, ni, nv);
    } else {
      hipLaunchKernelGGL((rtk::k_primary_visibility<false, false>), dim3(ptiles, n), dim3(64), 0, c->stream, S, Fp, c->uniforms, dslots, ptiles, nn, nt, ni, nv);
    }
  }
  if (ev) HIP_TRY(c, hipEventRecord(ev->b, c->stream));

  // 2. path trace
  if (wavefront) {
    r = launch_wavefront(c, S, F, dslots, n, fits_lds);
    if (r < 0) return r;
  } else if (c->variant == 0) {
    ev = next_events(c, RT_TIMER_PATHTRACE);
    if (ev) HIP_TRY(c, hipEventRecord(ev->a, c->stream));
    if (c->detailed_counters)
      hipLaunchKernelGGL(rtk::k_pathtrace<true>, dim3(tiles), dim3(64), 0, c->stream, S, F, c->uniforms);
    else
      hipLaunchKernelGGL(rtk::k_pathtrace<false>, dim3(tiles), dim3(64), 0, c->stream, S, F, c->uniforms);
    if (ev) HIP_TRY(c, hipEventRecord(ev->b, c->stream));
  } else {
    // persistent kernel: grid = resident workgroups, tiles handed out through a ticket counter
    HIP_TRY(c, hipMemsetAsync(c->ticket.ptr, 0, 4, c->stream));
    const size_t lds_bytes = rtk::scene_lds_slots(c->n_nodes, c->n_tris, c->n_instances, c->n_verts, c->n_lights) * 16;
    const bool use_lds = !c->no_lds_staging && lds_bytes + (size_t)4 * RT_WORK_BYTES_PER_WAVE <= 64 * 1024;
    size_t dyn = (size_t)4 * RT_WORK_BYTES_PER_WAVE + lds_bytes;  // work queues + records
    rtk::LdsPlan plan;
    plan.k_nodes = c->n_nodes;
    plan.stage_inst = plan.stage_tri = 1;
    plan.pad = 0;
    // a scene that does not fit as a whole: six 256-thread workgroups per CU (6 waves / SIMD), each with its share of the
    // CU's LDS for the top of the tree
    if (!use_lds) plan = plan_lds(c, c->lds_per_cu / 6, (size_t)4 * RT_WORK_BYTES_PER_WAVE, &dyn);
    const int vi = (c->detailed_counters ? 2 : 0) + (use_lds ? 1 : 0);
    static const void* const fns[4] = {(const void*)rtk::k_pathtrace_persistent<false, false>,
                                       (const void*)rtk::k_pathtrace_persistent<false, true>,
                                       (const void*)rtk::k_pathtrace_persistent<true, false>,
                                       (const void*)rtk::k_pathtrace_persistent<true, true>};
    const void* fn = fns[vi];
    // resident workgroups per CU: queried once per (variant, LDS size)
    if (c->occ_dyn[vi] != dyn || c->occ_blocks[vi] == 0) {
      int per_cu = 0;
      HIP_TRY(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
      HIP_TRY(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, dyn));
      c->occ_blocks[vi] = per_cu < 1 ? 1 : per_cu;
      c->occ_dyn[vi] = dyn;
    }
    uint32_t blocks = (uint32_t)c->occ_blocks[vi] * (uint32_t)c->num_cus;
    const uint32_t own_tiles = F.own_period ? ((c->width + 7) / 8) * F.own_tile_rows : tiles;
    // as many waves as there are tickets (tiles x frames), up to the resident limit; fewer, longer-lived waves measured worse
    const uint32_t max_useful = (own_tiles * n + 3) / 4;
    if (blocks > max_useful) blocks = max_useful ? max_useful : 1;
    uint32_t* ticket = (uint32_t*)c->ticket.ptr;
    uint32_t nn = c->n_nodes, nt = c->n_tris, ni = c->n_instances, nv = c->n_verts, ns = n;
    void* args[] = {&S, &F, &c->uniforms, &ticket, &nn, &nt, &ni, &nv, &dslots, &ns, &plan};
    ev = next_events(c, RT_TIMER_PATHTRACE);
    if (ev) HIP_TRY(c, hipEventRecord(ev->a, c->stream));
    HIP_TRY(c, hipLaunchKernel(fn, dim3(blocks), dim3(256), args, dyn, c->stream));
    if (ev) HIP_TRY(c, hipEventRecord(ev->b, c->stream));
    if (n > 1)  // ordered accumulation of the batch's frame colours
      hipLaunchKernelGGL(rtk::k_accumulate_frames, dim3((uint32_t)((npx + 255) / 256)), dim3(256), 0, c->stream, F, dslots,
                         c->acc_frames, c->width, c->height);
  }
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipEventRecord(c->slot_ring_ev[ring], c->stream));
  c->slot_ring_used[ring] = true;
  if (n_commit < n) {   // frames traced ahead: remember where their colours and G-buffers are
    c->spec.valid = true;
    c->spec.epoch = c->epoch;
    c->spec.n = n;
    c->spec.k = n_commit;
    c->spec.fc0 = frame_counts[0];
    c->spec.tf0 = tf_first;
    c->spec.dslots = dslots;
    c->spec.slots = slots;
    c->spec.frame = F;
    c->gbuf_slot = (int)n_commit - 1;
  }
  return RT_OK;
}

// compute(frame_count) when that frame has been traced ahead: commit its host state, add its colours.
static int consume_ahead(rt_ctx* c, uint32_t frame_count) {
  auto& sp = c->spec;
  const uint32_t k = sp.k;
  c->total_frames++;
  step_jitter(c, c->total_frames, frame_count);
  write_mixed(c, frame_count);
  if (c->uniforms.jitter[0] != sp.slots[k].jitter_x || c->uniforms.jitter[1] != sp.slots[k].jitter_y)
    return fail(c, RT_ERR_INTERNAL, "lookahead: the frame traced ahead does not match the frame asked for");
  HIP_TRY(c, hipSetDevice(c->device));
  DevFrame F = sp.frame;
  const size_t npx = (size_t)c->width * c->height;
  F.accum = accum_ptr(c);
  F.frame_col = sp.frame.frame_col + (size_t)k * npx;
  hipLaunchKernelGGL(rtk::k_accumulate_frames, dim3((uint32_t)((npx + 255) / 256)), dim3(256), 0, c->stream, F, sp.dslots + k, 1u,
                     c->width, c->height);
  HIP_TRY(c, hipGetLastError());
  c->gbuf_slot = (int)k;
  sp.k++;
  if (sp.k >= sp.n) sp.valid = false;
  return RT_OK;
}

int rt_compute(rt_ctx* c, uint32_t frame_count) {
  if (!c) return RT_ERR_INVALID;
  if (c->lookahead_max <= 1 || c->detailed_counters || c->variant == 0 || c->accum_stale) return compute_frames(c, &frame_count, 1, 1);
  // a frame traced ahead by an earlier call of this run?
  if (c->spec.valid && c->spec.epoch == c->epoch && frame_count == c->spec.fc0 + c->spec.k &&
      c->total_frames + 1 == c->spec.tf0 + c->spec.k && scene_ready(c)) {
    const int r = consume_ahead(c, frame_count);
    c->run_last_fc = frame_count;
    c->run_last_tf = c->total_frames;
    return r;
  }
  // does this call continue a run of consecutive frames?  Then trace ahead, twice as far as last time.
  const bool continues = c->run_epoch == c->epoch && frame_count == c->run_last_fc + 1 && c->total_frames == c->run_last_tf;
  c->look = continues ? std::min<uint32_t>(std::min<uint32_t>(c->look * 2u, c->lookahead_max), 64u) : 1u;
  if (c->look_limit) c->look = std::max<uint32_t>(1u, std::min(c->look, c->look_limit));   // rt_set_lookahead_limit: the caller knows when the run ends
  c->run_epoch = c->epoch;
  uint32_t fcs[64];
  for (uint32_t i = 0; i < c->look; i++) fcs[i] = frame_count + i;
  const int r = compute_frames(c, fcs, c->look, 1);
  c->run_last_fc = frame_count;
  c->run_last_tf = c->total_frames;
  return r;
}

int rt_compute_batch(rt_ctx* c, const uint32_t* frame_counts, uint32_t n) {
  if (!c) return RT_ERR_INVALID;
  if (n > 64) return fail(c, RT_ERR_INVALID, "at most 64 frames per batch");
  c->epoch++;   // an explicit batch ends a run
  return compute_frames(c, frame_counts, n, n);
}

int rt_set_lookahead(rt_ctx* c, uint32_t max_frames) {
  if (!c) return RT_ERR_INVALID;
  if (max_frames > 64) return fail(c, RT_ERR_INVALID, "at most 64 frames of lookahead");
  c->lookahead_max = max_frames;
  c->epoch++;
  return RT_OK;
}

int rt_set_lookahead_limit(rt_ctx* c, uint32_t frames_left) {
  if (!c) return RT_ERR_INVALID;
  c->look_limit = frames_left;   // changes no result and drops nothing: only how far the next dispatches trace ahead
  return RT_OK;
}

int rt_present(rt_ctx* c) {
  if (!c) return RT_ERR_INVALID;
  if (!c->width || !c->render_target.ptr || !c->accum.ptr) return RT_SKIPPED;  // PostProcessPass.ts:32-38
  if (c->accum_stale)
    return fail(c, RT_ERR_INVALID, "the bound accumulation buffer was dropped by rt_resize: call rt_bind_accum again");
  if (c->present_stale)
    return fail(c, RT_ERR_INVALID, "the bound present source was dropped by rt_resize: call rt_bind_present_source again");
  HIP_TRY(c, hipSetDevice(c->device));
  DevPost P;
  P.accum = c->present_source ? (const float4*)c->present_source : accum_ptr(c);
  P.history_in = (const ushort4*)c->history[1 - c->history_index].ptr;  // previous frame (read)
  P.history_out = (ushort4*)c->history[c->history_index].ptr;           // current frame (write)
  P.out_rgba8 = (uint32_t*)c->render_target.ptr;
  dim3 grid((c->width + 15) / 16, (c->height + 15) / 16);
  EventPair* ev = next_events(c, RT_TIMER_POST);
  if (ev) HIP_TRY(c, hipEventRecord(ev->a, c->stream));
  hipLaunchKernelGGL(rtk::k_postprocess, grid, dim3(256), 0, c->stream, P, c->uniforms);
  if (ev) HIP_TRY(c, hipEventRecord(ev->b, c->stream));
  HIP_TRY(c, hipGetLastError());
  c->history_index = 1 - c->history_index;  // swap history index for TAA
  return RT_OK;
}

int rt_capture(rt_ctx* c, uint8_t* out_rgba, size_t cap) {
  if (!c || !out_rgba) return RT_ERR_INVALID;
  if (!c->render_target.ptr) return fail(c, RT_ERR_NOT_READY, "No render target");  // WebGPUContext.ts:43
  const size_t bytes = (size_t)c->width * c->height * 4;
  if (cap < bytes) return fail(c, RT_ERR_INVALID, "capture buffer too small");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipMemcpyAsync(out_rgba, c->render_target.ptr, bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return RT_OK;
}

int rt_sync(rt_ctx* c) {
  if (!c) return RT_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return RT_OK;
}

// ---------------------------------------------------------------- additions
int rt_read_accum(rt_ctx* c, float* out, size_t cap) {
  if (!c || !out) return RT_ERR_INVALID;
  const size_t bytes = (size_t)c->width * c->height * 16;
  if (!c->accum.ptr) return fail(c, RT_ERR_NOT_READY, "no accumulation buffer");
  if (cap < bytes) return fail(c, RT_ERR_INVALID, "buffer too small");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipMemcpyAsync(out, accum_ptr(c), bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return RT_OK;
}
int rt_write_accum(rt_ctx* c, const float* in, size_t bytes) {
  if (!c || !in) return RT_ERR_INVALID;
  c->epoch++;   // drops frames traced ahead (rt_set_lookahead)
  if (!c->accum.ptr || bytes != (size_t)c->width * c->height * 16)
    return fail(c, RT_ERR_INVALID, "accumulation size mismatch");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipMemcpyAsync(accum_ptr(c), in, bytes, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return RT_OK;
}
int rt_read_gbuffer(rt_ctx* c, uint8_t* albedo, float* normal_id, float* depth) {
  if (!c) return RT_ERR_INVALID;
  if (!c->render_target.ptr) return fail(c, RT_ERR_NOT_READY, "no G-buffer");
  const size_t n = (size_t)c->width * c->height;
  HIP_TRY(c, hipSetDevice(c->device));
  const void *pa = c->render_target.ptr, *pn = c->g_normal.ptr, *pd = c->g_depth.ptr;
  // the frame of the last compute() was traced as part of a batch; rt_resize and the next dispatch (the only calls that free
  // or reuse the batch's planes) reset gbuf_slot
  if (c->gbuf_slot >= 0 && (size_t)c->gbuf_slot < c->spec.slots.size()) {
    const DevFrameSlot& sl = c->spec.slots[c->gbuf_slot];
    pa = sl.albedo;
    pn = sl.normal_id;
    pd = sl.depth;
  }
  if (albedo) HIP_TRY(c, hipMemcpyAsync(albedo, pa, n * 4, hipMemcpyDeviceToHost, c->stream));
  if (normal_id) HIP_TRY(c, hipMemcpyAsync(normal_id, pn, n * 16, hipMemcpyDeviceToHost, c->stream));
  if (depth) HIP_TRY(c, hipMemcpyAsync(depth, pd, n * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return RT_OK;
}
int rt_read_history(rt_ctx* c, uint16_t* out, size_t cap) {
  if (!c || !out) return RT_ERR_INVALID;
  const size_t bytes = (size_t)c->width * c->height * 8;
  if (!c->history[0].ptr) return fail(c, RT_ERR_NOT_READY, "no history");
  if (cap < bytes) return fail(c, RT_ERR_INVALID, "buffer too small");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipMemcpyAsync(out, c->history[1 - c->history_index].ptr, bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return RT_OK;
}
int rt_read_uniforms(rt_ctx* c, void* out256) {
  if (!c || !out256) return RT_ERR_INVALID;
  std::memcpy(out256, &c->uniforms, 256);
  return RT_OK;
}
static int read_counters(rt_ctx* c, int bank_lo, int bank_hi, rt_counters* out) {
  HIP_TRY(c, hipSetDevice(c->device));
  std::vector<uint64_t> host((size_t)2 * RT_COUNTER_SHARDS * 6);
  HIP_TRY(c, hipMemcpyAsync(host.data(), c->counters.ptr, host.size() * 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  uint64_t sum[6] = {0, 0, 0, 0, 0, 0};
  for (int b = bank_lo; b <= bank_hi; b++)
    for (size_t s = 0; s < RT_COUNTER_SHARDS; s++)
      for (int k = 0; k < 6; k++) sum[k] += host[((size_t)b * RT_COUNTER_SHARDS + s) * 6 + k];
  out->primary_rays = sum[0];
  out->extension_rays = sum[1];
  out->shadow_rays = sum[2];
  out->nodes_visited = sum[3];
  out->tris_tested = sum[4];
  out->shaded_hits = sum[5];
  return RT_OK;
}
int rt_get_counters(rt_ctx* c, rt_counters* out) {
  if (!c || !out) return RT_ERR_INVALID;
  return read_counters(c, 0, 1, out);
}
int rt_get_kernel_counters(rt_ctx* c, int kernel, rt_counters* out) {
  if (!c || !out || kernel < 0 || kernel > 1) return RT_ERR_INVALID;
  return read_counters(c, kernel, kernel, out);
}
int rt_reset_counters(rt_ctx* c) {
  if (!c) return RT_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipMemsetAsync(c->counters.ptr, 0, c->counters.size, c->stream));
  return RT_OK;
}
int rt_set_counting(rt_ctx* c, int detailed) {
  if (!c) return RT_ERR_INVALID;
  c->epoch++;   // drops frames traced ahead (rt_set_lookahead)
  c->detailed_counters = detailed != 0;
  return RT_OK;
}
int rt_set_stripes(rt_ctx* c, uint32_t stripe_rows, uint32_t rank, uint32_t count) {
  if (!c) return RT_ERR_INVALID;
  if (count > 1 && (stripe_rows == 0 || rank >= count)) return fail(c, RT_ERR_INVALID, "invalid stripe spec");
  c->epoch++;
  c->stripe_rows = stripe_rows;
  c->stripe_rank = rank;
  c->stripe_count = count ? count : 1;
  return RT_OK;
}
void* rt_accum_device_ptr(rt_ctx* c) { return c ? (void*)accum_ptr(c) : nullptr; }
int rt_bind_accum(rt_ctx* c, void* device_ptr) {
  if (!c) return RT_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->external_accum = device_ptr;
  c->accum_stale = false;
  return RT_OK;
}
int rt_bind_present_source(rt_ctx* c, void* device_ptr) {
  if (!c) return RT_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->present_source = device_ptr;
  c->present_stale = false;
  return RT_OK;
}
int rt_set_stream(rt_ctx* c, void* hip_stream) {
  if (!c) return RT_ERR_INVALID;
  c->epoch++;   // drops frames traced ahead (rt_set_lookahead)
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
  return RT_OK;
}
int rt_set_kernel_variant(rt_ctx* c, int variant) {
  if (!c || variant < 0 || variant > 3) return RT_ERR_INVALID;
  c->variant = variant;
  c->epoch++;
  return RT_OK;
}
int rt_set_walk(rt_ctx* c, int walk) {
  if (!c || walk < 0 || walk > 2) return RT_ERR_INVALID;
  c->walk = walk;
  c->epoch++;
  return RT_OK;
}
int rt_set_kernel_timing(rt_ctx* c, int enabled) {
  if (!c) return RT_ERR_INVALID;
  c->epoch++;   // drops frames traced ahead (rt_set_lookahead)
  c->timing = enabled != 0;
  return RT_OK;
}
int rt_debug_read_traversal_nodes(rt_ctx* c, float* tnodes_out, uint32_t* new_index_out, uint32_t* inst_root_out, uint32_t cap_nodes) {
  if (!c) return RT_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  int r = prepare_scene(c);
  if (r < 0) return r;
  if (cap_nodes < c->n_nodes || !c->tnodes.ptr) return fail(c, RT_ERR_INVALID, "rt_debug_read_traversal_nodes: no nodes or buffer too small");
  if (tnodes_out) HIP_TRY(c, hipMemcpyAsync(tnodes_out, c->tnodes.ptr, (size_t)c->n_nodes * 32, hipMemcpyDeviceToHost, c->stream));
  if (new_index_out) HIP_TRY(c, hipMemcpyAsync(new_index_out, c->node_newidx.ptr, (size_t)c->n_nodes * 4, hipMemcpyDeviceToHost, c->stream));
  if (inst_root_out) HIP_TRY(c, hipMemcpyAsync(inst_root_out, c->inst_root.ptr, (size_t)c->n_instances * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return (int)c->n_nodes;
}
int rt_debug_read_pairs(rt_ctx* c, float* pairs_out, float* root_rec_out, uint32_t cap_pairs) {
  if (!c) return RT_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  c->pairs_wanted = true;
  int r = prepare_scene(c);
  c->pairs_wanted = false;
  if (r < 0) return r;
  if (cap_pairs < c->n_pairs || !c->pairs.ptr || !c->root_rec.ptr) return fail(c, RT_ERR_INVALID, "rt_debug_read_pairs: no records or buffer too small");
  if (pairs_out && c->n_pairs) HIP_TRY(c, hipMemcpyAsync(pairs_out, c->pairs.ptr, (size_t)c->n_pairs * 64, hipMemcpyDeviceToHost, c->stream));
  if (root_rec_out) HIP_TRY(c, hipMemcpyAsync(root_rec_out, c->root_rec.ptr, ((size_t)c->n_instances + 1) * 32, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return (int)c->n_pairs;
}
int rt_debug_trace_sections(rt_ctx* c, uint64_t* out16, int reset) {
  if (!c || !out16) return RT_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipMemcpyFromSymbol(out16, HIP_SYMBOL(rtk::g_trace_sections), 128, 0, hipMemcpyDeviceToHost));
  if (reset) {
    uint64_t zero[16] = {0};
    HIP_TRY(c, hipMemcpyToSymbol(HIP_SYMBOL(rtk::g_trace_sections), zero, 128, 0, hipMemcpyHostToDevice));
  }
#ifdef RT_TRACE_STAMPS
  return 1;
#else
  return 0;
#endif
}
int rt_debug_pt_sections(rt_ctx* c, uint64_t* out8, int reset) {
  if (!c || !out8) return RT_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipMemcpyFromSymbol(out8, HIP_SYMBOL(rtk::g_pt_sections), 64, 0, hipMemcpyDeviceToHost));
  if (reset) {
    uint64_t zero[8] = {0};
    HIP_TRY(c, hipMemcpyToSymbol(HIP_SYMBOL(rtk::g_pt_sections), zero, 64, 0, hipMemcpyHostToDevice));
  }
#ifdef RT_PT_STAMPS
  return 1;
#else
  return 0;
#endif
}
int rt_debug_lane_stats(rt_ctx* c, uint64_t* out32, int reset) {
  if (!c || !out32) return RT_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipMemcpyFromSymbol(out32, HIP_SYMBOL(rtk::g_lane_stats), 256, 0, hipMemcpyDeviceToHost));
  if (reset) {
    uint64_t zero[32] = {0};
    HIP_TRY(c, hipMemcpyToSymbol(HIP_SYMBOL(rtk::g_lane_stats), zero, 256, 0, hipMemcpyHostToDevice));
  }
#ifdef RT_LANE_STATS
  return 1;
#else
  return 0;
#endif
}
int rt_debug_clock_stamps(rt_ctx* c, uint64_t* out_pairs, uint32_t cap_pairs) {
  if (!c || !out_pairs) return RT_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  const uint32_t n = cap_pairs < RT_CLOCK_STAMP_SLOTS ? cap_pairs : RT_CLOCK_STAMP_SLOTS;
  HIP_TRY(c, hipMemcpyFromSymbol(out_pairs, HIP_SYMBOL(rtk::g_clock_stamps), (size_t)n * 16, 0, hipMemcpyDeviceToHost));
#ifdef RT_CLOCK_STAMP
  return (int)n;
#else
  return 0;  // product build: the kernels execute no stamp
#endif
}
int rt_debug_ieee_check(rt_ctx* c, int op, uint64_t first, uint64_t count, rt_ieee_report* out) {
  if (!c || !out) return RT_ERR_INVALID;
  static_assert(sizeof(rt_ieee_report) == sizeof(rtk::IeeeReport), "report layouts differ");
  std::memset(out, 0, sizeof(*out));
#ifdef RT_IEEE_PLAIN
  return fail(c, RT_ERR_INVALID, "rt_debug_ieee_check: built with RT_IEEE_PLAIN, the kernels use the plain operators");
#else
  if (op < 0 || op >= RT_IEEE_OP_COUNT) return fail(c, RT_ERR_INVALID, "rt_debug_ieee_check: unknown op");
  if (count == 0) return RT_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  rtk::IeeeReport* d = nullptr;
  HIP_TRY(c, hipMalloc((void**)&d, sizeof(rtk::IeeeReport)));
  hipError_t e = hipMemsetAsync(d, 0, sizeof(rtk::IeeeReport), c->stream);
  const uint64_t want = (count + 255) / 256;
  const dim3 grid((uint32_t)std::min<uint64_t>(want, (uint64_t)c->num_cus * 32u)), block(256);
  if (e == hipSuccess) {
    switch (op) {
      case RT_IEEE_OP_RCP: hipLaunchKernelGGL(rtk::k_ieee_check<RT_IEEE_OP_RCP>, grid, block, 0, c->stream, first, count, d); break;
      case RT_IEEE_OP_SQRT: hipLaunchKernelGGL(rtk::k_ieee_check<RT_IEEE_OP_SQRT>, grid, block, 0, c->stream, first, count, d); break;
      case RT_IEEE_OP_RSQRT: hipLaunchKernelGGL(rtk::k_ieee_check<RT_IEEE_OP_RSQRT>, grid, block, 0, c->stream, first, count, d); break;
      case RT_IEEE_OP_DIV: hipLaunchKernelGGL(rtk::k_ieee_check<RT_IEEE_OP_DIV>, grid, block, 0, c->stream, first, count, d); break;
      case RT_IEEE_OP_DIV3: hipLaunchKernelGGL(rtk::k_ieee_check<RT_IEEE_OP_DIV3>, grid, block, 0, c->stream, first, count, d); break;
      case RT_IEEE_OP_DIV3Z: hipLaunchKernelGGL(rtk::k_ieee_check<RT_IEEE_OP_DIV3Z>, grid, block, 0, c->stream, first, count, d); break;
      case RT_IEEE_OP_DIV_PI: hipLaunchKernelGGL(rtk::k_ieee_check<RT_IEEE_OP_DIV_PI>, grid, block, 0, c->stream, first, count, d); break;
      default: hipLaunchKernelGGL(rtk::k_ieee_check<RT_IEEE_OP_UNORM8>, grid, block, 0, c->stream, first, count, d); break;
    }
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(out, d, sizeof(*out), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(d);
  if (e != hipSuccess) return hip_fail(c, e, "rt_debug_ieee_check");
  out->n = count;
  return RT_OK;
#endif
}
int rt_kernel_times(rt_ctx* c, double* sum_ms, uint32_t* launches, uint32_t n) {
  if (!c || !sum_ms || !launches) return RT_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  for (uint32_t k = 0; k < n; k++) {
    sum_ms[k] = 0.0;
    launches[k] = 0;
  }
  for (auto& t : c->ev_tags) {
    float ms = 0;
    if ((uint32_t)t.second < n && hipEventElapsedTime(&ms, c->ev_pool[t.first].a, c->ev_pool[t.first].b) == hipSuccess) {
      sum_ms[t.second] += ms;
      launches[t.second]++;
    }
  }
  c->ev_tags.clear();
  c->ev_used = 0;
  return RT_OK;
}
int rt_kernel_time_ms(rt_ctx* c, double* avg_pt, double* avg_pv, uint32_t* launches) {
  double sum[RT_TIMER_COUNT];
  uint32_t cnt[RT_TIMER_COUNT];
  int r = rt_kernel_times(c, sum, cnt, RT_TIMER_COUNT);
  if (r < 0) return r;
  if (avg_pv) *avg_pv = cnt[RT_TIMER_PRIMARY] ? sum[RT_TIMER_PRIMARY] / cnt[RT_TIMER_PRIMARY] : 0.0;
  if (avg_pt) *avg_pt = cnt[RT_TIMER_PATHTRACE] ? sum[RT_TIMER_PATHTRACE] / cnt[RT_TIMER_PATHTRACE] : 0.0;
  if (launches) *launches = cnt[RT_TIMER_PATHTRACE];
  return RT_OK;
}

}  // extern "C"
