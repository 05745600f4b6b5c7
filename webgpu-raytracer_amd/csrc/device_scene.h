// device_scene.h — device-side view of the scene as the HIP kernels read it.
//
// The bridge arrays arrive in the reference's layouts (include/mi355rt_layout.h).  At
// upload time they are re-laid-out once on the GPU into gather-friendly records (all
// bit-exact: only f32 subtractions the reference shader would perform per test, or
// pure permutations):
//
//   nodes      2 x float4 / node   {min.xyz, skip} {max.xyz, data}           (unchanged, 32 B)
//   tnodes     2 x float4 / node   the same boxes and leaf words with BOTH successors explicit ({.., skip'} {.., inner |
//              first child'}), ordered most-visited first so that a prefix can live in LDS (k_treelet.hip.h)
//   inst_root  u32 / inst          index in tnodes of the instance's BLAS root
//   pairs      4 x float4 / INNER node   {L.min, wordL} {L.max, 0} {R.min, wordR} {R.max, skipX}: what the wave-level walk of
//              the trace kernels reads (k_pairs.hip.h, k_pairwalk.hip.h) — both children in one 64-byte-aligned record
//   root_rec   2 x float4 / inst (+1)    box and word of the instance's BLAS root; the last record is the TLAS root
//   tri_geom   3 x float4 / tri    {v0, _} {e1 = v1-v0, _} {e2 = v2-v0, _}   (48 B instead of the
//              80-B topology row + 3 dependent 16-B position gathers, Raytracer.wgsl:476-477)
//   tri_shade  8 x float4 / tri    rows 1..4 of the topology record (material attributes), then the three vertex normals and
//              uvs: {n0, uv0.x} {n1, uv0.y} {n2, uv1.x} {uv1.y, uv2.x, uv2.y, 0}.  One aligned 128-byte line per shaded hit
//              instead of the 80-B topology row + six dependent 16-/8-byte vertex gathers through its index row (ten cache
//              lines): the shade kernels of the large scenes are bound by exactly that line traffic
//   inst_trav  4 x float4 / inst   rows 0..2 of the inverse matrix (so M*p is 3 dot-like rows),
//              {blas_node_offset, inv[3], inv[7], inv[11]}                    (64 B instead of 144 B)
//   light_rec  4 x float4 / light  world-space light triangle, its unit normal and area (what sample_light_source
//              recomputes per NEE sample from topology + positions + instance matrix, Raytracer.wgsl:354-373)
//   topo/pos/nrm/uv/inst/lights    raw arrays, read once per shaded hit
#ifndef MI355RT_DEVICE_SCENE_H
#define MI355RT_DEVICE_SCENE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mi355rt_layout.h"
#include "k_ieee.hip.h"   // before the math header: the device build's correctly rounded rcp / div / sqrt sequences
#include "../../include/mi355rt_math.h"

// 16-byte slots per triangle record.  4 (one aligned 64-byte line per triangle; a 48-byte record straddles two lines
// 37 % of the time) was measured: no gain on any scene, 33 % more memory
#ifndef RT_TRI_STRIDE
#define RT_TRI_STRIDE 3
#endif

struct DevScene {
  const float4* nodes;      // 2 per node, TLAS ++ BLAS (bridge layout; the per-lane walks of the primary pass read it)
  const float4* tnodes;     // 2 per node: the same nodes with explicit successors, treelet first (k_treelet.hip.h)
  const uint32_t* inst_root;  // 1 per instance: index in tnodes of the instance's BLAS root
  const float4* pairs;      // 4 per INNER node: both children's boxes and words + the stackless skip (k_pairs.hip.h)
  const float4* root_rec;   // 2 per instance: its BLAS root's box and word; record n_instances = the TLAS root
  const float4* tri_geom;   // RT_TRI_STRIDE per triangle: {v0} {e1} {e2} (+ padding to one 64-byte line)
  const float4* tri_shade;  // 8 per triangle: what shading reads about a hit, in ONE 128-byte line
  const float4* inst_trav;  // 4 per instance
  const float4* topo;       // 5 per triangle (raw MeshTopology rows)
  const float4* pos;        // 1 per vertex
  const float4* nrm;        // 1 per vertex
  const float2* uv;         // 1 per vertex
  const float4* inst;       // 9 per instance (raw Instance)
  const uint2* lights;      // LightRef
  const float4* light_rec;  // 4 per light: {v0.xyz, area} {v1.xyz, n.x} {v2.xyz, n.y} {n.z, tri, 0, 0} in world space
  const uint8_t* tex;       // layers x 1024 x 1024 x 4, or nullptr => 1x1 white default
  uint32_t tex_layers;
  uint32_t n_lights;        // elements in `lights` (for the robust-access clamp)
};

struct DevFrame {
  float4* accum;        // W*H float4
  float4* frame_col;    // batched dispatch only: n_slots x W*H frame colours (else nullptr)
  uint32_t* albedo;     // W*H rgba8 (render target: G-buffer albedo, later post output)
  float4* normal_id;    // W*H rgba32f
  float* depth;         // W*H f32
  uint64_t* counters;   // 6 x u64
  uint32_t max_depth, spp;
  uint32_t stripe_rows, stripe_rank, stripe_count;
  // dense enumeration of the tile rows this rank owns (set when stripe_rows % 8 == 0, else own_period = 0):
  // k-th owned tile row = (k / own_run) * own_period + own_first + (k % own_run); own_tile_rows of them exist
  uint32_t own_run, own_period, own_first, own_tile_rows;
};

// One frame of a batched dispatch (rt_compute_batch): what changes from one compute() to the next.
struct DevFrameSlot {
  uint32_t frame_count;
  float jitter_x, jitter_y;
  uint32_t pad;
  uint32_t* albedo;    // this frame's G-buffer planes
  float4* normal_id;
  float* depth;
  uint64_t pad2;
};

// Wavefront form (large scenes): path state lives in HBM between the shade and trace stages.
struct WfPath {       // one 64-B record (one half cache line) per live path, in the order of the depth's active list.
  float4 c;           // The ray itself is NOT here: the extension ray a
  float4 d;           // path continues along is in the ray queue at the path's slot (k_wf_shade reads it back from there).
  float4 e;           // c: throughput.xyz, bitcast(rng)   d: radiance.xyz, bitcast(flags): depth | specular << 8 |
  uint4 m;            // ended << 9 | nee_valid << 10   e: pending NEE term .xyz, prev_bsdf_pdf
};                    // m: queue slot of the pending shadow ray, of the extension ray, 0, 0
struct WfState {
  WfPath* p[2];       // [depth & 1]: the record of the path at position i of that depth's active list — the shade pass of depth
};                    // d reads p[d & 1] in list order and writes p[(d + 1) & 1] at the slots it appends to the next list, so the
                      // path state streams (round 2 kept one record per PATH ID: 64-byte gathers and scatters into 4 GB)
struct WfQueues {
  uint32_t* active[2];    // path ids alive at the current / next depth
  uint32_t* shadow_ids;   // shadow-ray queue: path id ...
  float4* shadow_rays;    // ... and {o.xyz, t_max} {d.xyz, 0}
  uint32_t* occluded;     // ... result of k_wf_trace<any hit>, by slot: 1 = something is in the way
  uint32_t* ext_ids;      // extension-ray queue: path id ...
  float4* ext_rays[2];    // ... {o.xyz, 0} {d.xyz, 0}; [depth & 1]: the shade pass of depth d + 1 reads the rays of depth d
                          // (they are the path's ray: WfPath does not repeat it) while it queues its own
  float4* ext_hit;        // ... result of k_wf_trace<closest hit>, by slot: {t, bits(triangle), bits(instance), 0}; instance < 0 = miss
  uint32_t* counters;     // 8 u32 per depth: n_active, n_shadow, n_ext, head_shadow, head_ext, 0, 0, 0
};
#define WF_FLAG_SPECULAR 0x100u
#define WF_FLAG_ENDED 0x200u
#define WF_FLAG_NEE_VALID 0x400u

struct DevPost {
  const float4* accum;
  const ushort4* history_in;  // rgba16f, previous frame
  ushort4* history_out;       // rgba16f, current frame
  uint32_t* out_rgba8;        // render target
};

#endif
