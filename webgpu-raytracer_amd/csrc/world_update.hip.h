// world_update.hip.h — the per-frame half of World::update(t) on the GPU (the rest of it).
//
// The reference's World::update (rust-shader-tools/src/lib.rs:149-270) runs, per displayed frame of an animated scene,
//   animation + scene graph          lib.rs:149-184        host, a few hundred bytes of output: stays on the host
//   linear-blend skinning            rebuilder.rs:36-91    k_skin
//   BLAS build per geometry          bvh/blas.rs           csrc/bvh_build.hip.h (same tree)
//   topology rows in BLAS order      rebuilder.rs:121-161  k_topology
//   emissive triangles               rebuilder.rs:163-168  k_topology flags + ordered compaction (k_emissive_*)
//   instance boxes, TLAS             lib.rs:194-235, bvh/tlas.rs:58-111   k_tlas (median split, stable sort by centre)
//   instances / lights / draw cmds   lib.rs:237-270        k_tlas tail + k_lights
// and the TypeScript side re-uploads every array (src/main.ts:133-163).  Here the arrays are written where the renderer
// reads them (rt_ctx's scene buffers) and never leave HBM; the arithmetic is the scene compiler's
// (csrc/scene/scene_compiler.cpp world_update), operation for operation, so every array is byte-identical to the host
// path (tests/test_gpu_world_update.py).  Host side: rt_world_update in rt_api.hip.
#ifndef MI355RT_WORLD_UPDATE_HIP_H
#define MI355RT_WORLD_UPDATE_HIP_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bvh_build.hip.h"

namespace wu {

using bvhb::float_of;
using bvhb::key_of;

// one geometry of the static scene description (device pointers into the static buffer)
struct Geom {
  const float* pos3;
  const float* nrm3;
  const float* uv2;
  const uint32_t* joints;
  const float* weights;
  const uint32_t* idx;
  const float* attr;
  uint32_t n_verts, n_uvs, n_tris, v_offset;
  uint32_t topo_start, skinned, joint_first, n_joints;   // joint matrices of its skin: [joint_first, joint_first + n_joints)
  uint32_t em_count, em_first, pad0, pad1;                // emissive triangles (static count), their slot in the emissive list
};
// what k_tlas needs to know of a geometry
struct GeomRow {
  uint32_t n_tris, topo_start, em_count, em_first;
};

__device__ __forceinline__ bool is_nan(float x) { return x != x; }

// rebuilder.rs:36-91 (scene_compiler.cpp world_update, the skinning loop): one vertex per lane
__global__ __launch_bounds__(256) void k_skin(Geom G, const float* __restrict__ joint_mats, float4* __restrict__ pos,
                                               float4* __restrict__ nrm, float2* __restrict__ uv) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= G.n_verts) return;
  float p[3] = {G.pos3[3 * (size_t)i], G.pos3[3 * (size_t)i + 1], G.pos3[3 * (size_t)i + 2]};
  float n[3] = {G.nrm3[3 * (size_t)i], G.nrm3[3 * (size_t)i + 1], G.nrm3[3 * (size_t)i + 2]};
  float2 t = make_float2(0.0f, 0.0f);
  if (i < G.n_uvs) t = make_float2(G.uv2[2 * (size_t)i], G.uv2[2 * (size_t)i + 1]);
  if (G.skinned) {
    float m[16];   // column-major, m[c * 4 + r]
    for (int k = 0; k < 16; k++) m[k] = 0.0f;
    for (int k = 0; k < 4; k++) {
      const float wk = G.weights[4 * (size_t)i + k];
      const uint32_t jk = G.joints[4 * (size_t)i + k];
      if (wk > 0.0f && jk < G.n_joints) {
        const float* jm = joint_mats + 16 * (size_t)(G.joint_first + jk);
        for (int e = 0; e < 16; e++) m[e] = m[e] + jm[e] * wk;
      }
    }
    bool any = false;
    for (int e = 0; e < 16; e++) any = any || m[e] != 0.0f;
    if (!any) {   // "if mat == Mat4::ZERO"
      for (int e = 0; e < 16; e++) m[e] = 0.0f;
      m[0] = m[5] = m[10] = m[15] = 1.0f;
    }
    float q[3], v[3];
    for (int r = 0; r < 3; r++) {   // glam Mat4::transform_point3 / transform_vector3
      float acc = m[r] * p[0];
      acc = m[4 + r] * p[1] + acc;
      acc = m[8 + r] * p[2] + acc;
      acc = m[12 + r] + acc;
      q[r] = acc;
      float bcc = m[r] * n[0];
      bcc = m[4 + r] * n[1] + bcc;
      bcc = m[8 + r] * n[2] + bcc;
      v[r] = bcc;
    }
    for (int r = 0; r < 3; r++) p[r] = q[r];
    const float len = __builtin_sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    const float rcp = 1.0f / len;
    const bool ok = !is_nan(rcp) && rcp < __uint_as_float(0x7f800000u) && rcp > 0.0f;   // is_finite && > 0
    for (int r = 0; r < 3; r++) n[r] = ok ? v[r] * rcp : 0.0f;
  }
  if (is_nan(p[0]) || is_nan(p[1]) || is_nan(p[2])) p[0] = p[1] = p[2] = 0.0f;
  if (is_nan(n[0]) || is_nan(n[1]) || is_nan(n[2])) {
    n[0] = n[1] = 0.0f;
    n[2] = 1.0f;
  }
  const size_t o = (size_t)G.v_offset + i;
  pos[o] = make_float4(p[0], p[1], p[2], 1.0f);
  nrm[o] = make_float4(n[0], n[1], n[2], 0.0f);
  uv[o] = t;
}

// rebuilder.rs:140-168: topology row i of the geometry = triangle order[i]; flag = emissive (material 3)
__global__ __launch_bounds__(256) void k_topology(Geom G, uint32_t gi, const uint32_t* __restrict__ order, float4* __restrict__ topo,
                                                   uint32_t* __restrict__ em_flag) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= G.n_tris) return;
  const uint32_t old_id = order[i];
  const float4* a = reinterpret_cast<const float4*>(G.attr + 16 * (size_t)old_id);
  float4* row = topo + 5 * ((size_t)G.topo_start + i);
  row[0] = make_float4(__uint_as_float(G.idx[3 * (size_t)old_id] + G.v_offset), __uint_as_float(G.idx[3 * (size_t)old_id + 1] + G.v_offset),
                       __uint_as_float(G.idx[3 * (size_t)old_id + 2] + G.v_offset), __uint_as_float(gi));
  const float4 a0 = a[0];
  row[1] = a0;
  row[2] = a[1];
  row[3] = a[2];
  row[4] = a[3];
  if (em_flag) em_flag[i] = __builtin_fabsf(a0.w - 3.0f) < 1e-6f ? 1u : 0u;
}
// emissive triangles in topology order -> em_list[G.em_first ...] (ranks from bvhb::k_scan_blocks / k_scan_top)
__global__ __launch_bounds__(1024) void k_emissive_apply(Geom G, const uint32_t* __restrict__ flag, const uint32_t* __restrict__ blk,
                                                          uint32_t* __restrict__ em_list) {
  __shared__ uint32_t s_wave[16];
  const uint32_t i = blockIdx.x * 1024u + threadIdx.x;
  const bool f = i < G.n_tris && flag[i] != 0u;
  const unsigned long long m = __ballot(f);
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
  if (lane == 0u) s_wave[wave] = (uint32_t)__builtin_popcountll(m);
  __syncthreads();
  uint32_t before = 0u;
  for (uint32_t w = 0; w < wave; w++) before += s_wave[w];
  const uint32_t k = blk[blockIdx.x] + before + rank;
  if (f && k < G.em_count) em_list[G.em_first + k] = G.topo_start + i;
}

// ------------------------------------------------------------------------------------------------------- TLAS
// bvh/tlas.rs:58-111 (scene_compiler.cpp TlasBuilder): recursive median split — box of the range, axis by the
// reference's rule (y if ext.y > ext.x, else z if it exceeds both, else x), STABLE sort of the range by box centre on
// that axis, halves [0, count/2) and the rest, the half with the larger area x count first.  All ranges of a recursion
// depth are worked on together, one position per lane (strided).  The tree is full and every leaf holds one instance, so a
// node's pre-order index follows from the counts alone: left child = node + 1, right child = node + 2 * (left count),
// skip = node + 2 * count - 1.
// Round 4: the stable sort of ALL ranges of a depth is ONE bitonic sort, in LDS, of 64-bit keys
//   (first position of the range : 14 bits | order-preserving key of the centre : 32 | current position : 14):
// the range is the leading field, so ranges stay where they are; the position is the last, so equal centres keep their
// order — the reference's stable sort — and every key is distinct, which makes the (unstable) network's result unique.
// 16 384 keys are 128 KB of the CU's 160 KB: the device path takes up to 16 384 instances, and a depth of that size
// costs 105 compare-exchange stages of 8 pairs per lane instead of the N^2 / 1024 global-memory comparisons per lane of
// round 3's rank count (1 001 instances: 1 M dependent loads on the first depth alone).  Range boxes are reduced in the
// wave (a wave of 64 consecutive positions inside one range: shuffles, then six atomics) instead of six same-address
// atomics per instance.  One workgroup: the depths are a chain of ~log2 N dependent rounds of a few microseconds each, and
// a grid-wide barrier per step would cost more than the step.
struct TlasArgs {
  const float4* raw;          // instances in declaration order, 9 float4 each (rt_instance)
  const GeomRow* geoms;
  const uint32_t* node_base;  // per geometry: first node of its BLAS (BLAS-local units)
  float4* nodes;              // the world's node array: TLAS written at 0, BLAS roots read at n_tlas + node_base[g]
  float4* inst_out;           // packed instances, TLAS order
  uint4* draw_out;            // draw commands, TLAS order
  uint32_t* light_off;        // per TLAS position: first light of the instance
  // scratch, n_inst entries each
  float* box;                 // 6 per instance (declaration order): world box
  float* ctr;                 // 3 per instance: its centre
  uint32_t* ord;              // position -> declaration index
  uint32_t* ord2;
  uint32_t* seg;              // 3 per position: first, count (0 = finished leaf), node
  uint32_t* skey;             // 18 per position (used at range heads): box / left-half box / right-half box as keys
  uint32_t* sinfo;            // 2 per position (range heads): axis, rotate
  uint32_t* status;           // [0] != 0: a NaN centre was seen (the host path must do this update); [1] lights written
  uint32_t n_inst, n_tlas, n_lights, pad;
};

__device__ __forceinline__ float min_nn(float a, float b) { return is_nan(b) ? a : (is_nan(a) ? b : (key_of(b) < key_of(a) ? b : a)); }
__device__ __forceinline__ float max_nn(float a, float b) { return is_nan(b) ? a : (is_nan(a) ? b : (key_of(b) > key_of(a) ? b : a)); }

#define RT_TLAS_MAX_INSTANCES 16384u   // 14-bit fields of the sort key; 128 KB of LDS

// Boxes of runs of consecutive positions (ranges, or halves of ranges) without global atomics: a wave holds 64 consecutive
// positions; a segmented scan by shuffles leaves every run's union in its last lane.  A run that lies whole inside the
// wave's 64 positions is finished there; the others are combined in LDS, in the slot of the 64-position chunk the run
// STARTS in (only one run can start in a chunk and leave it: the slot is that run's alone).
struct KeyBox {
  uint32_t k[6];   // min keys, max keys (order-preserving keys of floats: key_of)
};
__device__ __forceinline__ void kb_identity(KeyBox& b) {
  b.k[0] = b.k[1] = b.k[2] = 0xffffffffu;
  b.k[3] = b.k[4] = b.k[5] = 0u;
}
__device__ __forceinline__ void kb_of_box(KeyBox& b, const float* box) {   // NaN bounds are ignored, like fmin_nn / fmax_nn
  for (int c = 0; c < 3; c++) {
    b.k[c] = is_nan(box[c]) ? 0xffffffffu : key_of(box[c]);
    b.k[c + 3] = is_nan(box[c + 3]) ? 0u : key_of(box[c + 3]);
  }
}
__device__ __forceinline__ float kb_area(const KeyBox& b) {   // primitives.rs AABB::area
  const float dx = float_of(b.k[3]) - float_of(b.k[0]), dy = float_of(b.k[4]) - float_of(b.k[1]), dz = float_of(b.k[5]) - float_of(b.k[2]);
  if (dx < 0.0f || dy < 0.0f || dz < 0.0f) return 0.0f;
  return 2.0f * (dx * dy + dy * dz + dz * dx);
}
// inclusive segmented scan over the wave: afterwards a lane holds the union of its run from the run's first lane IN THIS
// WAVE up to itself (`id` is equal exactly within a run; runs are contiguous)
__device__ __forceinline__ void kb_run_scan(KeyBox& b, uint32_t id) {
  const uint32_t lane = threadIdx.x & 63u;
  for (uint32_t off = 1u; off < 64u; off <<= 1) {
    const uint32_t id_up = __shfl_up(id, off, 64);
    KeyBox u;
    for (int c = 0; c < 6; c++) u.k[c] = __shfl_up(b.k[c], off, 64);
    if (lane >= off && id_up == id)
      for (int c = 0; c < 3; c++) {
        b.k[c] = u.k[c] < b.k[c] ? u.k[c] : b.k[c];
        b.k[c + 3] = u.k[c + 3] > b.k[c + 3] ? u.k[c + 3] : b.k[c + 3];
      }
  }
}
__device__ __forceinline__ void kb_lds_add(uint32_t* slot, const KeyBox& b) {
  for (int c = 0; c < 3; c++) {
    atomicMin(&slot[c], b.k[c]);
    atomicMax(&slot[c + 3], b.k[c + 3]);
  }
}
__device__ __forceinline__ void kb_lds_get(KeyBox& b, const uint32_t* slot) {
  for (int c = 0; c < 6; c++) b.k[c] = slot[c];
}
// the node of a range whose box is known: leaf (count 1) or inner node + the axis its positions will be sorted on
__device__ __forceinline__ void tlas_emit_node(const TlasArgs& A, uint32_t first, uint32_t count, uint32_t node, const KeyBox& b,
                                               uint32_t* axis_slot, uint32_t* any_inner) {
  const float mn[3] = {float_of(b.k[0]), float_of(b.k[1]), float_of(b.k[2])};
  const float mx[3] = {float_of(b.k[3]), float_of(b.k[4]), float_of(b.k[5])};
  const uint32_t skip = count == 1u ? node + 1u : node + 2u * count - 1u;
  const uint32_t data = count == 1u ? ((first << 3) | 1u) : 0u;
  A.nodes[2 * (size_t)node] = make_float4(mn[0], mn[1], mn[2], __uint_as_float(skip));
  A.nodes[2 * (size_t)node + 1] = make_float4(mx[0], mx[1], mx[2], __uint_as_float(data));
  if (count > 1u) {
    const float ex = mx[0] - mn[0], ey = mx[1] - mn[1], ez = mx[2] - mn[2];
    *axis_slot = ey > ex ? 1u : ((ez > ex && ez > ey) ? 2u : 0u);   // tlas.rs:76
    *any_inner = 1u;
  }
}

// lib.rs:237-270: instances, draw commands and the lights' offsets, in TLAS order (A.ord[p] = the instance at position p)
__device__ __forceinline__ void tlas_pack_tail(const TlasArgs& A, uint32_t* s_scan, uint32_t* s_carry_p) {
  const uint32_t tid = threadIdx.x, N = A.n_inst;
  uint32_t& s_carry = *s_carry_p;
  if (tid == 0u) s_carry = 0u;
  __syncthreads();
  for (uint32_t p0 = 0; p0 < N; p0 += 1024u) {
    const uint32_t p = p0 + tid;
    uint32_t e = 0;
    if (p < N) {
      const float4* I = A.raw + 9 * (size_t)A.ord[p];
      const float4 tail = I[8];
      const uint32_t g = __float_as_uint(tail.z);
      const GeomRow G = A.geoms[g];
      float4* O = A.inst_out + 9 * (size_t)p;
      for (int k = 0; k < 8; k++) O[k] = I[k];
      O[8] = make_float4(__uint_as_float(A.node_base[g]), tail.y, tail.z, tail.w);
      A.draw_out[p] = make_uint4(G.n_tris * 3u, 1u, G.topo_start * 3u, p);
      e = G.em_count;
    }
    s_scan[tid] = e;
    __syncthreads();
    for (uint32_t off = 1u; off < 1024u; off <<= 1) {
      const uint32_t w = tid >= off ? s_scan[tid - off] : 0u;
      __syncthreads();
      s_scan[tid] += w;
      __syncthreads();
    }
    if (p < N) A.light_off[p] = s_carry + s_scan[tid] - e;
    __syncthreads();
    if (tid == 1023u) s_carry += s_scan[1023];
    __syncthreads();
  }
  if (tid == 0u) A.status[1] = s_carry;
}

// world box of instance i (declaration order) and its centre: the BLAS root box through the instance transform
// (primitives.rs AABB::transform: 8 corners); a NaN centre is reported in status[0] (the host path must do this update)
__device__ __forceinline__ void tlas_instance_box(const TlasArgs& A, uint32_t i, float* box, float* ctr) {
  const float inf = __uint_as_float(0x7f800000u);
  const float4* I = A.raw + 9 * (size_t)i;
  const uint32_t g = __float_as_uint(I[8].z);
  const size_t root = (size_t)A.n_tlas + A.node_base[g];
  const float4 lo = A.nodes[2 * root], hi = A.nodes[2 * root + 1];
  const float4 c0 = I[0], c1 = I[1], c2 = I[2], c3 = I[3];
  float mn[3] = {inf, inf, inf}, mx[3] = {-inf, -inf, -inf};
  for (int k = 0; k < 8; k++) {
    const float px = (k & 1) ? hi.x : lo.x, py = (k & 2) ? hi.y : lo.y, pz = (k & 4) ? hi.z : lo.z;
    const float m0[3] = {c0.x, c0.y, c0.z}, m1[3] = {c1.x, c1.y, c1.z}, m2[3] = {c2.x, c2.y, c2.z}, m3[3] = {c3.x, c3.y, c3.z};
    for (int r = 0; r < 3; r++) {
      float acc = m0[r] * px;
      acc = m1[r] * py + acc;
      acc = m2[r] * pz + acc;
      acc = m3[r] + acc;
      mn[r] = min_nn(mn[r], acc);
      mx[r] = max_nn(mx[r], acc);
    }
  }
  bool nan_centre = false;
  for (int r = 0; r < 3; r++) {
    box[r] = mn[r];
    box[3 + r] = mx[r];
    const float ce = (mn[r] + mx[r]) * 0.5f;
    ctr[r] = ce;
    nan_centre = nan_centre || is_nan(ce);
  }
  if (nan_centre) atomicOr(&A.status[0], 1u);
}

// dynamic LDS: next_pow2(max(n_inst, 1024)) 64-bit sort keys
__global__ __launch_bounds__(1024) void k_tlas(TlasArgs A) {
  extern __shared__ unsigned long long s_sort[];
  __shared__ uint32_t s_acc[256 * 6];   // one key box per 64-position chunk: the run that starts in the chunk and leaves it
  __shared__ uint32_t s_any;
  __shared__ uint32_t s_scan[1024];
  __shared__ uint32_t s_carry;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, N = A.n_inst;
  uint32_t M = 1024u;   // keys sorted per depth: a power of two >= N
  while (M < N) M <<= 1;
  const uint32_t NP = (N + 1023u) & ~1023u;   // the strided loops run whole waves (shuffles need every lane)
  // ---- instance boxes
  for (uint32_t i = tid; i < N; i += 1024u) {
    tlas_instance_box(A, i, A.box + 6 * (size_t)i, A.ctr + 3 * (size_t)i);
    A.ord[i] = i;
    A.seg[3 * (size_t)i] = 0u;
    A.seg[3 * (size_t)i + 1] = N;
    A.seg[3 * (size_t)i + 2] = 0u;
  }
  __syncthreads();
  for (uint32_t round = 0; round < 40u; round++) {   // depth <= ceil(log2 N) + 1
    // ---- 1. box of every live range -> its node; leaf, or the axis of its sort
    for (uint32_t i = tid; i < 256u * 6u; i += 1024u) s_acc[i] = (i % 6u) < 3u ? 0xffffffffu : 0u;
    if (tid == 0u) s_any = 0u;
    __syncthreads();
    for (uint32_t p = tid; p < NP; p += 1024u) {
      const uint32_t chunk0 = p - lane;
      const bool in = p < N;
      const uint32_t first = in ? A.seg[3 * (size_t)p] : 0u, count = in ? A.seg[3 * (size_t)p + 1] : 0u, node = in ? A.seg[3 * (size_t)p + 2] : 0u;
      const bool live = count != 0u;
      KeyBox b;
      kb_identity(b);
      if (live) kb_of_box(b, A.box + 6 * (size_t)A.ord[p]);
      const uint32_t id = live ? first : 0xffffffffu - lane;   // positions of finished ranges: runs of their own
      kb_run_scan(b, id);
      const uint32_t id_next = __shfl_down(id, 1u, 64);
      const bool run_end = live && (lane == 63u || id_next != id);
      if (run_end) {
        if (first >= chunk0 && first + count <= chunk0 + 64u)
          tlas_emit_node(A, first, count, node, b, &A.sinfo[2 * (size_t)first], &s_any);      // the whole range is in this wave
        else
          kb_lds_add(&s_acc[6u * (first >> 6)], b);
      }
    }
    __syncthreads();
    for (uint32_t p = tid; p < NP; p += 1024u) {   // the range that starts in a chunk and leaves it: finished by the chunk's last position
      const uint32_t chunk0 = p - lane;
      const bool last = p < N && (lane == 63u || p + 1u == N);
      if (!last) continue;
      const uint32_t first = A.seg[3 * (size_t)p], count = A.seg[3 * (size_t)p + 1], node = A.seg[3 * (size_t)p + 2];
      if (count != 0u && first >= chunk0 && first + count > chunk0 + 64u) {
        KeyBox b;
        kb_lds_get(b, &s_acc[6u * (first >> 6)]);
        tlas_emit_node(A, first, count, node, b, &A.sinfo[2 * (size_t)first], &s_any);
      }
    }
    __syncthreads();
    if (!s_any) break;
    // ---- 2. stable sort of every range by centre on its axis (tlas.rs:78-83): one bitonic sort of all positions.
    // A position of a finished range (count <= 1) is a range of its own: it stays where it is.  The centre goes in as
    // c + 0.0f: partial_cmp calls -0 and +0 equal (their order is then the positions'), the integer key would not.
    for (uint32_t p = tid; p < M; p += 1024u) {
      unsigned long long key = ~0ull;   // padding sorts to the end
      if (p < N) {
        const uint32_t first = A.seg[3 * (size_t)p], count = A.seg[3 * (size_t)p + 1];
        if (count > 1u) {
          const uint32_t axis = A.sinfo[2 * (size_t)first];
          const float ce = A.ctr[3 * (size_t)A.ord[p] + axis] + 0.0f;
          key = ((unsigned long long)first << 46) | ((unsigned long long)key_of(ce) << 14) | p;
        } else {
          key = ((unsigned long long)p << 46) | p;
        }
      }
      s_sort[p] = key;
    }
    for (uint32_t i = tid; i < 256u * 6u; i += 1024u) s_acc[i] = (i % 6u) < 3u ? 0xffffffffu : 0u;
    __syncthreads();
    for (uint32_t k = 2u; k <= M; k <<= 1)
      for (uint32_t j = k >> 1; j > 0u; j >>= 1) {
        for (uint32_t i = tid; i < (M >> 1); i += 1024u) {
          const uint32_t l = 2u * i - (i & (j - 1u)), r = l + j;
          const unsigned long long a = s_sort[l], b = s_sort[r];
          if ((a > b) == ((l & k) == 0u)) {
            s_sort[l] = b;
            s_sort[r] = a;
          }
        }
        __syncthreads();
      }
    // ---- 3. the new order; boxes of the two halves of every range -> their areas (skey[18 s] of the half's first position s)
    for (uint32_t p = tid; p < NP; p += 1024u) {
      const uint32_t chunk0 = p - lane;
      const bool in = p < N;
      const uint32_t first = in ? A.seg[3 * (size_t)p] : 0u, count = in ? A.seg[3 * (size_t)p + 1] : 0u;
      const bool live = count > 1u;
      uint32_t e = 0u;
      if (in) {
        e = A.ord[(uint32_t)s_sort[p] & 0x3fffu];
        A.ord2[p] = e;
      }
      const uint32_t mid = count / 2u;
      const uint32_t half = live && (p - first) >= mid ? 1u : 0u;
      const uint32_t hs = half ? first + mid : first, hn = half ? count - mid : mid;   // the half: first position, length
      KeyBox b;
      kb_identity(b);
      if (live) kb_of_box(b, A.box + 6 * (size_t)e);
      const uint32_t id = live ? hs : 0xffffffffu - lane;
      kb_run_scan(b, id);
      const uint32_t id_next = __shfl_down(id, 1u, 64);
      const bool run_end = live && (lane == 63u || id_next != id);
      if (run_end) {
        if (hs >= chunk0 && hs + hn <= chunk0 + 64u)
          A.skey[18 * (size_t)hs] = __float_as_uint(kb_area(b));
        else
          kb_lds_add(&s_acc[6u * (hs >> 6)], b);
      }
    }
    __syncthreads();
    for (uint32_t p = tid; p < NP; p += 1024u) {
      const uint32_t chunk0 = p - lane;
      const bool last = p < N && (lane == 63u || p + 1u == N);
      if (!last) continue;
      const uint32_t first = A.seg[3 * (size_t)p], count = A.seg[3 * (size_t)p + 1];
      if (count <= 1u) continue;
      const uint32_t mid = count / 2u;
      const uint32_t half = (p - first) >= mid ? 1u : 0u;
      const uint32_t hs = half ? first + mid : first, hn = half ? count - mid : mid;
      if (hs >= chunk0 && hs + hn > chunk0 + 64u) {
        KeyBox b;
        kb_lds_get(b, &s_acc[6u * (hs >> 6)]);
        A.skey[18 * (size_t)hs] = __float_as_uint(kb_area(b));
      }
    }
    __syncthreads();
    // ---- 4. the costlier half goes first (tlas.rs:95-104): rotate, then every position joins its child range
    for (uint32_t p = tid; p < N; p += 1024u) {
      const uint32_t first = A.seg[3 * (size_t)p], count = A.seg[3 * (size_t)p + 1], node = A.seg[3 * (size_t)p + 2];
      if (count == 1u) {
        A.seg[3 * (size_t)p + 1] = 0u;   // its leaf node was written in step 1
        continue;
      }
      if (count == 0u) continue;
      const uint32_t l_count = count / 2u, r_count = count - l_count, rel = p - first;
      const float la = __uint_as_float(A.skey[18 * (size_t)first]), ra = __uint_as_float(A.skey[18 * (size_t)(first + l_count)]);
      const bool rot = ra * (float)r_count > la * (float)l_count;
      const uint32_t nrel = rot ? (rel >= l_count ? rel - l_count : rel + r_count) : rel;   // std::rotate(first, first + l, end)
      A.ord[first + nrel] = A.ord2[p];
      const uint32_t lp = rot ? r_count : l_count;   // size of the first child after the rotation
      if (rel < lp) {
        A.seg[3 * (size_t)p] = first;
        A.seg[3 * (size_t)p + 1] = lp;
        A.seg[3 * (size_t)p + 2] = node + 1u;
      } else {
        A.seg[3 * (size_t)p] = first + lp;
        A.seg[3 * (size_t)p + 1] = count - lp;
        A.seg[3 * (size_t)p + 2] = node + 2u * lp;
      }
    }
    __syncthreads();
  }
  __syncthreads();
  tlas_pack_tail(A, s_scan, &s_carry);
}

// ---- up to 1 024 instances (config 3 has 1 001): ONE position per lane, everything that is read more than once in LDS
// or registers — boxes and centres by instance, the order, the axis and half-area words — and the sort network's 45
// stages that stay inside a wave (partner = lane ^ j, j < 64) run on shuffles without LDS or barriers; only the 10
// stages that cross waves go through LDS.  A depth costs ~3 us instead of the ~20 us of dependent global-memory round trips
// of k_tlas; same steps, same keys, same results.
__global__ __launch_bounds__(1024) void k_tlas_small(TlasArgs A) {
  __shared__ float s_box[1024 * 6];
  __shared__ float s_ctr[1024 * 3];
  __shared__ unsigned long long s_x[1024];
  __shared__ unsigned long long s_x2[1024];   // second exchange buffer of the sort; its words are s_area (steps 3-4) and s_scan (tail) otherwise
  __shared__ uint32_t s_axis[1024], s_ord[1024];
  uint32_t* const s_area = reinterpret_cast<uint32_t*>(s_x2);
  uint32_t* const s_scan = reinterpret_cast<uint32_t*>(s_x2) + 1024;
  __shared__ uint32_t s_acc[16 * 6];
  __shared__ uint32_t s_any, s_carry;
  const uint32_t p = threadIdx.x, lane = p & 63u, chunk0 = p - lane, N = A.n_inst;
  const bool in = p < N;
  if (in) tlas_instance_box(A, p, &s_box[6u * p], &s_ctr[3u * p]);
  s_ord[p] = p;
  uint32_t e = p;                                             // the instance at this position
  uint32_t first = 0u, count = in ? N : 0u, node = 0u;        // the range this position belongs to (count 0: finished)
  __syncthreads();
  for (uint32_t round = 0; round < 40u; round++) {
    if (p < 96u) s_acc[p] = (p % 6u) < 3u ? 0xffffffffu : 0u;
    if (p == 0u) s_any = 0u;
    __syncthreads();
    // ---- 1. box of every live range -> its node; leaf, or the axis of its sort
    {
      const bool live = count != 0u;
      KeyBox b;
      kb_identity(b);
      if (live) kb_of_box(b, &s_box[6u * e]);
      const uint32_t id = live ? first : 0xffffffffu - lane;
      kb_run_scan(b, id);
      const uint32_t id_next = __shfl_down(id, 1u, 64);
      if (live && (lane == 63u || id_next != id)) {
        if (first >= chunk0 && first + count <= chunk0 + 64u)
          tlas_emit_node(A, first, count, node, b, &s_axis[first], &s_any);
        else
          kb_lds_add(&s_acc[6u * (first >> 6)], b);
      }
    }
    __syncthreads();
    if (in && (lane == 63u || p + 1u == N) && count != 0u && first >= chunk0 && first + count > chunk0 + 64u) {
      KeyBox b;
      kb_lds_get(b, &s_acc[6u * (first >> 6)]);
      tlas_emit_node(A, first, count, node, b, &s_axis[first], &s_any);
    }
    __syncthreads();
    if (!s_any) break;
    // ---- 2. the stable sort of every range (tlas.rs:78-83): bitonic network over the 1 024 positions, key in a register
    unsigned long long key = ~0ull;
    if (in) {
      if (count > 1u) {
        const float ce = s_ctr[3u * e + s_axis[first]] + 0.0f;   // -0 and +0 compare equal in the reference
        key = ((unsigned long long)first << 46) | ((unsigned long long)key_of(ce) << 14) | p;
      } else {
        key = ((unsigned long long)p << 46) | p;
      }
    }
    if (p < 96u) s_acc[p] = (p % 6u) < 3u ? 0xffffffffu : 0u;
    uint32_t flip = 0u;   // the cross-wave stages alternate between two buffers: one barrier each (a barrier of 16 waves costs ~0.4 us)
    // (both loops unrolled: with j a constant the in-wave exchanges become DPP / swizzle moves instead of ds_bpermute)
#pragma unroll
    for (uint32_t k = 2u; k <= 1024u; k <<= 1)
#pragma unroll
      for (uint32_t j = k >> 1; j > 0u; j >>= 1) {
        unsigned long long other;
        if (j < 64u) {
          const uint32_t lo = __shfl_xor((uint32_t)key, (int)j, 64), hi = __shfl_xor((uint32_t)(key >> 32), (int)j, 64);
          other = ((unsigned long long)hi << 32) | lo;
        } else {
          unsigned long long* const buf = flip ? s_x2 : s_x;
          flip ^= 1u;
          buf[p] = key;
          __syncthreads();
          other = buf[p ^ j];
        }
        const bool keep_min = ((p & j) == 0u) == ((p & k) == 0u);   // the lower position of the pair keeps the minimum in an ascending block
        key = keep_min ? (key < other ? key : other) : (key > other ? key : other);
      }
    __syncthreads();   // every wave has read the last exchange buffer: its words are s_area from here on
    const uint32_t e2 = in ? s_ord[(uint32_t)key & 0x3fffu] : 0u;   // s_ord is not written during the sort
    // ---- 3. boxes of the two halves of every range -> their areas (s_area[first position of the half])
    const uint32_t mid = count / 2u;
    const bool live2 = count > 1u;
    const uint32_t half = live2 && (p - first) >= mid ? 1u : 0u;
    const uint32_t hs = half ? first + mid : first, hn = half ? count - mid : mid;
    {
      KeyBox b;
      kb_identity(b);
      if (live2) kb_of_box(b, &s_box[6u * e2]);
      const uint32_t id = live2 ? hs : 0xffffffffu - lane;
      kb_run_scan(b, id);
      const uint32_t id_next = __shfl_down(id, 1u, 64);
      if (live2 && (lane == 63u || id_next != id)) {
        if (hs >= chunk0 && hs + hn <= chunk0 + 64u)
          s_area[hs] = __float_as_uint(kb_area(b));
        else
          kb_lds_add(&s_acc[6u * (hs >> 6)], b);
      }
    }
    __syncthreads();
    if (in && (lane == 63u || p + 1u == N) && live2 && hs >= chunk0 && hs + hn > chunk0 + 64u) {
      KeyBox b;
      kb_lds_get(b, &s_acc[6u * (hs >> 6)]);
      s_area[hs] = __float_as_uint(kb_area(b));
    }
    __syncthreads();
    // ---- 4. the costlier half goes first (tlas.rs:95-104): rotate, then every position joins its child range
    if (in) {
      if (count > 1u) {
        const uint32_t l_count = mid, r_count = count - mid, rel = p - first;
        const float la = __uint_as_float(s_area[first]), ra = __uint_as_float(s_area[first + l_count]);
        const bool rot = ra * (float)r_count > la * (float)l_count;
        const uint32_t nrel = rot ? (rel >= l_count ? rel - l_count : rel + r_count) : rel;
        s_ord[first + nrel] = e2;
        const uint32_t lp = rot ? r_count : l_count;
        if (rel < lp) {
          count = lp;
          node = node + 1u;
        } else {
          count = count - lp;
          node = node + 2u * lp;
          first = first + lp;
        }
      } else {
        s_ord[p] = e2;
        count = 0u;   // a leaf: its node was written in step 1
      }
    }
    __syncthreads();
    e = s_ord[p];
  }
  __syncthreads();
  if (in) A.ord[p] = e;
  __syncthreads();
  tlas_pack_tail(A, s_scan, &s_carry);
}

// lib.rs:247-252: the emissive triangles of the geometry of every instance, TLAS order — one workgroup per instance
__global__ __launch_bounds__(256) void k_lights(TlasArgs A, const uint32_t* __restrict__ em_list, uint2* __restrict__ lights) {
  const uint32_t p = blockIdx.x;
  if (p >= A.n_inst) return;
  const uint32_t g = __float_as_uint(A.raw[9 * (size_t)A.ord[p] + 8].z);
  const GeomRow G = A.geoms[g];
  const uint32_t off = A.light_off[p];
  for (uint32_t k = threadIdx.x; k < G.em_count; k += 256u)
    if (off + k < A.n_lights) lights[off + k] = make_uint2(p, em_list[G.em_first + k]);
}

// A geometry that is not skinned has the same vertices, hence the same BLAS, topology rows and emissive list in every
// frame: after the first update of a static description its rows are left where they are and its node block is copied
// from a cache to where this frame's prefix of node counts puts it (node_base[0] in, node_base[1] out).
__global__ __launch_bounds__(256) void k_static_nodes(const float4* __restrict__ cache, uint32_t n_nodes, uint32_t* node_base,
                                                       float4* __restrict__ out_blas) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  const uint32_t base = node_base[0];
  if (i < 2u * n_nodes) out_blas[2 * (size_t)base + i] = cache[i];
  if (i == 0u) node_base[1] = base + n_nodes;
}

// what the host wants to know of one finished build, kept where the next build does not overwrite it
__global__ void k_build_stats(const bvhb::Ctl* __restrict__ ctl, uint32_t levels, uint32_t* __restrict__ out) {
  if (threadIdx.x != 0u || blockIdx.x != 0u) return;
  uint32_t depth = 0;
  while (depth < bvhb::kMaxLevels && ctl->cnt[depth]) depth++;
  out[0] = ctl->cnt[levels];   // != 0: the tree is deeper than the levels launched — build again
  out[1] = depth;
  out[2] = ctl->big_levels;
  out[3] = ctl->n_nodes;
}

}  // namespace wu
#endif
