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
// that axis, halves [0, count/2) and the rest, the half with the larger area x count first.  One workgroup; all
// ranges of a recursion depth are worked on together, one position per lane (strided).  The tree is full and every leaf
// holds one instance, so a node's pre-order index follows from the counts alone: left child = node + 1, right child =
// node + 2 * (left count), skip = node + 2 * count - 1.  The stable sort is a rank count inside the range (N^2 / 1024
// comparisons per lane on the first level; instance counts are thousands at most).
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

__device__ __forceinline__ void key_box_init(uint32_t* k) {
  k[0] = k[1] = k[2] = 0xffffffffu;
  k[3] = k[4] = k[5] = 0u;
}
// union into a key box; NaN components are ignored like fmin_nn / fmax_nn of the scene compiler do
__device__ __forceinline__ void key_box_add(uint32_t* k, const float* b) {
  for (int c = 0; c < 3; c++) {
    if (!is_nan(b[c])) atomicMin(&k[c], key_of(b[c]));
    if (!is_nan(b[c + 3])) atomicMax(&k[c + 3], key_of(b[c + 3]));
  }
}
// the key boxes are made by atomics (performed in L2): read them past the L1
__device__ __forceinline__ uint32_t ld_key(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float key_box_area(const uint32_t* k) {   // primitives.rs AABB::area
  const float dx = float_of(ld_key(k + 3)) - float_of(ld_key(k)), dy = float_of(ld_key(k + 4)) - float_of(ld_key(k + 1)),
              dz = float_of(ld_key(k + 5)) - float_of(ld_key(k + 2));
  if (dx < 0.0f || dy < 0.0f || dz < 0.0f) return 0.0f;
  return 2.0f * (dx * dy + dy * dz + dz * dx);
}
__device__ __forceinline__ float min_nn(float a, float b) { return is_nan(b) ? a : (is_nan(a) ? b : (key_of(b) < key_of(a) ? b : a)); }
__device__ __forceinline__ float max_nn(float a, float b) { return is_nan(b) ? a : (is_nan(a) ? b : (key_of(b) > key_of(a) ? b : a)); }

__global__ __launch_bounds__(1024) void k_tlas(TlasArgs A) {
  __shared__ uint32_t s_any;
  __shared__ uint32_t s_scan[1024];
  __shared__ uint32_t s_carry;
  const uint32_t tid = threadIdx.x, N = A.n_inst;
  const float inf = __uint_as_float(0x7f800000u);
  // ---- instance boxes: the BLAS root box through the instance transform (primitives.rs AABB transform: 8 corners)
  for (uint32_t i = tid; i < N; i += 1024u) {
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
      A.box[6 * (size_t)i + r] = mn[r];
      A.box[6 * (size_t)i + 3 + r] = mx[r];
      const float ce = (mn[r] + mx[r]) * 0.5f;
      A.ctr[3 * (size_t)i + r] = ce;
      nan_centre = nan_centre || is_nan(ce);
    }
    if (nan_centre) atomicOr(&A.status[0], 1u);
    A.ord[i] = i;
    A.seg[3 * (size_t)i] = 0u;
    A.seg[3 * (size_t)i + 1] = N;
    A.seg[3 * (size_t)i + 2] = 0u;
  }
  __syncthreads();
  for (uint32_t round = 0; round < 40u; round++) {   // depth <= ceil(log2 N) + 1
    // ---- 1. box of every live range
    for (uint32_t p = tid; p < N; p += 1024u)
      if (A.seg[3 * (size_t)p + 1] && A.seg[3 * (size_t)p] == p) key_box_init(A.skey + 18 * (size_t)p);
    if (tid == 0u) s_any = 0u;
    __syncthreads();
    for (uint32_t p = tid; p < N; p += 1024u)
      if (A.seg[3 * (size_t)p + 1]) key_box_add(A.skey + 18 * (size_t)A.seg[3 * (size_t)p], A.box + 6 * (size_t)A.ord[p]);
    __syncthreads();
    // ---- 2. the node; leaf or axis
    for (uint32_t p = tid; p < N; p += 1024u) {
      const uint32_t count = A.seg[3 * (size_t)p + 1];
      if (!count || A.seg[3 * (size_t)p] != p) continue;
      const uint32_t node = A.seg[3 * (size_t)p + 2];
      const uint32_t* k = A.skey + 18 * (size_t)p;
      const float mn[3] = {float_of(ld_key(k)), float_of(ld_key(k + 1)), float_of(ld_key(k + 2))};
      const float mx[3] = {float_of(ld_key(k + 3)), float_of(ld_key(k + 4)), float_of(ld_key(k + 5))};
      const uint32_t skip = count == 1u ? node + 1u : node + 2u * count - 1u;
      const uint32_t data = count == 1u ? ((p << 3) | 1u) : 0u;
      A.nodes[2 * (size_t)node] = make_float4(mn[0], mn[1], mn[2], __uint_as_float(skip));
      A.nodes[2 * (size_t)node + 1] = make_float4(mx[0], mx[1], mx[2], __uint_as_float(data));
      if (count > 1u) {
        const float ex = mx[0] - mn[0], ey = mx[1] - mn[1], ez = mx[2] - mn[2];
        A.sinfo[2 * (size_t)p] = ey > ex ? 1u : ((ez > ex && ez > ey) ? 2u : 0u);   // tlas.rs:76
        s_any = 1u;
      }
    }
    __syncthreads();
    if (!s_any) break;
    // ---- 3. stable sort of every range by centre on its axis: rank = elements that go before this one
    for (uint32_t p = tid; p < N; p += 1024u) {
      const uint32_t first = A.seg[3 * (size_t)p], count = A.seg[3 * (size_t)p + 1];
      if (count <= 1u) continue;
      const uint32_t axis = A.sinfo[2 * (size_t)first];
      const float kp = A.ctr[3 * (size_t)A.ord[p] + axis];
      uint32_t rank = 0;
      for (uint32_t q = first; q < first + count; q++) {
        const float kq = A.ctr[3 * (size_t)A.ord[q] + axis];
        rank += (kq < kp || (kq == kp && q < p)) ? 1u : 0u;
      }
      A.ord2[first + rank] = A.ord[p];
    }
    for (uint32_t p = tid; p < N; p += 1024u)
      if (A.seg[3 * (size_t)p + 1] > 1u && A.seg[3 * (size_t)p] == p) {
        key_box_init(A.skey + 18 * (size_t)p + 6);
        key_box_init(A.skey + 18 * (size_t)p + 12);
      }
    __syncthreads();
    // ---- 4. boxes of the two halves
    for (uint32_t p = tid; p < N; p += 1024u) {
      const uint32_t first = A.seg[3 * (size_t)p], count = A.seg[3 * (size_t)p + 1];
      if (count <= 1u) continue;
      const uint32_t mid = count / 2u;
      key_box_add(A.skey + 18 * (size_t)first + ((p - first) < mid ? 6 : 12), A.box + 6 * (size_t)A.ord2[p]);
    }
    __syncthreads();
    // ---- 5. the costlier half goes first (tlas.rs:95-104)
    for (uint32_t p = tid; p < N; p += 1024u) {
      const uint32_t count = A.seg[3 * (size_t)p + 1];
      if (count <= 1u || A.seg[3 * (size_t)p] != p) continue;
      const uint32_t l_count = count / 2u, r_count = count - l_count;
      const float la = key_box_area(A.skey + 18 * (size_t)p + 6), ra = key_box_area(A.skey + 18 * (size_t)p + 12);
      A.sinfo[2 * (size_t)p + 1] = (ra * (float)r_count > la * (float)l_count) ? 1u : 0u;
    }
    __syncthreads();
    // ---- 6. rotate, then every position joins its child range
    for (uint32_t p = tid; p < N; p += 1024u) {
      const uint32_t first = A.seg[3 * (size_t)p], count = A.seg[3 * (size_t)p + 1], node = A.seg[3 * (size_t)p + 2];
      if (count == 1u) {
        A.seg[3 * (size_t)p + 1] = 0u;   // its leaf node was written in step 2
        continue;
      }
      if (count == 0u) continue;
      const uint32_t l_count = count / 2u, r_count = count - l_count, rel = p - first;
      const bool rot = A.sinfo[2 * (size_t)first + 1] != 0u;
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
  // ---- lib.rs:237-270: instances, draw commands and the lights' offsets, in TLAS order
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
