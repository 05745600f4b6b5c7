// k_pairs.hip.h — upload-time re-layout of the node array into CHILD-PAIR records (what k_pairwalk.hip.h walks).
// Part of the kernel set of csrc/kernels.hip.h (included from there, in order; not a stand-alone header).
//
// The bridge's node arrays (TLAS ++ BLAS, bvh/mod.rs StacklessBVHNode) encode the tree by POSITION: an inner node's
// first child is the next element, `skip` names the node that follows its subtree (BLAS: relative to the BLAS root;
// Raytracer.wgsl:455-528).  Derived here, once per upload, all on the GPU:
//
//   pairs      4 x float4 per INNER node X, records in array order of their nodes (64-byte aligned):
//                {L.min.xyz, wordL} {L.max.xyz, 0} {R.min.xyz, wordR} {R.max.xyz, skipX}
//              L = X + 1, R = the node L's skip pointer names (the sibling).  word(C) = RT_PAIR_INNER | record of C
//              (C inner) or C's own leaf word (BLAS: first << 3 | count; TLAS: instance << 3 | 1).  skipX = record whose
//              RIGHT child follows X's subtree in pre-order, RT_REF_END when that leaves the TLAS / the BLAS: the
//              stackless fall-back of the walk follows it.
//   root_rec   2 x float4 per instance: {BLAS root.min, word(root)} {root.max, 0}, and one more for the TLAS root
//              (record index n_instances): roots are tested from these, they have no parent pair.
//
// tests/pair_layout.py restates the same construction in numpy; test_gpu_parity.py compares the arrays byte for byte.
#ifndef MI355RT_K_PAIRS_HIP_H
#define MI355RT_K_PAIRS_HIP_H

namespace rtk {

struct PairArgs {
  const float4* nodes;     // original, 2 per node, TLAS ++ BLAS
  float4* pairs;           // 4 per inner node
  uint32_t* pair_of;       // n_nodes: node -> record index of an inner node (exclusive count of inner nodes before it)
  uint32_t* parent;        // n_nodes: node -> its parent node, 0xffffffff for roots / unreachable nodes
  const uint32_t* roots;   // sorted unique BLAS-local root offsets of the instances (validated)
  uint32_t n_nodes, n_tlas, n_roots, pad;
};

__device__ __forceinline__ bool pair_is_inner(const PairArgs& A, uint32_t i) { return __float_as_uint(A.nodes[2 * (size_t)i + 1].w) == 0u; }

// largest r with roots[r] <= local (roots sorted); no root at or below -> 0xffffffff
__device__ __forceinline__ uint32_t pair_root_of(const PairArgs& A, uint32_t local) {
  if (A.n_roots == 0u) return 0xffffffffu;
  uint32_t lo = 0u, hi = A.n_roots;
  while (hi - lo > 1u) {
    const uint32_t mid = (lo + hi) >> 1;
    if (A.roots[mid] <= local) lo = mid; else hi = mid;
  }
  return A.roots[lo] <= local ? A.roots[lo] : 0xffffffffu;
}
// the level node i walks in: first node and end of its array, and the absolute index its skip pointer names
__device__ __forceinline__ void pair_level(const PairArgs& A, uint32_t i, uint32_t& start, uint32_t& end, uint32_t& target) {
  const uint32_t skip = __float_as_uint(A.nodes[2 * (size_t)i].w);
  if (i < A.n_tlas) {
    start = 0u;
    end = __float_as_uint(A.nodes[0].w);
    target = skip;
  } else {
    const uint32_t r = pair_root_of(A, i - A.n_tlas);
    if (r == 0xffffffffu) {           // a node no instance reaches: never walked; keep every index in range
      start = i;
      end = i;
      target = 0xffffffffu;
      return;
    }
    start = A.n_tlas + r;
    end = start + __float_as_uint(A.nodes[2 * (size_t)start].w);
    target = start + skip;
  }
  if (end > A.n_nodes) end = A.n_nodes;
}

// numbering of the inner nodes: per-1024-block counts -> k_treelet_blockscan's scan -> ranks (work[RT_TREELET_WORK_HEAD + b])
__global__ __launch_bounds__(1024) void k_pair_count(PairArgs A, uint32_t* __restrict__ work) {
  __shared__ uint32_t s_cnt;
  if (threadIdx.x == 0u) s_cnt = 0u;
  __syncthreads();
  const uint32_t i = blockIdx.x * 1024u + threadIdx.x;
  const bool inner = i < A.n_nodes && pair_is_inner(A, i);
  const unsigned long long m = __ballot(inner);
  if ((threadIdx.x & 63u) == 0u && m) atomicAdd(&s_cnt, (uint32_t)__builtin_popcountll(m));
  __syncthreads();
  if (threadIdx.x == 0u) work[RT_TREELET_WORK_HEAD + blockIdx.x] = s_cnt;
}
__global__ __launch_bounds__(1024) void k_pair_number(PairArgs A, const uint32_t* __restrict__ work) {
  __shared__ uint32_t s_wave[16];
  const uint32_t i = blockIdx.x * 1024u + threadIdx.x;
  const bool inner = i < A.n_nodes && pair_is_inner(A, i);
  const unsigned long long m = __ballot(inner);
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
  if (lane == 0u) s_wave[wave] = (uint32_t)__builtin_popcountll(m);
  __syncthreads();
  uint32_t before = 0u;
  for (uint32_t w = 0; w < wave; w++) before += s_wave[w];
  if (i >= A.n_nodes) return;
  A.pair_of[i] = work[RT_TREELET_WORK_HEAD + blockIdx.x] + before + rank;
  A.parent[i] = 0xffffffffu;
}
// children of inner node i (absolute indices), or false when the node is not a walkable inner node
__device__ __forceinline__ bool pair_children(const PairArgs& A, uint32_t i, uint32_t& l, uint32_t& r) {
  uint32_t start, end, target;
  pair_level(A, i, start, end, target);
  l = i + 1u;
  if (l >= end) return false;
  uint32_t ls, le, lt;
  pair_level(A, l, ls, le, lt);
  r = lt;
  return r > l && r < end;
}
__global__ __launch_bounds__(256) void k_pair_parent(PairArgs A) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= A.n_nodes || !pair_is_inner(A, i)) return;
  uint32_t l, r;
  if (!pair_children(A, i, l, r)) return;
  A.parent[l] = i;
  A.parent[r] = i;
}
__device__ __forceinline__ uint32_t pair_word(const PairArgs& A, uint32_t c) {
  const uint32_t data = __float_as_uint(A.nodes[2 * (size_t)c + 1].w);
  return data == 0u ? (RT_PAIR_INNER | A.pair_of[c]) : data;
}
__global__ __launch_bounds__(256) void k_pair_emit(PairArgs A) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= A.n_nodes || !pair_is_inner(A, i)) return;
  uint32_t l, r;
  const bool ok = pair_children(A, i, l, r);
  if (!ok) l = r = 0u;       // unreachable / malformed (never walked): any in-range record
  uint32_t start, end, target;
  pair_level(A, i, start, end, target);
  uint32_t skip_x = RT_REF_END;
  if (ok && target < end) {
    const uint32_t p = A.parent[target];
    if (p != 0xffffffffu) skip_x = A.pair_of[p];
  }
  const float4 llo = A.nodes[2 * (size_t)l], lhi = A.nodes[2 * (size_t)l + 1];
  const float4 rlo = A.nodes[2 * (size_t)r], rhi = A.nodes[2 * (size_t)r + 1];
  float4* out = A.pairs + 4 * (size_t)A.pair_of[i];
  out[0] = make_float4(llo.x, llo.y, llo.z, __uint_as_float(pair_word(A, l)));
  out[1] = make_float4(lhi.x, lhi.y, lhi.z, 0.0f);
  out[2] = make_float4(rlo.x, rlo.y, rlo.z, __uint_as_float(pair_word(A, r)));
  out[3] = make_float4(rhi.x, rhi.y, rhi.z, __uint_as_float(skip_x));
}
// root records: one per instance (its BLAS root), then the TLAS root at index n_inst
__global__ __launch_bounds__(256) void k_pair_roots(PairArgs A, const float4* __restrict__ inst, float4* __restrict__ root_rec,
                                                    uint32_t n_inst) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i > n_inst) return;
  uint32_t node = 0u;
  bool valid = A.n_tlas != 0u;
  if (i < n_inst) {
    node = A.n_tlas + __float_as_uint(inst[9 * (size_t)i + 8].x);
    valid = node < A.n_nodes;
  }
  float4 lo = make_float4(0.0f, 0.0f, 0.0f, 0.0f), hi = lo;
  if (valid) {
    lo = A.nodes[2 * (size_t)node];
    hi = A.nodes[2 * (size_t)node + 1];
    lo.w = __uint_as_float(pair_word(A, node));
    hi.w = 0.0f;
  }   // else: no such node (validate_scene refuses such a scene): word 0 = a leaf without triangles
  root_rec[2 * (size_t)i] = lo;
  root_rec[2 * (size_t)i + 1] = hi;
}

}  // namespace rtk
#endif
