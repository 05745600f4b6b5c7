// k_treelet.hip.h — upload-time re-layout of the node array for traversal: explicit child pointers + treelet-first order.
// Part of the kernel set of csrc/kernels.hip.h (included from there, in order; not a stand-alone header).
//
// The bridge's node arrays (TLAS ++ BLAS, bvh/mod.rs StacklessBVHNode) encode the tree by POSITION: an inner node's first
// child is the next element and `skip` is an offset from the BLAS root (Raytracer.wgsl:455-528).  The traversal kernels
// read a derived array instead, `tnodes`, with the same boxes and leaf words but both successors explicit:
//
//   tnodes[2j]   = {min.xyz, bits(skip')}    skip' = index in tnodes of the node the walk goes to when this one is missed
//                                            or finished, RT_NODE_END when that leaves the TLAS / the BLAS
//   tnodes[2j+1] = {max.xyz, bits(data')}    inner node: data' = 0x80000000 | index of the first child
//                                            leaf: data' = the original word (first << 3 | count; TLAS: instance << 3 | 1)
//
// Every walk visits the same nodes in the same order and makes the same tests (bit-identical results and counters), but
// the ORDER of the array is now free.  It is chosen so that the nodes a ray is most likely to visit come first: the
// k_max nodes with the largest surface area (world-space for the TLAS, object-space area x the summed squared scale of
// the instances that use the BLAS) — the classic SAH visit-probability estimate — followed by all other nodes in their
// original depth-first order.  The trace kernels stage a prefix of `tnodes` in LDS (whatever fits beside the wave
// queues in the CU's 160 KB): on the 263 k-triangle config the first 3 200 nodes take 73 % of all node visits, on the
// 1 000-instance config the whole TLAS and every BLAS fit.
#ifndef MI355RT_K_TREELET_HIP_H
#define MI355RT_K_TREELET_HIP_H

namespace rtk {

#define RT_NODE_END 0xffffffffu
#define RT_NODE_INNER 0x80000000u

struct TreeletArgs {
  const float4* nodes;     // original, 2 per node, TLAS ++ BLAS
  float4* tnodes;          // derived, 2 per node
  uint32_t* key;           // n_nodes: order-preserving weight key
  uint32_t* new_index;     // n_nodes: original index -> index in tnodes
  const uint32_t* roots;   // sorted unique BLAS-local root offsets of the instances (validated)
  const float* root_w;     // per root: sum over the instances that use it of |det(M3x3)|^(2/3)
  uint32_t n_nodes, n_tlas, n_roots, k_max;
};

// largest r with roots[r] <= local (roots sorted, roots[0] == 0 whenever a BLAS node exists); n_roots == 0 -> 0xffffffff
__device__ __forceinline__ uint32_t treelet_root_of(const TreeletArgs& A, uint32_t local) {
  if (A.n_roots == 0u) return 0xffffffffu;
  uint32_t lo = 0u, hi = A.n_roots;
  while (hi - lo > 1u) {
    const uint32_t mid = (lo + hi) >> 1;
    if (A.roots[mid] <= local) lo = mid; else hi = mid;
  }
  return A.roots[lo] <= local ? lo : 0xffffffffu;
}

__global__ __launch_bounds__(256) void k_treelet_weight(TreeletArgs A) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= A.n_nodes) return;
  const float4 lo = A.nodes[2 * (size_t)i], hi = A.nodes[2 * (size_t)i + 1];
  const float dx = hi.x - lo.x, dy = hi.y - lo.y, dz = hi.z - lo.z;
  float w = (dx >= 0.0f && dy >= 0.0f && dz >= 0.0f) ? 2.0f * (dx * dy + dy * dz + dz * dx) : 0.0f;
  if (i >= A.n_tlas) {
    const uint32_t r = treelet_root_of(A, i - A.n_tlas);
    w = (r == 0xffffffffu) ? 0.0f : w * A.root_w[r];
  }
  if (!(w >= 0.0f)) w = 0.0f;            // NaN -> 0
  if (w > 3.0e38f) w = 3.0e38f;
  uint32_t k = __float_as_uint(w);       // non-negative floats order like their bit patterns
  if (i == 0u) k = 0xffffffffu;          // the TLAS root is always node 0 of tnodes
  A.key[i] = k;
}

// One workgroup: pick a key threshold such that at most k_max nodes lie at or above it (two 4096-bin histogram levels:
// the top 24 bits of the key), then number the nodes — selected ones first, in original order; the others after them,
// in original order as well.
__global__ __launch_bounds__(1024) void k_treelet_order(TreeletArgs A) {
  __shared__ uint32_t hist[4096];
  __shared__ uint32_t s_bin, s_above, s_scan[1024], s_base_sel, s_base_rest, s_total_sel;
  const uint32_t t = threadIdx.x;
  uint32_t threshold = 0u;
  for (int level = 0; level < 2; level++) {
    for (uint32_t b = t; b < 4096u; b += 1024u) hist[b] = 0u;
    __syncthreads();
    const uint32_t prefix_bin = level == 0 ? 0u : s_bin;
    for (uint32_t i = t; i < A.n_nodes; i += 1024u) {
      const uint32_t k = A.key[i];
      if (level == 0) atomicAdd(&hist[k >> 20], 1u);
      else if ((k >> 20) == prefix_bin) atomicAdd(&hist[(k >> 8) & 4095u], 1u);
    }
    __syncthreads();
    if (t == 0u) {
      // walk the bins from the top: stop at the first bin that would take the count above k_max
      uint32_t above = level == 0 ? 0u : s_above;
      int b = 4095;
      for (; b >= 0; b--) {
        if (above + hist[b] > A.k_max) break;
        above += hist[b];
      }
      s_above = above;                       // nodes strictly above bin b (they are all selected)
      s_bin = b < 0 ? 0u : (uint32_t)b;      // the bin that does not fit as a whole
      if (b < 0) s_bin = 0xffffffffu;        // everything fits
    }
    __syncthreads();
    if (s_bin == 0xffffffffu) {
      threshold = 0u;                        // select every node
      break;
    }
    if (level == 0) threshold = (s_bin + 1u) << 20;            // provisional: all bins above the one that overflows
    else threshold = (prefix_bin << 20) | ((s_bin + 1u) << 8);  // refined inside that bin
    if (level == 1 && s_bin == 4095u) threshold = (prefix_bin + 1u) << 20;
    __syncthreads();
  }
  // numbering: chunked exclusive scan of the selection flags
  if (t == 0u) {
    s_base_sel = 0u;
    s_base_rest = 0u;
  }
  __syncthreads();
  // first pass: count the selected nodes (needed for the base of the others)
  uint32_t mine = 0u;
  for (uint32_t i = t; i < A.n_nodes; i += 1024u) mine += (A.key[i] >= threshold && threshold != 0u) || threshold == 0u ? 1u : 0u;
  s_scan[t] = mine;
  __syncthreads();
  for (uint32_t off = 512u; off > 0u; off >>= 1) {
    if (t < off) s_scan[t] += s_scan[t + off];
    __syncthreads();
  }
  if (t == 0u) s_total_sel = s_scan[0];
  __syncthreads();
  const uint32_t total_sel = s_total_sel;
  for (uint32_t c0 = 0u; c0 < A.n_nodes; c0 += 1024u) {
    const uint32_t i = c0 + t;
    const bool valid = i < A.n_nodes;
    const bool sel = valid && (threshold == 0u || A.key[i] >= threshold);
    s_scan[t] = sel ? 1u : 0u;
    __syncthreads();
    for (uint32_t off = 1u; off < 1024u; off <<= 1) {   // Hillis-Steele inclusive scan
      const uint32_t v = t >= off ? s_scan[t - off] : 0u;
      __syncthreads();
      s_scan[t] += v;
      __syncthreads();
    }
    const uint32_t incl = s_scan[t], excl = incl - (sel ? 1u : 0u);
    if (valid) A.new_index[i] = sel ? s_base_sel + excl : total_sel + s_base_rest + (t - excl);
    __syncthreads();
    if (t == 1023u) {
      s_base_sel += incl;
      s_base_rest += 1024u - incl;   // only consulted for later chunks, whose nodes are all valid up to the last one
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void k_treelet_remap(TreeletArgs A) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= A.n_nodes) return;
  const float4 lo = A.nodes[2 * (size_t)i], hi = A.nodes[2 * (size_t)i + 1];
  const uint32_t skip = __float_as_uint(lo.w), data = __float_as_uint(hi.w);
  uint32_t skip_new = RT_NODE_END, data_new = data;
  if (i < A.n_tlas) {
    // TLAS: absolute skip pointers, the walk ends at nodes[0].skip (Raytracer.wgsl:499-501)
    const uint32_t end = __float_as_uint(A.nodes[0].w);
    if (skip < end && skip < A.n_tlas) skip_new = A.new_index[skip];
    if (data == 0u) data_new = (i + 1u < A.n_tlas) ? (RT_NODE_INNER | A.new_index[i + 1u]) : RT_NODE_INNER;
  } else {
    const uint32_t local = i - A.n_tlas;
    const uint32_t r = treelet_root_of(A, local);
    if (r != 0xffffffffu) {
      const uint32_t root = A.roots[r];
      const uint32_t end = root + __float_as_uint(A.nodes[2 * (size_t)(A.n_tlas + root)].w);  // BLAS-local end
      const uint32_t target = root + skip;
      if (local < end && target < end && A.n_tlas + target < A.n_nodes) skip_new = A.new_index[A.n_tlas + target];
      if (data == 0u) data_new = (i + 1u < A.n_nodes) ? (RT_NODE_INNER | A.new_index[i + 1u]) : RT_NODE_INNER;
    } else if (data == 0u) {
      data_new = RT_NODE_INNER;  // unreachable node: never visited
    }
  }
  const uint32_t j = A.new_index[i];
  A.tnodes[2 * (size_t)j] = make_float4(lo.x, lo.y, lo.z, __uint_as_float(skip_new));
  A.tnodes[2 * (size_t)j + 1] = make_float4(hi.x, hi.y, hi.z, __uint_as_float(data_new));
}

// per instance: index in tnodes of its BLAS root
__global__ __launch_bounds__(256) void k_treelet_inst_roots(const float4* __restrict__ inst, const uint32_t* __restrict__ new_index,
                                                            uint32_t* __restrict__ inst_root, uint32_t n_inst, uint32_t n_tlas,
                                                            uint32_t n_nodes) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n_inst) return;
  const uint32_t off = __float_as_uint(inst[9 * (size_t)i + 8].x);
  const uint32_t idx = n_tlas + off;
  inst_root[i] = idx < n_nodes ? new_index[idx] : RT_NODE_END;
}

}  // namespace rtk
#endif
