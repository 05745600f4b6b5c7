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

// Numbering in seven small launches (one workgroup doing everything took 0.5 ms for 110 k nodes, which an animated scene
// would pay on every update):
//   k_treelet_hist<0>, k_treelet_pick<0>   4096-bin histogram of the keys' top 12 bits; the bin that does not fit k_max whole
//   k_treelet_hist<1>, k_treelet_pick<1>   the same inside that bin on the next 12 bits -> key threshold (24 bits deep)
//   k_treelet_count, k_treelet_blockscan, k_treelet_number
//                                          selected nodes (key >= threshold) are numbered first, in original order, the
//                                          others after them in original order: per-1024-node block counts, their scan,
//                                          ballot ranks inside a block
// work: [0, 4096) level-0 histogram, [4096, 8192) level-1 histogram, then the TreeletPick words, then one count per block.
struct TreeletPick {
  uint32_t bin0, above0, threshold, total_sel, all, pad[3];
};
#define RT_TREELET_WORK_HEAD (8192u + 8u)

template <int LEVEL>
__global__ __launch_bounds__(256) void k_treelet_hist(TreeletArgs A, uint32_t* __restrict__ work) {
  __shared__ uint32_t s_hist[4096];
  for (uint32_t b = threadIdx.x; b < 4096u; b += 256u) s_hist[b] = 0u;
  __syncthreads();
  const TreeletPick* P = reinterpret_cast<const TreeletPick*>(work + 8192);
  const uint32_t bin0 = LEVEL == 0 ? 0u : P->bin0;
  if (LEVEL == 1 && P->all) return;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < A.n_nodes; i += gridDim.x * 256u) {
    const uint32_t k = A.key[i];
    if (LEVEL == 0) atomicAdd(&s_hist[k >> 20], 1u);
    else if ((k >> 20) == bin0) atomicAdd(&s_hist[(k >> 8) & 4095u], 1u);
  }
  __syncthreads();
  uint32_t* hist = work + (LEVEL == 0 ? 0u : 4096u);
  for (uint32_t b = threadIdx.x; b < 4096u; b += 256u)
    if (s_hist[b]) atomicAdd(&hist[b], s_hist[b]);
}

// one workgroup of 256 threads: walk the 4096 bins from the top until the count would exceed k_max
template <int LEVEL>
__global__ __launch_bounds__(256) void k_treelet_pick(TreeletArgs A, uint32_t* __restrict__ work) {
  __shared__ uint32_t s_part[256];
  TreeletPick* P = reinterpret_cast<TreeletPick*>(work + 8192);
  if (LEVEL == 1 && P->all) return;
  const uint32_t* hist = work + (LEVEL == 0 ? 0u : 4096u);
  const uint32_t t = threadIdx.x;
  // thread t owns bins [4080 - 16 t, 4096 - 16 t): thread 0 the topmost sixteen
  const uint32_t top = 4096u - 16u * t;
  uint32_t sum = 0u;
  for (uint32_t k = 0; k < 16u; k++) sum += hist[top - 1u - k];
  s_part[t] = sum;
  __syncthreads();
  for (uint32_t off = 1u; off < 256u; off <<= 1) {
    const uint32_t v = t >= off ? s_part[t - off] : 0u;
    __syncthreads();
    s_part[t] += v;
    __syncthreads();
  }
  const uint32_t base = LEVEL == 0 ? 0u : P->above0;
  const uint32_t before = base + s_part[t] - sum;   // nodes in all bins above this thread's sixteen
  const uint32_t total = base + s_part[255];
  __syncthreads();
  if (LEVEL == 0 && t == 0u) {
    P->all = total <= A.k_max ? 1u : 0u;
    if (total <= A.k_max) {
      P->threshold = 0u;
      P->total_sel = total;
    }
  }
  if (total <= A.k_max) return;
  if (before <= A.k_max && before + sum > A.k_max) {   // the overflowing bin is one of mine: exactly one thread gets here
    uint32_t above = before;
    uint32_t b = top - 1u;
    for (uint32_t k = 0; k < 16u; k++, b--) {
      if (above + hist[b] > A.k_max) break;
      above += hist[b];
    }
    if (LEVEL == 0) {
      P->bin0 = b;
      P->above0 = above;
    } else {
      P->threshold = (P->bin0 << 20) + ((b + 1u) << 8);   // keys >= threshold: every bin above bin0, and inside it the sub-bins above b
      P->total_sel = above;
    }
  }
}

__device__ __forceinline__ bool treelet_selected(const TreeletArgs& A, const TreeletPick* P, uint32_t i) {
  return i < A.n_nodes && (P->all != 0u || A.key[i] >= P->threshold);
}
__global__ __launch_bounds__(1024) void k_treelet_count(TreeletArgs A, uint32_t* __restrict__ work) {
  __shared__ uint32_t s_cnt;
  const TreeletPick* P = reinterpret_cast<const TreeletPick*>(work + 8192);
  if (threadIdx.x == 0u) s_cnt = 0u;
  __syncthreads();
  const bool sel = treelet_selected(A, P, blockIdx.x * 1024u + threadIdx.x);
  const unsigned long long m = __ballot(sel);
  if ((threadIdx.x & 63u) == 0u && m) atomicAdd(&s_cnt, (uint32_t)__builtin_popcountll(m));
  __syncthreads();
  if (threadIdx.x == 0u) work[RT_TREELET_WORK_HEAD + blockIdx.x] = s_cnt;
}
// exclusive scan of the per-block counts, in place (one workgroup; n_blocks is small: 1 per 1024 nodes)
__global__ __launch_bounds__(1024) void k_treelet_blockscan(uint32_t* __restrict__ work, uint32_t n_blocks) {
  __shared__ uint32_t s_scan[1024];
  __shared__ uint32_t s_carry;
  uint32_t* cnt = work + RT_TREELET_WORK_HEAD;
  if (threadIdx.x == 0u) s_carry = 0u;
  __syncthreads();
  for (uint32_t c0 = 0u; c0 < n_blocks; c0 += 1024u) {
    const uint32_t i = c0 + threadIdx.x;
    const uint32_t v = i < n_blocks ? cnt[i] : 0u;
    s_scan[threadIdx.x] = v;
    __syncthreads();
    for (uint32_t off = 1u; off < 1024u; off <<= 1) {
      const uint32_t w = threadIdx.x >= off ? s_scan[threadIdx.x - off] : 0u;
      __syncthreads();
      s_scan[threadIdx.x] += w;
      __syncthreads();
    }
    if (i < n_blocks) cnt[i] = s_carry + s_scan[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023u) s_carry += s_scan[1023];
    __syncthreads();
  }
}
__global__ __launch_bounds__(1024) void k_treelet_number(TreeletArgs A, const uint32_t* __restrict__ work) {
  __shared__ uint32_t s_wave[16];
  const TreeletPick* P = reinterpret_cast<const TreeletPick*>(work + 8192);
  const uint32_t i = blockIdx.x * 1024u + threadIdx.x;
  const bool sel = treelet_selected(A, P, i);
  const unsigned long long m = __ballot(sel);
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t rank_in_wave = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
  if (lane == 0u) s_wave[wave] = (uint32_t)__builtin_popcountll(m);
  __syncthreads();
  uint32_t before = 0u;   // selected nodes of this block in earlier waves
  for (uint32_t w = 0; w < wave; w++) before += s_wave[w];
  if (i >= A.n_nodes) return;
  const uint32_t sel_before = work[RT_TREELET_WORK_HEAD + blockIdx.x] + before + rank_in_wave;   // selected nodes with a smaller index
  const uint32_t total_sel = P->all ? A.n_nodes : P->total_sel;
  A.new_index[i] = sel ? sel_before : total_sel + (i - sel_before);
}

// identity numbering (tnodes in the bridge's depth-first order): what a device-resident update(t) uses — the numbering
// passes above are 7 of the 10 launches of this re-layout, and their order only pays when a PREFIX of tnodes is staged in
// LDS (MI355RT_TREELET_MAX), which the default plan never does (rt_api.hip plan_lds)
__global__ __launch_bounds__(256) void k_treelet_iota(uint32_t* __restrict__ new_index, uint32_t n) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < n) new_index[i] = i;
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
