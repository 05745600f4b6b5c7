// bvh_build.hip.h — binned-SAH BLAS build on the GPU, bit-identical to the scene compiler's CPU builder.
//
// SURVEY.md §8f N1 (GPU half).  The reference rebuilds every BLAS on one CPU thread inside World::update(t)
// (rust-shader-tools/src/lib.rs:186-193 -> rebuilder.rs:93-98 -> bvh/blas.rs), 147 ms for 263 k triangles here — 26x the
// time this renderer needs to trace a frame of that scene, so an animated scene is build-bound.  A different builder
// (LBVH, refit) would change the traversal order and with it tie-breaks and the node / triangle counters, so this one
// makes the SAME tree: 16 bins on the axis the reference picks (blas.rs:106: y if extent.y > extent.x, else z if it exceeds both, else x — NOT
// the longest axis), SAH sweep, two-pointer partition, costlier child
// first, leaves at <= 4 triangles (blas.rs:99-234 as restated in csrc/scene/scene_compiler.cpp BlasBuilder; the
// parity test compares node arrays and triangle order byte for byte).
//
// Shape: breadth-first, one launch per tree level, one 256-thread workgroup per active node.  Per node:
//   1. box of the range (block reduction over order-mapped triangle boxes)
//   2. leaf test / axis / bin scale (one lane), 16 bins of {count, box} in LDS (LDS atomics on order-preserving keys)
//   3. SAH sweep and split choice (one lane; 15 candidates)
//   4. the two-pointer partition of blas.rs:179-199 without its sequential loop: the k-th misplaced element from the left
//      is exchanged with the k-th misplaced element from the right, which is exactly what the pointer walk does;
//      ranks come from block prefix sums over the two regions, then the swaps run in parallel
//   5. children's ranges written to the other order buffer, rotated when the right child is the costlier one
// Node ids are handed out breadth-first (atomic counter); the stackless pre-order layout the traversal needs is made at
// the end: subtree sizes bottom-up, pre-order indices top-down (one tiny launch per level each), then one emit pass.
// f32 min/max run on order-preserving u32 keys, i.e. -0 < +0: the sign of a zero bound is the one thing a sequential
// f32::min chain leaves to the visiting order (Rust documents it as unspecified); the CPU builder uses the same rule.
#ifndef MI355RT_BVH_BUILD_HIP_H
#define MI355RT_BVH_BUILD_HIP_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bvhb {

constexpr int kBins = 16;

struct BNode {  // breadth-first build record
  float mn[3];
  uint32_t first;
  float mx[3];
  uint32_t count;
  int32_t left, right;  // BFS ids, -1 for a leaf
  uint32_t size, dfs;   // subtree size (nodes), pre-order index
};

struct Build {
  const float4* tri_mn;  // per triangle: padded box and centre (blas.rs:63-90)
  const float4* tri_mx;
  const float4* tri_c;
  uint32_t* order_in;    // position -> triangle id at this level (swapped in place, then copied out)
  uint32_t* order_out;
  uint32_t* order_final; // leaves park their range here
  uint32_t* scratch_l;   // positions of misplaced elements, by rank
  uint32_t* scratch_r;
  uint8_t* bin_cache;    // bin of the triangle at each position, written by the bin pass of the level
  BNode* nodes;
  uint32_t* counters;    // [0] next node id
};

__device__ __forceinline__ uint32_t key_of(float f) {  // order-preserving: a < b  <=>  key(a) < key(b); -0 < +0
  const uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float float_of(uint32_t k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
__device__ __forceinline__ float tmin(float a, float b) { return key_of(b) < key_of(a) ? b : a; }
__device__ __forceinline__ float tmax(float a, float b) { return key_of(b) > key_of(a) ? b : a; }

// blas.rs:63-90: triangle box, padded by 1e-5 on a flat axis, and its centre
__global__ __launch_bounds__(256) void k_tri_boxes(const float4* __restrict__ pos, const uint32_t* __restrict__ idx, uint32_t n_tris,
                                                    float4* __restrict__ tri_mn, float4* __restrict__ tri_mx,
                                                    float4* __restrict__ tri_c, uint32_t* __restrict__ order) {
  const uint32_t t = blockIdx.x * 256u + threadIdx.x;
  if (t >= n_tris) return;
  const float4 a = pos[idx[3 * t]], b = pos[idx[3 * t + 1]], c = pos[idx[3 * t + 2]];
  float mn[3] = {tmin(tmin(a.x, b.x), c.x), tmin(tmin(a.y, b.y), c.y), tmin(tmin(a.z, b.z), c.z)};
  float mx[3] = {tmax(tmax(a.x, b.x), c.x), tmax(tmax(a.y, b.y), c.y), tmax(tmax(a.z, b.z), c.z)};
  float ce[3];
  for (int k = 0; k < 3; k++) {
    const float size = mx[k] - mn[k];
    const float pad = size < 1e-5f ? 1e-5f : 0.0f;
    mn[k] = mn[k] - pad * 0.5f;
    mx[k] = mx[k] + pad * 0.5f;
    ce[k] = (mn[k] + mx[k]) * 0.5f;
  }
  tri_mn[t] = make_float4(mn[0], mn[1], mn[2], 0.0f);
  tri_mx[t] = make_float4(mx[0], mx[1], mx[2], 0.0f);
  tri_c[t] = make_float4(ce[0], ce[1], ce[2], 0.0f);
  order[t] = t;
}

__device__ __forceinline__ uint32_t bin_of(float val, float split_min, float scale) {  // `as usize` then min(BINS - 1)
  const float f = (val - split_min) * scale;
  if (!(f > 0.0f)) return 0u;  // NaN or <= 0
  if (f >= (float)(kBins - 1)) return (uint32_t)(kBins - 1);
  return (uint32_t)f;
}
__device__ __forceinline__ float axis_of(const float4& v, int axis) { return axis == 0 ? v.x : (axis == 1 ? v.y : v.z); }
__device__ __forceinline__ float area_of(const float mn[3], const float mx[3]) {  // primitives.rs AABB::area
  const float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
  if (dx < 0.0f || dy < 0.0f || dz < 0.0f) return 0.0f;
  return 2.0f * (dx * dy + dy * dz + dz * dx);
}

// exclusive prefix sum of a flag over the T threads of the block; returns this thread's rank and the block total
template <int T>
__device__ __forceinline__ uint32_t block_rank(bool flag, uint32_t* s_wave, uint32_t& total) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const unsigned long long m = __ballot(flag);
  const uint32_t in_wave = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
  __syncthreads();  // s_wave may still be read by the previous call
  if (lane == 0u) s_wave[wave] = (uint32_t)__builtin_popcountll(m);
  __syncthreads();
  uint32_t before = 0, all = 0;
  for (uint32_t w = 0; w < (uint32_t)(T / 64); w++) {
    const uint32_t v = s_wave[w];
    before += w < wave ? v : 0u;
    all += v;
  }
  total = all;
  return before + in_wave;
}

// blas.rs:149-177 + 201-217 for one node from its bins: best split, left count, child order
__device__ __forceinline__ void sah_split(const uint32_t* bin_cnt, const uint32_t (*bin_box)[6], uint32_t count, int& leaf, int& split,
                                          uint32_t& L, int& rotate, float* lbox, float* rbox) {
  float l_area[kBins], r_area[kBins];
  uint32_t l_cnt[kBins], r_cnt[kBins];
  const float inf = __uint_as_float(0x7f800000u);
  float cmn[3] = {inf, inf, inf}, cmx[3] = {-inf, -inf, -inf};
  uint32_t sum = 0;
  for (int i = 0; i < kBins; i++) {
    sum += bin_cnt[i];
    for (int c = 0; c < 3; c++) {
      const float bmn = bin_cnt[i] ? float_of(bin_box[i][c]) : inf, bmx = bin_cnt[i] ? float_of(bin_box[i][c + 3]) : -inf;
      cmn[c] = tmin(cmn[c], bmn);
      cmx[c] = tmax(cmx[c], bmx);
    }
    l_area[i] = area_of(cmn, cmx);
    l_cnt[i] = sum;
  }
  for (int c = 0; c < 3; c++) { cmn[c] = inf; cmx[c] = -inf; }
  sum = 0;
  for (int i = kBins - 1; i >= 0; i--) {
    sum += bin_cnt[i];
    for (int c = 0; c < 3; c++) {
      const float bmn = bin_cnt[i] ? float_of(bin_box[i][c]) : inf, bmx = bin_cnt[i] ? float_of(bin_box[i][c + 3]) : -inf;
      cmn[c] = tmin(cmn[c], bmn);
      cmx[c] = tmax(cmx[c], bmx);
    }
    r_area[i] = area_of(cmn, cmx);
    r_cnt[i] = sum;
  }
  float best = inf;
  int best_split = -1;
  for (int i = 0; i < kBins - 1; i++) {
    if (l_cnt[i] == 0u || r_cnt[i + 1] == 0u) continue;
    const float cost = l_area[i] * (float)l_cnt[i] + r_area[i + 1] * (float)r_cnt[i + 1];
    if (cost < best) {
      best = cost;
      best_split = i;
    }
  }
  leaf = best_split < 0;
  L = 0;
  rotate = 0;
  if (!leaf) {
    L = l_cnt[best_split];
    if (L == 0u || L == count) leaf = 1;
    const float l_cost = l_area[best_split] * (float)L, r_cost = r_area[best_split + 1] * (float)(count - L);
    rotate = r_cost > l_cost;  // the costlier child goes first
    // the children's boxes are the partial unions of the sweep: a child holds exactly the triangles of bins <= split
    // (resp. > split), and min / max on the key order do not depend on the order of the operands
    for (int c = 0; c < 3; c++) { lbox[c] = inf; lbox[c + 3] = -inf; rbox[c] = inf; rbox[c + 3] = -inf; }
    for (int i = 0; i < kBins; i++) {
      if (!bin_cnt[i]) continue;
      float* bx = i <= best_split ? lbox : rbox;
      for (int c = 0; c < 3; c++) {
        bx[c] = tmin(bx[c], float_of(bin_box[i][c]));
        bx[c + 3] = tmax(bx[c + 3], float_of(bin_box[i][c + 3]));
      }
    }
  }
  split = best_split;
}

// T = 1024 for the first levels (few, large nodes: one workgroup walks up to the whole mesh, so it needs all the waves a
// CU can hold to hide the order -> triangle gathers), 256 below
template <int T>
__global__ __launch_bounds__(T) void k_level(Build B, const uint32_t* __restrict__ ids, uint32_t id0, uint32_t n_active) {
  constexpr uint32_t kThreads = (uint32_t)T;
  __shared__ uint32_t s_red[T / 64][6];
  __shared__ uint32_t s_bin_cnt[kBins];
  __shared__ uint32_t s_bin_box[kBins][6];
  __shared__ uint32_t s_wave[T / 64];
  __shared__ float s_f[2];        // split_min, scale
  __shared__ float s_box[2][6];   // boxes of the left / right part
  __shared__ int32_t s_i[6];      // leaf flag, axis, best split, L, rotate, nbad
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (blockIdx.x >= n_active) return;
  const uint32_t id = ids ? ids[blockIdx.x] : id0 + blockIdx.x;  // the BFS ids of a level are contiguous
  const uint32_t first = B.nodes[id].first, count = B.nodes[id].count, end = first + count;

  // ---- 1. box of the range: the root reduces its triangles' boxes, every other node got its box from its parent's sweep
  uint32_t k[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
  for (uint32_t p = first + tid; id == 0u && p < end; p += kThreads) {
    const uint32_t t = B.order_in[p];
    const float4 a = B.tri_mn[t], b = B.tri_mx[t];
    k[0] = min(k[0], key_of(a.x)); k[1] = min(k[1], key_of(a.y)); k[2] = min(k[2], key_of(a.z));
    k[3] = max(k[3], key_of(b.x)); k[4] = max(k[4], key_of(b.y)); k[5] = max(k[5], key_of(b.z));
  }
  for (int off = 32; off > 0; off >>= 1)
    for (int c = 0; c < 6; c++) {
      const uint32_t o = (uint32_t)__shfl_xor((int)k[c], off, 64);
      k[c] = c < 3 ? min(k[c], o) : max(k[c], o);
    }
  if (lane == 0u)
    for (int c = 0; c < 6; c++) s_red[wave][c] = k[c];
  __syncthreads();
  if (tid == 0u) {
    float mn[3], mx[3];
    for (int c = 0; c < 3; c++) {
      uint32_t kmn = s_red[0][c], kmx = s_red[0][c + 3];
      for (int w = 1; w < T / 64; w++) {
        kmn = min(kmn, s_red[w][c]);
        kmx = max(kmx, s_red[w][c + 3]);
      }
      if (id == 0u) {
        B.nodes[id].mn[c] = float_of(kmn);
        B.nodes[id].mx[c] = float_of(kmx);
      }
      mn[c] = B.nodes[id].mn[c];
      mx[c] = B.nodes[id].mx[c];
    }
    int leaf = count <= 4u;
    const float ex = mx[0] - mn[0], ey = mx[1] - mn[1], ez = mx[2] - mn[2];
    const int axis = ey > ex ? 1 : ((ez > ex && ez > ey) ? 2 : 0);   // blas.rs:127-133
    const float split_len = axis == 0 ? ex : (axis == 1 ? ey : ez);
    if (split_len < 1e-6f) leaf = 1;
    s_i[0] = leaf;
    s_i[1] = axis;
    s_f[0] = mn[axis];
    s_f[1] = (float)kBins / split_len;
  }
  if (tid < (uint32_t)kBins) {
    s_bin_cnt[tid] = 0u;
    for (int c = 0; c < 3; c++) {
      s_bin_box[tid][c] = 0xffffffffu;   // above every key: an empty bin is (+inf, -inf) after decoding
      s_bin_box[tid][c + 3] = 0u;
    }
  }
  __syncthreads();
  const int axis = s_i[1];
  const float split_min = s_f[0], scale = s_f[1];
  if (!s_i[0]) {
    // ---- 2. bins
    for (uint32_t p = first + tid; p < end; p += kThreads) {
      const uint32_t t = B.order_in[p];
      const uint32_t b = bin_of(axis_of(B.tri_c[t], axis), split_min, scale);
      B.bin_cache[p] = (uint8_t)b;
      const float4 a = B.tri_mn[t], c = B.tri_mx[t];
      atomicAdd(&s_bin_cnt[b], 1u);
      atomicMin(&s_bin_box[b][0], key_of(a.x)); atomicMin(&s_bin_box[b][1], key_of(a.y)); atomicMin(&s_bin_box[b][2], key_of(a.z));
      atomicMax(&s_bin_box[b][3], key_of(c.x)); atomicMax(&s_bin_box[b][4], key_of(c.y)); atomicMax(&s_bin_box[b][5], key_of(c.z));
    }
    __syncthreads();
    // ---- 3. SAH sweep and split choice (blas.rs:149-177, 201-217)
    if (tid == 0u) {
      int leaf, split, rotate;
      uint32_t L;
      float lbox[6], rbox[6];
      sah_split(s_bin_cnt, s_bin_box, count, leaf, split, L, rotate, lbox, rbox);
      for (int c = 0; c < 6; c++) {
        s_box[0][c] = lbox[c];
        s_box[1][c] = rbox[c];
      }
      s_i[0] = leaf;
      s_i[2] = split;
      s_i[3] = (int32_t)L;
      s_i[4] = rotate;
    }
    __syncthreads();
  }
  if (s_i[0]) {  // leaf: blas.rs:111-115 (a count above 7 overflows the 3-bit field exactly like the reference)
    for (uint32_t p = first + tid; p < end; p += kThreads) B.order_final[p] = B.order_in[p];
    if (tid == 0u) {
      B.nodes[id].left = -1;
      B.nodes[id].right = -1;
    }
    return;
  }
  const uint32_t split = (uint32_t)s_i[2], L = (uint32_t)s_i[3];
  const bool rotate = s_i[4] != 0;
  // ---- 4. partition: rank the misplaced elements of both regions
  uint32_t run_l = 0, run_r = 0;
  for (uint32_t base = 0; base < L; base += kThreads) {
    const uint32_t p = first + base + tid;
    bool bad = false;
    if (base + tid < L) bad = B.bin_cache[p] > split;
    uint32_t total;
    const uint32_t r = block_rank<T>(bad, s_wave, total);
    if (bad) B.scratch_l[first + run_l + r] = p;
    run_l += total;
  }
  const uint32_t R = count - L;
  for (uint32_t base = 0; base < R; base += kThreads) {
    const uint32_t p = end - 1u - (base + tid);  // from the right end
    bool bad = false;
    if (base + tid < R) bad = B.bin_cache[p] <= split;
    uint32_t total;
    const uint32_t r = block_rank<T>(bad, s_wave, total);
    if (bad) B.scratch_r[first + run_r + r] = p;
    run_r += total;
  }
  __syncthreads();  // scratch writes of this block are visible to this block
  for (uint32_t q = tid; q < run_l; q += kThreads) {
    const uint32_t a = B.scratch_l[first + q], b = B.scratch_r[first + q];
    const uint32_t ta = B.order_in[a], tb = B.order_in[b];
    B.order_in[a] = tb;
    B.order_in[b] = ta;
  }
  __syncthreads();
  // ---- 5. children
  for (uint32_t p = first + tid; p < end; p += kThreads) {
    const uint32_t rel = p - first;
    const uint32_t nrel = rotate ? (rel >= L ? rel - L : rel + R) : rel;
    B.order_out[first + nrel] = B.order_in[p];
  }
  if (tid == 0u) {
    const uint32_t l_count = rotate ? R : L;
    const uint32_t ids = atomicAdd(&B.counters[0], 2u);
    B.nodes[ids].first = first;
    B.nodes[ids].count = l_count;
    B.nodes[ids + 1u].first = first + l_count;
    B.nodes[ids + 1u].count = count - l_count;
    for (int c = 0; c < 3; c++) {  // after a rotation the former right part is the first child
      B.nodes[ids].mn[c] = s_box[rotate ? 1 : 0][c];
      B.nodes[ids].mx[c] = s_box[rotate ? 1 : 0][c + 3];
      B.nodes[ids + 1u].mn[c] = s_box[rotate ? 0 : 1][c];
      B.nodes[ids + 1u].mx[c] = s_box[rotate ? 0 : 1][c + 3];
    }
    B.nodes[id].left = (int32_t)ids;
    B.nodes[id].right = (int32_t)(ids + 1u);
  }
}


// ------------------------------------------------------------------------------------------- large nodes
// A node with more than kBig triangles is worked on by many workgroups: its range is cut into chunks of kChunk
// positions and every step of k_level becomes its own launch over all chunks of all large nodes of the level
// (bounds -> setup -> bins -> split -> count -> scan -> scatter -> swap -> copy).  Same arithmetic, same result.
constexpr uint32_t kBig = 4096u;
constexpr uint32_t kChunk = 2048u;

struct BigNode {
  uint32_t id, first, count, chunk0, nchunks;
  int32_t leaf, axis, split, rotate;
  uint32_t L, nbad;
  float split_min, scale;
  uint32_t box[6];
  uint32_t bin_cnt[kBins];
  uint32_t bin_box[kBins][6];
  float lbox[6], rbox[6];  // boxes of the left / right part (from the sweep)
};
struct Chunk {
  uint32_t big, j;  // index into the level's BigNode array, chunk index inside the node
};

__global__ __launch_bounds__(256) void k_big_bounds(Build B, BigNode* bn, const Chunk* __restrict__ chunks) {
  __shared__ uint32_t s_red[4][6];
  const Chunk ch = chunks[blockIdx.x];
  BigNode& N = bn[ch.big];
  const uint32_t lo = N.first + ch.j * kChunk, hi = min(lo + kChunk, N.first + N.count);
  uint32_t k[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
  for (uint32_t p = lo + threadIdx.x; p < hi; p += 256u) {
    const uint32_t t = B.order_in[p];
    const float4 a = B.tri_mn[t], b = B.tri_mx[t];
    k[0] = min(k[0], key_of(a.x)); k[1] = min(k[1], key_of(a.y)); k[2] = min(k[2], key_of(a.z));
    k[3] = max(k[3], key_of(b.x)); k[4] = max(k[4], key_of(b.y)); k[5] = max(k[5], key_of(b.z));
  }
  for (int off = 32; off > 0; off >>= 1)
    for (int c = 0; c < 6; c++) {
      const uint32_t o = (uint32_t)__shfl_xor((int)k[c], off, 64);
      k[c] = c < 3 ? min(k[c], o) : max(k[c], o);
    }
  if ((threadIdx.x & 63u) == 0u)
    for (int c = 0; c < 6; c++) s_red[threadIdx.x >> 6][c] = k[c];
  __syncthreads();
  if (threadIdx.x < 6u) {
    const uint32_t c = threadIdx.x;
    if (c < 3u) atomicMin(&N.box[c], min(min(s_red[0][c], s_red[1][c]), min(s_red[2][c], s_red[3][c])));
    else atomicMax(&N.box[c], max(max(s_red[0][c], s_red[1][c]), max(s_red[2][c], s_red[3][c])));
  }
}

__global__ __launch_bounds__(64) void k_big_setup(Build B, BigNode* bn, uint32_t n_big) {
  const uint32_t i = blockIdx.x * 64u + threadIdx.x;
  if (i >= n_big) return;
  BigNode& N = bn[i];
  float mn[3], mx[3];
  for (int c = 0; c < 3; c++) {
    if (N.id == 0u) {  // only the root reduced its box (k_big_bounds); the others carry their parent's partial union
      B.nodes[N.id].mn[c] = float_of(N.box[c]);
      B.nodes[N.id].mx[c] = float_of(N.box[c + 3]);
    }
    mn[c] = B.nodes[N.id].mn[c];
    mx[c] = B.nodes[N.id].mx[c];
  }
  const float ex = mx[0] - mn[0], ey = mx[1] - mn[1], ez = mx[2] - mn[2];
  const int axis = ey > ex ? 1 : ((ez > ex && ez > ey) ? 2 : 0);
  const float split_len = axis == 0 ? ex : (axis == 1 ? ey : ez);
  N.leaf = (N.count <= 4u || split_len < 1e-6f) ? 1 : 0;
  N.axis = axis;
  N.split_min = mn[axis];
  N.scale = (float)kBins / split_len;
}

__global__ __launch_bounds__(256) void k_big_bin(Build B, BigNode* bn, const Chunk* __restrict__ chunks) {
  __shared__ uint32_t s_cnt[kBins];
  __shared__ uint32_t s_box[kBins][6];
  const Chunk ch = chunks[blockIdx.x];
  BigNode& N = bn[ch.big];
  if (N.leaf) return;
  if (threadIdx.x < (uint32_t)kBins) {
    s_cnt[threadIdx.x] = 0u;
    for (int c = 0; c < 3; c++) {
      s_box[threadIdx.x][c] = 0xffffffffu;
      s_box[threadIdx.x][c + 3] = 0u;
    }
  }
  __syncthreads();
  const uint32_t lo = N.first + ch.j * kChunk, hi = min(lo + kChunk, N.first + N.count);
  const int axis = N.axis;
  const float split_min = N.split_min, scale = N.scale;
  for (uint32_t p = lo + threadIdx.x; p < hi; p += 256u) {
    const uint32_t t = B.order_in[p];
    const uint32_t b = bin_of(axis_of(B.tri_c[t], axis), split_min, scale);
    B.bin_cache[p] = (uint8_t)b;
    const float4 a = B.tri_mn[t], c = B.tri_mx[t];
    atomicAdd(&s_cnt[b], 1u);
    atomicMin(&s_box[b][0], key_of(a.x)); atomicMin(&s_box[b][1], key_of(a.y)); atomicMin(&s_box[b][2], key_of(a.z));
    atomicMax(&s_box[b][3], key_of(c.x)); atomicMax(&s_box[b][4], key_of(c.y)); atomicMax(&s_box[b][5], key_of(c.z));
  }
  __syncthreads();
  if (threadIdx.x < (uint32_t)kBins && s_cnt[threadIdx.x]) {
    const uint32_t b = threadIdx.x;
    atomicAdd(&N.bin_cnt[b], s_cnt[b]);
    for (int c = 0; c < 3; c++) {
      atomicMin(&N.bin_box[b][c], s_box[b][c]);
      atomicMax(&N.bin_box[b][c + 3], s_box[b][c + 3]);
    }
  }
}

__global__ __launch_bounds__(64) void k_big_split(BigNode* bn, uint32_t n_big) {
  const uint32_t i = blockIdx.x * 64u + threadIdx.x;
  if (i >= n_big) return;
  BigNode& N = bn[i];
  if (N.leaf) return;
  int leaf, split, rotate;
  uint32_t L;
  sah_split(N.bin_cnt, N.bin_box, N.count, leaf, split, L, rotate, N.lbox, N.rbox);
  N.leaf = leaf;
  N.split = split;
  N.L = L;
  N.rotate = rotate;
}

// misplaced elements per chunk: left-region positions (< first + L) that belong right, right-region ones that belong left
__global__ __launch_bounds__(256) void k_big_count(Build B, const BigNode* __restrict__ bn, const Chunk* __restrict__ chunks,
                                                    uint32_t* __restrict__ chunk_cnt) {
  __shared__ uint32_t s_c[2];
  const Chunk ch = chunks[blockIdx.x];
  const BigNode& N = bn[ch.big];
  if (N.leaf) return;
  if (threadIdx.x < 2u) s_c[threadIdx.x] = 0u;
  __syncthreads();
  const uint32_t lo = N.first + ch.j * kChunk, hi = min(lo + kChunk, N.first + N.count), mid = N.first + N.L;
  uint32_t cl = 0, cr = 0;
  for (uint32_t p = lo + threadIdx.x; p < hi; p += 256u) {
    const bool right = B.bin_cache[p] > (uint32_t)N.split;
    cl += (p < mid && right) ? 1u : 0u;
    cr += (p >= mid && !right) ? 1u : 0u;
  }
  for (int off = 32; off > 0; off >>= 1) {
    cl += (uint32_t)__shfl_xor((int)cl, off, 64);
    cr += (uint32_t)__shfl_xor((int)cr, off, 64);
  }
  if ((threadIdx.x & 63u) == 0u) {
    atomicAdd(&s_c[0], cl);
    atomicAdd(&s_c[1], cr);
  }
  __syncthreads();
  if (threadIdx.x < 2u) chunk_cnt[2 * (N.chunk0 + ch.j) + threadIdx.x] = s_c[threadIdx.x];
}

// rank bases per chunk: misplaced-left ranks grow with the position, misplaced-right ranks grow towards the left
__global__ __launch_bounds__(64) void k_big_scan(BigNode* bn, uint32_t n_big, const uint32_t* __restrict__ chunk_cnt,
                                                  uint32_t* __restrict__ chunk_base) {
  const uint32_t i = blockIdx.x * 64u + threadIdx.x;
  if (i >= n_big) return;
  BigNode& N = bn[i];
  if (N.leaf) return;
  uint32_t run = 0;
  for (uint32_t j = 0; j < N.nchunks; j++) {
    chunk_base[2 * (N.chunk0 + j)] = run;
    run += chunk_cnt[2 * (N.chunk0 + j)];
  }
  N.nbad = run;
  run = 0;
  for (uint32_t j = N.nchunks; j-- > 0;) {
    chunk_base[2 * (N.chunk0 + j) + 1] = run;
    run += chunk_cnt[2 * (N.chunk0 + j) + 1];
  }
}

__global__ __launch_bounds__(256) void k_big_scatter(Build B, const BigNode* __restrict__ bn, const Chunk* __restrict__ chunks,
                                                      const uint32_t* __restrict__ chunk_cnt, const uint32_t* __restrict__ chunk_base) {
  __shared__ uint32_t s_wave[4];
  const Chunk ch = chunks[blockIdx.x];
  const BigNode& N = bn[ch.big];
  if (N.leaf) return;
  const uint32_t lo = N.first + ch.j * kChunk, hi = min(lo + kChunk, N.first + N.count), mid = N.first + N.L;
  const uint32_t cidx = N.chunk0 + ch.j;
  const uint32_t base_l = chunk_base[2 * cidx], base_r = chunk_base[2 * cidx + 1], tot_r = chunk_cnt[2 * cidx + 1];
  uint32_t run_l = 0, run_r = 0;
  for (uint32_t q = lo; q < hi; q += 256u) {
    const uint32_t p = q + threadIdx.x;
    bool bl = false, br = false;
    if (p < hi) {
      const bool right = B.bin_cache[p] > (uint32_t)N.split;
      bl = p < mid && right;
      br = p >= mid && !right;
    }
    uint32_t total;
    uint32_t r = block_rank<256>(bl, s_wave, total);
    if (bl) B.scratch_l[N.first + base_l + run_l + r] = p;
    run_l += total;
    r = block_rank<256>(br, s_wave, total);
    // ranks of the right region count from the node's right end: inside the chunk the highest position comes first
    if (br) B.scratch_r[N.first + base_r + (tot_r - 1u - (run_r + r))] = p;
    run_r += total;
  }
}

__global__ __launch_bounds__(256) void k_big_swap(Build B, const BigNode* __restrict__ bn, const Chunk* __restrict__ chunks) {
  const Chunk ch = chunks[blockIdx.x];
  const BigNode& N = bn[ch.big];
  if (N.leaf) return;
  const uint32_t hi = min((ch.j + 1u) * kChunk, N.nbad);
  for (uint32_t q = ch.j * kChunk + threadIdx.x; q < hi; q += 256u) {
    const uint32_t a = B.scratch_l[N.first + q], b = B.scratch_r[N.first + q];
    const uint32_t ta = B.order_in[a], tb = B.order_in[b];
    B.order_in[a] = tb;
    B.order_in[b] = ta;
  }
}

__global__ __launch_bounds__(256) void k_big_copy(Build B, const BigNode* __restrict__ bn, const Chunk* __restrict__ chunks) {
  const Chunk ch = chunks[blockIdx.x];
  const BigNode& N = bn[ch.big];
  const uint32_t lo = N.first + ch.j * kChunk, hi = min(lo + kChunk, N.first + N.count);
  if (N.leaf) {
    for (uint32_t p = lo + threadIdx.x; p < hi; p += 256u) B.order_final[p] = B.order_in[p];
    if (ch.j == 0u && threadIdx.x == 0u) {
      B.nodes[N.id].left = -1;
      B.nodes[N.id].right = -1;
    }
    return;
  }
  const uint32_t L = N.L, R = N.count - N.L;
  const bool rotate = N.rotate != 0;
  for (uint32_t p = lo + threadIdx.x; p < hi; p += 256u) {
    const uint32_t rel = p - N.first;
    const uint32_t nrel = rotate ? (rel >= L ? rel - L : rel + R) : rel;
    B.order_out[N.first + nrel] = B.order_in[p];
  }
  if (ch.j == 0u && threadIdx.x == 0u) {
    const uint32_t l_count = rotate ? R : L;
    const uint32_t ids = atomicAdd(&B.counters[0], 2u);
    B.nodes[ids].first = N.first;
    B.nodes[ids].count = l_count;
    B.nodes[ids + 1u].first = N.first + l_count;
    B.nodes[ids + 1u].count = N.count - l_count;
    const float* fb = rotate ? N.rbox : N.lbox;
    const float* sb = rotate ? N.lbox : N.rbox;
    for (int c = 0; c < 3; c++) {
      B.nodes[ids].mn[c] = fb[c];
      B.nodes[ids].mx[c] = fb[c + 3];
      B.nodes[ids + 1u].mn[c] = sb[c];
      B.nodes[ids + 1u].mx[c] = sb[c + 3];
    }
    B.nodes[N.id].left = (int32_t)ids;
    B.nodes[N.id].right = (int32_t)(ids + 1u);
  }
}

// subtree sizes, one level at a time from the deepest to the root
__global__ __launch_bounds__(256) void k_sizes(BNode* nodes, uint32_t id0, uint32_t n) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  BNode& nd = nodes[id0 + i];
  nd.size = nd.left < 0 ? 1u : 1u + nodes[nd.left].size + nodes[nd.right].size;
}
// pre-order indices, one level at a time from the root down (the root's is 0)
__global__ __launch_bounds__(256) void k_preorder(BNode* nodes, uint32_t id0, uint32_t n) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const BNode& nd = nodes[id0 + i];
  if (nd.left < 0) return;
  nodes[nd.left].dfs = nd.dfs + 1u;
  nodes[nd.right].dfs = nd.dfs + 1u + nodes[nd.left].size;
}
// the node array the traversal reads: {min, skip} {max, data}, skip = index after the subtree (BLAS-local)
__global__ __launch_bounds__(256) void k_emit(const BNode* __restrict__ nodes, uint32_t n, float4* __restrict__ out) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const BNode nd = nodes[i];
  const uint32_t skip = nd.dfs + nd.size;
  const uint32_t data = nd.left < 0 ? ((nd.first << 3) | nd.count) : 0u;
  out[2 * nd.dfs] = make_float4(nd.mn[0], nd.mn[1], nd.mn[2], __uint_as_float(skip));
  out[2 * nd.dfs + 1] = make_float4(nd.mx[0], nd.mx[1], nd.mx[2], __uint_as_float(data));
}

}  // namespace bvhb
#endif
