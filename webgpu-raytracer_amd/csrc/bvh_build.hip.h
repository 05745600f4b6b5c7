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
// Shape: breadth-first, one launch per tree level, NO host synchronisation between levels (round 3): the number of
// nodes of every level lives in device memory (Ctl::cnt), a level's kernel reads its range from there, walks it with a
// grid-stride loop and hands its children ids base(level + 1) + atomicAdd(cnt[level + 1], 2); the host only decides
// how many levels to launch (from the previous build of the mesh) and checks afterwards that the last one was empty.
// Per node:
//   1. box of the range (the root reduces its triangles' boxes; every other node got its box from its parent's sweep)
//   2. leaf test / axis / bin scale (one lane), 16 bins of {count, box} in LDS (LDS atomics on order-preserving keys)
//   3. SAH sweep and split choice: prefix / suffix unions over the 16 bins on 16 lanes (wave shuffles), 15 candidates
//   4. the two-pointer partition of blas.rs:179-199 without its sequential loop: the k-th misplaced element from the left
//      is exchanged with the k-th misplaced element from the right, which is exactly what the pointer walk does;
//      ranks come from block prefix sums over the two regions, then the swaps run in parallel
//   5. children's ranges written to the other order buffer, rotated when the right child is the costlier one
// The stackless pre-order layout the traversal needs comes from ONE prefix sum at the end instead of two passes per
// level: with LB(p) = number of leaves that start before position p of the final triangle order, a node over
// [first, first + count) that is the LEFT descendant of `leftdepth` of its ancestors has
//     pre-order index = 2 LB(first) + leftdepth        (every leaf start but the first is the split point of exactly one
//     subtree size    = 2 (LB(first + count) - LB(first)) - 1     inner node: those entirely to the left and those
//                                                                  ancestors the node hangs to the right of)
// f32 min/max run on order-preserving u32 keys, i.e. -0 < +0: the sign of a zero bound is the one thing a sequential
// f32::min chain leaves to the visiting order (Rust documents it as unspecified); the CPU builder uses the same rule.
#ifndef MI355RT_BVH_BUILD_HIP_H
#define MI355RT_BVH_BUILD_HIP_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bvhb {

constexpr int kBins = 16;
#ifndef RT_BLAS_BIG
#define RT_BLAS_BIG 4096u
#endif
#ifndef RT_BLAS_CHUNK
#define RT_BLAS_CHUNK 2048u
#endif
constexpr uint32_t kBig = RT_BLAS_BIG;      // nodes above this many triangles are worked on by many workgroups (k_big_*)
constexpr uint32_t kChunk = RT_BLAS_CHUNK;  // ... in chunks of this many positions

constexpr uint32_t kMaxLevels = 1024;  // deeper trees are refused (Ctl::cnt has one word per level)

struct BNode {  // breadth-first build record
  float mn[3];
  uint32_t first;
  float mx[3];
  uint32_t count;
  int32_t left, right;   // BFS ids, -1 for a leaf
  uint32_t leftdepth;    // ancestors of which this node is in the LEFT subtree
  uint32_t gen;          // Build::gen of the build that made this record (the id space is not dense: k_emit skips the rest)
};

// Device-side bookkeeping of one build (zeroed by k_tri_boxes)
struct Ctl {
  uint32_t n_big, n_chunks, n_small, big_levels;  // plan of the current large-node level; levels that had a node > kBig
  uint32_t n_leaves, n_nodes, n_sub, pad1;        // results (k_scan_top); ids handed to the in-wave subtrees (k_subtree)
  uint32_t base[kMaxLevels + 2];                  // first BFS id of every level
  uint32_t cnt[kMaxLevels + 2];                   // nodes of every level
};

struct Build {
  const float4* tri_mn;  // per triangle: padded box and centre (blas.rs:63-90)
  const float4* tri_mx;
  const float4* tri_c;
  uint32_t* ord[2];      // position -> triangle id: level L reads ord[L & 1] (swapped in place) and writes ord[~L & 1]
  uint32_t* order_final; // leaves park their range here
  uint32_t* scratch_l;   // positions of misplaced elements, by rank
  uint32_t* scratch_r;
  uint8_t* bin_cache;    // bin of the triangle at each position, written by the bin pass of the level
  uint32_t* leaf_flag;   // 1 at every position a leaf starts at; scanned into LB (n + 1 entries) at the end
  uint32_t* small_ids;   // nodes <= kBig of a large-node level
  BNode* nodes;          // 4 n records: [0, 2n) ids of the level kernels, [2n, 4n) blocks of the in-wave subtrees
  Ctl* ctl;
  uint32_t n_tris, gen;  // gen: stamp of this build's node records
};
constexpr uint32_t kSubtree = 64u;   // a node of at most this many triangles is finished by ONE wave (k_subtree)

__device__ __forceinline__ uint32_t key_of(float f) {  // order-preserving: a < b  <=>  key(a) < key(b); -0 < +0
  const uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float float_of(uint32_t k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
__device__ __forceinline__ float tmin(float a, float b) { return key_of(b) < key_of(a) ? b : a; }
__device__ __forceinline__ float tmax(float a, float b) { return key_of(b) > key_of(a) ? b : a; }

// blas.rs:63-90: triangle box, padded by 1e-5 on a flat axis, and its centre; also the start state of the build
__global__ __launch_bounds__(256) void k_tri_boxes(const float4* __restrict__ pos, const uint32_t* __restrict__ idx, uint32_t n_tris,
                                                    Build B) {
  if (blockIdx.x == 0u) {
    uint32_t* w = reinterpret_cast<uint32_t*>(B.ctl);
    for (uint32_t k = threadIdx.x; k < (uint32_t)(sizeof(Ctl) / 4); k += 256u) w[k] = 0u;
    __syncthreads();
    if (threadIdx.x == 0u) {
      B.ctl->cnt[0] = 1u;
      BNode& r = B.nodes[0];
      for (int c = 0; c < 3; c++) r.mn[c] = r.mx[c] = 0.0f;
      r.first = 0u;
      r.count = n_tris;
      r.left = r.right = -1;
      r.leftdepth = 0u;
      r.gen = B.gen;
    }
  }
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
  const_cast<float4*>(B.tri_mn)[t] = make_float4(mn[0], mn[1], mn[2], 0.0f);
  const_cast<float4*>(B.tri_mx)[t] = make_float4(mx[0], mx[1], mx[2], 0.0f);
  const_cast<float4*>(B.tri_c)[t] = make_float4(ce[0], ce[1], ce[2], 0.0f);
  B.ord[0][t] = t;
  B.leaf_flag[t] = 0u;
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

// blas.rs:149-177 + 201-217 for one node from its bins: best split, left count, child order.  Called by ALL 64 lanes of one
// wave; lane i < 16 owns bin i.  The sweep's running unions are prefix (left) and suffix (right) scans over the bins — min /
// max on the key order do not depend on the order of the operands, so the scans give the bits of the sequential loops —
// and the 15 candidate costs are compared across lanes (strictly smaller wins, the lower bin on a tie, like `cost < best`
// walking upwards).  Results are returned in every lane.
struct SahOut {
  int leaf, split, rotate;
  uint32_t L;
  float lbox[6], rbox[6];
};
__device__ __forceinline__ SahOut sah_split_wave(const uint32_t* bin_cnt, const uint32_t (*bin_box)[6], uint32_t count) {
  const uint32_t lane = threadIdx.x & 63u;
  const float inf = __uint_as_float(0x7f800000u);
  const bool own = lane < (uint32_t)kBins;
  const uint32_t c0 = own ? bin_cnt[lane] : 0u;
  float bmn[3], bmx[3];
  for (int c = 0; c < 3; c++) {
    bmn[c] = (own && c0) ? float_of(bin_box[lane][c]) : inf;
    bmx[c] = (own && c0) ? float_of(bin_box[lane][c + 3]) : -inf;
  }
  // inclusive prefix over bins 0..lane (left) and inclusive suffix over bins lane..15 (right)
  float lmn[3] = {bmn[0], bmn[1], bmn[2]}, lmx[3] = {bmx[0], bmx[1], bmx[2]};
  float rmn[3] = {bmn[0], bmn[1], bmn[2]}, rmx[3] = {bmx[0], bmx[1], bmx[2]};
  uint32_t lc = c0, rc = c0;
  for (int off = 1; off < kBins; off <<= 1) {
    const uint32_t ulc = (uint32_t)__shfl_up((int)lc, off, 64), drc = (uint32_t)__shfl_down((int)rc, off, 64);
    float u_mn[3], u_mx[3], d_mn[3], d_mx[3];
    for (int c = 0; c < 3; c++) {
      u_mn[c] = __shfl_up(lmn[c], off, 64);
      u_mx[c] = __shfl_up(lmx[c], off, 64);
      d_mn[c] = __shfl_down(rmn[c], off, 64);
      d_mx[c] = __shfl_down(rmx[c], off, 64);
    }
    if (lane >= (uint32_t)off) {
      lc += ulc;
      for (int c = 0; c < 3; c++) { lmn[c] = tmin(lmn[c], u_mn[c]); lmx[c] = tmax(lmx[c], u_mx[c]); }
    }
    if (lane + (uint32_t)off < (uint32_t)kBins) {
      rc += drc;
      for (int c = 0; c < 3; c++) { rmn[c] = tmin(rmn[c], d_mn[c]); rmx[c] = tmax(rmx[c], d_mx[c]); }
    }
  }
  const float l_area = area_of(lmn, lmx), r_area = area_of(rmn, rmx);
  // candidate `lane`: left = bins <= lane, right = bins > lane (the suffix of lane + 1)
  const float r_area1 = __shfl_down(r_area, 1, 64);
  const uint32_t rc1 = (uint32_t)__shfl_down((int)rc, 1, 64);
  float r1mn[3], r1mx[3];
  for (int c = 0; c < 3; c++) {
    r1mn[c] = __shfl_down(rmn[c], 1, 64);
    r1mx[c] = __shfl_down(rmx[c], 1, 64);
  }
  float cost = inf;
  if (lane < (uint32_t)(kBins - 1) && lc != 0u && rc1 != 0u) {
    cost = l_area * (float)lc + r_area1 * (float)rc1;
    if (!(cost < inf)) cost = inf;   // NaN / inf never beat `best = inf`
  }
  float best = cost;
  uint32_t best_lane = lane;
  for (int off = 8; off > 0; off >>= 1) {   // lanes 0..15 hold candidates; 16..63 hold inf
    const float oc = __shfl_xor(best, off, 64);
    const uint32_t ol = (uint32_t)__shfl_xor((int)best_lane, off, 64);
    if (oc < best || (oc == best && ol < best_lane)) {
      best = oc;
      best_lane = ol;
    }
  }
  // lanes 0..15 agree; everybody reads lane 0
  best = __shfl(best, 0, 64);
  best_lane = (uint32_t)__shfl((int)best_lane, 0, 64);
  SahOut o;
  o.split = best < inf ? (int)best_lane : -1;
  o.leaf = o.split < 0;
  o.L = 0u;
  o.rotate = 0;
  for (int c = 0; c < 6; c++) { o.lbox[c] = 0.0f; o.rbox[c] = 0.0f; }
  const int src = o.leaf ? 0 : o.split;
  const uint32_t L = (uint32_t)__shfl((int)lc, src, 64);
  const float la = __shfl(l_area, src, 64), ra = __shfl(r_area1, src, 64);
  float lb[6], rb[6];
  for (int c = 0; c < 3; c++) {
    lb[c] = __shfl(lmn[c], src, 64);
    lb[c + 3] = __shfl(lmx[c], src, 64);
    rb[c] = __shfl(r1mn[c], src, 64);
    rb[c + 3] = __shfl(r1mx[c], src, 64);
  }
  if (!o.leaf) {
    o.L = L;
    if (L == 0u || L == count) o.leaf = 1;
    const float l_cost = la * (float)L, r_cost = ra * (float)(count - L);
    o.rotate = r_cost > l_cost;  // the costlier child goes first
    // the children's boxes are the partial unions of the sweep: a child holds exactly the triangles of bins <= split
    // (resp. > split)
    for (int c = 0; c < 6; c++) { o.lbox[c] = lb[c]; o.rbox[c] = rb[c]; }
  }
  return o;
}

// where a level's nodes are and where its children's ids start (both final once the previous level's kernels are done)
__device__ __forceinline__ void level_range(const Ctl* ctl, uint32_t level, uint32_t& id0, uint32_t& n_active, uint32_t& child_base) {
  id0 = ctl->base[level];
  n_active = ctl->cnt[level];
  child_base = id0 + n_active;
}
// two children of node `id` over [first, first + l_count) and the rest
__device__ __forceinline__ void make_children(const Build& B, uint32_t level, uint32_t child_base, uint32_t id, uint32_t first,
                                              uint32_t count, uint32_t l_count, const float* fbox, const float* sbox) {
  const uint32_t ids = child_base + atomicAdd(&B.ctl->cnt[level + 1u], 2u);
  const uint32_t ld = B.nodes[id].leftdepth;
  BNode& l = B.nodes[ids];
  BNode& r = B.nodes[ids + 1u];
  l.first = first;
  l.count = l_count;
  r.first = first + l_count;
  r.count = count - l_count;
  for (int c = 0; c < 3; c++) {
    l.mn[c] = fbox[c];
    l.mx[c] = fbox[c + 3];
    r.mn[c] = sbox[c];
    r.mx[c] = sbox[c + 3];
  }
  l.left = l.right = r.left = r.right = -1;
  l.leftdepth = ld + 1u;
  r.leftdepth = ld;
  l.gen = r.gen = B.gen;
  B.nodes[id].left = (int32_t)ids;
  B.nodes[id].right = (int32_t)(ids + 1u);
}

// A node of at most 64 triangles, finished by one wave without going back to the level loop: lane l holds the triangle at
// position first + l (id, box, centre: 10 registers); every node of the subtree runs the same five steps as k_level on those
// registers — bins by LDS atomics, the 16-lane sweep, the two-pointer partition as ballot ranks, the rotation — the
// triangles change lanes through LDS, pending nodes wait on a small LDS stack, and node records / leaf ranges are
// stored without anything being read back.  Ids come from a block of 2 count - 2 reserved with ONE atomic (the id
// space is not dense; records carry the build's stamp).  Replaces the deepest ~8 levels of launches, whose nodes each
// cost a chain of ten dependent global-memory round trips for a few dozen triangles.
struct SubShared {
  uint32_t bin_cnt[kBins];
  uint32_t bin_box[kBins][6];
  uint32_t tri[kSubtree][10];
  uint32_t stack[kSubtree + 2][9];   // rel | cnt << 8, id, leftdepth, box
  uint32_t pl[kSubtree], pr[kSubtree];
};
__device__ __forceinline__ void k_subtree(const Build& B, SubShared& S, uint32_t id, uint32_t first, uint32_t count, const uint32_t* order_in) {
  const uint32_t lane = threadIdx.x & 63u;
  const float inf = __uint_as_float(0x7f800000u);
  uint32_t t = 0;
  float mn[3] = {inf, inf, inf}, mx[3] = {-inf, -inf, -inf}, ce[3] = {0.0f, 0.0f, 0.0f};
  if (lane < count) {
    t = order_in[first + lane];
    const float4 a = B.tri_mn[t], b = B.tri_mx[t], c = B.tri_c[t];
    mn[0] = a.x; mn[1] = a.y; mn[2] = a.z;
    mx[0] = b.x; mx[1] = b.y; mx[2] = b.z;
    ce[0] = c.x; ce[1] = c.y; ce[2] = c.z;
  }
  float rbox[6];
  if (id == 0u) {   // the whole mesh: only the root reduces its triangles' boxes
    uint32_t k[6] = {key_of(mn[0]), key_of(mn[1]), key_of(mn[2]), key_of(mx[0]), key_of(mx[1]), key_of(mx[2])};
    if (lane >= count) { k[0] = k[1] = k[2] = 0xffffffffu; k[3] = k[4] = k[5] = 0u; }
    for (int off = 32; off > 0; off >>= 1)
      for (int c = 0; c < 6; c++) {
        const uint32_t o = (uint32_t)__shfl_xor((int)k[c], off, 64);
        k[c] = c < 3 ? min(k[c], o) : max(k[c], o);
      }
    for (int c = 0; c < 6; c++) rbox[c] = float_of(k[c]);
    if (lane == 0u)
      for (int c = 0; c < 3; c++) {
        B.nodes[0].mn[c] = rbox[c];
        B.nodes[0].mx[c] = rbox[c + 3];
      }
  } else {
    for (int c = 0; c < 3; c++) {
      rbox[c] = B.nodes[id].mn[c];
      rbox[c + 3] = B.nodes[id].mx[c];
    }
  }
  uint32_t blk = 0;
  if (lane == 0u) blk = atomicAdd(&B.ctl->n_sub, 2u * count - 2u);
  const uint32_t id_base = 2u * B.n_tris + (uint32_t)__shfl((int)blk, 0, 64);
  uint32_t next_local = 0u;
  if (lane == 0u) {
    S.stack[0][0] = 0u | (count << 8);
    S.stack[0][1] = id;
    S.stack[0][2] = B.nodes[id].leftdepth;
    for (int c = 0; c < 6; c++) S.stack[0][3 + c] = __float_as_uint(rbox[c]);
  }
  uint32_t sp = 1u;
  __syncthreads();
  while (sp > 0u) {
    sp--;
    const uint32_t e0 = S.stack[sp][0], nid = S.stack[sp][1], ld = S.stack[sp][2];
    float nb[6];
    for (int c = 0; c < 6; c++) nb[c] = __uint_as_float(S.stack[sp][3 + c]);
    const uint32_t rel0 = e0 & 255u, cnt = e0 >> 8;
    const bool in_range = lane >= rel0 && lane < rel0 + cnt;
    const float ex = nb[3] - nb[0], ey = nb[4] - nb[1], ez = nb[5] - nb[2];
    const int axis = ey > ex ? 1 : ((ez > ex && ez > ey) ? 2 : 0);   // blas.rs:127-133
    const float split_len = axis == 0 ? ex : (axis == 1 ? ey : ez);
    bool leaf = cnt <= 4u || split_len < 1e-6f;
    SahOut o;
    uint32_t bin = 0u;
    if (!leaf) {
      if (lane < (uint32_t)kBins) {
        S.bin_cnt[lane] = 0u;
        for (int c = 0; c < 3; c++) {
          S.bin_box[lane][c] = 0xffffffffu;
          S.bin_box[lane][c + 3] = 0u;
        }
      }
      __syncthreads();
      if (in_range) {
        bin = bin_of(axis == 0 ? ce[0] : (axis == 1 ? ce[1] : ce[2]), nb[axis], (float)kBins / split_len);
        atomicAdd(&S.bin_cnt[bin], 1u);
        for (int c = 0; c < 3; c++) {
          atomicMin(&S.bin_box[bin][c], key_of(mn[c]));
          atomicMax(&S.bin_box[bin][c + 3], key_of(mx[c]));
        }
      }
      __syncthreads();
      o = sah_split_wave(S.bin_cnt, S.bin_box, cnt);
      leaf = o.leaf != 0;
    }
    if (leaf) {   // blas.rs:111-115
      if (in_range) B.order_final[first + lane] = t;
      if (lane == 0u) {
        B.nodes[nid].left = -1;
        B.nodes[nid].right = -1;
        B.leaf_flag[first + rel0] = 1u;
      }
      __syncthreads();   // the stack entry and the bins were read by every lane before anything below rewrites them
      continue;
    }
    // the two-pointer partition: the k-th misplaced element from the left changes places with the k-th from the right
    const uint32_t L = o.L, R = cnt - L, rel = lane - rel0;
    const bool bad_l = in_range && rel < L && bin > (uint32_t)o.split;
    const bool bad_r = in_range && rel >= L && bin <= (uint32_t)o.split;
    const unsigned long long m_l = __ballot(bad_l), m_r = __ballot(bad_r);
    const unsigned long long below = lane ? (~0ull >> (64u - lane)) : 0ull, above = lane < 63u ? (~0ull << (lane + 1u)) : 0ull;
    const uint32_t rank_l = (uint32_t)__builtin_popcountll(m_l & below), rank_r = (uint32_t)__builtin_popcountll(m_r & above);
    if (bad_l) S.pl[rank_l] = lane;
    if (bad_r) S.pr[rank_r] = lane;
    __syncthreads();
    uint32_t pos = lane;
    if (bad_l) pos = S.pr[rank_l];
    if (bad_r) pos = S.pl[rank_r];
    const bool rotate = o.rotate != 0;
    if (in_range) {
      const uint32_t r0 = pos - rel0;
      pos = rel0 + (rotate ? (r0 >= L ? r0 - L : r0 + R) : r0);
    }
    // the triangles move to their new lanes
    S.tri[pos][0] = t;
    for (int c = 0; c < 3; c++) {
      S.tri[pos][1 + c] = __float_as_uint(mn[c]);
      S.tri[pos][4 + c] = __float_as_uint(mx[c]);
      S.tri[pos][7 + c] = __float_as_uint(ce[c]);
    }
    __syncthreads();
    t = S.tri[lane][0];
    for (int c = 0; c < 3; c++) {
      mn[c] = __uint_as_float(S.tri[lane][1 + c]);
      mx[c] = __uint_as_float(S.tri[lane][4 + c]);
      ce[c] = __uint_as_float(S.tri[lane][7 + c]);
    }
    // children: after a rotation the former right part is the first child
    const uint32_t l_count = rotate ? R : L;
    const float* fbox = rotate ? o.rbox : o.lbox;
    const float* sbox = rotate ? o.lbox : o.rbox;
    const uint32_t cid = id_base + next_local;
    next_local += 2u;
    if (lane == 0u) {
      BNode& l = B.nodes[cid];
      BNode& r = B.nodes[cid + 1u];
      l.first = first + rel0;
      l.count = l_count;
      r.first = first + rel0 + l_count;
      r.count = cnt - l_count;
      for (int c = 0; c < 3; c++) {
        l.mn[c] = fbox[c];
        l.mx[c] = fbox[c + 3];
        r.mn[c] = sbox[c];
        r.mx[c] = sbox[c + 3];
      }
      l.left = l.right = r.left = r.right = -1;
      l.leftdepth = ld + 1u;
      r.leftdepth = ld;
      l.gen = r.gen = B.gen;
      B.nodes[nid].left = (int32_t)cid;
      B.nodes[nid].right = (int32_t)(cid + 1u);
      S.stack[sp][0] = rel0 | (l_count << 8);
      S.stack[sp][1] = cid;
      S.stack[sp][2] = ld + 1u;
      S.stack[sp + 1u][0] = (rel0 + l_count) | ((cnt - l_count) << 8);
      S.stack[sp + 1u][1] = cid + 1u;
      S.stack[sp + 1u][2] = ld;
      for (int c = 0; c < 6; c++) {
        S.stack[sp][3 + c] = __float_as_uint(fbox[c]);
        S.stack[sp + 1u][3 + c] = __float_as_uint(sbox[c]);
      }
    }
    sp += 2u;
    __syncthreads();
  }
}

// One tree level: workgroup b takes nodes b, b + gridDim.x, ... of the level (IDS: of the level's small-node list, made by
// k_big_plan).  T = 256 for the levels of few, larger nodes, T = 64 (one wave per node) for the deep levels: tens of
// thousands of nodes of a few dozen triangles.  A node of any size is handled correctly (a large one slowly): the
// large-node kernels below are an optimisation the host schedules for the levels it expects such nodes on.
template <int T, bool IDS>
__global__ __launch_bounds__(T) void k_level(Build B, uint32_t level) {
  constexpr uint32_t kThreads = (uint32_t)T;
  __shared__ uint32_t s_red[T / 64][6];
  __shared__ uint32_t s_bin_cnt[kBins];
  __shared__ uint32_t s_bin_box[kBins][6];
  __shared__ uint32_t s_wave[T / 64];
  __shared__ float s_f[2];        // split_min, scale
  __shared__ float s_box[2][6];   // boxes of the left / right part
  __shared__ int32_t s_i[6];      // leaf flag, axis, best split, L, rotate
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  uint32_t id0, n_level, child_base;
  level_range(B.ctl, level, id0, n_level, child_base);
  if (blockIdx.x == 0u && tid == 0u) B.ctl->base[level + 1u] = child_base;
  const uint32_t n_active = IDS ? B.ctl->n_small : n_level;
  uint32_t* order_in = B.ord[level & 1u];
  uint32_t* order_out = B.ord[(level & 1u) ^ 1u];
  for (uint32_t blk = blockIdx.x; blk < n_active; blk += gridDim.x) {
    const uint32_t id = IDS ? B.small_ids[blk] : id0 + blk;
    const uint32_t first = B.nodes[id].first, count = B.nodes[id].count, end = first + count;
    if (!IDS && count > kBig && tid == 0u) atomicMax(&B.ctl->big_levels, level + 1u);
    if constexpr (T == 64) {
      __shared__ SubShared s_sub;
      if (count <= kSubtree) {   // the whole subtree, now
        k_subtree(B, s_sub, id, first, count, order_in);
        continue;
      }
    }

    // ---- 1. box of the range: the root reduces its triangles' boxes, every other node got its box from its parent's sweep
    if (id == 0u) {
      uint32_t k[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
      for (uint32_t p = first + tid; p < end; p += kThreads) {
        const uint32_t t = order_in[p];
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
      if (tid == 0u)
        for (int c = 0; c < 3; c++) {
          uint32_t kmn = s_red[0][c], kmx = s_red[0][c + 3];
          for (int w = 1; w < T / 64; w++) {
            kmn = min(kmn, s_red[w][c]);
            kmx = max(kmx, s_red[w][c + 3]);
          }
          B.nodes[id].mn[c] = float_of(kmn);
          B.nodes[id].mx[c] = float_of(kmx);
        }
    }
    if (tid == 0u) {
      float mn[3], mx[3];
      for (int c = 0; c < 3; c++) {
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
        const uint32_t t = order_in[p];
        const uint32_t b = bin_of(axis_of(B.tri_c[t], axis), split_min, scale);
        B.bin_cache[p] = (uint8_t)b;
        const float4 a = B.tri_mn[t], c = B.tri_mx[t];
        atomicAdd(&s_bin_cnt[b], 1u);
        atomicMin(&s_bin_box[b][0], key_of(a.x)); atomicMin(&s_bin_box[b][1], key_of(a.y)); atomicMin(&s_bin_box[b][2], key_of(a.z));
        atomicMax(&s_bin_box[b][3], key_of(c.x)); atomicMax(&s_bin_box[b][4], key_of(c.y)); atomicMax(&s_bin_box[b][5], key_of(c.z));
      }
      __syncthreads();
      // ---- 3. SAH sweep and split choice (blas.rs:149-177, 201-217): wave 0, bins on its first 16 lanes
      if (wave == 0u) {
        const SahOut o = sah_split_wave(s_bin_cnt, s_bin_box, count);
        if (lane == 0u) {
          for (int c = 0; c < 6; c++) {
            s_box[0][c] = o.lbox[c];
            s_box[1][c] = o.rbox[c];
          }
          s_i[0] = o.leaf;
          s_i[2] = o.split;
          s_i[3] = (int32_t)o.L;
          s_i[4] = o.rotate;
        }
      }
      __syncthreads();
    }
    if (s_i[0]) {  // leaf: blas.rs:111-115 (a count above 7 overflows the 3-bit field exactly like the reference)
      for (uint32_t p = first + tid; p < end; p += kThreads) B.order_final[p] = order_in[p];
      if (tid == 0u) {
        B.nodes[id].left = -1;
        B.nodes[id].right = -1;
        B.leaf_flag[first] = 1u;
      }
      __syncthreads();   // s_i is rewritten by the next node of this workgroup
      continue;
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
      const uint32_t ta = order_in[a], tb = order_in[b];
      order_in[a] = tb;
      order_in[b] = ta;
    }
    __syncthreads();
    // ---- 5. children
    for (uint32_t p = first + tid; p < end; p += kThreads) {
      const uint32_t rel = p - first;
      const uint32_t nrel = rotate ? (rel >= L ? rel - L : rel + R) : rel;
      order_out[first + nrel] = order_in[p];
    }
    if (tid == 0u)   // after a rotation the former right part is the first child
      make_children(B, level, child_base, id, first, count, rotate ? R : L, s_box[rotate ? 1 : 0], s_box[rotate ? 0 : 1]);
    __syncthreads();   // the shared arrays are rewritten by the next node of this workgroup
  }
}


// ------------------------------------------------------------------------------------------- large nodes
// A node with more than kBig triangles is worked on by many workgroups: its range is cut into chunks of kChunk
// positions and every step of k_level becomes its own launch over all chunks of all large nodes of the level
// (plan -> bounds -> setup -> bins -> split -> count -> scan -> scatter -> swap -> copy).  Same arithmetic, same result.
// The list of large nodes and their chunks is made on the device (k_big_plan); the launches are sized for the most
// chunks / nodes a level of the mesh can have and read the actual numbers from Ctl.
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
// most large nodes / chunks of one level of a mesh of n triangles
__host__ __device__ inline uint32_t big_cap(uint32_t n) { return n / kBig + 2u; }
__host__ __device__ inline uint32_t chunk_cap(uint32_t n) { return n / kChunk + n / kBig + 4u; }

// one workgroup: sort the level's nodes into large ones (BigNode records + their chunks) and the rest (small_ids)
__global__ __launch_bounds__(1024) void k_big_plan(Build B, uint32_t level, BigNode* bn, Chunk* chunks) {
  __shared__ uint32_t s_wave[16];
  __shared__ uint32_t s_scan[1024];
  __shared__ uint32_t s_carry[3];   // large nodes, chunks, small nodes so far
  uint32_t id0, n_active, child_base;
  level_range(B.ctl, level, id0, n_active, child_base);
  const uint32_t tid = threadIdx.x;
  if (tid < 3u) s_carry[tid] = 0u;
  __syncthreads();
  for (uint32_t i0 = 0; i0 < n_active; i0 += 1024u) {
    const uint32_t i = i0 + tid;
    const bool live = i < n_active;
    uint32_t first = 0, count = 0;
    if (live) {
      first = B.nodes[id0 + i].first;
      count = B.nodes[id0 + i].count;
    }
    const bool big = live && count > kBig;
    const uint32_t nch = big ? (count + kChunk - 1u) / kChunk : 0u;
    uint32_t n_big_here, n_small_here;
    const uint32_t r_big = block_rank<1024>(big, s_wave, n_big_here);
    const uint32_t r_small = block_rank<1024>(live && !big, s_wave, n_small_here);
    s_scan[tid] = nch;
    __syncthreads();
    for (uint32_t off = 1u; off < 1024u; off <<= 1) {
      const uint32_t w = tid >= off ? s_scan[tid - off] : 0u;
      __syncthreads();
      s_scan[tid] += w;
      __syncthreads();
    }
    const uint32_t ch_before = s_scan[tid] - nch, ch_here = s_scan[1023];
    const uint32_t c_big = s_carry[0], c_ch = s_carry[1], c_small = s_carry[2];
    if (big) {
      BigNode& N = bn[c_big + r_big];
      N.id = id0 + i;
      N.first = first;
      N.count = count;
      N.chunk0 = c_ch + ch_before;
      N.nchunks = nch;
      N.leaf = N.axis = N.split = N.rotate = 0;
      N.L = N.nbad = 0u;
      N.split_min = N.scale = 0.0f;
      for (int k = 0; k < 3; k++) {
        N.box[k] = 0xffffffffu;
        N.box[k + 3] = 0u;
      }
      for (int bi = 0; bi < kBins; bi++) {
        N.bin_cnt[bi] = 0u;
        for (int k = 0; k < 3; k++) {
          N.bin_box[bi][k] = 0xffffffffu;
          N.bin_box[bi][k + 3] = 0u;
        }
      }
      for (int k = 0; k < 6; k++) N.lbox[k] = N.rbox[k] = 0.0f;
    } else if (live) {
      B.small_ids[c_small + r_small] = id0 + i;
    }
    __syncthreads();
    if (tid == 0u) {
      s_carry[0] = c_big + n_big_here;
      s_carry[1] = c_ch + ch_here;
      s_carry[2] = c_small + n_small_here;
    }
    __syncthreads();
  }
  const uint32_t n_big = s_carry[0];
  for (uint32_t b = 0; b < n_big; b++) {   // the BigNode records were written by this workgroup: visible after the barrier
    const uint32_t c0 = bn[b].chunk0, nch = bn[b].nchunks;
    for (uint32_t j = tid; j < nch; j += 1024u) chunks[c0 + j] = Chunk{b, j};
  }
  if (tid == 0u) {
    B.ctl->n_big = n_big;
    B.ctl->n_chunks = s_carry[1];
    B.ctl->n_small = s_carry[2];
    if (n_big) atomicMax(&B.ctl->big_levels, level + 1u);
  }
}

__global__ __launch_bounds__(256) void k_big_bounds(Build B, uint32_t level, BigNode* bn, const Chunk* __restrict__ chunks) {
  __shared__ uint32_t s_red[4][6];
  if (blockIdx.x >= B.ctl->n_chunks) return;
  const uint32_t* order_in = B.ord[level & 1u];
  const Chunk ch = chunks[blockIdx.x];
  BigNode& N = bn[ch.big];
  const uint32_t lo = N.first + ch.j * kChunk, hi = min(lo + kChunk, N.first + N.count);
  uint32_t k[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
  for (uint32_t p = lo + threadIdx.x; p < hi; p += 256u) {
    const uint32_t t = order_in[p];
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

__global__ __launch_bounds__(64) void k_big_setup(Build B, BigNode* bn) {
  const uint32_t i = blockIdx.x * 64u + threadIdx.x;
  if (i >= B.ctl->n_big) return;
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

__global__ __launch_bounds__(256) void k_big_bin(Build B, uint32_t level, BigNode* bn, const Chunk* __restrict__ chunks) {
  __shared__ uint32_t s_cnt[kBins];
  __shared__ uint32_t s_box[kBins][6];
  if (blockIdx.x >= B.ctl->n_chunks) return;
  const uint32_t* order_in = B.ord[level & 1u];
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
    const uint32_t t = order_in[p];
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

// one wave per large node
__global__ __launch_bounds__(64) void k_big_split(Build B, BigNode* bn) {
  if (blockIdx.x >= B.ctl->n_big) return;
  BigNode& N = bn[blockIdx.x];
  if (N.leaf) return;
  const SahOut o = sah_split_wave(N.bin_cnt, N.bin_box, N.count);
  if (threadIdx.x == 0u) {
    N.leaf = o.leaf;
    N.split = o.split;
    N.L = o.L;
    N.rotate = o.rotate;
    for (int c = 0; c < 6; c++) {
      N.lbox[c] = o.lbox[c];
      N.rbox[c] = o.rbox[c];
    }
  }
}

// misplaced elements per chunk: left-region positions (< first + L) that belong right, right-region ones that belong left
__global__ __launch_bounds__(256) void k_big_count(Build B, const BigNode* __restrict__ bn, const Chunk* __restrict__ chunks,
                                                    uint32_t* __restrict__ chunk_cnt) {
  __shared__ uint32_t s_c[2];
  if (blockIdx.x >= B.ctl->n_chunks) return;
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
__global__ __launch_bounds__(64) void k_big_scan(Build B, BigNode* bn, const uint32_t* __restrict__ chunk_cnt,
                                                  uint32_t* __restrict__ chunk_base) {
  const uint32_t i = blockIdx.x * 64u + threadIdx.x;
  if (i >= B.ctl->n_big) return;
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
  if (blockIdx.x >= B.ctl->n_chunks) return;
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

__global__ __launch_bounds__(256) void k_big_swap(Build B, uint32_t level, const BigNode* __restrict__ bn, const Chunk* __restrict__ chunks) {
  if (blockIdx.x >= B.ctl->n_chunks) return;
  uint32_t* order_in = B.ord[level & 1u];
  const Chunk ch = chunks[blockIdx.x];
  const BigNode& N = bn[ch.big];
  if (N.leaf) return;
  const uint32_t hi = min((ch.j + 1u) * kChunk, N.nbad);
  for (uint32_t q = ch.j * kChunk + threadIdx.x; q < hi; q += 256u) {
    const uint32_t a = B.scratch_l[N.first + q], b = B.scratch_r[N.first + q];
    const uint32_t ta = order_in[a], tb = order_in[b];
    order_in[a] = tb;
    order_in[b] = ta;
  }
}

__global__ __launch_bounds__(256) void k_big_copy(Build B, uint32_t level, const BigNode* __restrict__ bn, const Chunk* __restrict__ chunks) {
  if (blockIdx.x >= B.ctl->n_chunks) return;
  const uint32_t* order_in = B.ord[level & 1u];
  uint32_t* order_out = B.ord[(level & 1u) ^ 1u];
  const Chunk ch = chunks[blockIdx.x];
  const BigNode& N = bn[ch.big];
  const uint32_t lo = N.first + ch.j * kChunk, hi = min(lo + kChunk, N.first + N.count);
  if (N.leaf) {
    for (uint32_t p = lo + threadIdx.x; p < hi; p += 256u) B.order_final[p] = order_in[p];
    if (ch.j == 0u && threadIdx.x == 0u) {
      B.nodes[N.id].left = -1;
      B.nodes[N.id].right = -1;
      B.leaf_flag[N.first] = 1u;
    }
    return;
  }
  const uint32_t L = N.L, R = N.count - N.L;
  const bool rotate = N.rotate != 0;
  for (uint32_t p = lo + threadIdx.x; p < hi; p += 256u) {
    const uint32_t rel = p - N.first;
    const uint32_t nrel = rotate ? (rel >= L ? rel - L : rel + R) : rel;
    order_out[N.first + nrel] = order_in[p];
  }
  if (ch.j == 0u && threadIdx.x == 0u) {
    uint32_t id0, n_active, child_base;
    level_range(B.ctl, level, id0, n_active, child_base);
    make_children(B, level, child_base, N.id, N.first, N.count, rotate ? R : L, rotate ? N.rbox : N.lbox, rotate ? N.lbox : N.rbox);
  }
}

// ------------------------------------------------------------------------------------------- pre-order layout
// LB = exclusive prefix sum of leaf_flag over the n positions (+ the total at index n), in three launches
__global__ __launch_bounds__(1024) void k_scan_blocks(const uint32_t* __restrict__ flag, uint32_t n, uint32_t* __restrict__ blk) {
  __shared__ uint32_t s_cnt;
  if (threadIdx.x == 0u) s_cnt = 0u;
  __syncthreads();
  const uint32_t i = blockIdx.x * 1024u + threadIdx.x;
  const unsigned long long m = __ballot(i < n && flag[i] != 0u);
  if ((threadIdx.x & 63u) == 0u && m) atomicAdd(&s_cnt, (uint32_t)__builtin_popcountll(m));
  __syncthreads();
  if (threadIdx.x == 0u) blk[blockIdx.x] = s_cnt;
}
// one workgroup: exclusive scan of the block counts in place; the totals of the build; where the NEXT mesh's nodes start
// in a shared node array (node_base[0] = this mesh's first node, node_base[1] is written)
__global__ __launch_bounds__(1024) void k_scan_top(uint32_t* __restrict__ blk, uint32_t n_blocks, Ctl* ctl, uint32_t* node_base) {
  __shared__ uint32_t s_scan[1024];
  __shared__ uint32_t s_carry;
  if (threadIdx.x == 0u) s_carry = 0u;
  __syncthreads();
  for (uint32_t c0 = 0u; c0 < n_blocks; c0 += 1024u) {
    const uint32_t i = c0 + threadIdx.x;
    const uint32_t v = i < n_blocks ? blk[i] : 0u;
    s_scan[threadIdx.x] = v;
    __syncthreads();
    for (uint32_t off = 1u; off < 1024u; off <<= 1) {
      const uint32_t w = threadIdx.x >= off ? s_scan[threadIdx.x - off] : 0u;
      __syncthreads();
      s_scan[threadIdx.x] += w;
      __syncthreads();
    }
    if (i < n_blocks) blk[i] = s_carry + s_scan[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023u) s_carry += s_scan[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0u && ctl) {   // ctl == NULL: a plain scan of flags (world_update.hip.h's emissive list)
    const uint32_t leaves = s_carry;
    ctl->n_leaves = leaves;
    ctl->n_nodes = leaves ? 2u * leaves - 1u : 0u;
    if (node_base) node_base[1] = node_base[0] + ctl->n_nodes;
  }
}
__global__ __launch_bounds__(1024) void k_scan_apply(const uint32_t* __restrict__ flag, uint32_t n, const uint32_t* __restrict__ blk,
                                                      const Ctl* __restrict__ ctl, uint32_t* __restrict__ lb) {
  __shared__ uint32_t s_wave[16];
  const uint32_t i = blockIdx.x * 1024u + threadIdx.x;
  const bool f = i < n && flag[i] != 0u;
  const unsigned long long m = __ballot(f);
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
  if (lane == 0u) s_wave[wave] = (uint32_t)__builtin_popcountll(m);
  __syncthreads();
  uint32_t before = 0u;
  for (uint32_t w = 0; w < wave; w++) before += s_wave[w];
  if (i < n) lb[i] = blk[blockIdx.x] + before + rank;
  if (i == 0u) lb[n] = ctl->n_leaves;
}
// The node array the traversal reads: {min, skip} {max, data}, skip = index after the subtree (BLAS-local).  Written at
// out + 2 * node_base[0] (node_base == NULL: at out); a leaf's `first` is made an index into the world's topology rows
// the way rebuilder.rs:123-134 does it: ((data >> 3) + topo_start) << 3 | (data & 7).
__global__ __launch_bounds__(256) void k_emit(const BNode* __restrict__ nodes, uint32_t n_ids, uint32_t gen, const Ctl* __restrict__ ctl,
                                              const uint32_t* __restrict__ lb, const uint32_t* __restrict__ node_base, uint32_t topo_start,
                                              float4* __restrict__ out) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n_ids) return;
  const BNode nd = nodes[i];
  if (nd.gen != gen) return;   // not a node of this build
  const uint32_t l0 = lb[nd.first], l1 = lb[nd.first + nd.count];
  const uint32_t pre = 2u * l0 + nd.leftdepth, size = 2u * (l1 - l0) - 1u;
  const uint32_t skip = pre + size;
  uint32_t data = nd.left < 0 ? ((nd.first << 3) | nd.count) : 0u;
  if (data != 0u) data = (((data >> 3) + topo_start) << 3) | (data & 7u);
  float4* o = out + 2 * ((size_t)(node_base ? node_base[0] : 0u) + pre);
  o[0] = make_float4(nd.mn[0], nd.mn[1], nd.mn[2], __uint_as_float(skip));
  o[1] = make_float4(nd.mx[0], nd.mx[1], nd.mx[2], __uint_as_float(data));
}

}  // namespace bvhb
#endif
