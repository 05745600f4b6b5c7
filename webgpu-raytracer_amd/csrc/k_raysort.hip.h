// k_raysort.hip.h — binning of the wavefront form's ray queues for coherence (counting sort, device-sized).
// Part of the kernel set of csrc/kernels.hip.h (included from there, in order; not a stand-alone header).
//
// After the first bounce the rays of a queue are incoherent: the 64 rays of a wave start all over the scene and head in
// all directions, so every node fetch of the wave touches 64 different cache lines and the vector L1 (32 KB per CU for
// ~1 500 rays in flight) misses a third of the time.  Before each trace stage the queue is therefore binned by
//   key = Morton code of the ray origin in a 16 x 16 x 16 grid over the scene box (12 bits) << 3 | direction octant
// with a counting sort: rays of one cell and octant — which walk the same part of the tree — become neighbours in the
// queue, whatever path they belong to.  Only the ORDER in which rays are pulled changes; each ray's walk, its result and
// every counter are unchanged (the trace kernel addresses results by path id), so the image stays bit-identical.
//
// Three launches per queue, sized on the device (no host read-back):
//   k_sort_hist     one 1024-thread workgroup per CU counts its contiguous slice of the queue into a 32 768-bin
//                   histogram held in LDS (128 KB of the CU's 160 KB), then adds the non-empty bins to the global one
//   k_sort_scan     one workgroup: exclusive scan of the global histogram -> bin offsets, cursors, total
//   k_sort_scatter  each workgroup recounts its slice, reserves its range of every bin with ONE atomic per (workgroup,
//                   bin) and writes its rays there (LDS cursors); within a bin, workgroups land in arrival order
#ifndef MI355RT_K_RAYSORT_HIP_H
#define MI355RT_K_RAYSORT_HIP_H

namespace rtk {

#define RT_WF_INVALID 0xffffffffu
#define RT_SORT_BINS 32768u
#define RT_SORT_BLOCK 1024

struct SortArgs {
  const uint32_t* ids;      // the queue: path ids, RT_WF_INVALID in the padding of a chunk
  const uint32_t* keys;     // one key per queue slot (written by k_wf_shade next to the id)
  uint32_t* sorted;         // out: queue slots (write_ids == 0) or the path ids themselves (write_ids == 1)
  const uint32_t* n_slots;  // device: slots in use (chunk-padded)
  uint32_t* hist;           // RT_SORT_BINS + 1 words, zero before k_sort_hist; after the scan hist[RT_SORT_BINS] = rays
  uint32_t* cursor;         // RT_SORT_BINS words
  uint32_t write_ids;
};

__device__ __forceinline__ uint32_t sort_part3(uint32_t v) {  // 4 bits -> every third bit
  v = (v | (v << 4)) & 0x0c3u;
  v = (v | (v << 2)) & 0x249u;
  return v;
}
// scene box: lo = box minimum, scale = 16 / extent per axis (0 on a flat axis)
__device__ __forceinline__ uint32_t ray_sort_key(rt3 lo, rt3 scale, rt3 o, rt3 d) {
  const float fx = (o.x - lo.x) * scale.x, fy = (o.y - lo.y) * scale.y, fz = (o.z - lo.z) * scale.z;
  const uint32_t cx = (uint32_t)rt_min(rt_max(fx, 0.0f), 15.0f), cy = (uint32_t)rt_min(rt_max(fy, 0.0f), 15.0f),
                 cz = (uint32_t)rt_min(rt_max(fz, 0.0f), 15.0f);
  const uint32_t morton = sort_part3(cx) | (sort_part3(cy) << 1) | (sort_part3(cz) << 2);
  const uint32_t oct = (d.x < 0.0f ? 1u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 4u : 0u);
  return (morton << 3) | oct;
}

__device__ __forceinline__ void sort_slice(const SortArgs& A, uint32_t& begin, uint32_t& end) {
  const uint32_t n = *A.n_slots;
  const uint32_t per = ((n + gridDim.x - 1u) / gridDim.x + 63u) & ~63u;
  begin = blockIdx.x * per;
  end = begin + per < n ? begin + per : n;
  if (begin > n) begin = n;
}

__global__ __launch_bounds__(RT_SORT_BLOCK) void k_sort_hist(SortArgs A) {
  extern __shared__ uint32_t s_bins[];
  for (uint32_t b = threadIdx.x; b < RT_SORT_BINS; b += RT_SORT_BLOCK) s_bins[b] = 0u;
  __syncthreads();
  uint32_t begin, end;
  sort_slice(A, begin, end);
  for (uint32_t i = begin + threadIdx.x; i < end; i += RT_SORT_BLOCK)
    if (A.ids[i] != RT_WF_INVALID) atomicAdd(&s_bins[A.keys[i] & (RT_SORT_BINS - 1u)], 1u);
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < RT_SORT_BINS; b += RT_SORT_BLOCK) {
    const uint32_t v = s_bins[b];
    if (v) atomicAdd(&A.hist[b], v);
  }
}

__global__ __launch_bounds__(RT_SORT_BLOCK) void k_sort_scan(SortArgs A) {
  __shared__ uint32_t s_part[RT_SORT_BLOCK];
  constexpr uint32_t PER = RT_SORT_BINS / RT_SORT_BLOCK;  // 32 consecutive bins per thread
  const uint32_t t = threadIdx.x;
  uint32_t local[PER];
  uint32_t sum = 0u;
#pragma unroll
  for (uint32_t k = 0; k < PER; k++) {
    local[k] = A.hist[t * PER + k];
    sum += local[k];
  }
  s_part[t] = sum;
  __syncthreads();
  for (uint32_t off = 1u; off < RT_SORT_BLOCK; off <<= 1) {
    const uint32_t v = t >= off ? s_part[t - off] : 0u;
    __syncthreads();
    s_part[t] += v;
    __syncthreads();
  }
  uint32_t run = s_part[t] - sum;  // exclusive prefix of this thread's 32 bins
#pragma unroll
  for (uint32_t k = 0; k < PER; k++) {
    A.hist[t * PER + k] = run;
    A.cursor[t * PER + k] = run;
    run += local[k];
  }
  if (t == RT_SORT_BLOCK - 1u) A.hist[RT_SORT_BINS] = run;  // number of rays in the queue
}

__global__ __launch_bounds__(RT_SORT_BLOCK) void k_sort_scatter(SortArgs A) {
  extern __shared__ uint32_t s_bins[];
  for (uint32_t b = threadIdx.x; b < RT_SORT_BINS; b += RT_SORT_BLOCK) s_bins[b] = 0u;
  __syncthreads();
  uint32_t begin, end;
  sort_slice(A, begin, end);
  for (uint32_t i = begin + threadIdx.x; i < end; i += RT_SORT_BLOCK)
    if (A.ids[i] != RT_WF_INVALID) atomicAdd(&s_bins[A.keys[i] & (RT_SORT_BINS - 1u)], 1u);
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < RT_SORT_BINS; b += RT_SORT_BLOCK) {
    const uint32_t v = s_bins[b];
    if (v) s_bins[b] = atomicAdd(&A.cursor[b], v);  // this workgroup's range of the bin starts here
  }
  __syncthreads();
  for (uint32_t i = begin + threadIdx.x; i < end; i += RT_SORT_BLOCK) {
    const uint32_t id = A.ids[i];
    if (id != RT_WF_INVALID) {
      const uint32_t pos = atomicAdd(&s_bins[A.keys[i] & (RT_SORT_BINS - 1u)], 1u);
      A.sorted[pos] = A.write_ids ? id : i;
    }
  }
}

}  // namespace rtk
#endif
