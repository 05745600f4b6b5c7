// gather_peak.hip — what one MI355X sustains on DEPENDENT, lane-divergent 16-byte gathers: the access pattern of a BVH
// node step (every lane of a wave reads a different 32-byte record, and the next address depends on what came back).
//
// Each lane chases a random permutation through a table of records: `loads` x global_load_dwordx4 per step from one
// record (1 = 16 B of a 32-B record, 2 = the 32-B node of k_traverse.hip.h, 4 = a 64-byte-aligned 64-B PAIR of nodes:
// the table is then a permutation of n/2 64-byte records, so a 4-load walk wanders over the whole table like the others
// — round 2's 4-load rows followed the successor of the even 32-B record only and collapsed into a short cached cycle),
// `lanes` lanes of every wave active, `blocks` 256-thread workgroups per CU.  "4q" (registers) and "4d" (LDS-DMA) = the same 64-byte records fetched
// QUAD-COOPERATIVELY: four lanes read the four 16-byte chunks of ONE ray's record in one instruction (four instructions
// serve the 64 rays of a wave; each instruction touches 16 lines instead of 64), the successor word handed to the
// owning lane by a DPP quad broadcast.  Table sizes walk the hierarchy: 16 KB (vector L1), 1 MB (L2), 4 MB
// (the node array of the 263 k-triangle scene, per-XCD L2 = 4 MB), 64 MB (Infinity Cache).  Printed per run: lane-steps
// per second chip-wide, and CU cycles per wave-level load instruction = clock x elapsed / (waves per CU x steps x
// loads) — the figure to hold against "the vector L1 looks up one line per cycle".
// A second table does the same walk with the records in LDS (ds_read_b128), the treelet's access pattern.
//
// build: hipcc --offload-arch=gfx950 -O3 -o gather_peak tools/gather_peak.hip      run: ./gather_peak
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                     \
      return 1;                                                                   \
    }                                                                             \
  } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef const f4 __attribute__((address_space(1))) * gptr;

// record r = 2 x f4 (LOADS 1, 2) or 4 x f4 (LOADS 4); .w of the first f4 holds the next record's index (as bits)
template <int LOADS>
__global__ __launch_bounds__(256) void k_chase(const f4* tab, uint32_t n_rec, uint32_t lanes, int iters, float* out,
                                               unsigned long long* stamps) {
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u % n_rec;
  float acc = 0.0f;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  if (lane < lanes) {
    for (int it = 0; it < iters; it++) {
      const size_t base = (LOADS == 4 ? 4 : 2) * (size_t)idx;
      f4 a = ((gptr)tab)[base];
      if (LOADS >= 2) {
        f4 b = ((gptr)tab)[base + 1];
        acc += b.x;
      }
      if (LOADS == 4) {
        f4 c = ((gptr)tab)[base + 2], d = ((gptr)tab)[base + 3];
        acc += c.y + d.z;
      }
      acc += a.x;
      idx = __float_as_uint(a.w);
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 256 + threadIdx.x] = acc + (float)idx;
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = c1 - c0;
    stamps[2 * blockIdx.x + 1] = r1 - r0;
  }
}

// 64-byte records, quad-cooperative: instruction k of a step serves the rays 4q + k (q = quad); lane 4q + c reads chunk c.
// Ray r is active iff r < lanes.  Every lane keeps its own ray's cursor; the cursor of ray 4q + k reaches the quad by a
// DPP quad broadcast, the successor word (chunk 0, .w) goes back to the owner the same way.
template <int QP>
__device__ __forceinline__ uint32_t quad_bcast(uint32_t v) {   // value of lane QP of every quad, in all four of its lanes
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, QP * 0x55, 0xf, 0xf, true);   // quad_perm:[QP,QP,QP,QP]
}
__global__ __launch_bounds__(256) void k_chase_quad(const f4* tab, uint32_t n_rec, uint32_t lanes, int iters, float* out,
                                                    unsigned long long* stamps) {
  const uint32_t lane = threadIdx.x & 63u, c = lane & 3u, qbase = lane & ~3u;
  uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u % n_rec;
  float acc = 0.0f;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it++) {
    const uint32_t i0 = quad_bcast<0>(idx), i1 = quad_bcast<1>(idx), i2 = quad_bcast<2>(idx), i3 = quad_bcast<3>(idx);
    f4 v0 = {0, 0, 0, 0}, v1 = v0, v2 = v0, v3 = v0;
    if (qbase + 0u < lanes) v0 = ((gptr)tab)[4 * (size_t)i0 + c];
    if (qbase + 1u < lanes) v1 = ((gptr)tab)[4 * (size_t)i1 + c];
    if (qbase + 2u < lanes) v2 = ((gptr)tab)[4 * (size_t)i2 + c];
    if (qbase + 3u < lanes) v3 = ((gptr)tab)[4 * (size_t)i3 + c];
    // successor of ray 4q + k = chunk 0 (.w) of v_k, held by lane 4q + 0
    const uint32_t n0 = quad_bcast<0>(__float_as_uint(v0.w)), n1 = quad_bcast<0>(__float_as_uint(v1.w));
    const uint32_t n2 = quad_bcast<0>(__float_as_uint(v2.w)), n3 = quad_bcast<0>(__float_as_uint(v3.w));
    const uint32_t nx = c == 0u ? n0 : (c == 1u ? n1 : (c == 2u ? n2 : n3));
    if (lane < lanes) idx = nx;
    acc += v0.x + v1.y + v2.z + v3.x;
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 256 + threadIdx.x] = acc + (float)idx;
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = c1 - c0;
    stamps[2 * blockIdx.x + 1] = r1 - r0;
  }
}

// "4d": the quad-cooperative fetch with the data landing in LDS (global_load_lds_dwordx4, per-lane source address): instruction
// k writes the 64 rays' ... no: the 16 records of the rays 4q + k as one contiguous KB at its own LDS region (M0), quad q's
// four lanes supplying the four chunks of record q; the owner then reads its record back (here: chunk 0 for the successor,
// the other three summed so that they are not dead).  No register shuffling at all — what the trace kernels use.
typedef __attribute__((address_space(1))) const void* gvptr;
typedef __attribute__((address_space(3))) void* lvptr;
__global__ __launch_bounds__(256) void k_chase_quad_dma(const f4* tab, uint32_t n_rec, uint32_t lanes, int iters, float* out,
                                                        unsigned long long* stamps) {
  extern __shared__ f4 dma_lds[];   // per wave 4 regions of 1 KB + 16 B of padding each (bank spread for the read-back)
  const uint32_t lane = threadIdx.x & 63u, c = lane & 3u, qbase = lane & ~3u, wave = threadIdx.x >> 6;
  f4* wl = dma_lds + wave * 4u * 65u;
  uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u % n_rec;
  float acc = 0.0f;
  const f4* mine = wl + c * 65u + (lane >> 2) * 4u;   // the owner's record: region (lane & 3), record (lane >> 2)
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it++) {
    const uint32_t i0 = quad_bcast<0>(idx), i1 = quad_bcast<1>(idx), i2 = quad_bcast<2>(idx), i3 = quad_bcast<3>(idx);
    if (qbase + 0u < lanes) __builtin_amdgcn_global_load_lds((gvptr)(tab + 4 * (size_t)i0 + c), (lvptr)(wl + 0u * 65u), 16, 0, 0);
    if (qbase + 1u < lanes) __builtin_amdgcn_global_load_lds((gvptr)(tab + 4 * (size_t)i1 + c), (lvptr)(wl + 1u * 65u), 16, 0, 0);
    if (qbase + 2u < lanes) __builtin_amdgcn_global_load_lds((gvptr)(tab + 4 * (size_t)i2 + c), (lvptr)(wl + 2u * 65u), 16, 0, 0);
    if (qbase + 3u < lanes) __builtin_amdgcn_global_load_lds((gvptr)(tab + 4 * (size_t)i3 + c), (lvptr)(wl + 3u * 65u), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane < lanes) {
      const f4 a = mine[0], b = mine[1], cc = mine[2], d = mine[3];
      acc += a.x + b.y + cc.z + d.x;
      idx = __float_as_uint(a.w);
    }
    __builtin_amdgcn_wave_barrier();
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 256 + threadIdx.x] = acc + (float)idx;
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = c1 - c0;
    stamps[2 * blockIdx.x + 1] = r1 - r0;
  }
}

template <int LOADS>
__global__ __launch_bounds__(256) void k_chase_lds(const f4* tab, uint32_t n_rec, uint32_t lanes, int iters, float* out) {
  extern __shared__ f4 lds[];
  for (uint32_t i = threadIdx.x; i < 2u * n_rec; i += 256u) lds[i] = tab[i];
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u % n_rec;
  float acc = 0.0f;
  if (lane < lanes) {
    for (int it = 0; it < iters; it++) {
      f4 a = lds[2u * idx];
      if (LOADS >= 2) {
        f4 b = lds[2u * idx + 1u];
        acc += b.x;
      }
      acc += a.x;
      idx = __float_as_uint(a.w);
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc + (float)idx;
}

int main() {
  CHECK(hipSetDevice(0));
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("device: %s, %d CUs\n", prop.gcnArchName, cus);
  const uint32_t sizes[] = {512u, 32768u, 131072u, 2097152u};   // records of 32 B: 16 KB, 1 MB, 4 MB, 64 MB
  const uint32_t max_rec = 2097152u;
  f4* tab;
  float* out;
  unsigned long long* stamps;
  CHECK(hipMalloc(&tab, (size_t)max_rec * 32));
  CHECK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
  CHECK(hipMalloc(&stamps, (size_t)cus * 8 * 16));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  std::vector<f4> host((size_t)max_rec * 2);
  std::vector<unsigned long long> hs((size_t)cus * 8 * 2);
  printf("%-8s %-6s %-6s %-7s %14s %16s %12s %10s\n", "table", "loads", "lanes", "blk/CU", "Glane-steps/s", "CUcyc/wave-load", "CUcyc/step", "clock GHz");
  for (uint32_t n_rec : sizes) {
    // one random cycle through all records (Sattolo), so every chain keeps wandering over the whole table.  layout 0:
    // n_rec records of 32 B; layout 1: n_rec / 2 records of 64 B (the successor in chunk 0) — the same bytes of table
    for (int layout = 0; layout < 2; layout++) {
      const uint32_t n_walk = layout == 0 ? n_rec : n_rec / 2;
      std::vector<uint32_t> perm(n_walk);
      for (uint32_t i = 0; i < n_walk; i++) perm[i] = i;
      uint64_t s = 88172645463325252ull;
      for (uint32_t i = n_walk - 1; i > 0; i--) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        uint32_t j = (uint32_t)(s % i);
        std::swap(perm[i], perm[j]);
      }
      const uint32_t per = layout == 0 ? 2u : 4u;
      for (uint32_t i = 0; i < n_walk; i++) {
        for (uint32_t k = 0; k < per; k++) {
          f4 a = {1e-9f, 0.f, 0.f, 0.f};
          host[(size_t)per * i + k] = a;
        }
        uint32_t nx = perm[i];
        float fw;
        memcpy(&fw, &nx, 4);
        host[(size_t)per * i].w = fw;
      }
      CHECK(hipMemcpy(tab, host.data(), (size_t)n_rec * 32, hipMemcpyHostToDevice));
      for (int loads : {1, 2, 4, 5, 6}) {   // 5 = "4q": the 64-byte records fetched quad-cooperatively; 6 = "4d": the same into LDS
        if ((layout == 0) != (loads <= 2)) continue;
        for (uint32_t lanes : {64u, 40u, 16u}) {
          for (int blocks : {2, 4, 6, 8}) {
            if (lanes != 40u && blocks != 6) continue;
            const int iters = n_rec <= 512u ? 20000 : 4000;
            const int grid = cus * blocks;
            float ms = 0.f;
            for (int rep = 0; rep < 2; rep++) {
              CHECK(hipEventRecord(e0));
              if (loads == 1) hipLaunchKernelGGL(k_chase<1>, dim3(grid), dim3(256), 0, 0, tab, n_walk, lanes, iters, out, stamps);
              if (loads == 2) hipLaunchKernelGGL(k_chase<2>, dim3(grid), dim3(256), 0, 0, tab, n_walk, lanes, iters, out, stamps);
              if (loads == 4) hipLaunchKernelGGL(k_chase<4>, dim3(grid), dim3(256), 0, 0, tab, n_walk, lanes, iters, out, stamps);
              if (loads == 5) hipLaunchKernelGGL(k_chase_quad, dim3(grid), dim3(256), 0, 0, tab, n_walk, lanes, iters, out, stamps);
              if (loads == 6) hipLaunchKernelGGL(k_chase_quad_dma, dim3(grid), dim3(256), 4 * 4 * 65 * 16, 0, tab, n_walk, lanes, iters, out, stamps);
              CHECK(hipEventRecord(e1));
              CHECK(hipEventSynchronize(e1));
              CHECK(hipEventElapsedTime(&ms, e0, e1));
            }
            CHECK(hipMemcpy(hs.data(), stamps, (size_t)grid * 16, hipMemcpyDeviceToHost));
            double clk = 0;
            for (int b = 0; b < grid; b++) clk += (double)hs[2 * b] / (double)hs[2 * b + 1] * 0.1;
            clk /= grid;
            const double steps = (double)grid * 4 * lanes * iters;
            const int instr = loads >= 5 ? 4 : loads;   // wave-level load instructions per step
            const double cyc_step = clk * 1e9 * ms * 1e-3 / ((double)blocks * 4 * iters);
            printf("%-8s %-6s %-6u %-7d %14.1f %16.1f %12.1f %10.2f\n",
                   n_rec == 512u ? "16KB" : n_rec == 32768u ? "1MB" : n_rec == 131072u ? "4MB" : "64MB",
                   loads == 6 ? "4d" : (loads == 5 ? "4q" : (loads == 4 ? "4" : (loads == 2 ? "2" : "1"))), lanes, blocks,
                   steps / (ms * 1e-3) * 1e-9, cyc_step / instr, cyc_step, clk);
            fflush(stdout);
          }
        }
      }
    }
  }
  // the same walk with the table in LDS (treelet pattern): 2048 records = 64 KB per workgroup, 2 workgroups per CU
  {
    const uint32_t n_rec = 2048u;
    CHECK(hipFuncSetAttribute((const void*)k_chase_lds<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CHECK(hipFuncSetAttribute((const void*)k_chase_lds<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CHECK(hipMemcpy(tab, host.data(), 0, hipMemcpyHostToDevice));
    std::vector<uint32_t> perm(n_rec);
    for (uint32_t i = 0; i < n_rec; i++) perm[i] = (i * 1103515245u + 12345u) % n_rec;
    for (uint32_t i = 0; i < n_rec; i++) {
      f4 a = {1e-9f, 0.f, 0.f, 0.f};
      uint32_t nx = (i * 677u + 13u) % n_rec;   // 677 is odd: a permutation of 0..2047 with long cycles
      { float fw; memcpy(&fw, &nx, 4); a.w = fw; }
      host[2 * (size_t)i] = a;
      host[2 * (size_t)i + 1] = a;
    }
    CHECK(hipMemcpy(tab, host.data(), (size_t)n_rec * 32, hipMemcpyHostToDevice));
    for (int loads : {1, 2}) {
      for (uint32_t lanes : {64u, 40u}) {
        const int iters = 20000, blocks = 2, grid = cus * blocks;
        float ms = 0.f;
        for (int rep = 0; rep < 2; rep++) {
          CHECK(hipEventRecord(e0));
          if (loads == 1) hipLaunchKernelGGL(k_chase_lds<1>, dim3(grid), dim3(256), 65536, 0, tab, n_rec, lanes, iters, out);
          if (loads == 2) hipLaunchKernelGGL(k_chase_lds<2>, dim3(grid), dim3(256), 65536, 0, tab, n_rec, lanes, iters, out);
          CHECK(hipEventRecord(e1));
          CHECK(hipEventSynchronize(e1));
          CHECK(hipEventElapsedTime(&ms, e0, e1));
        }
        const double steps = (double)grid * 4 * lanes * iters;
        printf("%-8s %-6d %-6u %-7d %14.1f %16.1f (at 2.4 GHz)\n", "LDS64KB", loads, lanes, blocks, steps / (ms * 1e-3) * 1e-9,
               2.4e9 * ms * 1e-3 / ((double)blocks * 4 * iters * loads));
      }
    }
  }
  return 0;
}
