// k_validate.hip.h — every index the kernels follow, checked once per scene upload.
// WebGPU gives the reference robust buffer access: an index out of range reads zeros or a clamped element and a skip
// pointer that goes backwards spins a shader until the browser's watchdog ends it.  A HIP kernel that follows such an
// index faults (or never drains), and a fault can reset the GPUs of the whole host — so a malformed upload is refused
// with an error instead.  Checked: vertex ids of every triangle; TLAS skips (forward, inside the TLAS) and instance
// ids; every reachable BLAS node (skip forward and inside its BLAS, an internal node's first child inside it, a leaf's
// triangle range inside the topology); instance BLAS offsets; light references.
#ifndef MI355RT_K_VALIDATE_HIP_H
#define MI355RT_K_VALIDATE_HIP_H

namespace rtk {

struct ValidateArgs {
  const float4* topo;     // 5 per triangle
  const float4* nodes;    // 2 per node
  const float4* inst;     // 9 per instance
  const uint2* lights;
  const uint32_t* roots;  // sorted unique BLAS-local root offsets of the instances
  uint32_t n_tris, n_verts, n_nodes, n_tlas, n_inst, n_lights, n_roots;
  uint32_t* bad;          // [5]: triangles, TLAS nodes, BLAS nodes, instances, lights
};

__global__ __launch_bounds__(256) void k_validate_scene(ValidateArgs A) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < A.n_tris) {
    const float4 t = A.topo[5 * (size_t)i];
    if (rt_f2u(t.x) >= A.n_verts || rt_f2u(t.y) >= A.n_verts || rt_f2u(t.z) >= A.n_verts) atomicAdd(&A.bad[0], 1u);
  }
  if (i < A.n_nodes) {
    const float4 lo = A.nodes[2 * (size_t)i], hi = A.nodes[2 * (size_t)i + 1];
    const uint32_t skip = rt_f2u(lo.w), data = rt_f2u(hi.w);
    if (i < A.n_tlas) {
      // TLAS: absolute skips that move forward and stay inside the TLAS; a leaf names an instance
      bool ok = skip > i && skip <= A.n_tlas;
      if (data != 0u) ok = ok && (data >> 3) < A.n_inst;
      else ok = ok && i + 1u < A.n_tlas;
      if (!ok) atomicAdd(&A.bad[1], 1u);
    } else if (A.n_roots) {
      // BLAS node: find the BLAS it belongs to (largest root <= its local index); nodes no instance reaches are ignored
      const uint32_t g = i - A.n_tlas;
      uint32_t lo_i = 0, hi_i = A.n_roots;
      while (lo_i < hi_i) {
        const uint32_t mid = (lo_i + hi_i) >> 1;
        if (A.roots[mid] <= g) lo_i = mid + 1u; else hi_i = mid;
      }
      if (lo_i > 0u) {
        const uint32_t root = A.roots[lo_i - 1u];
        if ((size_t)A.n_tlas + root < A.n_nodes) {
          const uint32_t size = rt_f2u(A.nodes[2 * ((size_t)A.n_tlas + root)].w);  // the root's skip = size of its BLAS
          const uint32_t local = g - root;
          if (local < size) {
            bool ok = skip > local && skip <= size;
            if (data != 0u) ok = ok && (size_t)(data >> 3) + (data & 7u) <= A.n_tris;
            else ok = ok && local + 1u < size;
            if (!ok) atomicAdd(&A.bad[2], 1u);
          }
        }
      }
    }
  }
  if (i < A.n_inst) {
    const uint32_t off = rt_f2u(A.inst[9 * (size_t)i + 8].x);
    bool ok = (size_t)A.n_tlas + off < A.n_nodes;
    if (ok) {
      const uint32_t size = rt_f2u(A.nodes[2 * ((size_t)A.n_tlas + off)].w);
      ok = size >= 1u && (size_t)A.n_tlas + off + size <= A.n_nodes;
    }
    if (!ok) atomicAdd(&A.bad[3], 1u);
  }
  if (i < A.n_lights) {
    const uint2 l = A.lights[i];
    if (l.x >= A.n_inst || l.y >= A.n_tris) atomicAdd(&A.bad[4], 1u);
  }
}

}  // namespace rtk
#endif
