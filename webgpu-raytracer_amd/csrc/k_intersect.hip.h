// k_intersect.hip.h — rays, slab and triangle tests, the per-lane stackless TLAS / BLAS walk (Raytracer.wgsl:433-600).
// Part of the kernel set of csrc/kernels.hip.h (included from there, in order; not a stand-alone header).
#ifndef MI355RT_K_INTERSECT_HIP_H
#define MI355RT_K_INTERSECT_HIP_H

namespace rtk {

// ------------------------------------------------------------------ traversal
struct LocalRay {
  rt3 o, d, inv_d, o_inv_d;
};
__device__ __forceinline__ LocalRay make_ray(rt3 o, rt3 d) {  // :83-86
  LocalRay r;
  r.o = o;
  r.d = d;
  r.inv_d = rt_rcp3(d);
  r.o_inv_d = o * r.inv_d;
  return r;
}
// slab test (:433-441): true when tm_near <= tm_far
__device__ __forceinline__ bool hit_box(float4 lo, float4 hi, const LocalRay& r, float t_min, float t_max) {
  float t1x = lo.x * r.inv_d.x - r.o_inv_d.x, t2x = hi.x * r.inv_d.x - r.o_inv_d.x;
  float t1y = lo.y * r.inv_d.y - r.o_inv_d.y, t2y = hi.y * r.inv_d.y - r.o_inv_d.y;
  float t1z = lo.z * r.inv_d.z - r.o_inv_d.z, t2z = hi.z * r.inv_d.z - r.o_inv_d.z;
  float nx = rt_min(t1x, t2x), ny = rt_min(t1y, t2y), nz = rt_min(t1z, t2z);
  float fx = rt_max(t1x, t2x), fy = rt_max(t1y, t2y), fz = rt_max(t1z, t2z);
  float tm_near = rt_max(t_min, rt_max(nx, rt_max(ny, nz)));
  float tm_far = rt_min(t_max, rt_min(fx, rt_min(fy, fz)));
  return tm_near <= tm_far;
}
// Möller–Trumbore on the precomputed (v0, e1, e2) record (:443-453); returns t or -1
__device__ __forceinline__ float hit_tri(float4 g0, float4 g1, float4 g2, const LocalRay& r, float t_min,
                                         float t_max) {
  rt3 v0 = xyz(g0), e1 = xyz(g1), e2 = xyz(g2);
  rt3 h = rt_cross(r.d, e2);
  float a = rt_dot(e1, h);
  if (rt_abs(a) < 1e-6f) return -1.0f;
  float f = rt_rcp(a);
  rt3 s = r.o - v0;
  float u = f * rt_dot(s, h);
  if (u < 0.0f || u > 1.0f) return -1.0f;
  rt3 q = rt_cross(s, e1);
  float v = f * rt_dot(r.d, q);
  if (v < 0.0f || u + v > 1.0f) return -1.0f;
  float t = f * rt_dot(e2, q);
  return (t > t_min && t < t_max) ? t : -1.0f;
}

struct Hit {
  float t;
  int32_t tri;   // -1 = none (the reference carries the id as f32; identical below 2^24 triangles)
  int32_t inst;  // -1 = none
};

// closest hit: intersect_tlas + intersect_blas (:455-528)
template <bool COUNT>
__device__ Hit trace_closest(const DevScene& S, uint32_t blas_base, rt3 o, rt3 d, float t_min, float t_max,
                             LaneCounters& c) {
  Hit res;
  res.t = t_max;
  res.tri = -1;
  res.inst = -1;
  if (blas_base == 0u) return res;
  LocalRay rw = make_ray(o, d);
  uint32_t curr = 0u;
  const uint32_t end_node = rt_f2u(S.nodes[0].w);
  while (curr < end_node) {
    float4 lo = S.nodes[2 * curr], hi = S.nodes[2 * curr + 1];
    if (COUNT) c.nodes++;
    uint32_t next = rt_f2u(lo.w);
    if (hit_box(lo, hi, rw, t_min, res.t)) {
      uint32_t data = rt_f2u(hi.w);
      if (data != 0u) {
        uint32_t inst = data >> 3;
        InvRows m = load_inv_rows(S, inst);
        LocalRay rl = make_ray(mul_point(m, o), mul_dir(m, d));
        const uint32_t start = blas_base + rt_f2u(m.tail.x);
        const uint32_t bend = start + rt_f2u(S.nodes[2 * start].w);
        uint32_t bc = start;
        float closest = res.t;
        int32_t best = -1;
        while (bc < bend) {
          float4 blo = S.nodes[2 * bc], bhi = S.nodes[2 * bc + 1];
          if (COUNT) c.nodes++;
          uint32_t bnext = start + rt_f2u(blo.w);
          if (hit_box(blo, bhi, rl, t_min, closest)) {
            uint32_t bdata = rt_f2u(bhi.w);
            if (bdata != 0u) {
              uint32_t first = bdata >> 3, count = bdata & 7u;
              for (uint32_t i = 0; i < count; i++) {
                uint32_t tri = first + i;
                if (COUNT) c.tris++;
                float t = hit_tri(S.tri_geom[RT_TRI_STRIDE * tri], S.tri_geom[RT_TRI_STRIDE * tri + 1], S.tri_geom[RT_TRI_STRIDE * tri + 2], rl, t_min,
                                  closest);
                if (t > 0.0f) {
                  closest = t;
                  best = (int32_t)tri;
                }
              }
            } else {
              bnext = bc + 1u;
            }
          }
          bc = bnext;
        }
        if (best >= 0) {
          res.t = closest;
          res.tri = best;
          res.inst = (int32_t)inst;
        }
      } else {
        next = curr + 1u;
      }
    }
    curr = next;
  }
  return res;
}

// any hit: intersect_tlas_shadow + intersect_blas_shadow (:532-600)
template <bool COUNT>
__device__ bool trace_any(const DevScene& S, uint32_t blas_base, rt3 o, rt3 d, float t_min, float t_max,
                          LaneCounters& c) {
  if (blas_base == 0u) return false;
  LocalRay rw = make_ray(o, d);
  uint32_t curr = 0u;
  const uint32_t end_node = rt_f2u(S.nodes[0].w);
  while (curr < end_node) {
    float4 lo = S.nodes[2 * curr], hi = S.nodes[2 * curr + 1];
    if (COUNT) c.nodes++;
    uint32_t next = rt_f2u(lo.w);
    if (hit_box(lo, hi, rw, t_min, t_max)) {
      uint32_t data = rt_f2u(hi.w);
      if (data != 0u) {
        uint32_t inst = data >> 3;
        InvRows m = load_inv_rows(S, inst);
        LocalRay rl = make_ray(mul_point(m, o), mul_dir(m, d));
        const uint32_t start = blas_base + rt_f2u(m.tail.x);
        const uint32_t bend = start + rt_f2u(S.nodes[2 * start].w);
        uint32_t bc = start;
        while (bc < bend) {
          float4 blo = S.nodes[2 * bc], bhi = S.nodes[2 * bc + 1];
          if (COUNT) c.nodes++;
          uint32_t bnext = start + rt_f2u(blo.w);
          if (hit_box(blo, bhi, rl, t_min, t_max)) {
            uint32_t bdata = rt_f2u(bhi.w);
            if (bdata != 0u) {
              uint32_t first = bdata >> 3, count = bdata & 7u;
              for (uint32_t i = 0; i < count; i++) {
                uint32_t tri = first + i;
                if (COUNT) c.tris++;
                float t = hit_tri(S.tri_geom[RT_TRI_STRIDE * tri], S.tri_geom[RT_TRI_STRIDE * tri + 1], S.tri_geom[RT_TRI_STRIDE * tri + 2], rl, t_min,
                                  t_max);
                if (t > 0.0f) return true;
              }
            } else {
              bnext = bc + 1u;
            }
          }
          bc = bnext;
        }
      } else {
        next = curr + 1u;
      }
    }
    curr = next;
  }
  return false;
}

}  // namespace rtk
#endif
