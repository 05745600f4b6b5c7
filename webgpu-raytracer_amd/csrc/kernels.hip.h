// kernels.hip.h — hand-written gfx950 kernels of the path-tracing hot path.
//
//   k_prepare_tris / k_prepare_instances   upload-time re-layout (device_scene.h)
//   k_primary_visibility   replaces the hardware-raster G-buffer pass (Rasterizer.wgsl:81-173,
//                          RasterizerPass.ts:97-140): one closest-hit cast per pixel
//   k_pathtrace            Raytracer.wgsl `main` + ray_color (:607-819)
//   k_postprocess          PostProcess.wgsl `main` (:103-176)
//
// All arithmetic is unfused IEEE f32 (-ffp-contract=off) with the builtin semantics of
// include/mi355rt_math.h, in the evaluation order of the WGSL source, so that every path
// takes the same branches as the CPU oracle and results agree bit for bit.
#ifndef MI355RT_KERNELS_HIP_H
#define MI355RT_KERNELS_HIP_H

#include "device_scene.h"

#define RT_T_MIN 0.001f
#define RT_T_MAX 1e30f
#define RT_COUNTER_SHARDS 1024

namespace rtk {

// ---------------------------------------------------------------- small helpers
__device__ __forceinline__ rt3 xyz(float4 v) { return rt3_make(v.x, v.y, v.z); }

struct LaneCounters {
  uint32_t primary, extension, shadow, nodes, tris, shaded;
};

// rows 0..2 of a column-major matrix M: (M*vec4(p,1)).xyz, (M*vec4(d,0)).xyz and (vec4(n,0)*M).xyz
struct InvRows {
  float4 r0, r1, r2, tail;  // tail = {bitcast(blas_node_offset), M[3], M[7], M[11]}
};
__device__ __forceinline__ InvRows load_inv_rows(const DevScene& S, uint32_t inst) {
  InvRows m;
  m.r0 = S.inst_trav[4 * inst + 0];
  m.r1 = S.inst_trav[4 * inst + 1];
  m.r2 = S.inst_trav[4 * inst + 2];
  m.tail = S.inst_trav[4 * inst + 3];
  return m;
}
__device__ __forceinline__ rt3 mul_point(const InvRows& m, rt3 p) {
  return rt3_make(m.r0.x * p.x + m.r0.y * p.y + m.r0.z * p.z + m.r0.w * 1.0f,
                  m.r1.x * p.x + m.r1.y * p.y + m.r1.z * p.z + m.r1.w * 1.0f,
                  m.r2.x * p.x + m.r2.y * p.y + m.r2.z * p.z + m.r2.w * 1.0f);
}
__device__ __forceinline__ rt3 mul_dir(const InvRows& m, rt3 d) {
  return rt3_make(m.r0.x * d.x + m.r0.y * d.y + m.r0.z * d.z + m.r0.w * 0.0f,
                  m.r1.x * d.x + m.r1.y * d.y + m.r1.z * d.z + m.r1.w * 0.0f,
                  m.r2.x * d.x + m.r2.y * d.y + m.r2.z * d.z + m.r2.w * 0.0f);
}
__device__ __forceinline__ rt3 normal_to_world(const InvRows& m, rt3 n) {  // (vec4(n,0) * inv).xyz
  return rt3_make(n.x * m.r0.x + n.y * m.r1.x + n.z * m.r2.x + 0.0f * m.tail.y,
                  n.x * m.r0.y + n.y * m.r1.y + n.z * m.r2.y + 0.0f * m.tail.z,
                  n.x * m.r0.z + n.y * m.r1.z + n.z * m.r2.z + 0.0f * m.tail.w);
}

// ------------------------------------------------------------------- textures
// textureSampleLevel(tex, smp, uv, layer, 0): bilinear / repeat / level 0 / unorm / no sRGB
__device__ __forceinline__ rt3 texel_rgb(const uint8_t* base, int x, int y) {
  uint32_t p = *reinterpret_cast<const uint32_t*>(base + ((size_t)y * RT_TEX_SIZE + (size_t)x) * 4);
  return rt3_make(rt_from_unorm8(p & 255u), rt_from_unorm8((p >> 8) & 255u), rt_from_unorm8((p >> 16) & 255u));
}
__device__ rt3 sample_tex(const DevScene& S, rt2 tuv, int32_t layer) {
  if (S.tex_layers == 0u) return rt3_splat(1.0f);
  if (layer < 0) layer = 0;
  if ((uint32_t)layer >= S.tex_layers) layer = (int32_t)S.tex_layers - 1;
  const int N = RT_TEX_SIZE;
  float x = tuv.x * (float)N - 0.5f, y = tuv.y * (float)N - 0.5f;
  float fx0 = rt_floor(x), fy0 = rt_floor(y);
  float fx = x - fx0, fy = y - fy0;
  int ix = rt_f2i32_sat(fx0), iy = rt_f2i32_sat(fy0);
  int x0 = (int)((uint32_t)ix & (uint32_t)(N - 1)), x1 = (int)((uint32_t)(ix + 1) & (uint32_t)(N - 1));
  int y0 = (int)((uint32_t)iy & (uint32_t)(N - 1)), y1 = (int)((uint32_t)(iy + 1) & (uint32_t)(N - 1));
  const uint8_t* base = S.tex + (size_t)layer * N * N * 4;
  rt3 top = rt_mix3(texel_rgb(base, x0, y0), texel_rgb(base, x1, y0), fx);
  rt3 bot = rt_mix3(texel_rgb(base, x0, y1), texel_rgb(base, x1, y1), fx);
  return rt_mix3(top, bot, fy);
}

// ------------------------------------------------------------------------ RNG
__device__ __forceinline__ uint32_t init_rng(uint32_t pixel_idx, uint32_t frame) {  // Raytracer.wgsl:178-183
  uint32_t s = pixel_idx + frame * 719393u;
  s ^= 2747636419u; s *= 2654435769u; s ^= (s >> 16);
  s *= 2654435769u; s ^= (s >> 16); s *= 2654435769u;
  return s;
}
__device__ __forceinline__ float rand_pcg(uint32_t& state) {  // :185-189
  uint32_t old = state;
  state = old * 747796405u + 2891336453u;
  uint32_t word = (state >> ((old >> 28) + 4u)) ^ state;
  // f32(u32) rounds to nearest even; 4294967295.0 is 2^32 as an f32 literal: the division is an exact scaling
  return (float)((word >> 22) ^ word) * 2.3283064365386962890625e-10f;
}

// ------------------------------------------------------------------ traversal
struct LocalRay {
  rt3 o, d, inv_d, o_inv_d;
};
__device__ __forceinline__ LocalRay make_ray(rt3 o, rt3 d) {  // :83-86
  LocalRay r;
  r.o = o;
  r.d = d;
  r.inv_d = rt3_splat(1.0f) / d;
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
  float f = 1.0f / a;
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
                float t = hit_tri(S.tri_geom[3 * tri], S.tri_geom[3 * tri + 1], S.tri_geom[3 * tri + 2], rl, t_min,
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
                float t = hit_tri(S.tri_geom[3 * tri], S.tri_geom[3 * tri + 1], S.tri_geom[3 * tri + 2], rl, t_min,
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

// -------------------------------------------------------------- surface frame
// What a bounce needs to know about the hit triangle (Raytracer.wgsl:625-654 and :738-779).
struct Surface {
  float hit_t;
  rt2 tex_uv;
  rt3 normal;        // shading normal, world space
  rt3 geom_n;        // geometric normal, world space
  rt3 albedo;
  float u_bar, v_bar, w_bar;
};

struct Bary {
  float u, v, w, t;
  rt3 e1, e2;
};
// unbounded ray/plane barycentrics of the local-space ray against triangle `tri` (:632-643)
__device__ __forceinline__ Bary barycentrics(const DevScene& S, uint32_t tri, rt3 lo, rt3 ld) {
  rt3 v0 = xyz(S.tri_geom[3 * tri]);
  Bary b;
  b.e1 = xyz(S.tri_geom[3 * tri + 1]);
  b.e2 = xyz(S.tri_geom[3 * tri + 2]);
  rt3 s = lo - v0;
  rt3 h = rt_cross(ld, b.e2);
  float f = 1.0f / rt_dot(b.e1, h);
  b.u = f * rt_dot(s, h);
  rt3 q = rt_cross(s, b.e1);
  b.v = f * rt_dot(ld, q);
  b.w = 1.0f - b.u - b.v;
  b.t = f * rt_dot(b.e2, q);
  return b;
}

__device__ __forceinline__ rt2 pack_normal(rt3 n) {  // Rasterizer.wgsl:71-74
  float s = 1.0f / (rt_abs(n.x) + rt_abs(n.y) + rt_abs(n.z));
  rt2 p = rt2_make(n.x * s, n.y * s);
  if (n.z < 0.0f) {
    float ox = (1.0f - rt_abs(p.y)) * (p.x >= 0.0f ? 1.0f : -1.0f);
    float oy = (1.0f - rt_abs(p.x)) * (p.y >= 0.0f ? 1.0f : -1.0f);
    return rt2_make(ox, oy);
  }
  return p;
}
__device__ __forceinline__ rt3 unpack_normal(float px, float py) {  // Raytracer.wgsl:121-127
  rt3 n = rt3_make(px, py, 1.0f - rt_abs(px) - rt_abs(py));
  float t = rt_saturate(-n.z);
  n.x += (n.x >= 0.0f) ? -t : t;
  n.y += (n.y >= 0.0f) ? -t : t;
  return rt_normalize(n);
}

// ---------------------------------------------------------------------- BSDFs
struct Onb {
  rt3 u, v, w;
};
__device__ __forceinline__ Onb build_onb(rt3 n) {  // :207-214
  float sign = (n.z >= 0.0f) ? 1.0f : -1.0f;
  float a = -1.0f / (sign + n.z);
  float b = n.x * n.y * a;
  Onb o;
  o.u = rt3_make(1.0f + sign * n.x * n.x * a, sign * b, -sign * n.x);
  o.v = rt3_make(b, sign + n.y * n.y * a, -n.y);
  o.w = n;
  return o;
}
__device__ __forceinline__ rt3 to_world(const Onb& o, rt3 a) { return a.x * o.u + a.y * o.v + a.z * o.w; }

__device__ __forceinline__ float ggx_d(float n_dot_h, float a2) {  // :236-239
  float d = (n_dot_h * a2 - n_dot_h) * n_dot_h + 1.0f;
  return a2 / (RT_PI * d * d);
}
__device__ __forceinline__ float ggx_g(float n_dot_v, float n_dot_l, float a2) {  // :241-245
  float g1_v = 2.0f * n_dot_v / (n_dot_v + rt_sqrt(a2 + (1.0f - a2) * n_dot_v * n_dot_v));
  float g1_l = 2.0f * n_dot_l / (n_dot_l + rt_sqrt(a2 + (1.0f - a2) * n_dot_l * n_dot_l));
  return g1_v * g1_l;
}
__device__ __forceinline__ float pow5(float x) {
  float x2 = x * x;
  return x2 * x2 * x;
}
__device__ __forceinline__ rt3 fresnel_schlick(float cos_theta, rt3 f0) {  // :252-254
  return f0 + (rt3_splat(1.0f) - f0) * pow5(rt_clamp(1.0f - cos_theta, 0.0f, 1.0f));
}
__device__ rt3 eval_ggx(rt3 n, rt3 v, rt3 l, float roughness, rt3 f0) {  // :256-269
  rt3 h = rt_normalize(v + l);
  float n_dot_v = rt_max(rt_dot(n, v), 1e-4f);
  float n_dot_l = rt_max(rt_dot(n, l), 1e-4f);
  float n_dot_h = rt_max(rt_dot(n, h), 1e-4f);
  float v_dot_h = rt_max(rt_dot(v, h), 1e-4f);
  float a2 = roughness * roughness;
  float d = ggx_d(n_dot_h, a2);
  float g = ggx_g(n_dot_v, n_dot_l, a2);
  rt3 f = fresnel_schlick(v_dot_h, f0);
  return (d * g * f) / (4.0f * n_dot_v * n_dot_l);
}

struct Scatter {
  rt3 dir;
  float pdf;
  rt3 throughput;
  bool specular;
};
__device__ Scatter sample_diffuse(rt3 normal, rt3 albedo, uint32_t& rng) {  // :228-233, :191-199
  Onb onb = build_onb(normal);
  float r1 = rand_pcg(rng);
  float r2 = rand_pcg(rng);
  float phi = RT_TWO_PI * r1;
  float cos_theta = rt_sqrt(1.0f - r2);
  float sin_theta = rt_sqrt(r2);
  float sp, cp;
  rt_sincos(phi, &sp, &cp);
  Scatter s;
  s.dir = to_world(onb, rt3_make(cp * sin_theta, sp * sin_theta, cos_theta));
  float c = rt_max(rt_dot(normal, s.dir), 0.0f);
  s.pdf = c / RT_PI;
  s.throughput = albedo;
  s.specular = false;
  return s;
}
__device__ Scatter sample_ggx(rt3 n, rt3 v, float roughness, rt3 f0, uint32_t& rng) {  // :271-306
  float a = roughness;
  float ux = rand_pcg(rng);
  float uy = rand_pcg(rng);
  float phi = RT_TWO_PI * ux;
  float cos_theta = rt_sqrt(rt_max(0.0f, (1.0f - uy) / (1.0f + (a * a - 1.0f) * uy)));
  float sin_theta = rt_sqrt(rt_max(0.0f, 1.0f - cos_theta * cos_theta));
  float sp, cp;
  rt_sincos(phi, &sp, &cp);
  Onb onb = build_onb(n);
  rt3 h = to_world(onb, rt3_make(sin_theta * cp, sin_theta * sp, cos_theta));
  rt3 l = rt_reflect(-v, h);
  Scatter s;
  if (rt_dot(n, l) <= 0.0f) {
    s.dir = rt3_splat(0.0f);
    s.pdf = 0.0f;
    s.throughput = rt3_splat(0.0f);
    s.specular = false;
    return s;
  }
  float n_dot_v = rt_max(rt_dot(n, v), 1e-4f);
  float n_dot_l = rt_max(rt_dot(n, l), 1e-4f);
  float n_dot_h = rt_max(rt_dot(n, h), 1e-4f);
  float v_dot_h = rt_max(rt_dot(v, h), 1e-4f);
  float a2 = a * a;
  float d = ggx_d(n_dot_h, a2);
  float g = ggx_g(n_dot_v, n_dot_l, a2);
  rt3 f = fresnel_schlick(v_dot_h, f0);
  s.dir = l;
  s.pdf = (d * n_dot_h) / (4.0f * v_dot_h);
  s.throughput = rt3_splat(0.0f);
  if (s.pdf > 1e-6f) s.throughput = (g * f * v_dot_h) / (n_dot_v * n_dot_h);
  s.specular = roughness < 0.01f;
  return s;
}
__device__ Scatter sample_dielectric(rt3 dir, rt3 normal, float ior, rt3 albedo, uint32_t& rng) {  // :320-339
  bool front_face = rt_dot(dir, normal) < 0.0f;
  float ratio = front_face ? (1.0f / ior) : ior;
  rt3 n = front_face ? normal : -normal;
  rt3 unit_dir = rt_normalize(dir);
  float cos_theta = rt_min(rt_dot(-unit_dir, n), 1.0f);
  float sin_theta = rt_sqrt(1.0f - cos_theta * cos_theta);
  bool cannot_refract = ratio * sin_theta > 1.0f;
  bool reflect_it = cannot_refract;
  if (!reflect_it) {  // short-circuit `||`: the draw happens only when refraction is possible
    float r0 = (1.0f - ratio) / (1.0f + ratio);
    r0 = r0 * r0;
    float refl = r0 + (1.0f - r0) * pow5(1.0f - cos_theta);
    reflect_it = refl > rand_pcg(rng);
  }
  Scatter s;
  s.dir = reflect_it ? rt_reflect(unit_dir, n) : rt_refract(unit_dir, n, ratio);
  s.pdf = 1.0f;
  s.throughput = albedo;
  s.specular = true;
  return s;
}

// ------------------------------------------------------------- light sampling
struct LightSample {
  rt3 L, dir;
  float dist, pdf;
};
struct WorldTri {
  rt3 v0, v1, v2;
};
__device__ __forceinline__ WorldTri world_triangle(const DevScene& S, uint32_t tri, uint32_t inst) {
  float4 idx = S.topo[5 * tri];
  const float* m = reinterpret_cast<const float*>(&S.inst[9 * inst]);  // forward transform, column-major
  WorldTri w;
  w.v0 = rt_mat_mul_point(m, xyz(S.pos[rt_f2u(idx.x)]));
  w.v1 = rt_mat_mul_point(m, xyz(S.pos[rt_f2u(idx.y)]));
  w.v2 = rt_mat_mul_point(m, xyz(S.pos[rt_f2u(idx.z)]));
  return w;
}
__device__ LightSample sample_light(const DevScene& S, uint32_t light_count, rt3 hit_p, uint32_t& rng) {  // :345-399
  LightSample none;
  none.L = rt3_splat(0.0f);
  none.dir = rt3_splat(0.0f);
  none.dist = 0.0f;
  none.pdf = 0.0f;
  if (light_count == 0u) return none;
  uint32_t pick = rt_f2u32_sat(rand_pcg(rng) * (float)light_count);
  if (pick >= S.n_lights) pick = S.n_lights - 1u;  // robust buffer access clamp (rand can be exactly 1.0)
  // world-space triangle, unit normal and area of the picked light: precomputed per light at upload time
  // (k_prepare_lights, same operations as Raytracer.wgsl:354-373, so bit-identical)
  const float4 q0 = S.light_rec[4 * pick], q1 = S.light_rec[4 * pick + 1], q2 = S.light_rec[4 * pick + 2],
               q3 = S.light_rec[4 * pick + 3];
  WorldTri w;
  w.v0 = xyz(q0);
  w.v1 = xyz(q1);
  w.v2 = xyz(q2);
  const rt3 n_raw = rt3_make(q1.w, q2.w, q3.x);
  const float area = q0.w;
  uint2 ref;
  ref.y = rt_f2u(q3.y);
  float r1 = rand_pcg(rng);
  float r2 = rand_pcg(rng);
  float sqrt_r1 = rt_sqrt(r1);
  float u = 1.0f - sqrt_r1;
  float v = r2 * sqrt_r1;
  float ww = 1.0f - u - v;
  rt3 p = w.v0 * u + w.v1 * v + w.v2 * ww;
  rt3 l_dir = p - hit_p;
  float dist_sq = rt_dot(l_dir, l_dir);
  float dist = rt_sqrt(dist_sq);
  rt3 unit_l = l_dir / dist;
  float cos_l = rt_max(rt_dot(n_raw, -unit_l), 0.0f);
  if (cos_l < 1e-6f) return none;
  float4 idx = S.topo[5 * ref.y], d0 = S.topo[5 * ref.y + 1], d2 = S.topo[5 * ref.y + 3];
  rt3 L = xyz(d0);
  if (d2.x > -0.5f) {
    float2 a = S.uv[rt_f2u(idx.x)], b = S.uv[rt_f2u(idx.y)], c = S.uv[rt_f2u(idx.z)];
    rt2 tuv = rt2_make(a.x, a.y) * u + rt2_make(b.x, b.y) * v + rt2_make(c.x, c.y) * ww;
    L = L * sample_tex(S, tuv, rt_f2i32_sat(d2.x));
  }
  LightSample s;
  s.L = L;
  s.dir = unit_l;
  s.dist = dist;
  s.pdf = (dist_sq / (cos_l * area)) / (float)light_count;
  return s;
}
__device__ float light_pdf(const DevScene& S, uint32_t light_count, uint32_t tri, uint32_t inst, float t,
                           rt3 l_dir) {  // :401-421
  WorldTri w = world_triangle(S, tri, inst);
  rt3 edge1 = w.v1 - w.v0;
  rt3 edge2 = w.v2 - w.v0;
  rt3 cr = rt_cross(edge1, edge2);
  float area = rt_length(cr) * 0.5f;
  rt3 normal = rt_normalize(cr);
  float cos_l = rt_max(rt_dot(normal, -l_dir), 0.0f);
  if (cos_l < 1e-4f) return 0.0f;
  float dist_sq = t * t;
  return (dist_sq / (cos_l * area)) / (float)light_count;
}
__device__ __forceinline__ float power_heuristic(float a, float b) {
  float a2 = a * a, b2 = b * b;
  return a2 / (a2 + b2);
}

// ------------------------------------------------------------- counters
__device__ __forceinline__ uint64_t wave_sum(uint32_t v) {
  uint64_t s = v;
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  return s;
}
template <bool DETAIL>
__device__ __forceinline__ void flush_counters(const LaneCounters& c, uint64_t* counters, uint32_t shard) {
  uint64_t* dst = counters + (size_t)(shard % RT_COUNTER_SHARDS) * 6;
  uint64_t p = wave_sum(c.primary), e = wave_sum(c.extension), s = wave_sum(c.shadow);
  uint64_t n = 0, t = 0, h = 0;
  if (DETAIL) {
    n = wave_sum(c.nodes);
    t = wave_sum(c.tris);
    h = wave_sum(c.shaded);
  }
  if ((threadIdx.x & 63u) == 0u) {
    if (p) atomicAdd((unsigned long long*)&dst[0], (unsigned long long)p);
    if (e) atomicAdd((unsigned long long*)&dst[1], (unsigned long long)e);
    if (s) atomicAdd((unsigned long long*)&dst[2], (unsigned long long)s);
    if (DETAIL) {
      atomicAdd((unsigned long long*)&dst[3], (unsigned long long)n);
      atomicAdd((unsigned long long*)&dst[4], (unsigned long long)t);
      atomicAdd((unsigned long long*)&dst[5], (unsigned long long)h);
    }
  }
}

__device__ __forceinline__ bool owns_row(const DevFrame& F, uint32_t y) {
  if (F.stripe_rows == 0u || F.stripe_count <= 1u) return true;
  return (y / F.stripe_rows) % F.stripe_count == F.stripe_rank;
}

// One wave = one 8x8 pixel tile (the reference's workgroup shape, RaytracePass.ts:96-103).
__device__ __forceinline__ bool tile_pixel(const rt_scene_uniforms& U, uint32_t& x, uint32_t& y) {
  const uint32_t tiles_x = (U.width + 7u) / 8u;
  const uint32_t tile = blockIdx.x;
  const uint32_t lane = threadIdx.x;
  x = (tile % tiles_x) * 8u + (lane & 7u);
  y = (tile / tiles_x) * 8u + (lane >> 3);
  return x < U.width && y < U.height;
}

// =========================================================== upload-time re-layout
__global__ void k_prepare_tris(const float4* __restrict__ topo, const float4* __restrict__ pos,
                               float4* __restrict__ tri_geom, uint32_t n_tris, uint32_t n_verts) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_tris) return;
  float4 idx = topo[5 * i];
  uint32_t i0 = rt_f2u(idx.x), i1 = rt_f2u(idx.y), i2 = rt_f2u(idx.z);
  if (i0 >= n_verts) i0 = n_verts - 1;  // robust buffer access: clamp instead of faulting
  if (i1 >= n_verts) i1 = n_verts - 1;
  if (i2 >= n_verts) i2 = n_verts - 1;
  rt3 v0 = xyz(pos[i0]), v1 = xyz(pos[i1]), v2 = xyz(pos[i2]);
  rt3 e1 = v1 - v0, e2 = v2 - v0;
  tri_geom[3 * i + 0] = make_float4(v0.x, v0.y, v0.z, 0.0f);
  tri_geom[3 * i + 1] = make_float4(e1.x, e1.y, e1.z, 0.0f);
  tri_geom[3 * i + 2] = make_float4(e2.x, e2.y, e2.z, 0.0f);
}
__global__ void k_prepare_lights(DevScene S, float4* __restrict__ light_rec, uint32_t n, uint32_t n_tris,
                                 uint32_t n_inst) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint2 ref = S.lights[i];
  if (ref.y >= n_tris) ref.y = n_tris - 1;  // robust buffer access: clamp instead of faulting
  if (ref.x >= n_inst) ref.x = n_inst - 1;
  WorldTri w = world_triangle(S, ref.y, ref.x);
  rt3 edge1 = w.v1 - w.v0;
  rt3 edge2 = w.v2 - w.v0;
  rt3 cr = rt_cross(edge1, edge2);
  rt3 n_raw = rt_normalize(cr);
  float area = rt_length(cr) * 0.5f;
  light_rec[4 * i + 0] = make_float4(w.v0.x, w.v0.y, w.v0.z, area);
  light_rec[4 * i + 1] = make_float4(w.v1.x, w.v1.y, w.v1.z, n_raw.x);
  light_rec[4 * i + 2] = make_float4(w.v2.x, w.v2.y, w.v2.z, n_raw.y);
  light_rec[4 * i + 3] = make_float4(n_raw.z, rt_u2f(ref.y), 0.0f, 0.0f);
}
__global__ void k_prepare_instances(const float4* __restrict__ inst, float4* __restrict__ inst_trav, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 c0 = inst[9 * i + 4], c1 = inst[9 * i + 5], c2 = inst[9 * i + 6], c3 = inst[9 * i + 7];
  float4 meta = inst[9 * i + 8];
  inst_trav[4 * i + 0] = make_float4(c0.x, c1.x, c2.x, c3.x);
  inst_trav[4 * i + 1] = make_float4(c0.y, c1.y, c2.y, c3.y);
  inst_trav[4 * i + 2] = make_float4(c0.z, c1.z, c2.z, c3.z);
  inst_trav[4 * i + 3] = make_float4(meta.x, c0.w, c1.w, c2.w);
}

// ================================================================ primary visibility
template <bool DETAIL>
__global__ __launch_bounds__(64) void k_primary_visibility(DevScene S, DevFrame F, rt_scene_uniforms U,
                                                           const DevFrameSlot* __restrict__ slots) {
  // batched dispatch: blockIdx.y selects the frame; its jitter and G-buffer planes come from the slot table
  if (slots) {
    const DevFrameSlot sl = slots[blockIdx.y];
    U.frame_count = sl.frame_count;
    U.jitter[0] = sl.jitter_x;
    U.jitter[1] = sl.jitter_y;
    F.albedo = sl.albedo;
    F.normal_id = sl.normal_id;
    F.depth = sl.depth;
  }
  uint32_t x, y;
  bool live;
  if (F.own_period) {
    // sharded render with tile-aligned stripes: blockIdx.x enumerates only the tiles of the rows this rank owns
    const uint32_t tiles_x = (U.width + 7u) / 8u;
    uint32_t trow = blockIdx.x / tiles_x;
    trow = (trow / F.own_run) * F.own_period + F.own_first + (trow % F.own_run);
    x = (blockIdx.x % tiles_x) * 8u + (threadIdx.x & 7u);
    y = trow * 8u + (threadIdx.x >> 3);
    live = x < U.width && y < U.height;
  } else {
    live = tile_pixel(U, x, y) && owns_row(F, y);
  }
  LaneCounters c = {0, 0, 0, 0, 0, 0};
  if (live) {
    const uint32_t p_idx = y * U.width + x;
    rt3 eye = rt3_make(U.camera.origin[0], U.camera.origin[1], U.camera.origin[2]);
    rt3 ll = rt3_make(U.camera.lower_left[0], U.camera.lower_left[1], U.camera.lower_left[2]);
    rt3 hor = rt3_make(U.camera.horizontal[0], U.camera.horizontal[1], U.camera.horizontal[2]);
    rt3 ver = rt3_make(U.camera.vertical[0], U.camera.vertical[1], U.camera.vertical[2]);
    rt3 center = ll + hor * 0.5f + ver * 0.5f;
    float focal_length = rt_length(center - eye);
    const float z_near = 0.001f, z_far = 10000.0f;
    float u = ((float)x + 0.5f + U.jitter[0] * (float)U.width) / (float)U.width;
    float v = 1.0f - ((float)y + 0.5f + U.jitter[1] * (float)U.height) / (float)U.height;
    rt3 d = ll + u * hor + v * ver - eye;
    c.primary = 1;
    Hit hit = trace_closest<DETAIL>(S, U.blas_base_idx, eye, d, z_near / focal_length, z_far / focal_length, c);
    if (hit.inst < 0) {
      F.albedo[p_idx] = 0u;
      F.normal_id[p_idx] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      F.depth[p_idx] = 1.0f;
    } else {
      InvRows m = load_inv_rows(S, (uint32_t)hit.inst);
      Bary b = barycentrics(S, (uint32_t)hit.tri, mul_point(m, eye), mul_dir(m, d));
      float4 idx = S.topo[5 * hit.tri], d0 = S.topo[5 * hit.tri + 1], d2 = S.topo[5 * hit.tri + 3];
      uint32_t i0 = rt_f2u(idx.x), i1 = rt_f2u(idx.y), i2 = rt_f2u(idx.z);
      rt3 wn0 = rt_normalize(normal_to_world(m, xyz(S.nrm[i0])));
      rt3 wn1 = rt_normalize(normal_to_world(m, xyz(S.nrm[i1])));
      rt3 wn2 = rt_normalize(normal_to_world(m, xyz(S.nrm[i2])));
      rt3 n = rt_normalize(wn0 * b.w + wn1 * b.u + wn2 * b.v);
      rt2 pn = pack_normal(n);
      rt3 albedo = xyz(d0);
      if (d2.x > -0.5f) {
        float2 a0 = S.uv[i0], a1 = S.uv[i1], a2 = S.uv[i2];
        rt2 tuv = rt2_make(a0.x, a0.y) * b.w + rt2_make(a1.x, a1.y) * b.u + rt2_make(a2.x, a2.y) * b.v;
        albedo = albedo * sample_tex(S, tuv, rt_f2i32_sat(d2.x));
      }
      F.albedo[p_idx] = rt_unorm8(albedo.x) | (rt_unorm8(albedo.y) << 8) | (rt_unorm8(albedo.z) << 16) | (255u << 24);
      F.normal_id[p_idx] = make_float4(pn.x, pn.y, rt_u2f((uint32_t)hit.tri), rt_u2f((uint32_t)hit.inst));
      float z_view = hit.t * focal_length;
      float z_clip = z_view * (z_far / (z_far - z_near)) - (z_far * z_near) / (z_far - z_near);
      F.depth[p_idx] = z_clip / z_view;
    }
  }
  flush_counters<DETAIL>(c, F.counters, blockIdx.x + blockIdx.y * 977u);
}

// ======================================================================= path tracer
// One sample: Raytracer.wgsl ray_color (:607-783).
template <bool DETAIL>
__device__ rt3 ray_color(const DevScene& S, const DevFrame& F, const rt_scene_uniforms& U, rt3 ro, rt3 rd,
                         uint32_t& rng, uint32_t p_idx, LaneCounters& c) {
  rt3 throughput = rt3_splat(1.0f);
  rt3 radiance = rt3_splat(0.0f);
  float prev_bsdf_pdf = 0.0f;
  bool specular_bounce = true;

  // depth 0 comes from the G-buffer
  if (F.depth[p_idx] >= 1.0f) return radiance;
  float4 g = F.normal_id[p_idx];
  uint32_t tri = rt_f2u(g.z);
  uint32_t inst = rt_f2u(g.w);
  InvRows m = load_inv_rows(S, inst);
  Bary b = barycentrics(S, tri, mul_point(m, ro), mul_dir(m, rd));
  float hit_t = b.t;
  float4 tidx = S.topo[5 * tri];
  float2 uv0 = S.uv[rt_f2u(tidx.x)], uv1 = S.uv[rt_f2u(tidx.y)], uv2 = S.uv[rt_f2u(tidx.z)];
  rt2 tex_uv = rt2_make(uv0.x, uv0.y) * b.w + rt2_make(uv1.x, uv1.y) * b.u + rt2_make(uv2.x, uv2.y) * b.v;
  rt3 normal = unpack_normal(g.x, g.y);
  uint32_t ga = F.albedo[p_idx];
  rt3 albedo = rt3_make(rt_from_unorm8(ga & 255u), rt_from_unorm8((ga >> 8) & 255u), rt_from_unorm8((ga >> 16) & 255u));
  rt3 world_geom_n = rt_normalize(normal_to_world(m, rt_normalize(rt_cross(b.e1, b.e2))));

  for (uint32_t depth = 0u; depth < F.max_depth; depth++) {
    if (DETAIL) c.shaded++;
    float4 d0 = S.topo[5 * tri + 1], d1 = S.topo[5 * tri + 2], d2 = S.topo[5 * tri + 3], d3 = S.topo[5 * tri + 4];
    uint32_t mat_type = rt_f2u32_sat(d0.w + 0.5f);
    rt3 hit_p = ro + rd * hit_t;

    normal = (rt_dot(rd, normal) < 0.0f) ? normal : -normal;
    world_geom_n = (rt_dot(rd, world_geom_n) < 0.0f) ? world_geom_n : -world_geom_n;

    float metallic = d1.x, roughness = d1.y;
    if (d2.y > -0.5f) {
      rt3 mr = sample_tex(S, tex_uv, rt_f2i32_sat(d2.y));
      metallic *= mr.z;
      roughness *= mr.y;
    }
    roughness = rt_max(roughness, 0.005f);
    rt3 emissive = xyz(d3);
    if (d2.w > -0.5f) emissive = emissive * sample_tex(S, tex_uv, rt_f2i32_sat(d2.w));
    rt3 f0 = rt_mix3(rt3_splat(0.04f), albedo, metallic);

    // emissive / light
    if (mat_type == 3u || rt_length(emissive) > 1e-4f) {
      rt3 em_val = (mat_type == 3u) ? albedo : emissive;
      if (specular_bounce) {
        radiance = radiance + throughput * em_val;
      } else {
        radiance = radiance +
                   throughput * em_val * power_heuristic(prev_bsdf_pdf, light_pdf(S, U.light_count, tri, inst, hit_t, rd));
      }
      if (mat_type == 3u) break;
    }

    // next-event estimation
    if (mat_type != 2u) {
      LightSample ls = sample_light(S, U.light_count, hit_p, rng);
      if (ls.pdf > 0.0f) {
        c.shadow++;
        if (!trace_any<DETAIL>(S, U.blas_base_idx, hit_p + world_geom_n * 1e-4f, ls.dir, RT_T_MIN, ls.dist - 2e-4f, c)) {
          rt3 bsdf_val = rt3_splat(0.0f);
          float bsdf_pdf = 0.0f;
          if (mat_type == 0u) {
            bsdf_val = albedo / RT_PI;
            bsdf_pdf = rt_max(rt_dot(normal, ls.dir), 0.0f) / RT_PI;
          } else if (mat_type == 1u) {
            bsdf_val = eval_ggx(normal, -rd, ls.dir, roughness, f0);
            rt3 H = rt_normalize(-rd + ls.dir);
            bsdf_pdf = (ggx_d(rt_dot(normal, H), roughness * roughness) * rt_max(rt_dot(normal, H), 0.0f)) /
                       (4.0f * rt_max(rt_dot(-rd, H), 0.0f));
          }
          if (bsdf_pdf > 0.0f) {
            radiance = radiance + throughput * bsdf_val * ls.L * power_heuristic(ls.pdf, bsdf_pdf) *
                                      rt_max(rt_dot(normal, ls.dir), 0.0f) / ls.pdf;
          }
        }
      }
    }

    Scatter sc;
    if (mat_type == 0u) {
      sc = sample_diffuse(normal, albedo, rng);
    } else if (mat_type == 1u) {
      sc = sample_ggx(normal, -rd, roughness, f0, rng);
    } else {
      sc = sample_dielectric(rd, normal, d1.z, albedo, rng);
    }
    if (mat_type != 2u && rt_dot(sc.dir, world_geom_n) <= 0.0f) {
      sc.pdf = 0.0f;
      sc.throughput = rt3_splat(0.0f);
    }
    if (sc.pdf <= 0.0f || rt_length(sc.throughput) <= 0.0f) break;

    throughput = throughput * sc.throughput;
    rt3 offset_n = (rt_dot(sc.dir, world_geom_n) > 0.0f) ? world_geom_n : -world_geom_n;
    ro = hit_p + offset_n * 1e-4f;
    rd = sc.dir;
    prev_bsdf_pdf = sc.pdf;
    specular_bounce = sc.specular;

    if (depth > 3u) {  // Russian roulette
      float p = rt_max(throughput.x, rt_max(throughput.y, throughput.z));
      if (rand_pcg(rng) > p) break;
      throughput = throughput / p;
    }

    if (depth < F.max_depth - 1u) {
      c.extension++;
      Hit hit = trace_closest<DETAIL>(S, U.blas_base_idx, ro, rd, RT_T_MIN, RT_T_MAX, c);
      if (hit.inst < 0) break;
      hit_t = hit.t;
      tri = (uint32_t)hit.tri;
      inst = (uint32_t)hit.inst;
      m = load_inv_rows(S, inst);
      b = barycentrics(S, tri, mul_point(m, ro), mul_dir(m, rd));
      tidx = S.topo[5 * tri];
      uint32_t i0 = rt_f2u(tidx.x), i1 = rt_f2u(tidx.y), i2 = rt_f2u(tidx.z);
      uv0 = S.uv[i0];
      uv1 = S.uv[i1];
      uv2 = S.uv[i2];
      tex_uv = rt2_make(uv0.x, uv0.y) * b.w + rt2_make(uv1.x, uv1.y) * b.u + rt2_make(uv2.x, uv2.y) * b.v;
      rt3 ln = rt_normalize(xyz(S.nrm[i0]) * b.w + xyz(S.nrm[i1]) * b.u + xyz(S.nrm[i2]) * b.v);
      normal = rt_normalize(normal_to_world(m, ln));
      float4 nd0 = S.topo[5 * tri + 1], nd2 = S.topo[5 * tri + 3];
      albedo = xyz(nd0);
      if (nd2.x > -0.5f) albedo = albedo * sample_tex(S, tex_uv, rt_f2i32_sat(nd2.x));
      if (nd2.z > -0.5f) {
        rt3 n_map = sample_tex(S, tex_uv, rt_f2i32_sat(nd2.z)) * 2.0f - rt3_splat(1.0f);
        rt3 T = rt_normalize(b.e1);
        rt3 B = rt_normalize(rt_cross(ln, T));
        rt3 ln_mapped = rt_normalize(T * n_map.x + B * n_map.y + ln * n_map.z);
        normal = rt_normalize(normal_to_world(m, ln_mapped));
      }
      world_geom_n = rt_normalize(normal_to_world(m, rt_normalize(rt_cross(b.e1, b.e2))));
    }
  }
  return radiance;
}

// Raytracer.wgsl `main` (:791-819)
template <bool DETAIL>
__global__ __launch_bounds__(64) void k_pathtrace(DevScene S, DevFrame F, rt_scene_uniforms U) {
  uint32_t x, y;
  bool live = tile_pixel(U, x, y) && owns_row(F, y);
  LaneCounters c = {0, 0, 0, 0, 0, 0};
  if (live) {
    const uint32_t p_idx = y * U.width + x;
    rt3 cam_o = rt3_make(U.camera.origin[0], U.camera.origin[1], U.camera.origin[2]);
    rt3 cam_ll = rt3_make(U.camera.lower_left[0], U.camera.lower_left[1], U.camera.lower_left[2]);
    rt3 cam_h = rt3_make(U.camera.horizontal[0], U.camera.horizontal[1], U.camera.horizontal[2]);
    rt3 cam_v = rt3_make(U.camera.vertical[0], U.camera.vertical[1], U.camera.vertical[2]);
    const float lens = U.camera.origin[3];
    rt3 col = rt3_splat(0.0f);
    for (uint32_t i = 0u; i < F.spp; i++) {
      uint32_t rng = init_rng(p_idx, U.frame_count * F.spp + i);
      rt3 off = rt3_splat(0.0f);
      if (lens > 0.0f) {  // random_in_unit_disk (:201-205)
        float r = rt_sqrt(rand_pcg(rng));
        float theta = RT_TWO_PI * rand_pcg(rng);
        float st, ct;
        rt_sincos(theta, &st, &ct);
        rt3 rdk = lens * rt3_make(r * ct, r * st, 0.0f);
        rt3 cu = rt3_make(U.camera.u[0], U.camera.u[1], U.camera.u[2]);
        rt3 cv = rt3_make(U.camera.v[0], U.camera.v[1], U.camera.v[2]);
        off = cu * rdk.x + cv * rdk.y;
      }
      float u = ((float)x + 0.5f + U.jitter[0] * (float)U.width) / (float)U.width;
      float v = 1.0f - ((float)y + 0.5f + U.jitter[1] * (float)U.height) / (float)U.height;
      rt3 d = cam_ll + u * cam_h + v * cam_v - cam_o - off;
      col = col + ray_color<DETAIL>(S, F, U, cam_o + off, d, rng, p_idx, c);
    }
    col = col / (float)F.spp;
    float4 acc = make_float4(col.x, col.y, col.z, 1.0f);
    if (U.frame_count > 1u) {
      float4 prev = F.accum[p_idx];
      acc = make_float4(prev.x + col.x, prev.y + col.y, prev.z + col.z, prev.w + 1.0f);
    }
    F.accum[p_idx] = acc;
  }
  flush_counters<DETAIL>(c, F.counters, blockIdx.x);
}

// ============================================================ path tracer, persistent form
// k_pathtrace_persistent: the production path-trace kernel.
//
//  * persistent waves: the grid is sized to the resident wave count; each wave pulls 8x8 pixel
//    tiles from a global ticket counter until the image is exhausted (one ray per lane);
//  * path regeneration: a lane whose path ended (light hit, miss, absorbed, Russian roulette,
//    depth limit) takes the next pixel of its wave's current tile, found with a ballot/mbcnt prefix
//    over the idle mask, so the 64 lanes stay busy instead of waiting for the longest path;
//  * per trip every live lane executes exactly one bounce: shade -> (NEE shadow ray) -> scatter ->
//    (extension ray), so the wave runs the two traversals and the shading code converged;
//  * traversal data (nodes, triangle records, instance records) is staged once per workgroup in LDS
//    when it fits (LDS = true); larger scenes read the same records through L1/L2;
//  * TLAS and BLAS are walked by ONE loop with an in-instance flag, so lanes in different
//    instances / levels share the node fetch + slab test.
// Per-path arithmetic and RNG draw order are exactly those of ray_color above (and of the oracle);
// only the scheduling differs, which cannot change any pixel because paths are independent.

typedef float f4 __attribute__((ext_vector_type(4)));

struct TravMem {  // pointers may be LDS or global; the template flag keeps the two code paths apart
  const f4* nodes;
  const f4* tri_geom;
  const f4* inst_trav;
};

__device__ __forceinline__ bool hit_box4(f4 lo, f4 hi, const LocalRay& r, float t_min, float t_max) {
  float t1x = lo.x * r.inv_d.x - r.o_inv_d.x, t2x = hi.x * r.inv_d.x - r.o_inv_d.x;
  float t1y = lo.y * r.inv_d.y - r.o_inv_d.y, t2y = hi.y * r.inv_d.y - r.o_inv_d.y;
  float t1z = lo.z * r.inv_d.z - r.o_inv_d.z, t2z = hi.z * r.inv_d.z - r.o_inv_d.z;
  float nx = rt_min(t1x, t2x), ny = rt_min(t1y, t2y), nz = rt_min(t1z, t2z);
  float fx = rt_max(t1x, t2x), fy = rt_max(t1y, t2y), fz = rt_max(t1z, t2z);
  float tm_near = rt_max(t_min, rt_max(nx, rt_max(ny, nz)));
  float tm_far = rt_min(t_max, rt_min(fx, rt_min(fy, fz)));
  return tm_near <= tm_far;
}
__device__ __forceinline__ LocalRay to_instance(const TravMem& M, uint32_t inst, rt3 o, rt3 d, uint32_t& blas_off) {
  f4 r0 = M.inst_trav[4 * inst + 0], r1 = M.inst_trav[4 * inst + 1], r2 = M.inst_trav[4 * inst + 2];
  blas_off = rt_f2u(M.inst_trav[4 * inst + 3].x);
  rt3 lo = rt3_make(r0.x * o.x + r0.y * o.y + r0.z * o.z + r0.w * 1.0f, r1.x * o.x + r1.y * o.y + r1.z * o.z + r1.w * 1.0f,
                    r2.x * o.x + r2.y * o.y + r2.z * o.z + r2.w * 1.0f);
  rt3 ld = rt3_make(r0.x * d.x + r0.y * d.y + r0.z * d.z + r0.w * 0.0f, r1.x * d.x + r1.y * d.y + r1.z * d.z + r1.w * 0.0f,
                    r2.x * d.x + r2.y * d.y + r2.z * d.z + r2.w * 0.0f);
  return make_ray(lo, ld);
}

// Branch-free Möller–Trumbore: same operations and the same accept/reject truth table as
// hit_triangle_raw (Raytracer.wgsl:443-453), evaluated without early exits so that a wave testing
// 64 different triangles stays converged.
__device__ __forceinline__ bool hit_tri_nb(f4 g0, f4 g1, f4 g2, const LocalRay& r, float t_min, float t_max, float& t_out) {
  rt3 v0 = rt3_make(g0.x, g0.y, g0.z), e1 = rt3_make(g1.x, g1.y, g1.z), e2 = rt3_make(g2.x, g2.y, g2.z);
  rt3 h = rt_cross(r.d, e2);
  float a = rt_dot(e1, h);
  float f = 1.0f / a;
  rt3 s = r.o - v0;
  float u = f * rt_dot(s, h);
  rt3 q = rt_cross(s, e1);
  float v = f * rt_dot(r.d, q);
  float t = f * rt_dot(e2, q);
  t_out = t;
  bool reject = (rt_abs(a) < 1e-6f) | (u < 0.0f) | (u > 1.0f) | (v < 0.0f) | (u + v > 1.0f);
  return !reject & (t > t_min) & (t < t_max);
}

// ---------------------------------------------------------------------------------------------
// traverse(): one walk over TLAS and BLAS nodes for the 64 rays of a wave.
//
// Divergence control.  A lane is SEARCHING (walking nodes: slab tests, instance entry/exit) or
// WAITING (it reached a BLAS leaf whose box it hits and has queued that leaf's triangles).  Every
// trip of the loop lets all searching lanes take ONE node step.  When enough triangle tests are queued
// (RT_FLUSH_ITEMS) or nobody is searching any more, the wave flushes the queue:
//   * (lane, triangle) work items are compacted into LDS with a ballot/mbcnt prefix sum over the
//     3-bit leaf counts, each owner also posts its instance-space ray;
//   * the items are tested 64 at a time, one item per lane, whatever lane they came from — a leaf
//     with 6 triangles no longer holds 63 other lanes hostage;
//   * each owner then folds its own results in leaf order with the reference's strict `t < closest`
//     rule and goes back to searching.
// Equivalence with the reference's sequential leaf loop (Raytracer.wgsl:474-482): a test is accepted
// there iff geometry passes, t > t_min and t < the running closest; the running closest never exceeds
// the closest at leaf entry, so testing every triangle against the leaf-entry bound in parallel and
// re-applying `t < running closest` in order during the fold makes exactly the same decisions.
// Per lane the sequence of visited nodes, tested triangles and tie-breaks is the reference's.
// ANY = shadow ray (first accepted hit ends the ray), else closest hit.
struct WaveWork {
  f4* rays;         // 64 x 2: {o.xyz, t_min} {d.xyz, bound at leaf entry}
  uint32_t* items;  // up to 64*7: (owner lane << 26) | triangle id; overwritten by the result t (f32 bits)
};
#define RT_WORK_BYTES_PER_WAVE (64 * 32 + 64 * 7 * 4)
#ifndef RT_WF_WAVES
#define RT_WF_WAVES 6
#endif
#ifndef RT_FLUSH_ITEMS
#define RT_FLUSH_ITEMS 24u  // queued triangle tests that trigger a flush; swept 1..128 on MI355X: flat optimum 16..32
                            // (fewer = partial 64-item chunks, more = lanes wait longer for their results)
#endif

template <bool ANY, bool COUNT>
__device__ __forceinline__ void traverse(const TravMem& M, const WaveWork& W, uint32_t blas_base, bool active, rt3 o,
                                         rt3 d, float t_min, float t_max, float& out_t, int32_t& out_tri,
                                         int32_t& out_inst, bool& out_any, uint32_t& n_nodes, uint32_t& n_tris) {
  const uint32_t lane = threadIdx.x & 63u;
  float closest = t_max;
  int32_t best_tri = -1, best_inst = -1;
  bool any = false;
  bool searching = active && blas_base != 0u;
  bool waiting = false;
  uint32_t leaf = 0u;
  LocalRay r = make_ray(o, d);
  const uint32_t tlas_end = rt_f2u(M.nodes[0].w);
  uint32_t curr = 0u, end = tlas_end, base = 0u, tlas_next = 0u;
  uint32_t cur_inst = 0u;
  bool in_blas = false;
  for (;;) {
#ifdef RT_WAVE_STATS
    {
      const bool any_search = __ballot(searching) != 0ull;
      if (COUNT && lane == 0u && any_search) n_nodes++;  // wave-level node steps
    }
#endif
    // ---- range exhausted (rare): leave the instance, or finish
    if (searching && curr >= end) {
      if (in_blas && tlas_next < tlas_end) {
        in_blas = false;  // back to the world-space ray and the TLAS cursor
        r = make_ray(o, d);
        curr = tlas_next;
        end = tlas_end;
        base = 0u;
      } else {
        searching = false;
      }
    }
    // ---- one node step for every searching lane (curr < end holds); select-based, two branches only
    if (searching) {
      const f4 lo = M.nodes[2 * curr], hi = M.nodes[2 * curr + 1];
#ifndef RT_WAVE_STATS
      if (COUNT) n_nodes++;
#endif
      const bool hit = hit_box4(lo, hi, r, t_min, closest);
      const uint32_t data = rt_f2u(hi.w);
      const bool leafhit = hit && data != 0u;
      uint32_t next = (hit && data == 0u) ? curr + 1u : base + rt_f2u(lo.w);
      const bool got_leaf = leafhit && in_blas;
      if (leafhit && !in_blas) {  // TLAS leaf: enter the instance
        cur_inst = data >> 3;
        uint32_t off;
        r = to_instance(M, cur_inst, o, d, off);
        tlas_next = next;
        base = blas_base + off;
        end = base + rt_f2u(M.nodes[2 * base].w);
        next = base;
        in_blas = true;
      }
      leaf = got_leaf ? data : leaf;
      waiting = got_leaf;
      searching = !got_leaf;
      curr = next;
    }
    // ---- flush the triangle queue?
    const unsigned long long smask = __ballot(searching);
    const unsigned long long wmask = __ballot(waiting);
    if ((smask | wmask) == 0ull) break;
    const uint32_t cnt = waiting ? (leaf & 7u) : 0u;
    const unsigned long long b0 = __ballot((cnt & 1u) != 0u), b1 = __ballot((cnt & 2u) != 0u), b2 = __ballot((cnt & 4u) != 0u);
    const uint32_t total = (uint32_t)__builtin_popcountll(b0) + 2u * (uint32_t)__builtin_popcountll(b1) +
                           4u * (uint32_t)__builtin_popcountll(b2);
    if (wmask != 0ull && (total >= RT_FLUSH_ITEMS || smask == 0ull)) {
      const uint32_t excl =
          __builtin_amdgcn_mbcnt_hi((uint32_t)(b0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b0, 0u)) +
          2u * __builtin_amdgcn_mbcnt_hi((uint32_t)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b1, 0u)) +
          4u * __builtin_amdgcn_mbcnt_hi((uint32_t)(b2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b2, 0u));
      const uint32_t first = leaf >> 3;
      if (waiting) {
        f4 ra, rb;
        ra.x = r.o.x; ra.y = r.o.y; ra.z = r.o.z; ra.w = t_min;
        rb.x = r.d.x; rb.y = r.d.y; rb.z = r.d.z; rb.w = closest;
        W.rays[2 * lane] = ra;
        W.rays[2 * lane + 1] = rb;
        const uint32_t tag = lane << 26;
#pragma unroll
        for (uint32_t i = 0; i < 7u; i++)
          if (i < cnt) W.items[excl + i] = tag | (first + i);
      }
      __builtin_amdgcn_wave_barrier();
      for (uint32_t c = 0; c < total; c += 64u) {
#ifdef RT_WAVE_STATS
        if (COUNT && lane == 0u) n_tris++;  // wave-level 64-item chunks
#endif
        const uint32_t j = c + lane;
        if (j < total) {
          const uint32_t it = W.items[j];
          const uint32_t owner = it >> 26, tri = it & 0x03ffffffu;
          f4 ra = W.rays[2 * owner], rb = W.rays[2 * owner + 1];
          LocalRay q;
          q.o = rt3_make(ra.x, ra.y, ra.z);
          q.d = rt3_make(rb.x, rb.y, rb.z);
          float t;
          bool ok = hit_tri_nb(M.tri_geom[3 * tri], M.tri_geom[3 * tri + 1], M.tri_geom[3 * tri + 2], q, ra.w, rb.w, t);
          W.items[j] = rt_f2u(ok ? t : -1.0f);
        }
      }
      __builtin_amdgcn_wave_barrier();
      if (waiting) {
        // fold this lane's results in leaf order (strict t < closest: the first of equal hits wins)
        bool stop = false;
#pragma unroll
        for (uint32_t i = 0; i < 7u; i++) {
          if (i < cnt && !stop) {
#ifndef RT_WAVE_STATS
            if (COUNT) n_tris++;
#endif
            const float t = rt_u2f(W.items[excl + i]);
            if (t > 0.0f && t < closest) {
              if (ANY) {
                any = true;
                stop = true;
              } else {
                closest = t;
                best_tri = (int32_t)(first + i);
                best_inst = (int32_t)cur_inst;
              }
            }
          }
        }
        waiting = false;
        searching = !(ANY && any);
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  out_t = closest;
  out_tri = best_tri;
  out_inst = best_inst;
  out_any = any;
}

struct PathState {
  uint32_t pixel, rng, depth, sample;
  rt3 ro, rd, throughput, radiance, col;
  float prev_pdf;
  bool specular;
  // current surface
  float hit_t;
  uint32_t tri, inst;
  rt3 normal, geom_n, albedo;
  rt2 tex_uv;
};

// surface frame of the hit (tri, inst) for the ray (ro, rd): Raytracer.wgsl:738-779
__device__ __forceinline__ void setup_surface(const DevScene& S, PathState& p, bool from_gbuffer, float gx, float gy,
                                              uint32_t galbedo) {
  InvRows m = load_inv_rows(S, p.inst);
  Bary b = barycentrics(S, p.tri, mul_point(m, p.ro), mul_dir(m, p.rd));
  float4 tidx = S.topo[5 * p.tri];
  uint32_t i0 = rt_f2u(tidx.x), i1 = rt_f2u(tidx.y), i2 = rt_f2u(tidx.z);
  float2 uv0 = S.uv[i0], uv1 = S.uv[i1], uv2 = S.uv[i2];
  p.tex_uv = rt2_make(uv0.x, uv0.y) * b.w + rt2_make(uv1.x, uv1.y) * b.u + rt2_make(uv2.x, uv2.y) * b.v;
  if (from_gbuffer) {
    p.hit_t = b.t;
    p.normal = unpack_normal(gx, gy);
    p.albedo = rt3_make(rt_from_unorm8(galbedo & 255u), rt_from_unorm8((galbedo >> 8) & 255u),
                        rt_from_unorm8((galbedo >> 16) & 255u));
  } else {
    rt3 ln = rt_normalize(xyz(S.nrm[i0]) * b.w + xyz(S.nrm[i1]) * b.u + xyz(S.nrm[i2]) * b.v);
    p.normal = rt_normalize(normal_to_world(m, ln));
    float4 nd0 = S.topo[5 * p.tri + 1], nd2 = S.topo[5 * p.tri + 3];
    p.albedo = xyz(nd0);
    if (nd2.x > -0.5f) p.albedo = p.albedo * sample_tex(S, p.tex_uv, rt_f2i32_sat(nd2.x));
    if (nd2.z > -0.5f) {
      rt3 n_map = sample_tex(S, p.tex_uv, rt_f2i32_sat(nd2.z)) * 2.0f - rt3_splat(1.0f);
      rt3 T = rt_normalize(b.e1);
      rt3 B = rt_normalize(rt_cross(ln, T));
      rt3 ln_mapped = rt_normalize(T * n_map.x + B * n_map.y + ln * n_map.z);
      p.normal = rt_normalize(normal_to_world(m, ln_mapped));
    }
  }
  p.geom_n = rt_normalize(normal_to_world(m, rt_normalize(rt_cross(b.e1, b.e2))));
}

// One bounce of ray_color for a path whose surface frame is ready (Raytracer.wgsl:656-728): emissive / MIS, the
// three NEE draws and the pending NEE term, BSDF sampling, throughput, ray offset, Russian roulette, depth limit.
// The shadow ray and the extension ray it asks for are traced by the caller (megakernel trip or wavefront stage).
struct BounceOut {
  bool want_shadow, want_extend, nee_valid, ended;
  rt3 sh_o, sh_d, nee;
  float sh_tmax;
};
__device__ __forceinline__ void shade_bounce(const DevScene& S, uint32_t light_count, uint32_t max_depth, PathState& p,
                                             BounceOut& o) {
  o.want_shadow = o.want_extend = o.nee_valid = false;
  o.sh_o = o.sh_d = o.nee = rt3_splat(0.0f);
  o.sh_tmax = 0.0f;
  float4 d0 = S.topo[5 * p.tri + 1], d1 = S.topo[5 * p.tri + 2], d2 = S.topo[5 * p.tri + 3], d3 = S.topo[5 * p.tri + 4];
  const uint32_t mat_type = rt_f2u32_sat(d0.w + 0.5f);
  const rt3 hit_p = p.ro + p.rd * p.hit_t;
  p.normal = (rt_dot(p.rd, p.normal) < 0.0f) ? p.normal : -p.normal;
  p.geom_n = (rt_dot(p.rd, p.geom_n) < 0.0f) ? p.geom_n : -p.geom_n;
  float metallic = d1.x, roughness = d1.y;
  if (d2.y > -0.5f) {
    rt3 mr = sample_tex(S, p.tex_uv, rt_f2i32_sat(d2.y));
    metallic *= mr.z;
    roughness *= mr.y;
  }
  roughness = rt_max(roughness, 0.005f);
  rt3 emissive = xyz(d3);
  if (d2.w > -0.5f) emissive = emissive * sample_tex(S, p.tex_uv, rt_f2i32_sat(d2.w));
  const rt3 f0 = rt_mix3(rt3_splat(0.04f), p.albedo, metallic);

  bool ended = false;
  if (mat_type == 3u || rt_length(emissive) > 1e-4f) {
    rt3 em_val = (mat_type == 3u) ? p.albedo : emissive;
    if (p.specular) {
      p.radiance = p.radiance + p.throughput * em_val;
    } else {
      p.radiance = p.radiance + p.throughput * em_val *
                                    power_heuristic(p.prev_pdf, light_pdf(S, light_count, p.tri, p.inst, p.hit_t, p.rd));
    }
    if (mat_type == 3u) ended = true;
  }
  if (!ended) {
    if (mat_type != 2u) {  // NEE: the 3 draws happen here, the shadow ray is traced below
      LightSample ls = sample_light(S, light_count, hit_p, p.rng);
      if (ls.pdf > 0.0f) {
        rt3 bsdf_val = rt3_splat(0.0f);
        float bsdf_pdf = 0.0f;
        if (mat_type == 0u) {
          bsdf_val = p.albedo / RT_PI;
          bsdf_pdf = rt_max(rt_dot(p.normal, ls.dir), 0.0f) / RT_PI;
        } else if (mat_type == 1u) {
          bsdf_val = eval_ggx(p.normal, -p.rd, ls.dir, roughness, f0);
          rt3 H = rt_normalize(-p.rd + ls.dir);
          bsdf_pdf = (ggx_d(rt_dot(p.normal, H), roughness * roughness) * rt_max(rt_dot(p.normal, H), 0.0f)) /
                     (4.0f * rt_max(rt_dot(-p.rd, H), 0.0f));
        }
        o.want_shadow = true;  // the reference traces the shadow ray before looking at bsdf_pdf
        o.sh_o = hit_p + p.geom_n * 1e-4f;
        o.sh_d = ls.dir;
        o.sh_tmax = ls.dist - 2e-4f;
        o.nee_valid = bsdf_pdf > 0.0f;
        if (o.nee_valid) {
          o.nee = p.throughput * bsdf_val * ls.L * power_heuristic(ls.pdf, bsdf_pdf) *
                rt_max(rt_dot(p.normal, ls.dir), 0.0f) / ls.pdf;
        }
      }
    }
    Scatter sc;
    if (mat_type == 0u) {
      sc = sample_diffuse(p.normal, p.albedo, p.rng);
    } else if (mat_type == 1u) {
      sc = sample_ggx(p.normal, -p.rd, roughness, f0, p.rng);
    } else {
      sc = sample_dielectric(p.rd, p.normal, d1.z, p.albedo, p.rng);
    }
    if (mat_type != 2u && rt_dot(sc.dir, p.geom_n) <= 0.0f) {
      sc.pdf = 0.0f;
      sc.throughput = rt3_splat(0.0f);
    }
    if (sc.pdf <= 0.0f || rt_length(sc.throughput) <= 0.0f) {
      ended = true;
    } else {
      p.throughput = p.throughput * sc.throughput;
      rt3 offset_n = (rt_dot(sc.dir, p.geom_n) > 0.0f) ? p.geom_n : -p.geom_n;
      p.ro = hit_p + offset_n * 1e-4f;
      p.rd = sc.dir;
      p.prev_pdf = sc.pdf;
      p.specular = sc.specular;
      if (p.depth > 3u) {
        float pr = rt_max(p.throughput.x, rt_max(p.throughput.y, p.throughput.z));
        if (rand_pcg(p.rng) > pr) {
          ended = true;
        } else {
          p.throughput = p.throughput / pr;
        }
      }
      if (!ended) {
        if (p.depth < max_depth - 1u) {
          o.want_extend = true;
        } else {
          ended = true;  // depth limit: the loop condition ends the path after this bounce
        }
      }
    }
  }
  o.ended = ended;
}

// number of 16-byte LDS slots the whole scene needs (traversal records + shading arrays)
__host__ __device__ inline size_t scene_lds_slots(uint32_t n_nodes, uint32_t n_tris, uint32_t n_inst, uint32_t n_verts,
                                                  uint32_t n_lights) {
  return (size_t)2 * n_nodes + (size_t)3 * n_tris + (size_t)4 * n_inst + (size_t)5 * n_tris + (size_t)2 * n_verts +
         ((size_t)n_verts + 1) / 2 + (size_t)9 * n_inst + ((size_t)n_lights + 1) / 2 + (size_t)4 * n_lights;
}

// Occupancy: the LDS-resident form is VALU-issue bound (3, 4, 5 waves/SIMD within 2 %), the global-memory form
// is latency bound and gains ~11 % from 6 waves/SIMD even with the spills that costs (measured on MI355X).
template <bool DETAIL, bool LDS>
__global__ __launch_bounds__(256, LDS ? 4 : 6) void k_pathtrace_persistent(DevScene Sg, DevFrame F, rt_scene_uniforms U,
                                                              uint32_t* __restrict__ ticket, uint32_t n_nodes_total,
                                                              uint32_t n_tris_total, uint32_t n_inst_total,
                                                              uint32_t n_verts_total,
                                                              const DevFrameSlot* __restrict__ slots, uint32_t n_slots) {
  // Batched dispatch (rt_compute_batch): the launch covers n_slots consecutive compute() frames. The work item is
  // one (frame, pixel): tickets enumerate (frame, tile) pairs, so a launch has n_slots times as many tickets and the
  // persistent waves stay fed and balanced even when a rank owns 1/8 of the image. With n_slots > 1 every item
  // writes its frame colour to F.frame_col and k_accumulate_frames adds the frames in frame order afterwards, which
  // makes the result bit-identical to n_slots separate dispatches; with n_slots == 1 the item accumulates directly.
  extern __shared__ f4 s_scene[];
  // per-wave triangle work queue at the start of LDS, staged scene after it
  WaveWork WW;
  {
    char* wbase = reinterpret_cast<char*>(s_scene) + (threadIdx.x >> 6) * RT_WORK_BYTES_PER_WAVE;
    WW.rays = reinterpret_cast<f4*>(wbase);
    WW.items = reinterpret_cast<uint32_t*>(wbase + 64 * 32);
  }
  f4* const s_records = s_scene + (4 * RT_WORK_BYTES_PER_WAVE) / 16;
  TravMem M;
  DevScene S = Sg;
  if (LDS) {
    // Small scene: the whole scene (traversal records AND the arrays shading reads) lives in LDS,
    // staged once per workgroup; only textures, the G-buffer and the accumulation buffer stay in HBM.
    f4* dst = s_records;
    auto stage = [&](const void* src, size_t slots) {
      const f4* g = reinterpret_cast<const f4*>(src);
      f4* base = dst;
      for (uint32_t i = threadIdx.x; i < slots; i += 256) base[i] = g[i];
      dst += slots;
      return base;
    };
    f4* ln = stage(Sg.nodes, (size_t)2 * n_nodes_total);
    f4* lt = stage(Sg.tri_geom, (size_t)3 * n_tris_total);
    f4* li = stage(Sg.inst_trav, (size_t)4 * n_inst_total);
    S.topo = reinterpret_cast<const float4*>(stage(Sg.topo, (size_t)5 * n_tris_total));
    S.pos = reinterpret_cast<const float4*>(stage(Sg.pos, n_verts_total));
    S.nrm = reinterpret_cast<const float4*>(stage(Sg.nrm, n_verts_total));
    // uv (8 B/vertex) and lights (8 B each): the device buffers are allocated with >= 16-byte slack
    S.uv = reinterpret_cast<const float2*>(stage(Sg.uv, ((size_t)n_verts_total + 1) / 2));
    S.inst = reinterpret_cast<const float4*>(stage(Sg.inst, (size_t)9 * n_inst_total));
    S.lights = reinterpret_cast<const uint2*>(stage(Sg.lights, ((size_t)Sg.n_lights + 1) / 2));
    S.light_rec = reinterpret_cast<const float4*>(stage(Sg.light_rec, (size_t)4 * Sg.n_lights));
    __syncthreads();
    M.nodes = ln;
    M.tri_geom = lt;
    M.inst_trav = li;
    S.nodes = reinterpret_cast<const float4*>(ln);
    S.tri_geom = reinterpret_cast<const float4*>(lt);
    S.inst_trav = reinterpret_cast<const float4*>(li);
  } else {
    M.nodes = reinterpret_cast<const f4*>(Sg.nodes);
    M.tri_geom = reinterpret_cast<const f4*>(Sg.tri_geom);
    M.inst_trav = reinterpret_cast<const f4*>(Sg.inst_trav);
  }

  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t tiles_x = (U.width + 7u) / 8u;
  // tickets enumerate only the tile rows this rank owns when the stripes are tile-aligned
  const uint32_t n_tiles = tiles_x * (F.own_period ? F.own_tile_rows : (U.height + 7u) / 8u);
  const rt3 cam_o = rt3_make(U.camera.origin[0], U.camera.origin[1], U.camera.origin[2]);
  const rt3 cam_ll = rt3_make(U.camera.lower_left[0], U.camera.lower_left[1], U.camera.lower_left[2]);
  const rt3 cam_h = rt3_make(U.camera.horizontal[0], U.camera.horizontal[1], U.camera.horizontal[2]);
  const rt3 cam_v = rt3_make(U.camera.vertical[0], U.camera.vertical[1], U.camera.vertical[2]);
  const float lens = U.camera.origin[3];

  // wave-uniform work cursor: pixels [tile_pos, 64) of tile `tile` are still unassigned
  uint32_t tile = 0xffffffffu, tile_pos = 64u;
  bool work_left = true;

  PathState p;
  uint32_t item_slot = 0u;  // frame of the batch the lane's current (frame, pixel) item belongs to
  bool alive = false;       // lane owns a running path
  bool have_pixel = false;  // lane owns a pixel whose samples are not all done
  uint32_t cnt_ext = 0, cnt_shadow = 0, cnt_nodes = 0, cnt_tris = 0, cnt_shaded = 0;
  p.pixel = 0; p.rng = 0; p.depth = 0; p.sample = 0; p.prev_pdf = 0.0f; p.specular = true; p.hit_t = 0.0f;
  p.tri = 0; p.inst = 0;
  p.ro = p.rd = p.throughput = p.radiance = p.col = p.normal = p.geom_n = p.albedo = rt3_splat(0.0f);
  p.tex_uv = rt2_make(0.0f, 0.0f);

  for (;;) {
    // ------------------------------------------------------------ regenerate
    // (a) wave-wide: every lane without a pixel takes the next unassigned one of the wave's tile.
    //     All lanes execute this loop (busy lanes with need = false) so that the wave-uniform cursor
    //     (tile, tile_pos, work_left) stays identical in every lane.
    {
      bool need = !alive && !have_pixel;
      for (;;) {
        const unsigned long long mask = __ballot(need);
        if (mask == 0ull || !work_left) break;
        if (tile_pos >= 64u) {
          const int leader = __builtin_ctzll(mask);
          uint32_t t = 0;
          if (lane == (uint32_t)leader) t = atomicAdd(ticket, 1u);
          t = __shfl(t, leader, 64);
          if (t >= n_tiles * n_slots) {
            work_left = false;
            break;
          }
          tile = t;  // frame-major ticket: frame = t / n_tiles, tile = t % n_tiles
          tile_pos = 0u;
        }
        // rank of this lane among the needy lanes
        const uint32_t rank =
            __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
        const uint32_t slot = tile_pos + rank;
        if (need && slot < 64u) {
          const uint32_t tile_in_frame = tile % n_tiles;
          uint32_t trow = tile_in_frame / tiles_x;
          if (F.own_period) trow = (trow / F.own_run) * F.own_period + F.own_first + (trow % F.own_run);
          const uint32_t x = (tile_in_frame % tiles_x) * 8u + (slot & 7u);
          const uint32_t y = trow * 8u + (slot >> 3);
          need = false;
          if (x < U.width && y < U.height && owns_row(F, y)) {
            have_pixel = true;
            p.pixel = y * U.width + x;
            p.sample = 0u;
            item_slot = tile / n_tiles;
            p.col = rt3_splat(0.0f);
          }
        }
        tile_pos += (uint32_t)__builtin_popcountll(mask);
      }
    }
    // (b) start the next sample of the owned pixel: camera ray + depth-0 surface from the G-buffer
    if (!alive && have_pixel) {
      const uint32_t x = p.pixel % U.width, y = p.pixel / U.width;
      const DevFrameSlot slot = slots[item_slot];
      p.rng = init_rng(p.pixel, slot.frame_count * F.spp + p.sample);
      rt3 off = rt3_splat(0.0f);
      if (lens > 0.0f) {
        float r = rt_sqrt(rand_pcg(p.rng));
        float theta = RT_TWO_PI * rand_pcg(p.rng);
        float st, ct;
        rt_sincos(theta, &st, &ct);
        rt3 rdk = lens * rt3_make(r * ct, r * st, 0.0f);
        rt3 cu = rt3_make(U.camera.u[0], U.camera.u[1], U.camera.u[2]);
        rt3 cv = rt3_make(U.camera.v[0], U.camera.v[1], U.camera.v[2]);
        off = cu * rdk.x + cv * rdk.y;
      }
      float u = ((float)x + 0.5f + slot.jitter_x * (float)U.width) / (float)U.width;
      float v = 1.0f - ((float)y + 0.5f + slot.jitter_y * (float)U.height) / (float)U.height;
      p.rd = cam_ll + u * cam_h + v * cam_v - cam_o - off;
      p.ro = cam_o + off;
      p.throughput = rt3_splat(1.0f);
      p.radiance = rt3_splat(0.0f);
      p.prev_pdf = 0.0f;
      p.specular = true;
      p.depth = 0u;
      // background pixel (or MAX_DEPTH = 0): the sample is black and ends at once
      if (!(slot.depth[p.pixel] >= 1.0f) && F.max_depth != 0u) {
        float4 g = slot.normal_id[p.pixel];
        p.tri = rt_f2u(g.z);
        p.inst = rt_f2u(g.w);
        setup_surface(S, p, true, g.x, g.y, slot.albedo[p.pixel]);
        alive = true;
      }
    }
    const bool running = alive;
    bool path_done = have_pixel && !alive;  // background sample ends immediately

    // ------------------------------------------------------------ shade one bounce
    bool want_shadow = false, want_extend = false;
    bool nee_valid = false;
    rt3 sh_o = rt3_splat(0.0f), sh_d = rt3_splat(0.0f), nee = rt3_splat(0.0f);
    float sh_tmax = 0.0f;
#ifdef RT_WAVE_STATS
    if (DETAIL && lane == 0u) cnt_shaded++;  // wave-level outer trips
#endif
    if (running) {
#ifndef RT_WAVE_STATS
      if (DETAIL) cnt_shaded++;
#endif
      BounceOut bo;
      shade_bounce(S, U.light_count, F.max_depth, p, bo);
      want_shadow = bo.want_shadow;
      want_extend = bo.want_extend;
      nee_valid = bo.nee_valid;
      sh_o = bo.sh_o;
      sh_d = bo.sh_d;
      sh_tmax = bo.sh_tmax;
      nee = bo.nee;
      const bool ended = bo.ended;
      if (ended) path_done = true;
    }

#ifdef RT_EXP_NOSHADOW
    want_shadow = false;  // timing experiment only
#endif
#ifdef RT_EXP_NOEXT
    if (want_extend) { want_extend = false; path_done = true; }  // timing experiment only
#endif
    // ------------------------------------------------------------ shadow rays (any hit)
    if (__ballot(want_shadow) != 0ull) {
      float t_;
      int32_t a_, b_;
      bool occluded;
      traverse<true, DETAIL>(M, WW, U.blas_base_idx, want_shadow, sh_o, sh_d, RT_T_MIN, sh_tmax, t_, a_, b_, occluded,
                             cnt_nodes, cnt_tris);
      if (want_shadow) {
        cnt_shadow++;
        if (!occluded && nee_valid) p.radiance = p.radiance + nee;  // nothing is added when bsdf_pdf <= 0
      }
    }

    // ------------------------------------------------------------ extension rays (closest hit)
    if (__ballot(want_extend) != 0ull) {
      float t_;
      int32_t tri_, inst_;
      bool any_;
      traverse<false, DETAIL>(M, WW, U.blas_base_idx, want_extend, p.ro, p.rd, RT_T_MIN, RT_T_MAX, t_, tri_, inst_, any_,
                              cnt_nodes, cnt_tris);
      if (want_extend) {
        cnt_ext++;
        if (inst_ < 0) {
          path_done = true;
        } else {
          p.hit_t = t_;
          p.tri = (uint32_t)tri_;
          p.inst = (uint32_t)inst_;
          setup_surface(S, p, false, 0.0f, 0.0f, 0u);
          p.depth++;
        }
      }
    }

    // ------------------------------------------------------------ sample / pixel finished
    if (path_done) {
      alive = false;
      p.col = p.col + p.radiance;
      p.sample++;
      if (p.sample >= F.spp) {  // the item's last sample: Raytracer.wgsl:811-818
        rt3 c = p.col / (float)F.spp;
        if (F.frame_col) {
          // batched: park the frame colour; k_accumulate_frames adds the frames in order
          F.frame_col[(size_t)item_slot * ((size_t)U.width * U.height) + p.pixel] = make_float4(c.x, c.y, c.z, 1.0f);
        } else {
          float4 acc = make_float4(c.x, c.y, c.z, 1.0f);
          if (slots[0].frame_count > 1u) {
            float4 prev = F.accum[p.pixel];
            acc = make_float4(prev.x + c.x, prev.y + c.y, prev.z + c.z, prev.w + 1.0f);
          }
          F.accum[p.pixel] = acc;
        }
        have_pixel = false;
      }
    }
    if (!work_left && __ballot(alive || have_pixel) == 0ull) break;
  }

  // counters: one flush per persistent wave
  LaneCounters c;
  c.primary = 0;
  c.extension = cnt_ext;
  c.shadow = cnt_shadow;
  c.nodes = cnt_nodes;
  c.tris = cnt_tris;
  c.shaded = cnt_shaded;
  flush_counters<DETAIL>(c, F.counters, blockIdx.x * 4u + (threadIdx.x >> 6));
}

// ================================================================== path tracer, wavefront form
// For scenes whose traversal records do not fit LDS (hundreds of thousands of triangles, a thousand instances) a ray
// visits 70+ nodes with a long tail, and the per-trip lockstep of the persistent kernel leaves 60 % of the lanes idle
// while they wait on L2 / Infinity-Cache latency.  The wavefront form splits a bounce into stages with the path state in
// HBM (84 B per path, 288 GB to spare):
//   k_wf_shade   one lane per live path: surface frame + shade_bounce(); appends the shadow ray and the extension ray
//                to device queues (wave-aggregated atomics), finishes paths that end
//   k_wf_trace   persistent waves, RAY-level regeneration: a lane that finishes its ray writes the result and pulls the
//                next ray from the queue (batched, >= RT_WF_REFILL lanes), so the slowest ray no longer holds 63 lanes;
//                same node step / LDS triangle queue as traverse()
// Stages of one depth run as separate launches in stream order; the host enqueues all depths without reading anything
// back (queue sizes stay on the device).  Per path the arithmetic, RNG order and f32 addition order are unchanged, so
// the result is bit-identical to the other forms; frame colours go through frame_col + k_accumulate_frames.
// Restriction: SPP == 1 (the reference's default); other SPP values use the persistent kernel.
#define RT_WF_REFILL 16

// Device queues are filled and drained in chunks of RT_WF_CHUNK entries: a wave reserves a chunk with ONE atomic
// and then appends with ballot/mbcnt ranks (a queue counter is a single address: ~88 atomics/us chip-wide, so one atomic
// per wave-append or per 16-ray pull caps a stage at a few Grays/s). Unused tail entries of a chunk hold RT_WF_INVALID.
#define RT_WF_CHUNK 256u
#define RT_WF_INVALID 0xffffffffu
struct WaveQueueWriter {
  uint32_t pos, end;  // wave-uniform cursor into the current chunk
};
// returns the slot for lanes with want == true (RT_WF_INVALID otherwise); call from wave-uniform control flow
__device__ __forceinline__ uint32_t wq_append(WaveQueueWriter& w, uint32_t* counter, uint32_t* ids, bool want) {
  const uint32_t lane = threadIdx.x & 63u;
  const unsigned long long mask = __ballot(want);
  if (mask == 0ull) return RT_WF_INVALID;
  const uint32_t n = (uint32_t)__builtin_popcountll(mask);
  if (w.pos + n > w.end) {
    for (uint32_t i = w.pos + lane; i < w.end; i += 64u) ids[i] = RT_WF_INVALID;
    uint32_t b = 0;
    if (lane == 0u) b = atomicAdd(counter, RT_WF_CHUNK);
    b = __shfl(b, 0, 64);
    w.pos = b;
    w.end = b + RT_WF_CHUNK;
  }
  const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
  const uint32_t slot = w.pos + rank;
  w.pos += n;
  return want ? slot : RT_WF_INVALID;
}
__device__ __forceinline__ void wq_finish(const WaveQueueWriter& w, uint32_t* ids) {
  const uint32_t lane = threadIdx.x & 63u;
  for (uint32_t i = w.pos + lane; i < w.end; i += 64u) ids[i] = RT_WF_INVALID;
}

__device__ __forceinline__ void wf_store_path(const WfState& W, uint32_t id, const PathState& p, uint32_t flags,
                                              rt3 nee) {
  W.a[id] = make_float4(p.ro.x, p.ro.y, p.ro.z, p.hit_t);
  W.b[id] = make_float4(p.rd.x, p.rd.y, p.rd.z, p.prev_pdf);
  W.c[id] = make_float4(p.throughput.x, p.throughput.y, p.throughput.z, rt_u2f(p.rng));
  W.d[id] = make_float4(p.radiance.x, p.radiance.y, p.radiance.z, rt_u2f(flags));
  W.e[id] = make_float4(nee.x, nee.y, nee.z, rt_u2f(p.tri));
}

template <bool FIRST, bool DETAIL>
__global__ __launch_bounds__(256) void k_wf_shade(DevScene S, DevFrame F, rt_scene_uniforms U, WfState W, WfQueues Q,
                                                  const DevFrameSlot* __restrict__ slots, uint32_t n_slots,
                                                  uint32_t depth) {
  const uint32_t npx = U.width * U.height;
  uint32_t* cnt = Q.counters + 8u * depth;
  const uint32_t count = FIRST ? npx * n_slots : cnt[0];
  const uint32_t* active_in = Q.active[depth & 1u];
  uint32_t cnt_shaded = 0;
  WaveQueueWriter wq_shadow = {0u, 0u}, wq_ext = {0u, 0u};
  // wave-uniform loop (every lane of a wave takes part in the queue appends)
  for (uint32_t base_idx = (blockIdx.x * 256u + (threadIdx.x & ~63u)); base_idx < count; base_idx += gridDim.x * 256u) {
    const uint32_t idx = base_idx + (threadIdx.x & 63u);
    bool live = idx < count;
    uint32_t id = 0u;
    if (live) {
      id = FIRST ? idx : active_in[idx];
      live = id != RT_WF_INVALID;
    }
    BounceOut bo;
    bo.want_shadow = bo.want_extend = bo.nee_valid = bo.ended = false;
    bo.sh_o = bo.sh_d = bo.nee = rt3_splat(0.0f);
    bo.sh_tmax = 0.0f;
    PathState p;
    p.col = rt3_splat(0.0f);
    p.sample = 0u;
    p.pixel = id % npx;
    p.tri = p.inst = p.depth = p.rng = 0u;
    p.hit_t = p.prev_pdf = 0.0f;
    p.specular = true;
    p.ro = p.rd = p.throughput = p.radiance = p.normal = p.geom_n = p.albedo = rt3_splat(0.0f);
    p.tex_uv = rt2_make(0.0f, 0.0f);
    if (live) {
    if (FIRST) {
      const uint32_t x = p.pixel % U.width, y = p.pixel / U.width;
      if (!owns_row(F, y)) live = false;
      const DevFrameSlot slot = slots[id / npx];
      p.rng = init_rng(p.pixel, slot.frame_count);  // SPP == 1: frame_count * SPP + 0
      rt3 cam_o = rt3_make(U.camera.origin[0], U.camera.origin[1], U.camera.origin[2]);
      rt3 off = rt3_splat(0.0f);
      const float lens = U.camera.origin[3];
      if (lens > 0.0f) {
        float r = rt_sqrt(rand_pcg(p.rng));
        float theta = RT_TWO_PI * rand_pcg(p.rng);
        float st, ct;
        rt_sincos(theta, &st, &ct);
        rt3 rdk = lens * rt3_make(r * ct, r * st, 0.0f);
        rt3 cu = rt3_make(U.camera.u[0], U.camera.u[1], U.camera.u[2]);
        rt3 cv = rt3_make(U.camera.v[0], U.camera.v[1], U.camera.v[2]);
        off = cu * rdk.x + cv * rdk.y;
      }
      rt3 cam_ll = rt3_make(U.camera.lower_left[0], U.camera.lower_left[1], U.camera.lower_left[2]);
      rt3 cam_h = rt3_make(U.camera.horizontal[0], U.camera.horizontal[1], U.camera.horizontal[2]);
      rt3 cam_v = rt3_make(U.camera.vertical[0], U.camera.vertical[1], U.camera.vertical[2]);
      float u = ((float)x + 0.5f + slot.jitter_x * (float)U.width) / (float)U.width;
      float v = 1.0f - ((float)y + 0.5f + slot.jitter_y * (float)U.height) / (float)U.height;
      p.rd = cam_ll + u * cam_h + v * cam_v - cam_o - off;
      p.ro = cam_o + off;
      p.throughput = rt3_splat(1.0f);
      p.radiance = rt3_splat(0.0f);
      p.prev_pdf = 0.0f;
      p.specular = true;
      p.depth = 0u;
      if (live && (slot.depth[p.pixel] >= 1.0f || F.max_depth == 0u)) {  // background (or MAX_DEPTH = 0): black sample
        F.frame_col[id] = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
        live = false;
      }
      if (live) {
        float4 g = slot.normal_id[p.pixel];
        p.tri = rt_f2u(g.z);
        p.inst = rt_f2u(g.w);
        setup_surface(S, p, true, g.x, g.y, slot.albedo[p.pixel]);
      }
    } else {
      const float4 a = W.a[id], b = W.b[id], c = W.c[id], d = W.d[id], e = W.e[id];
      p.ro = xyz(a);
      p.hit_t = a.w;
      p.rd = xyz(b);
      p.prev_pdf = b.w;
      p.throughput = xyz(c);
      p.rng = rt_f2u(c.w);
      p.radiance = xyz(d);
      const uint32_t fl = rt_f2u(d.w);
      p.depth = fl & 0xffu;
      p.specular = (fl & WF_FLAG_SPECULAR) != 0u;
      p.tri = rt_f2u(e.w);
      p.inst = W.inst[id];
      setup_surface(S, p, false, 0.0f, 0.0f, 0u);
    }
    if (live) {
      if (DETAIL) cnt_shaded++;
      shade_bounce(S, U.light_count, F.max_depth, p, bo);
    }
    }  // if (live) — everything below runs for the whole wave
    const uint32_t sslot = wq_append(wq_shadow, &cnt[1], Q.shadow_ids, live && bo.want_shadow);
    if (sslot != RT_WF_INVALID) {
      Q.shadow_ids[sslot] = id;
      Q.shadow_rays[2 * sslot] = make_float4(bo.sh_o.x, bo.sh_o.y, bo.sh_o.z, bo.sh_tmax);
      Q.shadow_rays[2 * sslot + 1] = make_float4(bo.sh_d.x, bo.sh_d.y, bo.sh_d.z, 0.0f);
    }
    const uint32_t eslot = wq_append(wq_ext, &cnt[2], Q.ext_ids, live && bo.want_extend);
    if (eslot != RT_WF_INVALID) Q.ext_ids[eslot] = id;
    if (live) {
      if (bo.ended && !bo.want_shadow) {
        F.frame_col[id] = make_float4(p.radiance.x, p.radiance.y, p.radiance.z, 1.0f);  // SPP == 1: col / 1
      } else {
        const uint32_t flags = (p.depth & 0xffu) | (p.specular ? WF_FLAG_SPECULAR : 0u) |
                               (bo.ended ? WF_FLAG_ENDED : 0u) | (bo.nee_valid ? WF_FLAG_NEE_VALID : 0u);
        wf_store_path(W, id, p, flags, bo.nee);
      }
    }
  }
  wq_finish(wq_shadow, Q.shadow_ids);
  wq_finish(wq_ext, Q.ext_ids);
  if (DETAIL) {
    LaneCounters c = {0, 0, 0, 0, 0, cnt_shaded};
    flush_counters<true>(c, F.counters, blockIdx.x * 4u + (threadIdx.x >> 6));
  }
}

// Persistent ray tracer over a device queue. ANY: shadow rays (result: NEE term added, ended paths finished);
// else extension rays (result: hit stored + path appended to the next depth's active list, or path finished on a miss).
template <bool ANY, bool DETAIL, bool LDS>
__global__ __launch_bounds__(256, LDS ? 4 : RT_WF_WAVES) void k_wf_trace(DevScene Sg, DevFrame F, rt_scene_uniforms U, WfState Ws,
                                                            WfQueues Q, uint32_t depth, uint32_t n_nodes_total,
                                                            uint32_t n_tris_total, uint32_t n_inst_total) {
  extern __shared__ f4 s_scene[];
  WaveWork W;
  {
    char* wbase = reinterpret_cast<char*>(s_scene) + (threadIdx.x >> 6) * RT_WORK_BYTES_PER_WAVE;
    W.rays = reinterpret_cast<f4*>(wbase);
    W.items = reinterpret_cast<uint32_t*>(wbase + 64 * 32);
  }
  TravMem M;
  if (LDS) {
    f4* dst = s_scene + (4 * RT_WORK_BYTES_PER_WAVE) / 16;
    auto stage = [&](const void* src, size_t slots) {
      const f4* g = reinterpret_cast<const f4*>(src);
      f4* base = dst;
      for (uint32_t i = threadIdx.x; i < slots; i += 256) base[i] = g[i];
      dst += slots;
      return base;
    };
    M.nodes = stage(Sg.nodes, (size_t)2 * n_nodes_total);
    M.tri_geom = stage(Sg.tri_geom, (size_t)3 * n_tris_total);
    M.inst_trav = stage(Sg.inst_trav, (size_t)4 * n_inst_total);
    __syncthreads();
  } else {
    M.nodes = reinterpret_cast<const f4*>(Sg.nodes);
    M.tri_geom = reinterpret_cast<const f4*>(Sg.tri_geom);
    M.inst_trav = reinterpret_cast<const f4*>(Sg.inst_trav);
  }
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t blas_base = U.blas_base_idx;
  uint32_t* cnt = Q.counters + 8u * depth;
  const uint32_t n_rays = ANY ? cnt[1] : cnt[2];
  uint32_t* head = ANY ? &cnt[3] : &cnt[4];
  uint32_t* next_active = Q.active[(depth + 1u) & 1u];
  uint32_t* next_count = Q.counters + 8u * (depth + 1u);
  const uint32_t tlas_end = blas_base ? rt_f2u(M.nodes[0].w) : 0u;

  // per-lane ray + traversal state
  bool have_ray = false, searching = false, waiting = false, in_blas = false, any = false;
  uint32_t id = 0u, leaf = 0u, curr = 0u, end = 0u, base = 0u, tlas_next = 0u, cur_inst = 0u;
  rt3 o = rt3_splat(0.0f), d = rt3_splat(0.0f);
  float t_max = 0.0f, closest = 0.0f;
  int32_t best_tri = -1, best_inst = -1;
  LocalRay r = make_ray(rt3_splat(1.0f), rt3_splat(1.0f));
  bool queue_left = true;
  uint32_t chunk_pos = 0u, chunk_end = 0u;  // wave-uniform cursor into the chunk of the input queue this wave holds
  WaveQueueWriter wq_next = {0u, 0u};       // output: the next depth's active list (extension rays only)
  uint32_t n_nodes = 0, n_tris = 0, n_traced = 0;

  for (;;) {
    // ---- retire finished rays and pull new ones (batched: a block that runs for one lane costs as much as for 64)
    const bool done = have_ray && !searching && !waiting;
    const bool idle = !have_ray || done;
    const unsigned long long idle_m = __ballot(idle), done_m = __ballot(done);
    const unsigned long long busy_m = __ballot(searching || waiting);
    if (idle_m != 0ull &&
        ((uint32_t)__builtin_popcountll(done_m) >= RT_WF_REFILL ||
         (queue_left && (uint32_t)__builtin_popcountll(idle_m) >= RT_WF_REFILL) || busy_m == 0ull)) {
      bool push_next = false;
      if (done) {
        if (ANY) {
          float4 dd = Ws.d[id];
          const uint32_t fl = rt_f2u(dd.w);
          if (!any && (fl & WF_FLAG_NEE_VALID) != 0u) {
            const float4 e = Ws.e[id];
            dd.x = dd.x + e.x;  // radiance += pending NEE term (nothing is added when bsdf_pdf <= 0)
            dd.y = dd.y + e.y;
            dd.z = dd.z + e.z;
          }
          if ((fl & WF_FLAG_ENDED) != 0u)
            F.frame_col[id] = make_float4(dd.x, dd.y, dd.z, 1.0f);
          else
            Ws.d[id] = dd;
        } else {
          if (best_inst < 0) {  // miss: the path ends with what it has
            const float4 dd = Ws.d[id];
            F.frame_col[id] = make_float4(dd.x, dd.y, dd.z, 1.0f);
          } else {
            float4 a = Ws.a[id];
            a.w = closest;
            Ws.a[id] = a;
            float4 e = Ws.e[id];
            e.w = rt_u2f((uint32_t)best_tri);
            Ws.e[id] = e;
            Ws.inst[id] = (uint32_t)best_inst;
            float4 dd = Ws.d[id];
            const uint32_t fl = rt_f2u(dd.w);
            dd.w = rt_u2f((fl & ~0xffu) | (((fl & 0xffu) + 1u) & 0xffu));  // depth++
            Ws.d[id] = dd;
            push_next = true;
          }
        }
        have_ray = false;
      }
      if (!ANY) {
        const uint32_t slot = wq_append(wq_next, &next_count[0], next_active, push_next);
        if (slot != RT_WF_INVALID) next_active[slot] = id;
      }
      // pull: needy lanes take consecutive entries of the wave's chunk; a new chunk costs one atomic
      const bool need = !have_ray;
      const unsigned long long need_m = __ballot(need);
      if (queue_left && need_m != 0ull) {
        if (chunk_pos >= chunk_end) {
          uint32_t bq = 0;
          if (lane == 0u) bq = atomicAdd(head, RT_WF_CHUNK);
          bq = __shfl(bq, 0, 64);
          if (bq >= n_rays) {
            queue_left = false;
          } else {
            chunk_pos = bq;
            chunk_end = bq + RT_WF_CHUNK < n_rays ? bq + RT_WF_CHUNK : n_rays;
          }
        }
        if (queue_left) {
          const uint32_t rank =
              __builtin_amdgcn_mbcnt_hi((uint32_t)(need_m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need_m, 0u));
          const uint32_t qi = chunk_pos + rank;
          chunk_pos += (uint32_t)__builtin_popcountll(need_m);
          if (need && qi < chunk_end) {
            const uint32_t rid = ANY ? Q.shadow_ids[qi] : Q.ext_ids[qi];
            if (rid != RT_WF_INVALID) {
              id = rid;
              if (ANY) {
                const float4 r0 = Q.shadow_rays[2 * qi], r1 = Q.shadow_rays[2 * qi + 1];
                o = xyz(r0);
                d = xyz(r1);
                t_max = r0.w;
              } else {
                o = xyz(Ws.a[id]);
                d = xyz(Ws.b[id]);
                t_max = RT_T_MAX;
              }
              n_traced++;
              have_ray = true;
              closest = t_max;
              best_tri = -1;
              best_inst = -1;
              any = false;
              r = make_ray(o, d);
              curr = 0u;
              end = tlas_end;
              base = 0u;
              in_blas = false;
              searching = blas_base != 0u;
              waiting = false;
            }
          }
        }
      }
    }
    if (!queue_left && __ballot(have_ray) == 0ull) break;  // queue exhausted and every ray retired

    // ---- range exhausted: leave the instance, or finish the ray
    if (searching && curr >= end) {
      if (in_blas && tlas_next < tlas_end) {
        in_blas = false;
        r = make_ray(o, d);
        curr = tlas_next;
        end = tlas_end;
        base = 0u;
      } else {
        searching = false;
      }
    }
    // ---- one node step
    if (searching) {
      const f4 lo = M.nodes[2 * curr], hi = M.nodes[2 * curr + 1];
      if (DETAIL) n_nodes++;
      const bool hit = hit_box4(lo, hi, r, RT_T_MIN, closest);
      const uint32_t data = rt_f2u(hi.w);
      const bool leafhit = hit && data != 0u;
      uint32_t next = (hit && data == 0u) ? curr + 1u : base + rt_f2u(lo.w);
      const bool got_leaf = leafhit && in_blas;
      if (leafhit && !in_blas) {
        cur_inst = data >> 3;
        uint32_t off;
        r = to_instance(M, cur_inst, o, d, off);
        tlas_next = next;
        base = blas_base + off;
        end = base + rt_f2u(M.nodes[2 * base].w);
        next = base;
        in_blas = true;
      }
      leaf = got_leaf ? data : leaf;
      waiting = got_leaf;
      searching = !got_leaf;
      curr = next;
    }
    // ---- flush the triangle queue?
    const unsigned long long smask = __ballot(searching);
    const unsigned long long wmask = __ballot(waiting);
    const uint32_t cntl = waiting ? (leaf & 7u) : 0u;
    const unsigned long long b0 = __ballot((cntl & 1u) != 0u), b1 = __ballot((cntl & 2u) != 0u), b2 = __ballot((cntl & 4u) != 0u);
    const uint32_t total = (uint32_t)__builtin_popcountll(b0) + 2u * (uint32_t)__builtin_popcountll(b1) +
                           4u * (uint32_t)__builtin_popcountll(b2);
    if (wmask != 0ull && (total >= RT_FLUSH_ITEMS || smask == 0ull)) {
      const uint32_t excl =
          __builtin_amdgcn_mbcnt_hi((uint32_t)(b0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b0, 0u)) +
          2u * __builtin_amdgcn_mbcnt_hi((uint32_t)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b1, 0u)) +
          4u * __builtin_amdgcn_mbcnt_hi((uint32_t)(b2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b2, 0u));
      const uint32_t first = leaf >> 3;
      if (waiting) {
        f4 ra, rb;
        ra.x = r.o.x; ra.y = r.o.y; ra.z = r.o.z; ra.w = RT_T_MIN;
        rb.x = r.d.x; rb.y = r.d.y; rb.z = r.d.z; rb.w = closest;
        W.rays[2 * lane] = ra;
        W.rays[2 * lane + 1] = rb;
        const uint32_t tag = lane << 26;
#pragma unroll
        for (uint32_t i = 0; i < 7u; i++)
          if (i < cntl) W.items[excl + i] = tag | (first + i);
      }
      __builtin_amdgcn_wave_barrier();
      for (uint32_t c = 0; c < total; c += 64u) {
        const uint32_t j = c + lane;
        if (j < total) {
          const uint32_t it = W.items[j];
          const uint32_t owner = it >> 26, tri = it & 0x03ffffffu;
          f4 ra = W.rays[2 * owner], rb = W.rays[2 * owner + 1];
          LocalRay q;
          q.o = rt3_make(ra.x, ra.y, ra.z);
          q.d = rt3_make(rb.x, rb.y, rb.z);
          float t;
          bool ok = hit_tri_nb(M.tri_geom[3 * tri], M.tri_geom[3 * tri + 1], M.tri_geom[3 * tri + 2], q, ra.w, rb.w, t);
          W.items[j] = rt_f2u(ok ? t : -1.0f);
        }
      }
      __builtin_amdgcn_wave_barrier();
      if (waiting) {
        bool stop = false;
#pragma unroll
        for (uint32_t i = 0; i < 7u; i++) {
          if (i < cntl && !stop) {
            if (DETAIL) n_tris++;
            const float t = rt_u2f(W.items[excl + i]);
            if (t > 0.0f && t < closest) {
              if (ANY) {
                any = true;
                stop = true;
              } else {
                closest = t;
                best_tri = (int32_t)(first + i);
                best_inst = (int32_t)cur_inst;
              }
            }
          }
        }
        waiting = false;
        searching = !stop;
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (!ANY) wq_finish(wq_next, next_active);
  LaneCounters c = {0, ANY ? 0u : n_traced, ANY ? n_traced : 0u, n_nodes, n_tris, 0};
  flush_counters<DETAIL>(c, F.counters, blockIdx.x * 4u + (threadIdx.x >> 6));
}

// Ordered accumulation of a batched dispatch: acc = (frame_count > 1 ? acc : 0) + (col_f, 1) for f = 0..n-1, the
// exact sequence of f32 additions n separate dispatches perform (Raytracer.wgsl:813-818).
__global__ __launch_bounds__(256) void k_accumulate_frames(DevFrame F, const DevFrameSlot* __restrict__ slots,
                                                           uint32_t n_slots, uint32_t width, uint32_t height) {
  const uint32_t p = blockIdx.x * 256u + threadIdx.x;
  const uint32_t npx = width * height;
  if (p >= npx || !owns_row(F, p / width)) return;
  float4 acc = F.accum[p];
  for (uint32_t f = 0; f < n_slots; f++) {
    const float4 c = F.frame_col[(size_t)f * npx + p];
    if (slots[f].frame_count > 1u)
      acc = make_float4(acc.x + c.x, acc.y + c.y, acc.z + c.z, acc.w + 1.0f);
    else
      acc = make_float4(c.x, c.y, c.z, 1.0f);
  }
  F.accum[p] = acc;
}

// ===================================================================== texture ingest
// One 1024 x 1024 layer from a w x h RGBA8 image (ResourceManager.ts:164-196): one thread per destination texel,
// four source texels each; rows of a wave are contiguous in the destination.  src == nullptr: white fallback bitmap.
__global__ __launch_bounds__(256) void k_resize_texture(const uint32_t* __restrict__ src, uint32_t w, uint32_t h,
                                                        uint32_t* __restrict__ dst) {
  const uint32_t x = blockIdx.x * 256u + threadIdx.x, y = blockIdx.y;
  if (x >= RT_TEX_SIZE || y >= RT_TEX_SIZE) return;
  uint32_t out = 0xffffffffu;
  if (src) {
    uint32_t x0, x1, y0, y1;
    const float fx = rt_resize_coord(x, w, RT_TEX_SIZE, &x0, &x1);
    const float fy = rt_resize_coord(y, h, RT_TEX_SIZE, &y0, &y1);
    const uint32_t c00 = src[(size_t)y0 * w + x0], c10 = src[(size_t)y0 * w + x1];
    const uint32_t c01 = src[(size_t)y1 * w + x0], c11 = src[(size_t)y1 * w + x1];
    out = 0u;
#pragma unroll
    for (uint32_t k = 0; k < 32u; k += 8u)
      out |= rt_bilinear_u8((c00 >> k) & 255u, (c10 >> k) & 255u, (c01 >> k) & 255u, (c11 >> k) & 255u, fx, fy) << k;
  }
  dst[(size_t)y * RT_TEX_SIZE + x] = out;
}

// ===================================================================== post process
__device__ __forceinline__ rt3 pp_radiance(const DevPost& P, const rt_scene_uniforms& U, int cx, int cy) {  // :41-47
  int x = cx < 0 ? 0 : (cx > (int)U.width - 1 ? (int)U.width - 1 : cx);
  int y = cy < 0 ? 0 : (cy > (int)U.height - 1 ? (int)U.height - 1 : cy);
  float4 a = P.accum[(size_t)y * U.width + (size_t)x];
  if (a.w <= 0.0f) return rt3_splat(0.0f);
  return rt3_make(a.x, a.y, a.z) / a.w;
}
// i32(floor(f)) for a texel coordinate, kept within +-2^30 so that the +-1 / tile-origin arithmetic that
// follows cannot overflow (every coordinate is clamped to the image afterwards, so this changes no result;
// a non-finite average jitter at frame_count == 0 does produce such values)
__device__ __forceinline__ int pp_texel(float f) {
  int i = rt_f2i32_sat(f);
  return i < -1073741824 ? -1073741824 : (i > 1073741823 ? 1073741823 : i);
}
__device__ rt3 pp_clean(const DevPost& P, const rt_scene_uniforms& U, int cx, int cy) {  // :49-68
  rt3 center = pp_radiance(P, U, cx, cy);
  rt3 max_nb = rt3_splat(-1e6f);
  for (int y = -1; y <= 1; y++)
    for (int x = -1; x <= 1; x++) {
      if (x == 0 && y == 0) continue;
      max_nb = rt_max3(max_nb, pp_radiance(P, U, cx + x, cy + y));
    }
  return rt_clamp3(center, rt3_splat(0.0f), max_nb * 3.0f + rt3_splat(0.1f));
}
__device__ rt3 pp_nearest(const DevPost& P, const rt_scene_uniforms& U, int cx, int cy) {  // :71-97
  if (U.frame_count > 16u) return pp_clean(P, U, cx, cy);
  float u = ((float)cx + 0.5f) / (float)U.width - U.average_jitter[0];
  float v = ((float)cy + 0.5f) / (float)U.height - U.average_jitter[1];
  float fx = u * (float)U.width - 0.5f, fy = v * (float)U.height - 0.5f;
  float flx = rt_floor(fx), fly = rt_floor(fy);
  int ix = pp_texel(flx), iy = pp_texel(fly);
  float wx = fx - flx, wy = fy - fly;
  rt3 c00 = pp_clean(P, U, ix, iy), c10 = pp_clean(P, U, ix + 1, iy);
  rt3 c01 = pp_clean(P, U, ix, iy + 1), c11 = pp_clean(P, U, ix + 1, iy + 1);
  return rt_mix3(rt_mix3(c00, c10, wx), rt_mix3(c01, c11, wx), wy);
}
__device__ __forceinline__ rt3 aces(rt3 color) {  // :36-39
  const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
  rt3 num = color * (a * color + rt3_splat(b));
  rt3 den = color * (c * color + rt3_splat(d)) + rt3_splat(e);
  return rt_clamp3(num / den, rt3_splat(0.0f), rt3_splat(1.0f));
}

// k_postprocess: LDS-tiled.  A 16x16 block needs get_radiance_nearest on an 18x18 region; each of those
// is (frame_count <= 16) a bilinear blend of get_radiance_clean at 4 texels or (later) one of them, and
// each clean value looks at a 3x3 neighbourhood of get_radiance.  As written in the WGSL that is 171 / 684
// accumulation-buffer reads per pixel; here the three levels are materialised once per block in LDS
// (22x22 radiance -> 20x20 clean -> 18x18 nearest), so each accumulation texel is read ~1.9x (halo) and
// the rest is LDS traffic.  Every value is computed by the same function as the straight form, so the
// output is bit-identical; a bilinear footprint that falls outside the tile (possible only for
// |average_jitter| > 0.5 px or the non-finite jitter of frame_count == 0) falls back to the direct path.
#define PP_B 16
#define PP_R (PP_B + 6)  // radiance tile edge, origin at block - 3
#define PP_C (PP_B + 4)  // clean tile edge, origin at block - 2
#define PP_N (PP_B + 2)  // nearest tile edge, origin at block - 1
__global__ __launch_bounds__(256) void k_postprocess(DevPost P, rt_scene_uniforms U) {
  __shared__ float s_rad[PP_R * PP_R * 3];
  __shared__ float s_clean[PP_C * PP_C * 3];
  __shared__ float s_near[PP_N * PP_N * 3];
  const int bx = (int)blockIdx.x * PP_B, by = (int)blockIdx.y * PP_B;
  const int tid = (int)threadIdx.x;

  // level 0: radiance (get_radiance clamps the coordinate, so the halo holds edge-replicated values)
  for (int i = tid; i < PP_R * PP_R; i += 256) {
    rt3 v = pp_radiance(P, U, bx - 3 + i % PP_R, by - 3 + i / PP_R);
    s_rad[3 * i] = v.x; s_rad[3 * i + 1] = v.y; s_rad[3 * i + 2] = v.z;
  }
  __syncthreads();
  // level 1: firefly-clamped radiance
  for (int i = tid; i < PP_C * PP_C; i += 256) {
    const int lx = i % PP_C + 1, ly = i / PP_C + 1;  // position in the radiance tile
    auto rad = [&](int x, int y) {
      const float* q = &s_rad[3 * (y * PP_R + x)];
      return rt3_make(q[0], q[1], q[2]);
    };
    rt3 center = rad(lx, ly);
    rt3 max_nb = rt3_splat(-1e6f);
    for (int y = -1; y <= 1; y++)
      for (int x = -1; x <= 1; x++) {
        if (x == 0 && y == 0) continue;
        max_nb = rt_max3(max_nb, rad(lx + x, ly + y));
      }
    rt3 c = rt_clamp3(center, rt3_splat(0.0f), max_nb * 3.0f + rt3_splat(0.1f));
    s_clean[3 * i] = c.x; s_clean[3 * i + 1] = c.y; s_clean[3 * i + 2] = c.z;
  }
  __syncthreads();
  // level 2: un-jittered ("nearest") radiance
  for (int i = tid; i < PP_N * PP_N; i += 256) {
    const int cx = bx - 1 + i % PP_N, cy = by - 1 + i / PP_N;
    auto clean = [&](int gx, int gy, bool& inside) {
      const int lx = gx - (bx - 2), ly = gy - (by - 2);
      inside = lx >= 0 && ly >= 0 && lx < PP_C && ly < PP_C;
      const float* q = &s_clean[3 * ((inside ? ly : 0) * PP_C + (inside ? lx : 0))];
      return rt3_make(q[0], q[1], q[2]);
    };
    rt3 v;
    bool ok;
    if (U.frame_count > 16u) {
      v = clean(cx, cy, ok);  // always inside
    } else {
      float u = ((float)cx + 0.5f) / (float)U.width - U.average_jitter[0];
      float w = ((float)cy + 0.5f) / (float)U.height - U.average_jitter[1];
      float fx = u * (float)U.width - 0.5f, fy = w * (float)U.height - 0.5f;
      float flx = rt_floor(fx), fly = rt_floor(fy);
      int ix = pp_texel(flx), iy = pp_texel(fly);
      float wx = fx - flx, wy = fy - fly;
      bool i00, i11;
      bool i10, i01;
      rt3 c00 = clean(ix, iy, i00), c10 = clean(ix + 1, iy, i10);
      rt3 c01 = clean(ix, iy + 1, i01), c11 = clean(ix + 1, iy + 1, i11);
      if (i00 && i10 && i01 && i11) {
        v = rt_mix3(rt_mix3(c00, c10, wx), rt_mix3(c01, c11, wx), wy);
      } else {
        v = pp_nearest(P, U, cx, cy);  // footprint left the tile: straight path, same arithmetic
      }
    }
    s_near[3 * i] = v.x; s_near[3 * i + 1] = v.y; s_near[3 * i + 2] = v.z;
  }
  __syncthreads();

  const uint32_t x = (uint32_t)bx + (uint32_t)(tid & 15), y = (uint32_t)by + (uint32_t)(tid >> 4);
  if (x >= U.width || y >= U.height) return;
  const int lx = (tid & 15) + 1, ly = (tid >> 4) + 1;
  auto nearest = [&](int dx, int dy) {
    const float* q = &s_near[3 * ((ly + dy) * PP_N + lx + dx)];
    return rt3_make(q[0], q[1], q[2]);
  };
  const rt3 center_color = nearest(0, 0);
  rt3 filtered_sum = rt3_splat(0.0f);
  float total_weight = 0.0f;
  rt3 m1 = rt3_splat(0.0f), m2 = rt3_splat(0.0f);
  for (int dy = -1; dy <= 1; dy++)
    for (int dx = -1; dx <= 1; dx++) {
      rt3 ncol = nearest(dx, dy);
      float w_s = rt_exp(-(float)(dx * dx + dy * dy) / 0.5f);  // 2 * SIGMA_S^2 = 0.5
      rt3 cd = ncol - center_color;
      float w_r = rt_exp(-rt_dot(cd, cd) / 0.2f);  // 2 * SIGMA_R * RADIUS^2 = f32(0.1) * 2
      float w = w_s * w_r;
      filtered_sum = filtered_sum + ncol * w;
      total_weight += w;
      m1 = m1 + ncol;
      m2 = m2 + ncol * ncol;
    }
  rt3 denoised = filtered_sum / rt_max(total_weight, 1e-4f);

  const size_t p_idx = (size_t)y * U.width + x;
  ushort4 hp = P.history_in[p_idx];
  rt3 hist = rt3_make(rt_f16_to_f32(hp.x), rt_f16_to_f32(hp.y), rt_f16_to_f32(hp.z));
  rt3 mean = m1 / 9.0f;
  rt3 var = rt_max3(m2 / 9.0f - mean * mean, rt3_splat(0.0f));
  rt3 stddev = rt3_make(rt_sqrt(var.x), rt_sqrt(var.y), rt_sqrt(var.z));
  float k = (U.frame_count > 16u) ? 60.0f : 1.0f;
  rt3 clamped = rt_clamp3(hist, mean - stddev * k, mean + stddev * k);
  float alpha = 1.0f / (float)U.frame_count;
  if (U.frame_count == 1u) alpha = 0.1f;
  alpha = rt_max(alpha, 0.0001f);
  rt3 final_hdr = rt_mix3(clamped, denoised, alpha);
  ushort4 ho;
  ho.x = rt_f32_to_f16(final_hdr.x);
  ho.y = rt_f32_to_f16(final_hdr.y);
  ho.z = rt_f32_to_f16(final_hdr.z);
  ho.w = rt_f32_to_f16(1.0f);
  P.history_out[p_idx] = ho;

  rt3 mapped = aces(final_hdr);
  rt3 sharpened = mapped + aces(center_color - denoised) * 0.3f;
  rt3 cl = rt_clamp3(sharpened, rt3_splat(0.0f), rt3_splat(1.0f));
  const float inv_gamma = 0.4545454680919647216796875f;  // f32(1.0 / 2.2)
  P.out_rgba8[p_idx] = rt_unorm8(rt_pow(cl.x, inv_gamma)) | (rt_unorm8(rt_pow(cl.y, inv_gamma)) << 8) |
                       (rt_unorm8(rt_pow(cl.z, inv_gamma)) << 16) | (255u << 24);
}

}  // namespace rtk
#endif
