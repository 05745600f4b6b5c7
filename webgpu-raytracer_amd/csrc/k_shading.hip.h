// k_shading.hip.h — surface frame, BSDFs, light sampling (Raytracer.wgsl:207-427, 625-654, 738-779) and the ray / node counters.
// Part of the kernel set of csrc/kernels.hip.h (included from there, in order; not a stand-alone header).
#ifndef MI355RT_K_SHADING_HIP_H
#define MI355RT_K_SHADING_HIP_H

namespace rtk {

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
  rt3 v0 = xyz(S.tri_geom[RT_TRI_STRIDE * tri]);
  Bary b;
  b.e1 = xyz(S.tri_geom[RT_TRI_STRIDE * tri + 1]);
  b.e2 = xyz(S.tri_geom[RT_TRI_STRIDE * tri + 2]);
  rt3 s = lo - v0;
  rt3 h = rt_cross(ld, b.e2);
  float f = rt_rcp(rt_dot(b.e1, h));
  b.u = f * rt_dot(s, h);
  rt3 q = rt_cross(s, b.e1);
  b.v = f * rt_dot(ld, q);
  b.w = 1.0f - b.u - b.v;
  b.t = f * rt_dot(b.e2, q);
  return b;
}

__device__ __forceinline__ rt2 pack_normal(rt3 n) {  // Rasterizer.wgsl:71-74
  float s = rt_rcp(rt_abs(n.x) + rt_abs(n.y) + rt_abs(n.z));
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
  float a = -rt_rcp(sign + n.z);   // -1 / x = -(1 / x): negation is exact
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
  return rt_div(a2, RT_PI * d * d);
}
__device__ __forceinline__ float ggx_g(float n_dot_v, float n_dot_l, float a2) {  // :241-245
  float g1_v = rt_div(2.0f * n_dot_v, n_dot_v + rt_sqrt(a2 + (1.0f - a2) * n_dot_v * n_dot_v));
  float g1_l = rt_div(2.0f * n_dot_l, n_dot_l + rt_sqrt(a2 + (1.0f - a2) * n_dot_l * n_dot_l));
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
  s.pdf = rt_div_pi(c);
  s.throughput = albedo;
  s.specular = false;
  return s;
}
__device__ Scatter sample_ggx(rt3 n, rt3 v, float roughness, rt3 f0, uint32_t& rng) {  // :271-306
  float a = roughness;
  float ux = rand_pcg(rng);
  float uy = rand_pcg(rng);
  float phi = RT_TWO_PI * ux;
  float cos_theta = rt_sqrt(rt_max(0.0f, rt_div(1.0f - uy, 1.0f + (a * a - 1.0f) * uy)));
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
  s.pdf = rt_div(d * n_dot_h, 4.0f * v_dot_h);
  s.throughput = rt3_splat(0.0f);
  if (s.pdf > 1e-6f) s.throughput = (g * f * v_dot_h) / (n_dot_v * n_dot_h);
  s.specular = roughness < 0.01f;
  return s;
}
__device__ Scatter sample_dielectric(rt3 dir, rt3 normal, float ior, rt3 albedo, uint32_t& rng) {  // :320-339
  bool front_face = rt_dot(dir, normal) < 0.0f;
  float ratio = front_face ? rt_rcp(ior) : ior;
  rt3 n = front_face ? normal : -normal;
  rt3 unit_dir = rt_normalize(dir);
  float cos_theta = rt_min(rt_dot(-unit_dir, n), 1.0f);
  float sin_theta = rt_sqrt(1.0f - cos_theta * cos_theta);
  bool cannot_refract = ratio * sin_theta > 1.0f;
  bool reflect_it = cannot_refract;
  if (!reflect_it) {  // short-circuit `||`: the draw happens only when refraction is possible
    float r0 = rt_div(1.0f - ratio, 1.0f + ratio);
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
  s.pdf = rt_div(rt_div(dist_sq, cos_l * area), (float)light_count);
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
  return rt_div(rt_div(dist_sq, cos_l * area), (float)light_count);
}
__device__ __forceinline__ float power_heuristic(float a, float b) {
  float a2 = a * a, b2 = b * b;
  return rt_div(a2, a2 + b2);
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
__device__ __forceinline__ bool tile_pixel(const rt_scene_uniforms& U, uint32_t tile, uint32_t lane, uint32_t& x, uint32_t& y) {
  const uint32_t tiles_x = (U.width + 7u) / 8u;
  x = (tile % tiles_x) * 8u + (lane & 7u);
  y = (tile / tiles_x) * 8u + (lane >> 3);
  return x < U.width && y < U.height;
}
__device__ __forceinline__ bool tile_pixel(const rt_scene_uniforms& U, uint32_t& x, uint32_t& y) {
  return tile_pixel(U, blockIdx.x, threadIdx.x, x, y);
}

}  // namespace rtk
#endif
