// rt_oracle.cpp — CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
// this library.  Nothing under webgpu-raytracer_amd/ links, imports or calls it.
//
// PARITY UNPINNED: the reference (kokutoupan/webgpu-raytracer) ships no tests,
// golden images or known-answer vectors for this path, its
// WGSL needs a WebGPU device and its Rust crate cannot be built here, so this
// oracle is a restatement by reading.  What pins it: the hand-derived RNG /
// Halton / layout vectors of SURVEY.md Appendix A.2 (tests/test_oracle_kat.py).
//
// It restates, statement by statement and in the same evaluation order:
//   src/shaders/Raytracer.wgsl      (all)            -> section "path tracer"
//   src/shaders/Rasterizer.wgsl:81-173 (semantics)   -> section "primary visibility"
//   src/shaders/PostProcess.wgsl    (all)            -> section "post process"
//   src/renderer/ResourceManager.ts:348-447          -> section "uniforms"
//   src/renderer/WebGPURenderer.ts:88-129            -> compute()/present()
// The numeric meaning of every WGSL builtin is fixed by include/mi355rt_math.h.
// Build: g++ -O2 -ffp-contract=off (see oracle/Makefile).  Scalar code, 128-pixel
// spans of the owned rows distributed over a persistent pool of std::thread workers
// (the timed CPU baseline).

#include "../include/mi355rt_layout.h"
#include "../include/mi355rt_math.h"

#include <sched.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

const float PI = RT_PI;        // Raytracer.wgsl:5
const float T_MIN = 0.001f;    // :6
const float T_MAX = 1e30f;     // :7

struct Counters {
  uint64_t primary_rays = 0, extension_rays = 0, shadow_rays = 0;
  uint64_t nodes_visited = 0, tris_tested = 0, shaded_hits = 0;
  void add(const Counters& o) {
    primary_rays += o.primary_rays;
    extension_rays += o.extension_rays;
    shadow_rays += o.shadow_rays;
    nodes_visited += o.nodes_visited;
    tris_tested += o.tris_tested;
    shaded_hits += o.shaded_hits;
  }
};

struct Ray {  // Raytracer.wgsl:76-86
  rt3 origin, direction, inv_d, origin_inv_d;
};
Ray make_ray(rt3 origin, rt3 direction) {
  rt3 inv_d = rt3_splat(1.0f) / direction;
  return Ray{origin, direction, inv_d, origin * inv_d};
}

struct HitResult {  // :88-92
  float t;
  float tri_idx;
  int32_t inst_idx;
};
struct ONB {
  rt3 u, v, w;
};
struct LightSample {
  rt3 L, dir;
  float dist, pdf;
};
struct ScatterResult {
  rt3 dir;
  float pdf;
  rt3 throughput;
  bool is_specular;
};

struct Oracle {
  // ---- device-side resources (ResourceManager.ts) ----
  std::vector<float> pos, nrm, uv;          // geometry_pos / geometry_norm / geometry_uv
  std::vector<rt_topology> topology;
  std::vector<rt_node> nodes;               // TLAS ++ BLAS
  std::vector<rt_instance> instances;
  std::vector<rt_light_ref> lights;
  std::vector<uint32_t> draw_commands;
  std::vector<uint8_t> tex;                 // layers x 1024 x 1024 x 4
  uint32_t tex_layers = 0;
  rt_scene_uniforms scene{};
  uint32_t width = 0, height = 0;
  std::vector<float> accum;                 // vec4 per pixel
  std::vector<uint8_t> render_target;       // rgba8: G-buffer albedo, then post output (same texture in the reference)
  std::vector<float> g_normal;              // rgba32f
  std::vector<float> g_depth;               // depth32f
  std::vector<uint16_t> history[2];         // rgba16f
  int history_index = 0;
  // ---- host-side state ----
  uint32_t max_depth = 10, spp = 1;
  uint32_t total_frames = 0;
  uint32_t blas_offset = 0, vertex_count = 0, light_count = 0;
  float prev_camera[24] = {0};
  double acc_jx = 0, acc_jy = 0, jx = 0, jy = 0, avg_jx = 0, avg_jy = 0;
  int threads = 0;
  // optional pixel-ownership mask for sharded rendering (stripes of rows)
  uint32_t stripe_rows = 0, stripe_rank = 0, stripe_count = 1;
  Counters counters;
  // optional per-node visit histogram (oracle_set_node_histogram): one relaxed atomic add per node visit
  std::atomic<uint32_t>* node_hist = nullptr;
  void visit(uint32_t node) const {
    if (node_hist) node_hist[node].fetch_add(1, std::memory_order_relaxed);
  }

  // ------------------------------------------------------------ accessors
  rt3 get_pos(uint32_t i) const { return rt3_make(pos[i * 4], pos[i * 4 + 1], pos[i * 4 + 2]); }
  rt3 get_normal(uint32_t i) const { return rt3_make(nrm[i * 4], nrm[i * 4 + 1], nrm[i * 4 + 2]); }
  rt2 get_uv(uint32_t i) const { return rt2_make(uv[i * 2], uv[i * 2 + 1]); }

  // textureSampleLevel(tex, smp, uv, layer, 0): bilinear, repeat, level 0, unorm, no sRGB.
  // Without textures the 1x1 white default is bound (ResourceManager.ts:81-96).
  rt3 sample_tex(rt2 tuv, int32_t layer) const {
    if (tex_layers == 0) return rt3_splat(1.0f);
    if (layer < 0) layer = 0;
    if ((uint32_t)layer >= tex_layers) layer = (int32_t)tex_layers - 1;
    const int N = RT_TEX_SIZE;
    float x = tuv.x * (float)N - 0.5f, y = tuv.y * (float)N - 0.5f;
    float fx0 = rt_floor(x), fy0 = rt_floor(y);
    float fx = x - fx0, fy = y - fy0;
    int ix = rt_f2i32_sat(fx0), iy = rt_f2i32_sat(fy0);
    auto wrap = [&](int v) { return (int)(((uint32_t)v) & (uint32_t)(N - 1)); };  // two's complement mod 1024
    int x0 = wrap(ix), x1 = wrap(ix + 1), y0 = wrap(iy), y1 = wrap(iy + 1);
    const uint8_t* base = &tex[(size_t)layer * N * N * 4];
    auto texel = [&](int xx, int yy) {
      const uint8_t* p = base + ((size_t)yy * N + xx) * 4;
      return rt3_make(rt_from_unorm8(p[0]), rt_from_unorm8(p[1]), rt_from_unorm8(p[2]));
    };
    rt3 top = rt_mix3(texel(x0, y0), texel(x1, y0), fx);
    rt3 bot = rt_mix3(texel(x0, y1), texel(x1, y1), fx);
    return rt_mix3(top, bot, fy);
  }

  // ----------------------------------------------------------- normals
  static rt2 pack_normal(rt3 n) {  // Raytracer.wgsl:116-119 / Rasterizer.wgsl:71-74
    float s = 1.0f / (rt_abs(n.x) + rt_abs(n.y) + rt_abs(n.z));
    rt2 p = rt2_make(n.x * s, n.y * s);
    if (n.z < 0.0f) {
      float ox = (1.0f - rt_abs(p.y)) * (p.x >= 0.0f ? 1.0f : -1.0f);
      float oy = (1.0f - rt_abs(p.x)) * (p.y >= 0.0f ? 1.0f : -1.0f);
      return rt2_make(ox, oy);
    }
    return p;
  }
  static rt3 unpack_normal(rt2 p) {  // :121-127
    rt3 n = rt3_make(p.x, p.y, 1.0f - rt_abs(p.x) - rt_abs(p.y));
    float t = rt_saturate(-n.z);
    n.x += (n.x >= 0.0f) ? -t : t;
    n.y += (n.y >= 0.0f) ? -t : t;
    return rt_normalize(n);
  }

  // --------------------------------------------------------------- RNG
  static uint32_t init_rng(uint32_t pixel_idx, uint32_t frame) {  // :178-183
    uint32_t seed = pixel_idx + frame * 719393u;
    seed ^= 2747636419u; seed *= 2654435769u; seed ^= (seed >> 16);
    seed *= 2654435769u; seed ^= (seed >> 16); seed *= 2654435769u;
    return seed;
  }
  static float rand_pcg(uint32_t* state) {  // :185-189 (non-standard PCG, literal)
    uint32_t old = *state;
    *state = old * 747796405u + 2891336453u;
    uint32_t word = ((*state) >> ((old >> 28) + 4u)) ^ (*state);
    return (float)((word >> 22) ^ word) / 4294967296.0f;  // literal 4294967295.0 rounds to 2^32 in f32
  }

  static rt3 local_to_world(const ONB& o, rt3 a) { return a.x * o.u + a.y * o.v + a.z * o.w; }  // :216-218
  static ONB build_onb(rt3 n) {  // :207-214
    float sign = (n.z >= 0.0f) ? 1.0f : -1.0f;
    float a = -1.0f / (sign + n.z);
    float b = n.x * n.y * a;
    rt3 u = rt3_make(1.0f + sign * n.x * n.x * a, sign * b, -sign * n.x);
    rt3 v = rt3_make(b, sign + n.y * n.y * a, -n.y);
    return ONB{u, v, n};
  }
  static rt3 random_unit_vector(const ONB& onb, uint32_t* rng) {  // :191-199
    float r1 = rand_pcg(rng);
    float r2 = rand_pcg(rng);
    float phi = RT_TWO_PI * r1;
    float cos_theta = rt_sqrt(1.0f - r2);
    float sin_theta = rt_sqrt(r2);
    float s, c;
    rt_sincos(phi, &s, &c);
    rt3 local_dir = rt3_make(c * sin_theta, s * sin_theta, cos_theta);
    return local_to_world(onb, local_dir);
  }
  static rt3 random_in_unit_disk(uint32_t* rng) {  // :201-205
    float r = rt_sqrt(rand_pcg(rng));
    float theta = RT_TWO_PI * rand_pcg(rng);
    float s, c;
    rt_sincos(theta, &s, &c);
    return rt3_make(r * c, r * s, 0.0f);
  }

  // -------------------------------------------------------------- BSDFs
  static rt3 eval_diffuse(rt3 albedo) { return albedo / PI; }  // :224-226
  static ScatterResult sample_diffuse(rt3 normal, rt3 albedo, uint32_t* rng) {  // :228-233
    ONB onb = build_onb(normal);
    rt3 dir = random_unit_vector(onb, rng);
    float cos_theta = rt_max(rt_dot(normal, dir), 0.0f);
    return ScatterResult{dir, cos_theta / PI, albedo, false};
  }
  static float ggx_d(float n_dot_h, float a2) {  // :236-239
    float d = (n_dot_h * a2 - n_dot_h) * n_dot_h + 1.0f;
    return a2 / (PI * d * d);
  }
  static float ggx_g(float n_dot_v, float n_dot_l, float a2) {  // :241-245
    float g1_v = 2.0f * n_dot_v / (n_dot_v + rt_sqrt(a2 + (1.0f - a2) * n_dot_v * n_dot_v));
    float g1_l = 2.0f * n_dot_l / (n_dot_l + rt_sqrt(a2 + (1.0f - a2) * n_dot_l * n_dot_l));
    return g1_v * g1_l;
  }
  static float pow5(float x) {  // :247-250
    float x2 = x * x;
    return x2 * x2 * x;
  }
  static rt3 fresnel_schlick(float cos_theta, rt3 f0) {  // :252-254
    return f0 + (rt3_splat(1.0f) - f0) * pow5(rt_clamp(1.0f - cos_theta, 0.0f, 1.0f));
  }
  static rt3 eval_ggx(rt3 n, rt3 v, rt3 l, float roughness, rt3 f0) {  // :256-269
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
  static ScatterResult sample_ggx(rt3 n, rt3 v, float roughness, rt3 f0, uint32_t* rng) {  // :271-306
    float a = roughness;
    float ux = rand_pcg(rng);
    float uy = rand_pcg(rng);
    float phi = RT_TWO_PI * ux;
    float cos_theta = rt_sqrt(rt_max(0.0f, (1.0f - uy) / (1.0f + (a * a - 1.0f) * uy)));
    float sin_theta = rt_sqrt(rt_max(0.0f, 1.0f - cos_theta * cos_theta));
    float sp, cp;
    rt_sincos(phi, &sp, &cp);
    rt3 h_local = rt3_make(sin_theta * cp, sin_theta * sp, cos_theta);
    ONB onb = build_onb(n);
    rt3 h = local_to_world(onb, h_local);
    rt3 l = rt_reflect(-v, h);
    if (rt_dot(n, l) <= 0.0f) return ScatterResult{rt3_splat(0.0f), 0.0f, rt3_splat(0.0f), false};
    float n_dot_v = rt_max(rt_dot(n, v), 1e-4f);
    float n_dot_l = rt_max(rt_dot(n, l), 1e-4f);
    float n_dot_h = rt_max(rt_dot(n, h), 1e-4f);
    float v_dot_h = rt_max(rt_dot(v, h), 1e-4f);
    float a2 = a * a;
    float d = ggx_d(n_dot_h, a2);
    float g = ggx_g(n_dot_v, n_dot_l, a2);
    rt3 f = fresnel_schlick(v_dot_h, f0);
    (void)n_dot_l;
    float pdf = (d * n_dot_h) / (4.0f * v_dot_h);
    rt3 throughput = rt3_splat(0.0f);
    if (pdf > 1e-6f) throughput = (g * f * v_dot_h) / (n_dot_v * n_dot_h);
    bool treat_as_specular = roughness < 0.01f;
    return ScatterResult{l, pdf, throughput, treat_as_specular};
  }
  static float reflectance_dielectric(float cosine, float ref_idx) {  // :314-318
    float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    r0 = r0 * r0;
    return r0 + (1.0f - r0) * pow5(1.0f - cosine);
  }
  static ScatterResult sample_dielectric(rt3 dir, rt3 normal, float ior, rt3 albedo, uint32_t* rng) {  // :320-339
    bool front_face = rt_dot(dir, normal) < 0.0f;
    float refraction_ratio = front_face ? (1.0f / ior) : ior;
    rt3 n = front_face ? normal : -normal;
    rt3 unit_dir = rt_normalize(dir);
    float cos_theta = rt_min(rt_dot(-unit_dir, n), 1.0f);
    float sin_theta = rt_sqrt(1.0f - cos_theta * cos_theta);
    bool cannot_refract = refraction_ratio * sin_theta > 1.0f;
    rt3 direction;
    // short-circuit ||: the RNG draw happens only when refraction is possible
    if (cannot_refract || reflectance_dielectric(cos_theta, refraction_ratio) > rand_pcg(rng)) {
      direction = rt_reflect(unit_dir, n);
    } else {
      direction = rt_refract(unit_dir, n, refraction_ratio);
    }
    return ScatterResult{direction, 1.0f, albedo, true};
  }

  // ------------------------------------------------------ light sampling
  LightSample sample_light_source(rt3 hit_p, uint32_t* rng) const {  // :345-399
    LightSample none{rt3_splat(0.0f), rt3_splat(0.0f), 0.0f, 0.0f};
    uint32_t lc = scene.light_count;
    if (lc == 0u) return none;
    uint32_t pick = rt_f2u32_sat(rand_pcg(rng) * (float)lc);
    // rand_pcg can return exactly 1.0 -> pick == light_count; WebGPU clamps the read
    if (pick >= (uint32_t)lights.size()) pick = (uint32_t)lights.size() - 1u;
    rt_light_ref l_ref = lights[pick];
    const rt_topology& tri = topology[l_ref.tri_idx];
    const rt_instance& inst = instances[l_ref.inst_idx];
    const float* m = inst.transform;
    rt3 v0 = rt_mat_mul_point(m, get_pos(tri.v0));
    rt3 v1 = rt_mat_mul_point(m, get_pos(tri.v1));
    rt3 v2 = rt_mat_mul_point(m, get_pos(tri.v2));
    float r1 = rand_pcg(rng);
    float r2 = rand_pcg(rng);
    float sqrt_r1 = rt_sqrt(r1);
    float u = 1.0f - sqrt_r1;
    float v = r2 * sqrt_r1;
    float w = 1.0f - u - v;
    rt3 p = v0 * u + v1 * v + v2 * w;
    rt3 edge1 = v1 - v0;
    rt3 edge2 = v2 - v0;
    rt3 n_raw = rt_normalize(rt_cross(edge1, edge2));
    float area = rt_length(rt_cross(edge1, edge2)) * 0.5f;
    rt3 l_dir = p - hit_p;
    float dist_sq = rt_dot(l_dir, l_dir);
    float dist = rt_sqrt(dist_sq);
    rt3 unit_l = l_dir / dist;
    float cos_theta_l = rt_max(rt_dot(n_raw, -unit_l), 0.0f);
    if (cos_theta_l < 1e-6f) return none;
    rt2 uv0 = get_uv(tri.v0), uv1 = get_uv(tri.v1), uv2 = get_uv(tri.v2);
    rt2 tex_uv = uv0 * u + uv1 * v + uv2 * w;
    rt3 L = rt3_make(tri.data0[0], tri.data0[1], tri.data0[2]);
    float base_tex = tri.data2[0];
    if (base_tex > -0.5f) L = L * sample_tex(tex_uv, rt_f2i32_sat(base_tex));
    float pdf = (dist_sq / (cos_theta_l * area)) / (float)lc;
    return LightSample{L, unit_l, dist, pdf};
  }
  float get_light_pdf(uint32_t tri_idx, uint32_t inst_idx, float t, rt3 l_dir) const {  // :401-421
    const rt_topology& tri = topology[tri_idx];
    const rt_instance& inst = instances[inst_idx];
    const float* m = inst.transform;
    rt3 v0 = rt_mat_mul_point(m, get_pos(tri.v0));
    rt3 v1 = rt_mat_mul_point(m, get_pos(tri.v1));
    rt3 v2 = rt_mat_mul_point(m, get_pos(tri.v2));
    rt3 edge1 = v1 - v0;
    rt3 edge2 = v2 - v0;
    float area = rt_length(rt_cross(edge1, edge2)) * 0.5f;
    rt3 normal = rt_normalize(rt_cross(edge1, edge2));
    float cos_theta_l = rt_max(rt_dot(normal, -l_dir), 0.0f);
    if (cos_theta_l < 1e-4f) return 0.0f;
    float dist_sq = t * t;
    return (dist_sq / (cos_theta_l * area)) / (float)scene.light_count;
  }
  static float power_heuristic(float pdf_a, float pdf_b) {  // :423-427
    float a2 = pdf_a * pdf_a;
    float b2 = pdf_b * pdf_b;
    return a2 / (a2 + b2);
  }

  // -------------------------------------------------------- intersection
  static float intersect_aabb(const rt_node& n, const Ray& r, float t_min, float t_max) {  // :433-441
    rt3 mn = rt3_make(n.min_b[0], n.min_b[1], n.min_b[2]);
    rt3 mx = rt3_make(n.max_b[0], n.max_b[1], n.max_b[2]);
    rt3 t1 = mn * r.inv_d - r.origin_inv_d;
    rt3 t2 = mx * r.inv_d - r.origin_inv_d;
    rt3 t_near = rt_min3(t1, t2);
    rt3 t_far = rt_max3(t1, t2);
    float tm_near = rt_max(t_min, rt_max(t_near.x, rt_max(t_near.y, t_near.z)));
    float tm_far = rt_min(t_max, rt_min(t_far.x, rt_min(t_far.y, t_far.z)));
    return (tm_near <= tm_far) ? tm_near : T_MAX;
  }
  static float hit_triangle_raw(rt3 v0, rt3 v1, rt3 v2, const Ray& r, float t_min, float t_max) {  // :443-453
    rt3 e1 = v1 - v0, e2 = v2 - v0;
    rt3 h = rt_cross(r.direction, e2);
    float a = rt_dot(e1, h);
    if (rt_abs(a) < 1e-6f) return -1.0f;
    float f = 1.0f / a;
    rt3 s = r.origin - v0;
    float u = f * rt_dot(s, h);
    if (u < 0.0f || u > 1.0f) return -1.0f;
    rt3 q = rt_cross(s, e1);
    float v = f * rt_dot(r.direction, q);
    if (v < 0.0f || u + v > 1.0f) return -1.0f;
    float t = f * rt_dot(e2, q);
    return (t > t_min && t < t_max) ? t : -1.0f;
  }
  void intersect_blas(const Ray& r, float t_min, float t_max, uint32_t node_start_idx, float* out_t, float* out_idx,
                      Counters& c) const {  // :455-494
    float closest_t = t_max;
    float hit_idx = -1.0f;
    uint32_t end_node = node_start_idx + nodes[node_start_idx].skip;
    uint32_t curr = node_start_idx;
    while (curr < end_node) {
      const rt_node& node = nodes[curr];
      c.nodes_visited++;
      visit(curr);
      float t_aabb = intersect_aabb(node, r, t_min, closest_t);
      if (t_aabb < T_MAX) {
        uint32_t data = node.data;
        if (data != 0u) {
          uint32_t first = data >> 3, count = data & 7u;
          for (uint32_t i = 0; i < count; i++) {
            uint32_t tri_id = first + i;
            const rt_topology& tr = topology[tri_id];
            c.tris_tested++;
            float t = hit_triangle_raw(get_pos(tr.v0), get_pos(tr.v1), get_pos(tr.v2), r, t_min, closest_t);
            if (t > 0.0f) {
              closest_t = t;
              hit_idx = (float)tri_id;
            }
          }
          curr = node_start_idx + node.skip;
        } else {
          curr = curr + 1u;
        }
      } else {
        curr = node_start_idx + node.skip;
      }
    }
    *out_t = closest_t;
    *out_idx = hit_idx;
  }
  HitResult intersect_tlas(const Ray& r, float t_min, float t_max, Counters& c) const {  // :496-528
    HitResult res{t_max, -1.0f, -1};
    if (scene.blas_base_idx == 0u) return res;
    uint32_t curr = 0u;
    uint32_t end_node = nodes[0].skip;
    while (curr < end_node) {
      const rt_node& node = nodes[curr];
      c.nodes_visited++;
      visit(curr);
      if (intersect_aabb(node, r, t_min, res.t) < T_MAX) {
        uint32_t data = node.data;
        if (data != 0u) {
          uint32_t inst_idx = data >> 3;
          const rt_instance& inst = instances[inst_idx];
          Ray r_local = make_ray(rt_mat_mul_point(inst.inverse, r.origin), rt_mat_mul_dir(inst.inverse, r.direction));
          float bt, bidx;
          intersect_blas(r_local, t_min, res.t, scene.blas_base_idx + inst.blas_node_offset, &bt, &bidx, c);
          if (bidx > -0.5f) {
            res.t = bt;
            res.tri_idx = bidx;
            res.inst_idx = (int32_t)inst_idx;
          }
          curr = node.skip;
        } else {
          curr = curr + 1u;
        }
      } else {
        curr = node.skip;
      }
    }
    return res;
  }
  bool intersect_blas_shadow(const Ray& r, float t_min, float t_max, uint32_t node_start_idx, Counters& c) const {  // :532-563
    uint32_t end_node = node_start_idx + nodes[node_start_idx].skip;
    uint32_t curr = node_start_idx;
    while (curr < end_node) {
      const rt_node& node = nodes[curr];
      c.nodes_visited++;
      visit(curr);
      float t_aabb = intersect_aabb(node, r, t_min, t_max);
      if (t_aabb < T_MAX) {
        uint32_t data = node.data;
        if (data != 0u) {
          uint32_t first = data >> 3, count = data & 7u;
          for (uint32_t i = 0; i < count; i++) {
            const rt_topology& tr = topology[first + i];
            c.tris_tested++;
            float t = hit_triangle_raw(get_pos(tr.v0), get_pos(tr.v1), get_pos(tr.v2), r, t_min, t_max);
            if (t > 0.0f) return true;
          }
          curr = node_start_idx + node.skip;
        } else {
          curr = curr + 1u;
        }
      } else {
        curr = node_start_idx + node.skip;
      }
    }
    return false;
  }
  bool intersect_tlas_shadow(const Ray& r, float t_min, float t_max, Counters& c) const {  // :566-600
    if (scene.blas_base_idx == 0u) return false;
    uint32_t curr = 0u;
    uint32_t end_node = nodes[0].skip;
    while (curr < end_node) {
      const rt_node& node = nodes[curr];
      c.nodes_visited++;
      visit(curr);
      float t_aabb = intersect_aabb(node, r, t_min, t_max);
      if (t_aabb < T_MAX) {
        uint32_t data = node.data;
        if (data != 0u) {
          uint32_t inst_idx = data >> 3;
          const rt_instance& inst = instances[inst_idx];
          Ray r_local = make_ray(rt_mat_mul_point(inst.inverse, r.origin), rt_mat_mul_dir(inst.inverse, r.direction));
          if (intersect_blas_shadow(r_local, t_min, t_max, scene.blas_base_idx + inst.blas_node_offset, c)) return true;
          curr = node.skip;
        } else {
          curr = curr + 1u;
        }
      } else {
        curr = node.skip;
      }
    }
    return false;
  }

  // --------------------------------------------------------- path tracer
  rt3 ray_color(const Ray& r_in, uint32_t* rng, uint32_t cx, uint32_t cy, Counters& c) const {  // :607-783
    Ray ray = r_in;
    rt3 throughput = rt3_splat(1.0f);
    rt3 radiance = rt3_splat(0.0f);
    uint32_t pixel_idx = cy * scene.width + cx;
    float prev_bsdf_pdf = 0.0f;
    bool specular_bounce = true;

    // --- depth 0: G-buffer read ---
    float depth_val = g_depth[pixel_idx];
    if (depth_val >= 1.0f) return radiance;
    const float* gn = &g_normal[(size_t)pixel_idx * 4];
    uint32_t tri_idx = rt_f2u(gn[2]);
    int32_t inst_idx = (int32_t)rt_f2u(gn[3]);

    const rt_topology* tri = &topology[tri_idx];
    const rt_instance* inst = &instances[inst_idx];
    const float* inv = inst->inverse;
    rt3 v0_pos = get_pos(tri->v0), v1_pos = get_pos(tri->v1), v2_pos = get_pos(tri->v2);

    Ray r_local = make_ray(rt_mat_mul_point(inv, ray.origin), rt_mat_mul_dir(inv, ray.direction));
    rt3 s = r_local.origin - v0_pos;
    rt3 e1 = v1_pos - v0_pos;
    rt3 e2 = v2_pos - v0_pos;
    rt3 h_val = rt_cross(r_local.direction, e2);
    float f_val = 1.0f / rt_dot(e1, h_val);
    float u_bar = f_val * rt_dot(s, h_val);
    rt3 q = rt_cross(s, e1);
    float v_bar = f_val * rt_dot(r_local.direction, q);
    float w_bar = 1.0f - u_bar - v_bar;
    float hit_t = f_val * rt_dot(e2, q);

    rt2 uv0 = get_uv(tri->v0), uv1 = get_uv(tri->v1), uv2 = get_uv(tri->v2);
    rt2 tex_uv = uv0 * w_bar + uv1 * u_bar + uv2 * v_bar;

    rt3 normal = unpack_normal(rt2_make(gn[0], gn[1]));
    const uint8_t* ga = &render_target[(size_t)pixel_idx * 4];
    rt3 albedo = rt3_make(rt_from_unorm8(ga[0]), rt_from_unorm8(ga[1]), rt_from_unorm8(ga[2]));

    rt3 local_geom_n = rt_normalize(rt_cross(e1, e2));
    rt3 world_geom_n = rt_normalize(rt_vec_mul_mat_dir(local_geom_n, inv));

    for (uint32_t depth = 0u; depth < max_depth; depth++) {
      c.shaded_hits++;
      uint32_t mat_type = rt_f2u32_sat(tri->data0[3] + 0.5f);
      rt3 hit_p = ray.origin + ray.direction * hit_t;

      normal = (rt_dot(ray.direction, normal) < 0.0f) ? normal : -normal;
      world_geom_n = (rt_dot(ray.direction, world_geom_n) < 0.0f) ? world_geom_n : -world_geom_n;

      float metallic = tri->data1[0];
      float roughness = tri->data1[1];
      if (tri->data2[1] > -0.5f) {
        rt3 mr = sample_tex(tex_uv, rt_f2i32_sat(tri->data2[1]));
        metallic *= mr.z;
        roughness *= mr.y;
      }
      roughness = rt_max(roughness, 0.005f);

      rt3 emissive = rt3_make(tri->data3[0], tri->data3[1], tri->data3[2]);
      if (tri->data2[3] > -0.5f) emissive = emissive * sample_tex(tex_uv, rt_f2i32_sat(tri->data2[3]));

      rt3 f0 = rt_mix3(rt3_splat(0.04f), albedo, metallic);

      // --- emissive / light ---
      if (mat_type == 3u || rt_length(emissive) > 1e-4f) {
        rt3 em_val = (mat_type == 3u) ? albedo : emissive;
        if (specular_bounce) {
          radiance = radiance + throughput * em_val;
        } else {
          radiance = radiance + throughput * em_val *
                                    power_heuristic(prev_bsdf_pdf, get_light_pdf(tri_idx, (uint32_t)inst_idx, hit_t,
                                                                                 ray.direction));
        }
        if (mat_type == 3u) break;
      }

      // --- next event estimation ---
      if (mat_type != 2u) {
        LightSample light_s = sample_light_source(hit_p, rng);
        if (light_s.pdf > 0.0f) {
          c.shadow_rays++;
          if (!intersect_tlas_shadow(make_ray(hit_p + world_geom_n * 1e-4f, light_s.dir), T_MIN,
                                     light_s.dist - 2e-4f, c)) {
            rt3 bsdf_val = rt3_splat(0.0f);
            float bsdf_pdf_val = 0.0f;
            if (mat_type == 0u) {
              bsdf_val = eval_diffuse(albedo);
              bsdf_pdf_val = rt_max(rt_dot(normal, light_s.dir), 0.0f) / PI;
            } else if (mat_type == 1u) {
              bsdf_val = eval_ggx(normal, -ray.direction, light_s.dir, roughness, f0);
              rt3 H = rt_normalize(-ray.direction + light_s.dir);
              bsdf_pdf_val = (ggx_d(rt_dot(normal, H), roughness * roughness) * rt_max(rt_dot(normal, H), 0.0f)) /
                             (4.0f * rt_max(rt_dot(-ray.direction, H), 0.0f));
            }
            if (bsdf_pdf_val > 0.0f) {
              radiance = radiance + throughput * bsdf_val * light_s.L * power_heuristic(light_s.pdf, bsdf_pdf_val) *
                                        rt_max(rt_dot(normal, light_s.dir), 0.0f) / light_s.pdf;
            }
          }
        }
      }

      ScatterResult scatter;
      if (mat_type == 0u) {
        scatter = sample_diffuse(normal, albedo, rng);
      } else if (mat_type == 1u) {
        scatter = sample_ggx(normal, -ray.direction, roughness, f0, rng);
      } else {
        scatter = sample_dielectric(ray.direction, normal, tri->data1[2], albedo, rng);
      }

      if (mat_type != 2u && rt_dot(scatter.dir, world_geom_n) <= 0.0f) {
        scatter.pdf = 0.0f;
        scatter.throughput = rt3_splat(0.0f);
      }
      if (scatter.pdf <= 0.0f || rt_length(scatter.throughput) <= 0.0f) break;

      throughput = throughput * scatter.throughput;

      rt3 ray_offset_normal = (rt_dot(scatter.dir, world_geom_n) > 0.0f) ? world_geom_n : -world_geom_n;
      ray = make_ray(hit_p + ray_offset_normal * 1e-4f, scatter.dir);

      prev_bsdf_pdf = scatter.pdf;
      specular_bounce = scatter.is_specular;

      if (depth > 3u) {
        float p = rt_max(throughput.x, rt_max(throughput.y, throughput.z));
        if (rand_pcg(rng) > p) break;
        throughput = throughput / p;
      }

      // --- next intersection ---
      if (depth < max_depth - 1u) {
        c.extension_rays++;
        HitResult hit = intersect_tlas(ray, T_MIN, T_MAX, c);
        if (hit.inst_idx < 0) break;
        hit_t = hit.t;
        tri_idx = rt_f2u32_sat(hit.tri_idx);
        inst_idx = hit.inst_idx;

        tri = &topology[tri_idx];
        inst = &instances[inst_idx];
        inv = inst->inverse;
        v0_pos = get_pos(tri->v0);
        v1_pos = get_pos(tri->v1);
        v2_pos = get_pos(tri->v2);

        r_local = make_ray(rt_mat_mul_point(inv, ray.origin), rt_mat_mul_dir(inv, ray.direction));
        s = r_local.origin - v0_pos;
        e1 = v1_pos - v0_pos;
        e2 = v2_pos - v0_pos;
        h_val = rt_cross(r_local.direction, e2);
        f_val = 1.0f / rt_dot(e1, h_val);
        u_bar = f_val * rt_dot(s, h_val);
        q = rt_cross(s, e1);
        v_bar = f_val * rt_dot(r_local.direction, q);
        w_bar = 1.0f - u_bar - v_bar;

        uv0 = get_uv(tri->v0);
        uv1 = get_uv(tri->v1);
        uv2 = get_uv(tri->v2);
        tex_uv = uv0 * w_bar + uv1 * u_bar + uv2 * v_bar;

        rt3 n0 = get_normal(tri->v0), n1 = get_normal(tri->v1), n2 = get_normal(tri->v2);
        rt3 ln = rt_normalize(n0 * w_bar + n1 * u_bar + n2 * v_bar);
        normal = rt_normalize(rt_vec_mul_mat_dir(ln, inv));

        albedo = rt3_make(tri->data0[0], tri->data0[1], tri->data0[2]);
        if (tri->data2[0] > -0.5f) albedo = albedo * sample_tex(tex_uv, rt_f2i32_sat(tri->data2[0]));

        if (tri->data2[2] > -0.5f) {
          rt3 n_map = sample_tex(tex_uv, rt_f2i32_sat(tri->data2[2])) * 2.0f - rt3_splat(1.0f);
          rt3 T = rt_normalize(e1);
          rt3 B = rt_normalize(rt_cross(ln, T));
          rt3 ln_mapped = rt_normalize(T * n_map.x + B * n_map.y + ln * n_map.z);
          normal = rt_normalize(rt_vec_mul_mat_dir(ln_mapped, inv));
        }

        local_geom_n = rt_normalize(rt_cross(e1, e2));
        world_geom_n = rt_normalize(rt_vec_mul_mat_dir(local_geom_n, inv));
      }
    }
    return radiance;
  }

  bool owns_row(uint32_t y) const {
    if (stripe_rows == 0 || stripe_count <= 1) return true;
    return (y / stripe_rows) % stripe_count == stripe_rank;
  }

  // Raytracer.wgsl:791-819 `main`, one invocation
  void trace_pixel(uint32_t x, uint32_t y, Counters& c) {
    uint32_t p_idx = y * scene.width + x;
    const rt_camera& cam = scene.camera;
    rt3 cam_o = rt3_make(cam.origin[0], cam.origin[1], cam.origin[2]);
    rt3 cam_ll = rt3_make(cam.lower_left[0], cam.lower_left[1], cam.lower_left[2]);
    rt3 cam_h = rt3_make(cam.horizontal[0], cam.horizontal[1], cam.horizontal[2]);
    rt3 cam_v = rt3_make(cam.vertical[0], cam.vertical[1], cam.vertical[2]);
    rt3 cam_uu = rt3_make(cam.u[0], cam.u[1], cam.u[2]);
    rt3 cam_vv = rt3_make(cam.v[0], cam.v[1], cam.v[2]);
    rt3 col = rt3_splat(0.0f);
    for (uint32_t i = 0u; i < spp; i++) {
      uint32_t rng = init_rng(p_idx, scene.frame_count * spp + i);
      rt3 off = rt3_splat(0.0f);
      if (cam.origin[3] > 0.0f) {
        rt3 rd = cam.origin[3] * random_in_unit_disk(&rng);
        off = cam_uu * rd.x + cam_vv * rd.y;
      }
      float u = ((float)x + 0.5f + scene.jitter[0] * (float)scene.width) / (float)scene.width;
      float v = 1.0f - ((float)y + 0.5f + scene.jitter[1] * (float)scene.height) / (float)scene.height;
      rt3 d = cam_ll + u * cam_h + v * cam_v - cam_o - off;
      col = col + ray_color(make_ray(cam_o + off, d), &rng, x, y, c);
    }
    col = col / (float)spp;
    float* acc = &accum[(size_t)p_idx * 4];
    if (scene.frame_count > 1u) {
      acc[0] = acc[0] + col.x;
      acc[1] = acc[1] + col.y;
      acc[2] = acc[2] + col.z;
      acc[3] = acc[3] + 1.0f;
    } else {
      acc[0] = col.x;
      acc[1] = col.y;
      acc[2] = col.z;
      acc[3] = 1.0f;
    }
  }

  // --------------------------------------------------- primary visibility
  // Restates the *semantics* of the hardware rasterizer pass (Rasterizer.wgsl:81-173 +
  // RasterizerPass.ts:97-140) as one closest-hit cast per pixel through the same TLAS/BLAS:
  //  - pinhole ray through the jittered pixel centre (same u,v as Raytracer.wgsl:806-808; the
  //    NDC shift at Rasterizer.wgsl:148-150 is algebraically that jitter), no lens offset;
  //  - view-space z of the hit = t * focal_length, clipped to [z_near, z_far] = [0.001, 10000];
  //  - depth = z_clip / z_view with the shader's own z mapping (:132-136); miss -> clear 1.0;
  //  - normal: per-vertex normalize((n,0) * inv) (:108), barycentric interpolation, normalize,
  //    octahedral pack (:171); tri_idx / instance index bit-cast into .zw;
  //  - albedo: base colour x base texture at the interpolated uv (:163-167), written to an
  //    rgba8unorm target (ResourceManager.ts:99-104) => clamped to [0,1] and 8-bit quantised.
  // Cleared values: albedo (0,0,0,0), normal_id (0,0,0,0), depth 1.0 (RasterizerPass.ts:99-121).
  void gbuffer_pixel(uint32_t x, uint32_t y, Counters& c) {
    uint32_t p_idx = y * scene.width + x;
    const rt_camera& cam = scene.camera;
    rt3 eye = rt3_make(cam.origin[0], cam.origin[1], cam.origin[2]);
    rt3 ll = rt3_make(cam.lower_left[0], cam.lower_left[1], cam.lower_left[2]);
    rt3 hor = rt3_make(cam.horizontal[0], cam.horizontal[1], cam.horizontal[2]);
    rt3 ver = rt3_make(cam.vertical[0], cam.vertical[1], cam.vertical[2]);
    rt3 center = ll + hor * 0.5f + ver * 0.5f;
    float focal_length = rt_length(center - eye);
    const float z_near = 0.001f, z_far = 10000.0f;

    float u = ((float)x + 0.5f + scene.jitter[0] * (float)scene.width) / (float)scene.width;
    float v = 1.0f - ((float)y + 0.5f + scene.jitter[1] * (float)scene.height) / (float)scene.height;
    rt3 d = ll + u * hor + v * ver - eye;
    Ray ray = make_ray(eye, d);

    uint8_t* ga = &render_target[(size_t)p_idx * 4];
    float* gn = &g_normal[(size_t)p_idx * 4];
    c.primary_rays++;
    HitResult hit = intersect_tlas(ray, z_near / focal_length, z_far / focal_length, c);
    if (hit.inst_idx < 0) {
      ga[0] = ga[1] = ga[2] = ga[3] = 0;
      gn[0] = gn[1] = gn[2] = gn[3] = 0.0f;
      g_depth[p_idx] = 1.0f;
      return;
    }
    uint32_t tri_idx = rt_f2u32_sat(hit.tri_idx);
    const rt_topology& tri = topology[tri_idx];
    const rt_instance& inst = instances[hit.inst_idx];
    const float* inv = inst.inverse;
    rt3 v0 = get_pos(tri.v0), v1 = get_pos(tri.v1), v2 = get_pos(tri.v2);
    Ray rl = make_ray(rt_mat_mul_point(inv, ray.origin), rt_mat_mul_dir(inv, ray.direction));
    rt3 s = rl.origin - v0, e1 = v1 - v0, e2 = v2 - v0;
    rt3 h = rt_cross(rl.direction, e2);
    float f = 1.0f / rt_dot(e1, h);
    float ub = f * rt_dot(s, h);
    rt3 q = rt_cross(s, e1);
    float vb = f * rt_dot(rl.direction, q);
    float wb = 1.0f - ub - vb;

    rt3 wn0 = rt_normalize(rt_vec_mul_mat_dir(get_normal(tri.v0), inv));
    rt3 wn1 = rt_normalize(rt_vec_mul_mat_dir(get_normal(tri.v1), inv));
    rt3 wn2 = rt_normalize(rt_vec_mul_mat_dir(get_normal(tri.v2), inv));
    rt3 n = rt_normalize(wn0 * wb + wn1 * ub + wn2 * vb);
    rt2 pn = pack_normal(n);

    rt3 albedo = rt3_make(tri.data0[0], tri.data0[1], tri.data0[2]);
    if (tri.data2[0] > -0.5f) {
      rt2 tuv = get_uv(tri.v0) * wb + get_uv(tri.v1) * ub + get_uv(tri.v2) * vb;
      albedo = albedo * sample_tex(tuv, rt_f2i32_sat(tri.data2[0]));
    }
    ga[0] = (uint8_t)rt_unorm8(albedo.x);
    ga[1] = (uint8_t)rt_unorm8(albedo.y);
    ga[2] = (uint8_t)rt_unorm8(albedo.z);
    ga[3] = 255;
    gn[0] = pn.x;
    gn[1] = pn.y;
    gn[2] = rt_u2f(tri_idx);
    gn[3] = rt_u2f((uint32_t)hit.inst_idx);
    float z_view = hit.t * focal_length;
    float z_clip = z_view * (z_far / (z_far - z_near)) - (z_far * z_near) / (z_far - z_near);
    g_depth[p_idx] = z_clip / z_view;
  }

  // -------------------------------------------------------- post process
  rt3 get_radiance(int cx, int cy) const {  // PostProcess.wgsl:41-47
    int x = cx < 0 ? 0 : (cx > (int)scene.width - 1 ? (int)scene.width - 1 : cx);
    int y = cy < 0 ? 0 : (cy > (int)scene.height - 1 ? (int)scene.height - 1 : cy);
    const float* acc = &accum[((size_t)y * scene.width + (size_t)x) * 4];
    if (acc[3] <= 0.0f) return rt3_splat(0.0f);
    return rt3_make(acc[0], acc[1], acc[2]) / acc[3];
  }
  rt3 get_radiance_clean(int cx, int cy) const {  // :49-68
    rt3 center = get_radiance(cx, cy);
    rt3 min_nb = rt3_splat(1e6f), max_nb = rt3_splat(-1e6f);
    for (int y = -1; y <= 1; y++)
      for (int x = -1; x <= 1; x++) {
        if (x == 0 && y == 0) continue;
        rt3 nb = get_radiance(cx + x, cy + y);
        min_nb = rt_min3(min_nb, nb);
        max_nb = rt_max3(max_nb, nb);
      }
    const float threshold = 3.0f;
    return rt_clamp3(center, rt3_splat(0.0f), max_nb * threshold + rt3_splat(0.1f));
  }
  rt3 get_radiance_bilinear(float u, float v) const {  // :71-83
    float fx = u * (float)scene.width - 0.5f, fy = v * (float)scene.height - 0.5f;
    float flx = rt_floor(fx), fly = rt_floor(fy);
    // i32(floor(f)), kept within +-2^30 so that the +-1 offsets below cannot overflow a signed int; every
    // coordinate is clamped to the image in get_radiance, so no result changes
    auto texel = [](float f) {
      int i = rt_f2i32_sat(f);
      return i < -1073741824 ? -1073741824 : (i > 1073741823 ? 1073741823 : i);
    };
    int ix = texel(flx), iy = texel(fly);
    float wx = fx - flx, wy = fy - fly;
    rt3 c00 = get_radiance_clean(ix, iy), c10 = get_radiance_clean(ix + 1, iy);
    rt3 c01 = get_radiance_clean(ix, iy + 1), c11 = get_radiance_clean(ix + 1, iy + 1);
    return rt_mix3(rt_mix3(c00, c10, wx), rt_mix3(c01, c11, wx), wy);
  }
  rt3 get_radiance_nearest(int cx, int cy) const {  // :87-97
    if (scene.frame_count > 16u) return get_radiance_clean(cx, cy);
    float u = ((float)cx + 0.5f) / (float)scene.width;
    float v = ((float)cy + 0.5f) / (float)scene.height;
    return get_radiance_bilinear(u - scene.average_jitter[0], v - scene.average_jitter[1]);
  }
  static rt3 aces(rt3 color) {  // :36-39
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    rt3 num = color * (a * color + rt3_splat(b));
    rt3 den = color * (c * color + rt3_splat(d)) + rt3_splat(e);
    return rt_clamp3(num / den, rt3_splat(0.0f), rt3_splat(1.0f));
  }
  void post_pixel(uint32_t x, uint32_t y) {  // :103-176
    int ix = (int)x, iy = (int)y;
    rt3 center_color = get_radiance_nearest(ix, iy);
    const float SIGMA_S = 0.5f, SIGMA_R = 0.1f;
    const int RADIUS = 1;
    rt3 filtered_sum = rt3_splat(0.0f);
    float total_weight = 0.0f;
    for (int dy = -RADIUS; dy <= RADIUS; dy++)
      for (int dx = -RADIUS; dx <= RADIUS; dx++) {
        rt3 nc = get_radiance_nearest(ix + dx, iy + dy);
        float w_s = rt_exp(-(float)(dx * dx + dy * dy) / (2.0f * SIGMA_S * SIGMA_S));
        rt3 cd = nc - center_color;
        float w_r = rt_exp(-rt_dot(cd, cd) / (2.0f * SIGMA_R * (float)RADIUS * (float)RADIUS));
        float w = w_s * w_r;
        filtered_sum = filtered_sum + nc * w;
        total_weight += w;
      }
    rt3 denoised_hdr = filtered_sum / rt_max(total_weight, 1e-4f);

    // TAA: history sampled at the exact texel centre with a linear sampler == that texel
    const uint16_t* hp = &history[1 - history_index][((size_t)y * scene.width + x) * 4];
    rt3 samples_history = rt3_make(rt_f16_to_f32(hp[0]), rt_f16_to_f32(hp[1]), rt_f16_to_f32(hp[2]));
    rt3 m1 = rt3_splat(0.0f), m2 = rt3_splat(0.0f);
    for (int dy = -1; dy <= 1; dy++)
      for (int dx = -1; dx <= 1; dx++) {
        rt3 nc = get_radiance_nearest(ix + dx, iy + dy);
        m1 = m1 + nc;
        m2 = m2 + nc * nc;
      }
    rt3 mean = m1 / 9.0f;
    rt3 var = rt_max3(m2 / 9.0f - mean * mean, rt3_splat(0.0f));
    rt3 stddev = rt3_make(rt_sqrt(var.x), rt_sqrt(var.y), rt_sqrt(var.z));
    float k = 1.0f;
    if (scene.frame_count > 16u) k = 60.0f;
    rt3 clamped_history = rt_clamp3(samples_history, mean - stddev * k, mean + stddev * k);
    float alpha = 1.0f / (float)scene.frame_count;
    if (scene.frame_count == 1u) alpha = 0.1f;
    alpha = rt_max(alpha, 0.0001f);
    rt3 final_hdr = rt_mix3(clamped_history, denoised_hdr, alpha);

    uint16_t* ho = &history[history_index][((size_t)y * scene.width + x) * 4];
    ho[0] = rt_f32_to_f16(final_hdr.x);
    ho[1] = rt_f32_to_f16(final_hdr.y);
    ho[2] = rt_f32_to_f16(final_hdr.z);
    ho[3] = rt_f32_to_f16(1.0f);

    rt3 mapped = aces(final_hdr);
    rt3 edge_detect = center_color - denoised_hdr;
    rt3 sharpened = mapped + aces(edge_detect) * 0.3f;
    rt3 cl = rt_clamp3(sharpened, rt3_splat(0.0f), rt3_splat(1.0f));
    const float inv_gamma = 0.4545454680919647216796875f;  // f32(1.0 / 2.2)
    rt3 ldr = rt3_make(rt_pow(cl.x, inv_gamma), rt_pow(cl.y, inv_gamma), rt_pow(cl.z, inv_gamma));
    uint8_t* out = &post_out[((size_t)y * scene.width + x) * 4];
    out[0] = (uint8_t)rt_unorm8(ldr.x);
    out[1] = (uint8_t)rt_unorm8(ldr.y);
    out[2] = (uint8_t)rt_unorm8(ldr.z);
    out[3] = 255;
  }
  std::vector<uint8_t> post_out;  // staged, then copied to render_target (output texture)

  // -------------------------------------------------------------- driver
  // Work items are 128-pixel spans of the rows this renderer owns (a 1080p third is 5 400 items, so a few hundred
  // threads stay balanced); every item counts into a Counters on the worker's stack and is folded into the thread's
  // own cache line once per item (per-thread slots used to sit 48 B apart and were bumped on every node visit:
  // false sharing).  The workers are a persistent pool (WorkerPool below), not threads spawned per pass.
  struct alignas(128) PaddedCounters {
    Counters c;
  };
  static const uint32_t SPAN = 128;
  template <class F>
  void parallel_spans(bool owned_only, F&& fn) {
    std::vector<uint32_t> rows;
    rows.reserve(height);
    for (uint32_t y = 0; y < height; y++)
      if (!owned_only || owns_row(y)) rows.push_back(y);
    const uint32_t spans_per_row = (width + SPAN - 1) / SPAN;
    const uint32_t n_items = (uint32_t)rows.size() * spans_per_row;
    int nt = threads > 0 ? threads : usable_threads();
    if (nt < 1) nt = 1;
    if ((uint32_t)nt > n_items) nt = n_items ? (int)n_items : 1;
    std::vector<PaddedCounters> local((size_t)nt);
    std::atomic<uint32_t> next{0};
    auto worker = [&](int tid) {
      for (;;) {
        const uint32_t it = next.fetch_add(1, std::memory_order_relaxed);
        if (it >= n_items) break;
        const uint32_t y = rows[it / spans_per_row], x0 = (it % spans_per_row) * SPAN;
        const uint32_t x1 = x0 + SPAN < width ? x0 + SPAN : width;
        Counters item;
        fn(y, x0, x1, item);
        local[(size_t)tid].c.add(item);
      }
    };
    pool.run(nt, worker);
    for (auto& l : local) counters.add(l.c);
  }

  // threads this process may actually run on: the affinity mask, capped by the cgroup CPU quota when one is set
  static int usable_threads() {
    int n = 0;
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
    if (n < 1) n = (int)std::thread::hardware_concurrency();
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota> <period>" or "max <period>"
      char q[64];
      long long period = 0;
      if (fscanf(f, "%63s %lld", q, &period) == 2 && period > 0 && strcmp(q, "max") != 0) {
        const long long quota = atoll(q);
        const int cap = (int)((quota + period - 1) / period);
        if (cap >= 1 && cap < n) n = cap;
      }
      fclose(f);
    }
    return n < 1 ? 1 : n;
  }

  // Persistent worker pool: threads are created once (and again only when more are asked for) and parked on a
  // condition variable between passes; the calling thread works as worker 0.
  struct WorkerPool {
    std::vector<std::thread> threads;
    std::mutex m;
    std::condition_variable cv_start, cv_done;
    std::function<void(int)> job;
    uint64_t generation = 0;
    int active = 0, pending = 0;
    bool quit = false;
    void loop(int tid) {
      uint64_t seen = 0;
      for (;;) {
        std::function<void(int)> j;
        {
          std::unique_lock<std::mutex> lk(m);
          cv_start.wait(lk, [&] { return quit || (generation != seen && tid < active); });
          if (quit) return;
          seen = generation;
          j = job;
        }
        j(tid);
        {
          std::lock_guard<std::mutex> lk(m);
          if (--pending == 0) cv_done.notify_one();
        }
      }
    }
    template <class W>
    void run(int nt, W& worker) {
      if (nt <= 1) {
        worker(0);
        return;
      }
      while ((int)threads.size() < nt - 1) {
        const int tid = (int)threads.size() + 1;
        threads.emplace_back([this, tid] { loop(tid); });
      }
      {
        std::lock_guard<std::mutex> lk(m);
        job = [&worker](int tid) { worker(tid); };
        active = nt;
        pending = nt - 1;
        generation++;
      }
      cv_start.notify_all();
      worker(0);
      std::unique_lock<std::mutex> lk(m);
      cv_done.wait(lk, [&] { return pending == 0; });
    }
    ~WorkerPool() {
      {
        std::lock_guard<std::mutex> lk(m);
        quit = true;
      }
      cv_start.notify_all();
      for (auto& t : threads) t.join();
    }
  };
  WorkerPool pool;

  // ----------------------------------------------------------- uniforms
  static double halton(uint32_t index, uint32_t base) {  // ResourceManager.ts:348-357
    double f = 1, r = 0;
    while (index > 0) {
      f = f / (double)base;
      r = r + f * (double)(index % base);
      index = index / base;
    }
    return r;
  }
  void write_mixed(uint32_t frame_count) {  // ResourceManager.ts:374-403 / 426-446
    scene.frame_count = frame_count;
    scene.blas_base_idx = blas_offset;
    scene.vertex_count = vertex_count;
    scene.rand_seed = 0;  // Math.random() in the reference; never read by a shader
    scene.light_count = light_count;
    scene.width = width;
    scene.height = height;
    scene.pad = 0;
    scene.jitter[0] = (float)jx;
    scene.jitter[1] = (float)jy;
    scene.average_jitter[0] = (float)avg_jx;
    scene.average_jitter[1] = (float)avg_jy;
  }
  void step_jitter(uint32_t halton_index_source, uint32_t frame_count) {
    jx = (halton((halton_index_source % 16u) + 1u, 2) - 0.5) / (double)width;
    jy = (halton((halton_index_source % 16u) + 1u, 3) - 0.5) / (double)height;
    if (frame_count == 1u) {
      acc_jx = jx;
      acc_jy = jy;
    } else {
      acc_jx += jx;
      acc_jy += jy;
    }
    avg_jx = acc_jx / (double)frame_count;
    avg_jy = acc_jy / (double)frame_count;
  }
};

}  // namespace

// ------------------------------------------------------------------ C API
// Mirrors the WebGPURenderer method surface (src/renderer/WebGPURenderer.ts:7-138)
// so tests drive the oracle and the HIP renderer with the same call sequence.
extern "C" {

struct oracle_ctx {
  Oracle o;
};

oracle_ctx* oracle_create(void) { return new oracle_ctx(); }
void oracle_destroy(oracle_ctx* c) { delete c; }
void oracle_set_threads(oracle_ctx* c, int n) { c->o.threads = n; }
int oracle_hardware_threads(void) { return Oracle::usable_threads(); }  // affinity mask and cgroup quota respected

void oracle_build_pipeline(oracle_ctx* c, uint32_t depth, uint32_t spp) {
  c->o.max_depth = depth;
  c->o.spp = spp;
}
void oracle_update_screen_size(oracle_ctx* c, uint32_t w, uint32_t h) {
  Oracle& o = c->o;
  o.width = w;
  o.height = h;
  size_t n = (size_t)w * h;
  o.accum.assign(n * 4, 0.0f);
  o.render_target.assign(n * 4, 0);
  o.post_out.assign(n * 4, 0);
  o.g_normal.assign(n * 4, 0.0f);
  o.g_depth.assign(n, 1.0f);
  o.history[0].assign(n * 4, 0);
  o.history[1].assign(n * 4, 0);
}
void oracle_reset_accumulation(oracle_ctx* c) { std::fill(c->o.accum.begin(), c->o.accum.end(), 0.0f); }
void oracle_upload_textures(oracle_ctx* c, const uint8_t* rgba, uint32_t layers) {
  c->o.tex_layers = layers;
  c->o.tex.assign(rgba, rgba + (size_t)layers * RT_TEX_SIZE * RT_TEX_SIZE * 4);
}
// One 1024 x 1024 layer from a w x h RGBA8 image: createImageBitmap(blob, {resizeWidth: 1024, resizeHeight: 1024})
// with the default resizeQuality "low" (ResourceManager.ts:164-168) restated as bilinear, pixel centres aligned,
// clamp to edge (include/mi355rt_math.h rt_resize_coord / rt_bilinear_u8).  rgba == nullptr: the white fallback
// bitmap (:200-208).  The browser's own filter is not pinned by the reference (parity unpinned, see header).
void oracle_resize_texture(const uint8_t* rgba, uint32_t w, uint32_t h, uint8_t* out) {
  for (uint32_t y = 0; y < RT_TEX_SIZE; y++) {
    for (uint32_t x = 0; x < RT_TEX_SIZE; x++) {
      uint8_t* o = out + ((size_t)y * RT_TEX_SIZE + x) * 4;
      if (!rgba) {
        o[0] = o[1] = o[2] = o[3] = 255;
        continue;
      }
      uint32_t x0, x1, y0, y1;
      const float fx = rt_resize_coord(x, w, RT_TEX_SIZE, &x0, &x1);
      const float fy = rt_resize_coord(y, h, RT_TEX_SIZE, &y0, &y1);
      for (int k = 0; k < 4; k++) {
        const uint32_t c00 = rgba[((size_t)y0 * w + x0) * 4 + k], c10 = rgba[((size_t)y0 * w + x1) * 4 + k];
        const uint32_t c01 = rgba[((size_t)y1 * w + x0) * 4 + k], c11 = rgba[((size_t)y1 * w + x1) * 4 + k];
        o[k] = (uint8_t)rt_bilinear_u8(c00, c10, c01, c11, fx, fy);
      }
    }
  }
}
void oracle_update_topology(oracle_ctx* c, const uint32_t* data, size_t n_u32) {
  c->o.topology.resize(n_u32 / 20);
  std::memcpy(c->o.topology.data(), data, (n_u32 / 20) * sizeof(rt_topology));
}
void oracle_update_instances(oracle_ctx* c, const float* data, size_t n_f32) {
  c->o.instances.resize(n_f32 / 36);
  std::memcpy(c->o.instances.data(), data, (n_f32 / 36) * sizeof(rt_instance));
}
void oracle_update_lights(oracle_ctx* c, const uint32_t* data, size_t n_u32) {
  c->o.lights.resize(n_u32 / 2);
  std::memcpy(c->o.lights.data(), data, (n_u32 / 2) * sizeof(rt_light_ref));
}
void oracle_update_draw_commands(oracle_ctx* c, const uint32_t* data, size_t n_u32) {
  c->o.draw_commands.assign(data, data + n_u32);
}
void oracle_update_geometry(oracle_ctx* c, const float* v, const float* n, const float* uv, uint32_t vertex_count) {
  c->o.pos.assign(v, v + (size_t)vertex_count * 4);
  c->o.nrm.assign(n, n + (size_t)vertex_count * 4);
  c->o.uv.assign(uv, uv + (size_t)vertex_count * 2);
  c->o.vertex_count = vertex_count;
}
void oracle_update_bvh(oracle_ctx* c, const float* tlas, uint32_t n_tlas, const float* blas, uint32_t n_blas) {
  c->o.nodes.resize((size_t)n_tlas + n_blas);
  std::memcpy(c->o.nodes.data(), tlas, (size_t)n_tlas * sizeof(rt_node));
  std::memcpy(c->o.nodes.data() + n_tlas, blas, (size_t)n_blas * sizeof(rt_node));
  c->o.blas_offset = n_tlas;
}
// ResourceManager.updateSceneUniforms (ResourceManager.ts:359-405)
void oracle_update_scene_uniforms(oracle_ctx* c, const float cam[24], uint32_t frame_count, uint32_t light_count) {
  Oracle& o = c->o;
  o.light_count = light_count;
  o.step_jitter(frame_count, frame_count);
  std::memcpy(&o.scene.camera, cam, 96);
  std::memcpy(&o.scene.prev_camera, o.prev_camera, 96);
  o.write_mixed(frame_count);
  std::memcpy(o.prev_camera, cam, 96);
}
void oracle_set_stripes(oracle_ctx* c, uint32_t stripe_rows, uint32_t rank, uint32_t count) {
  c->o.stripe_rows = stripe_rows;
  c->o.stripe_rank = rank;
  c->o.stripe_count = count ? count : 1;
}
// WebGPURenderer.compute (WebGPURenderer.ts:88-102)
void oracle_compute(oracle_ctx* c, uint32_t frame_count) {
  Oracle& o = c->o;
  o.total_frames++;
  o.step_jitter(o.total_frames, frame_count);  // updateFrameUniforms: Halton index from totalFrames
  o.write_mixed(frame_count);
  if (o.nodes.empty() || o.topology.empty() || o.instances.empty() || o.width == 0) return;
  o.parallel_spans(true, [&](uint32_t y, uint32_t x0, uint32_t x1, Counters& cn) {
    for (uint32_t x = x0; x < x1; x++) o.gbuffer_pixel(x, y, cn);
  });
  o.parallel_spans(true, [&](uint32_t y, uint32_t x0, uint32_t x1, Counters& cn) {
    for (uint32_t x = x0; x < x1; x++) o.trace_pixel(x, y, cn);
  });
}
// WebGPURenderer.present (WebGPURenderer.ts:104-129)
void oracle_present(oracle_ctx* c) {
  Oracle& o = c->o;
  if (o.width == 0) return;
  o.parallel_spans(false, [&](uint32_t y, uint32_t x0, uint32_t x1, Counters&) {
    for (uint32_t x = x0; x < x1; x++) o.post_pixel(x, y);
  });
  o.render_target = o.post_out;
  o.history_index = 1 - o.history_index;
}
void oracle_capture_frame(oracle_ctx* c, uint8_t* out) {
  std::memcpy(out, c->o.render_target.data(), c->o.render_target.size());
}
void oracle_read_accum(oracle_ctx* c, float* out) { std::memcpy(out, c->o.accum.data(), c->o.accum.size() * 4); }
void oracle_write_accum(oracle_ctx* c, const float* in) { std::memcpy(c->o.accum.data(), in, c->o.accum.size() * 4); }
void oracle_read_gbuffer(oracle_ctx* c, uint8_t* albedo, float* normal_id, float* depth) {
  if (albedo) std::memcpy(albedo, c->o.render_target.data(), c->o.render_target.size());
  if (normal_id) std::memcpy(normal_id, c->o.g_normal.data(), c->o.g_normal.size() * 4);
  if (depth) std::memcpy(depth, c->o.g_depth.data(), c->o.g_depth.size() * 4);
}
// last-written history texture (the one the next present() will read)
void oracle_read_history(oracle_ctx* c, uint16_t* out) {
  const std::vector<uint16_t>& h = c->o.history[1 - c->o.history_index];
  std::memcpy(out, h.data(), h.size() * 2);
}
void oracle_read_uniforms(oracle_ctx* c, void* out256) { std::memcpy(out256, &c->o.scene, 256); }
void oracle_get_counters(oracle_ctx* c, uint64_t out[6]) {
  const Counters& k = c->o.counters;
  out[0] = k.primary_rays;
  out[1] = k.extension_rays;
  out[2] = k.shadow_rays;
  out[3] = k.nodes_visited;
  out[4] = k.tris_tested;
  out[5] = k.shaded_hits;
}
void oracle_reset_counters(oracle_ctx* c) { c->o.counters = Counters(); }

// ---- BVH validity check (tests/test_bvh_independent.py): the restated traversal against brute force ----
// rays: n x 8 f32 {o.xyz, t_min, d.xyz, t_max}.  out_bvh: n x 3 f32 {t, tri, inst} of intersect_tlas (Raytracer.wgsl:496-528).
// out_brute: n x 4 f32 {t, tri, inst, ties}: hit_triangle_raw (:443-453) on EVERY triangle of EVERY instance, the
// triangle range of an instance taken from its draw command (lib.rs:237-262) — no node array involved; the smallest t
// wins, on equal t the lowest (instance, triangle); ties = number of pairs sharing that smallest t.
// skip_tri (may be NULL): one byte per triangle, non-zero = brute force leaves it out (the triangles a fallback leaf with
// more than 7 entries loses to the 3-bit count overflow of blas.rs:111-115, which the reference's traversal never sees).
void oracle_trace_vs_brute_force(oracle_ctx* c, const float* rays, uint32_t n, float* out_bvh, float* out_brute,
                                 const uint8_t* skip_tri) {
  Oracle& o = c->o;
  const uint32_t saved_w = o.width, saved_h = o.height;
  const uint32_t sr = o.stripe_rows;
  o.stripe_rows = 0;       // parallel_spans over a fake n x 1 image: one item per 128 rays
  o.width = n;
  o.height = 1;
  const uint32_t n_inst = (uint32_t)o.instances.size();
  o.parallel_spans(false, [&](uint32_t, uint32_t x0, uint32_t x1, Counters& cn) {
    for (uint32_t k = x0; k < x1; k++) {
      const float* q = rays + (size_t)k * 8;
      Ray r = make_ray(rt3_make(q[0], q[1], q[2]), rt3_make(q[4], q[5], q[6]));
      HitResult h = o.intersect_tlas(r, q[3], q[7], cn);
      out_bvh[k * 3 + 0] = h.t;
      out_bvh[k * 3 + 1] = h.tri_idx;
      out_bvh[k * 3 + 2] = (float)h.inst_idx;
      float best_t = q[7], best_tri = -1.0f, best_inst = -1.0f, ties = 0.0f;
      for (uint32_t i = 0; i < n_inst && o.draw_commands.size() >= (size_t)4 * n_inst; i++) {
        const rt_instance& inst = o.instances[i];
        Ray rl = make_ray(rt_mat_mul_point(inst.inverse, r.origin), rt_mat_mul_dir(inst.inverse, r.direction));
        const uint32_t first = o.draw_commands[4 * i + 2] / 3u, count = o.draw_commands[4 * i] / 3u;
        for (uint32_t t = first; t < first + count; t++) {
          if (skip_tri && skip_tri[t]) continue;
          const rt_topology& tr = o.topology[t];
          float tt = Oracle::hit_triangle_raw(o.get_pos(tr.v0), o.get_pos(tr.v1), o.get_pos(tr.v2), rl, q[3], q[7]);
          if (tt > 0.0f) {
            if (tt < best_t) {
              best_t = tt;
              best_tri = (float)t;
              best_inst = (float)i;
              ties = 1.0f;
            } else if (tt == best_t) {
              ties += 1.0f;
            }
          }
        }
      }
      out_brute[k * 4 + 0] = best_t;
      out_brute[k * 4 + 1] = best_tri;
      out_brute[k * 4 + 2] = best_inst;
      out_brute[k * 4 + 3] = ties;
    }
  });
  o.width = saved_w;
  o.height = saved_h;
  o.stripe_rows = sr;
}

// Single rays through the restated traversal, with each ray's own counters (tests/test_pairwalk_model.py checks the
// child-pair walk of the HIP kernels against it ray by ray).  rays: n x 8 f32 {o.xyz, t_min, d.xyz, t_max};
// any = 0: intersect_tlas (Raytracer.wgsl:496-528) -> out n x 4 f32 {t, tri, inst, 0};
// any = 1: intersect_tlas_shadow (:566-600) -> {0, 0, 0, occluded}.  counts: n x 2 u64 {nodes_visited, tris_tested}.
void oracle_trace_rays(oracle_ctx* c, const float* rays, uint32_t n, int any, float* out, uint64_t* counts) {
  Oracle& o = c->o;
  for (uint32_t k = 0; k < n; k++) {
    const float* q = rays + (size_t)k * 8;
    Ray r = make_ray(rt3_make(q[0], q[1], q[2]), rt3_make(q[4], q[5], q[6]));
    Counters cn;
    float* w = out + (size_t)k * 4;
    w[0] = w[1] = w[2] = w[3] = 0.0f;
    if (any) {
      w[3] = o.intersect_tlas_shadow(r, q[3], q[7], cn) ? 1.0f : 0.0f;
    } else {
      HitResult h = o.intersect_tlas(r, q[3], q[7], cn);
      w[0] = h.t;
      w[1] = h.tri_idx;
      w[2] = (float)h.inst_idx;
    }
    counts[2 * (size_t)k] = cn.nodes_visited;
    counts[2 * (size_t)k + 1] = cn.tris_tested;
  }
}

// per-node visit counts of the following compute() calls (n_nodes u32, zeroed here); NULL switches the histogram off
void oracle_set_node_histogram(oracle_ctx* c, uint32_t* hist) {
  c->o.node_hist = reinterpret_cast<std::atomic<uint32_t>*>(hist);
  if (hist) std::memset(hist, 0, c->o.nodes.size() * 4);
}

// ---- unit-level entry points for known-answer tests ----
uint32_t oracle_init_rng(uint32_t pixel_idx, uint32_t frame) { return Oracle::init_rng(pixel_idx, frame); }
float oracle_rand_pcg(uint32_t* state) { return Oracle::rand_pcg(state); }
double oracle_halton(uint32_t index, uint32_t base) { return Oracle::halton(index, base); }
void oracle_sincos(float x, float* s, float* c) { rt_sincos(x, s, c); }
float oracle_exp(float x) { return rt_exp(x); }
float oracle_log(float x) { return rt_log(x); }
float oracle_pow(float x, float y) { return rt_pow(x, y); }
uint16_t oracle_f32_to_f16(float x) { return rt_f32_to_f16(x); }
float oracle_f16_to_f32(uint16_t h) { return rt_f16_to_f32(h); }
float oracle_min(float a, float b) { return rt_min(a, b); }
float oracle_max(float a, float b) { return rt_max(a, b); }
void oracle_pack_normal(const float n[3], float out[2]) {
  rt2 p = Oracle::pack_normal(rt3_make(n[0], n[1], n[2]));
  out[0] = p.x;
  out[1] = p.y;
}
void oracle_unpack_normal(const float p[2], float out[3]) {
  rt3 n = Oracle::unpack_normal(rt2_make(p[0], p[1]));
  out[0] = n.x;
  out[1] = n.y;
  out[2] = n.z;
}
float oracle_hit_triangle(const float v0[3], const float v1[3], const float v2[3], const float o[3], const float d[3],
                          float t_min, float t_max) {
  Ray r = make_ray(rt3_make(o[0], o[1], o[2]), rt3_make(d[0], d[1], d[2]));
  return Oracle::hit_triangle_raw(rt3_make(v0[0], v0[1], v0[2]), rt3_make(v1[0], v1[1], v1[2]),
                                  rt3_make(v2[0], v2[1], v2[2]), r, t_min, t_max);
}
float oracle_intersect_aabb(const float mn[3], const float mx[3], const float o[3], const float d[3], float t_min,
                            float t_max) {
  rt_node n;
  std::memcpy(n.min_b, mn, 12);
  std::memcpy(n.max_b, mx, 12);
  Ray r = make_ray(rt3_make(o[0], o[1], o[2]), rt3_make(d[0], d[1], d[2]));
  return Oracle::intersect_aabb(n, r, t_min, t_max);
}

}  // extern "C"
