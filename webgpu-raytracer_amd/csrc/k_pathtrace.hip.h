// k_pathtrace.hip.h — k_pathtrace (one pixel per lane), the wave-level traverse() with its LDS triangle queue, shade_bounce() and k_pathtrace_persistent.
// Part of the kernel set of csrc/kernels.hip.h (included from there, in order; not a stand-alone header).
#ifndef MI355RT_K_PATHTRACE_HIP_H
#define MI355RT_K_PATHTRACE_HIP_H

namespace rtk {

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
            bsdf_val = rt_div_pi3(albedo);
            bsdf_pdf = rt_div_pi(rt_max(rt_dot(normal, ls.dir), 0.0f));
          } else if (mat_type == 1u) {
            bsdf_val = eval_ggx(normal, -rd, ls.dir, roughness, f0);
            rt3 H = rt_normalize(-rd + ls.dir);
            bsdf_pdf = rt_div(ggx_d(rt_dot(normal, H), roughness * roughness) * rt_max(rt_dot(normal, H), 0.0f),
                              4.0f * rt_max(rt_dot(-rd, H), 0.0f));
          }
          if (bsdf_pdf > 0.0f) {
            radiance = radiance + rt_div3z(throughput * bsdf_val * ls.L * power_heuristic(ls.pdf, bsdf_pdf) *
                                               rt_max(rt_dot(normal, ls.dir), 0.0f), ls.pdf);
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
      throughput = rt_div3z(throughput, p);
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
      float u = rt_div((float)x + 0.5f + U.jitter[0] * (float)U.width, (float)U.width);
      float v = 1.0f - rt_div((float)y + 0.5f + U.jitter[1] * (float)U.height, (float)U.height);
      rt3 d = cam_ll + u * cam_h + v * cam_v - cam_o - off;
      col = col + ray_color<DETAIL>(S, F, U, cam_o + off, d, rng, p_idx, c);
    }
    if (F.spp != 1u) col = rt_div3z(col, (float)F.spp);   // x / 1 = x
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

// (the wave-level walk itself — TravMem, trav_step, trav_flush, traverse() — is in k_traverse.hip.h)

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
  const float4* ts = S.tri_shade + 8 * (size_t)p.tri;   // the hit's shading record: one 128-byte line
  const float4 q4 = ts[4], q5 = ts[5], q6 = ts[6], q7 = ts[7];
  p.tex_uv = rt2_make(q4.w, q5.w) * b.w + rt2_make(q6.w, q7.x) * b.u + rt2_make(q7.y, q7.z) * b.v;
  if (from_gbuffer) {
    p.hit_t = b.t;
    p.normal = unpack_normal(gx, gy);
    p.albedo = rt3_make(rt_from_unorm8(galbedo & 255u), rt_from_unorm8((galbedo >> 8) & 255u),
                        rt_from_unorm8((galbedo >> 16) & 255u));
  } else {
    rt3 ln = rt_normalize(xyz(q4) * b.w + xyz(q5) * b.u + xyz(q6) * b.v);
    p.normal = rt_normalize(normal_to_world(m, ln));
    float4 nd0 = ts[0], nd2 = ts[2];
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
  const float4* ts = S.tri_shade + 8 * (size_t)p.tri;
  float4 d0 = ts[0], d1 = ts[1], d2 = ts[2], d3 = ts[3];
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
          bsdf_val = rt_div_pi3(p.albedo);
          bsdf_pdf = rt_div_pi(rt_max(rt_dot(p.normal, ls.dir), 0.0f));
        } else if (mat_type == 1u) {
          bsdf_val = eval_ggx(p.normal, -p.rd, ls.dir, roughness, f0);
          rt3 H = rt_normalize(-p.rd + ls.dir);
          bsdf_pdf = rt_div(ggx_d(rt_dot(p.normal, H), roughness * roughness) * rt_max(rt_dot(p.normal, H), 0.0f),
                            4.0f * rt_max(rt_dot(-p.rd, H), 0.0f));
        }
        o.want_shadow = true;  // the reference traces the shadow ray before looking at bsdf_pdf
        o.sh_o = hit_p + p.geom_n * 1e-4f;
        o.sh_d = ls.dir;
        o.sh_tmax = ls.dist - 2e-4f;
        o.nee_valid = bsdf_pdf > 0.0f;
        if (o.nee_valid) {
          o.nee = rt_div3z(p.throughput * bsdf_val * ls.L * power_heuristic(ls.pdf, bsdf_pdf) *
                               rt_max(rt_dot(p.normal, ls.dir), 0.0f), ls.pdf);
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
          p.throughput = rt_div3z(p.throughput, pr);
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
  // tnodes, tri_geom, inst_trav, inst_root | tri_shade | topo, pos, uv (light_pdf / light sampling of emissive hits), inst,
  // lights, light_rec
  return (size_t)2 * n_nodes + (size_t)RT_TRI_STRIDE * n_tris + (size_t)4 * n_inst + ((size_t)n_inst + 3) / 4 + (size_t)8 * n_tris +
         (size_t)5 * n_tris + (size_t)n_verts + ((size_t)n_verts + 1) / 2 + (size_t)9 * n_inst + ((size_t)n_lights + 1) / 2 +
         (size_t)4 * n_lights;
}

// Diagnostic build only (-DRT_CLOCK_STAMP, tools/clock_check.py): every workgroup of the persistent kernel stamps
// s_memtime / s_memrealtime around its work loop into this array, which nothing else reads; the in-kernel clock is
// delta(memtime) / delta(memrealtime) x 100 MHz.  In the product build no stamp executes.
#define RT_CLOCK_STAMP_SLOTS 4096
__device__ unsigned long long g_clock_stamps[2 * RT_CLOCK_STAMP_SLOTS];
// Diagnostic build only (-DRT_PT_STAMPS, tools/pt_sections.py): s_memtime cycles the persistent kernel's waves spend in
// the five sections of a trip (regenerate + start a sample, shade, shadow traversal, extension traversal + surface frame,
// finish), summed over all waves; [5] = trips, [6] = waves.
__device__ unsigned long long g_pt_sections[8];

// Occupancy: the LDS-resident form is VALU-issue bound (3, 4, 5 waves/SIMD within 2 %), the global-memory form
// is latency bound and gains ~11 % from 6 waves/SIMD even with the spills that costs (measured on MI355X).
// What a workgroup stages in LDS behind its wave queues (decided on the host from the scene's size, rt_api.hip plan_lds):
// the first k_nodes records of tnodes, and — when they fit as a whole — the instance rows + BLAS roots and the triangle
// records.  LDS = true (the whole scene fits, shading arrays included) ignores it.
struct LdsPlan {
  uint32_t k_nodes, stage_inst, stage_tri, pad;
};

// Fill TravMem for the mixed mode and stage what the plan names; returns the number of 16-byte slots used.
__device__ __forceinline__ uint32_t trav_stage_mixed(TravMem& M, f4* lds, uint32_t slot0, const DevScene& Sg, const LdsPlan& P,
                                                     uint32_t n_tris_total, uint32_t n_inst_total) {
  uint32_t slot = slot0;
  M.gnodes = reinterpret_cast<const f4*>(Sg.tnodes);
  M.gtri = reinterpret_cast<const f4*>(Sg.tri_geom);
  M.ginst = reinterpret_cast<const f4*>(Sg.inst_trav);
  M.groot = Sg.inst_root;
  M.k_lds = P.k_nodes;
  M.l_nodes = slot;
  lds_stage(lds + slot, Sg.tnodes, (size_t)2 * P.k_nodes);
  slot += 2u * P.k_nodes;
  M.l_inst = M.l_root = M.l_tri = RT_LDS_NONE;
  if (P.stage_inst) {
    M.l_inst = slot;
    lds_stage(lds + slot, Sg.inst_trav, (size_t)4 * n_inst_total);
    slot += 4u * n_inst_total;
    M.l_root = slot;
    lds_stage(lds + slot, Sg.inst_root, ((size_t)n_inst_total + 3) / 4);   // the buffer has 16 bytes of slack
    slot += (n_inst_total + 3u) / 4u;
  }
  if (P.stage_tri) {
    M.l_tri = slot;
    lds_stage(lds + slot, Sg.tri_geom, (size_t)RT_TRI_STRIDE * n_tris_total);
    slot += (uint32_t)RT_TRI_STRIDE * n_tris_total;
  }
  return slot - slot0;
}

// Occupancy: the LDS-resident form is latency bound at 4 waves/SIMD (registers), the global-memory form gains ~11 %
// from 6 waves/SIMD even with the spills that costs (measured on MI355X).
#ifndef RT_PT_LDS_WAVES
#define RT_PT_LDS_WAVES 4      // waves per SIMD of the LDS-resident form (128 VGPRs)
#endif
#ifndef RT_PT_GLOBAL_WAVES
#define RT_PT_GLOBAL_WAVES 6   // workgroups of 4 waves per CU = waves per SIMD of the global-memory form (tools/SWEEPS.md)
#endif
template <bool DETAIL, bool LDS>
__global__ __launch_bounds__(256, LDS ? RT_PT_LDS_WAVES : RT_PT_GLOBAL_WAVES) void k_pathtrace_persistent(DevScene Sg, DevFrame F, rt_scene_uniforms U,
                                                              uint32_t* __restrict__ ticket, uint32_t n_nodes_total,
                                                              uint32_t n_tris_total, uint32_t n_inst_total,
                                                              uint32_t n_verts_total,
                                                              const DevFrameSlot* __restrict__ slots, uint32_t n_slots,
                                                              LdsPlan plan) {
  // Batched dispatch (rt_compute_batch): the launch covers n_slots consecutive compute() frames. The work item is
  // one (frame, pixel): tickets enumerate (frame, tile) pairs, so a launch has n_slots times as many tickets and the
  // persistent waves stay fed and balanced even when a rank owns 1/8 of the image. With n_slots > 1 every item
  // writes its frame colour to F.frame_col and k_accumulate_frames adds the frames in frame order afterwards, which
  // makes the result bit-identical to n_slots separate dispatches; with n_slots == 1 the item accumulates directly.
  extern __shared__ f4 s_scene[];
  // per-wave triangle work queue at the start of LDS, staged scene after it
  WaveWork WW;
  {
    wave_work_at(WW, reinterpret_cast<char*>(s_scene) + (threadIdx.x >> 6) * RT_WORK_BYTES_PER_WAVE);
  }
  const uint32_t rec0 = (4 * RT_WORK_BYTES_PER_WAVE) / 16;   // first slot behind the wave queues
  TravMem M;
  DevScene S = Sg;
  if (LDS) {
    // Small scene: the whole scene (traversal records AND the arrays shading reads) lives in LDS,
    // staged once per workgroup; only textures, the G-buffer and the accumulation buffer stay in HBM.
    uint32_t slot = rec0;
    auto stage = [&](const void* src, size_t n) {
      f4* base = s_scene + slot;
      lds_stage(base, src, n);
      slot += (uint32_t)n;
      return base;
    };
    M.gnodes = M.gtri = M.ginst = nullptr;
    M.groot = nullptr;
    M.k_lds = n_nodes_total;
    M.l_nodes = slot;
    f4* ln = stage(Sg.tnodes, (size_t)2 * n_nodes_total);
    M.l_tri = slot;
    f4* lt = stage(Sg.tri_geom, (size_t)RT_TRI_STRIDE * n_tris_total);
    M.l_inst = slot;
    f4* li = stage(Sg.inst_trav, (size_t)4 * n_inst_total);
    M.l_root = slot;
    stage(Sg.inst_root, ((size_t)n_inst_total + 3) / 4);
    S.tri_shade = reinterpret_cast<const float4*>(stage(Sg.tri_shade, (size_t)8 * n_tris_total));
    S.topo = reinterpret_cast<const float4*>(stage(Sg.topo, (size_t)5 * n_tris_total));
    S.pos = reinterpret_cast<const float4*>(stage(Sg.pos, n_verts_total));   // S.nrm stays in global memory: no reader left here
    // uv (8 B/vertex) and lights (8 B each): the device buffers are allocated with >= 16-byte slack
    S.uv = reinterpret_cast<const float2*>(stage(Sg.uv, ((size_t)n_verts_total + 1) / 2));
    S.inst = reinterpret_cast<const float4*>(stage(Sg.inst, (size_t)9 * n_inst_total));
    S.lights = reinterpret_cast<const uint2*>(stage(Sg.lights, ((size_t)Sg.n_lights + 1) / 2));
    S.light_rec = reinterpret_cast<const float4*>(stage(Sg.light_rec, (size_t)4 * Sg.n_lights));
    __syncthreads();
    S.tnodes = reinterpret_cast<const float4*>(ln);
    S.tri_geom = reinterpret_cast<const float4*>(lt);
    S.inst_trav = reinterpret_cast<const float4*>(li);
  } else {
    trav_stage_mixed(M, s_scene, rec0, Sg, plan, n_tris_total, n_inst_total);
    __syncthreads();
  }
  constexpr int MODE = LDS ? RT_TRAV_LDS : RT_TRAV_MIXED;

#ifdef RT_CLOCK_STAMP
  const unsigned long long stamp_c0 = __builtin_amdgcn_s_memtime(), stamp_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t tiles_x = (U.width + 7u) / 8u;
  // tickets enumerate only the tile rows this rank owns when the stripes are tile-aligned
  const uint32_t n_tiles = tiles_x * (F.own_period ? F.own_tile_rows : (U.height + 7u) / 8u);
  const rt3 cam_o = rt3_make(U.camera.origin[0], U.camera.origin[1], U.camera.origin[2]);
  const rt3 cam_ll = rt3_make(U.camera.lower_left[0], U.camera.lower_left[1], U.camera.lower_left[2]);
  const rt3 cam_h = rt3_make(U.camera.horizontal[0], U.camera.horizontal[1], U.camera.horizontal[2]);
  const rt3 cam_v = rt3_make(U.camera.vertical[0], U.camera.vertical[1], U.camera.vertical[2]);
  const float lens = U.camera.origin[3];

  // wave-uniform work cursor: pixels [tile_pos, 64) of the wave's tile are still unassigned; the tile's origin and frame are
  // worked out once per ticket (two divisions by run-time values, 20 instructions each: not once per regenerated lane)
  uint32_t tile_pos = 64u, tile_x0 = 0u, tile_y0 = 0u, tile_slot = 0u;
  bool work_left = true;

  PathState p;
  uint32_t item_slot = 0u;  // frame of the batch the lane's current (frame, pixel) item belongs to
  uint32_t pixel_xy = 0u;   // x | y << 16 of p.pixel
  bool alive = false;       // lane owns a running path
  bool have_pixel = false;  // lane owns a pixel whose samples are not all done
  uint32_t cnt_ext = 0, cnt_shadow = 0, cnt_nodes = 0, cnt_tris = 0, cnt_shaded = 0;
  p.pixel = 0; p.rng = 0; p.depth = 0; p.sample = 0; p.prev_pdf = 0.0f; p.specular = true; p.hit_t = 0.0f;
  p.tri = 0; p.inst = 0;
  p.ro = p.rd = p.throughput = p.radiance = p.col = p.normal = p.geom_n = p.albedo = rt3_splat(0.0f);
  p.tex_uv = rt2_make(0.0f, 0.0f);

#ifdef RT_PT_STAMPS
  unsigned long long pt_cyc[5] = {0, 0, 0, 0, 0}, pt_trips = 0;
#endif
  for (;;) {
#ifdef RT_PT_STAMPS
    const unsigned long long ps0 = __builtin_amdgcn_s_memtime();
#endif
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
          // frame-major ticket: frame = t / n_tiles, tile = t % n_tiles
          tile_slot = t / n_tiles;
          const uint32_t tile_in_frame = t - tile_slot * n_tiles;
          uint32_t trow = tile_in_frame / tiles_x;
          tile_x0 = (tile_in_frame - trow * tiles_x) * 8u;
          if (F.own_period) trow = (trow / F.own_run) * F.own_period + F.own_first + (trow % F.own_run);
          tile_y0 = trow * 8u;
          tile_pos = 0u;
        }
        // rank of this lane among the needy lanes
        const uint32_t rank =
            __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
        const uint32_t slot = tile_pos + rank;
        if (need && slot < 64u) {
          const uint32_t x = tile_x0 + (slot & 7u);
          const uint32_t y = tile_y0 + (slot >> 3);
          need = false;
          if (x < U.width && y < U.height && owns_row(F, y)) {
            have_pixel = true;
            p.pixel = y * U.width + x;
            pixel_xy = x | (y << 16);   // width, height <= 65535: rt_resize refuses more
            p.sample = 0u;
            item_slot = tile_slot;
            p.col = rt3_splat(0.0f);
          }
        }
        tile_pos += (uint32_t)__builtin_popcountll(mask);
      }
    }
    // (b) start the next sample of the owned pixel: camera ray + depth-0 surface from the G-buffer
    RT_LSTAT(6, !alive && have_pixel);
    if (!alive && have_pixel) {
      const uint32_t x = pixel_xy & 0xffffu, y = pixel_xy >> 16;
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
      float u = rt_div((float)x + 0.5f + slot.jitter_x * (float)U.width, (float)U.width);
      float v = 1.0f - rt_div((float)y + 0.5f + slot.jitter_y * (float)U.height, (float)U.height);
      p.rd = cam_ll + u * cam_h + v * cam_v - cam_o - off;
      p.ro = cam_o + off;
      p.throughput = rt3_splat(1.0f);
      p.radiance = rt3_splat(0.0f);
      p.prev_pdf = 0.0f;
      p.specular = true;
      p.depth = 0u;
      // background pixel (or MAX_DEPTH = 0): the sample is black and ends at once.  The three G-buffer words of the
      // pixel are requested together (they come from HBM: one round trip instead of depth first, then the rest)
      const float gdepth = slot.depth[p.pixel];
      const float4 g = slot.normal_id[p.pixel];
      const uint32_t galbedo = slot.albedo[p.pixel];
      if (!(gdepth >= 1.0f) && F.max_depth != 0u) {
        p.tri = rt_f2u(g.z);
        p.inst = rt_f2u(g.w);
        setup_surface(S, p, true, g.x, g.y, galbedo);
        alive = true;
      }
    }
    const bool running = alive;
    bool path_done = have_pixel && !alive;  // background sample ends immediately

#ifdef RT_PT_STAMPS
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long ps1 = __builtin_amdgcn_s_memtime();
#endif
    // ------------------------------------------------------------ shade one bounce
    bool want_shadow = false, want_extend = false;
    bool nee_valid = false;
    rt3 sh_o = rt3_splat(0.0f), sh_d = rt3_splat(0.0f), nee = rt3_splat(0.0f);
    float sh_tmax = 0.0f;
    RT_LSTAT(0, running);
    if (running) {
      if (DETAIL) cnt_shaded++;
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
#ifdef RT_PT_STAMPS
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long ps2 = __builtin_amdgcn_s_memtime();
#endif
    // ------------------------------------------------------------ shadow rays (any hit)
    if (__ballot(want_shadow) != 0ull) {
      float t_;
      int32_t a_, b_;
      bool occluded;
      traverse<true, DETAIL, MODE>(M, s_scene, WW, U.blas_base_idx, want_shadow, sh_o, sh_d, sh_tmax, t_, a_, b_,
                                   occluded, cnt_nodes, cnt_tris);
      if (want_shadow) {
        cnt_shadow++;
        if (!occluded && nee_valid) p.radiance = p.radiance + nee;  // nothing is added when bsdf_pdf <= 0
      }
    }

#ifdef RT_PT_STAMPS
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long ps3 = __builtin_amdgcn_s_memtime();
#endif
    // ------------------------------------------------------------ extension rays (closest hit)
    if (__ballot(want_extend) != 0ull) {
      float t_;
      int32_t tri_, inst_;
      bool any_;
      traverse<false, DETAIL, MODE>(M, s_scene, WW, U.blas_base_idx, want_extend, p.ro, p.rd, RT_T_MAX, t_, tri_,
                                    inst_, any_, cnt_nodes, cnt_tris);
      RT_LSTAT(5, want_extend && inst_ >= 0);
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

#ifdef RT_PT_STAMPS
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long ps4 = __builtin_amdgcn_s_memtime();
#endif
    // ------------------------------------------------------------ sample / pixel finished
    RT_LSTAT(7, path_done);
    if (path_done) {
      alive = false;
      p.col = p.col + p.radiance;
      p.sample++;
      if (p.sample >= F.spp) {  // the item's last sample: Raytracer.wgsl:811-818
        rt3 c = p.col;
        if (F.spp != 1u) c = rt_div3z(p.col, (float)F.spp);   // x / 1 = x, bit for bit
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
#ifdef RT_PT_STAMPS
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    {
      const unsigned long long ps5 = __builtin_amdgcn_s_memtime();
      pt_cyc[0] += ps1 - ps0; pt_cyc[1] += ps2 - ps1; pt_cyc[2] += ps3 - ps2; pt_cyc[3] += ps4 - ps3; pt_cyc[4] += ps5 - ps4;
      pt_trips++;
    }
#endif
    if (!work_left && __ballot(alive || have_pixel) == 0ull) break;
  }
#ifdef RT_PT_STAMPS
  if (lane == 0u) {
    for (int k = 0; k < 5; k++) atomicAdd(&g_pt_sections[k], pt_cyc[k]);
    atomicAdd(&g_pt_sections[5], pt_trips);
    atomicAdd(&g_pt_sections[6], 1ull);
  }
#endif

#ifdef RT_CLOCK_STAMP
  if (threadIdx.x == 0 && blockIdx.x < RT_CLOCK_STAMP_SLOTS) {
    g_clock_stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - stamp_c0;
    g_clock_stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - stamp_r0;
  }
#endif
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

}  // namespace rtk
#endif
