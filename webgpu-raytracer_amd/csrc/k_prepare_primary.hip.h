// k_prepare_primary.hip.h — upload-time re-layout kernels and k_primary_visibility (Rasterizer.wgsl:81-173 restated as a ray cast).
// Part of the kernel set of csrc/kernels.hip.h (included from there, in order; not a stand-alone header).
#ifndef MI355RT_K_PREPARE_PRIMARY_HIP_H
#define MI355RT_K_PREPARE_PRIMARY_HIP_H

namespace rtk {

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
  tri_geom[RT_TRI_STRIDE * i + 0] = make_float4(v0.x, v0.y, v0.z, 0.0f);
  tri_geom[RT_TRI_STRIDE * i + 1] = make_float4(e1.x, e1.y, e1.z, 0.0f);
  tri_geom[RT_TRI_STRIDE * i + 2] = make_float4(e2.x, e2.y, e2.z, 0.0f);
}
// per-triangle shading record (device_scene.h): pure copies
__global__ void k_prepare_tri_shade(const float4* __restrict__ topo, const float4* __restrict__ nrm,
                                    const float2* __restrict__ uv, float4* __restrict__ tri_shade, uint32_t n_tris,
                                    uint32_t n_verts) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_tris) return;
  float4 idx = topo[5 * i];
  uint32_t i0 = rt_f2u(idx.x), i1 = rt_f2u(idx.y), i2 = rt_f2u(idx.z);
  if (i0 >= n_verts) i0 = n_verts - 1;  // robust buffer access: clamp instead of faulting
  if (i1 >= n_verts) i1 = n_verts - 1;
  if (i2 >= n_verts) i2 = n_verts - 1;
  const float4 n0 = nrm[i0], n1 = nrm[i1], n2 = nrm[i2];
  const float2 t0 = uv[i0], t1 = uv[i1], t2 = uv[i2];
  float4* dst = tri_shade + 8 * (size_t)i;
  dst[0] = topo[5 * i + 1];
  dst[1] = topo[5 * i + 2];
  dst[2] = topo[5 * i + 3];
  dst[3] = topo[5 * i + 4];
  dst[4] = make_float4(n0.x, n0.y, n0.z, t0.x);
  dst[5] = make_float4(n1.x, n1.y, n1.z, t0.y);
  dst[6] = make_float4(n2.x, n2.y, n2.z, t1.x);
  dst[7] = make_float4(t1.y, t2.x, t2.y, 0.0f);
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
// 16-byte LDS slots of the records this kernel reads (nodes, triangle records, instance rows, per-triangle shading records)
__host__ __device__ inline size_t primary_lds_slots(uint32_t n_nodes, uint32_t n_tris, uint32_t n_inst, uint32_t /*n_verts*/) {
  return (size_t)2 * n_nodes + (size_t)RT_TRI_STRIDE * n_tris + (size_t)4 * n_inst + (size_t)8 * n_tris;
}

// LDS = false: one wave (one 8x8 tile) per workgroup, records through L1 / L2.  LDS = true (small scenes): four tiles per
// 256-thread workgroup and the records staged in LDS first — the walk is a chain of dependent fetches, and an LDS
// fetch returns in a fraction of an L1 hit's time.
template <bool DETAIL, bool LDS>
__global__ __launch_bounds__(LDS ? 256 : 64) void k_primary_visibility(DevScene Sg, DevFrame F, rt_scene_uniforms U,
                                                                       const DevFrameSlot* __restrict__ slots, uint32_t n_tiles,
                                                                       uint32_t n_nodes_total, uint32_t n_tris_total,
                                                                       uint32_t n_inst_total, uint32_t n_verts_total) {
  extern __shared__ float4 s_primary[];
  DevScene S = Sg;
  if (LDS) {
    float4* dst = s_primary;
    auto stage = [&](const void* src, size_t slots_n) {
      const float4* g = reinterpret_cast<const float4*>(src);
      float4* base = dst;
      for (uint32_t i = threadIdx.x; i < slots_n; i += 256) base[i] = g[i];
      dst += slots_n;
      return base;
    };
    S.nodes = stage(Sg.nodes, (size_t)2 * n_nodes_total);
    S.tri_geom = stage(Sg.tri_geom, (size_t)RT_TRI_STRIDE * n_tris_total);
    S.inst_trav = stage(Sg.inst_trav, (size_t)4 * n_inst_total);
    S.tri_shade = stage(Sg.tri_shade, (size_t)8 * n_tris_total);
    __syncthreads();
  }
  const uint32_t tile_id = LDS ? blockIdx.x * 4u + (threadIdx.x >> 6) : blockIdx.x;
  const uint32_t lane_id = threadIdx.x & 63u;
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
    uint32_t trow = tile_id / tiles_x;
    trow = (trow / F.own_run) * F.own_period + F.own_first + (trow % F.own_run);
    x = (tile_id % tiles_x) * 8u + (lane_id & 7u);
    y = trow * 8u + (lane_id >> 3);
    live = x < U.width && y < U.height;
  } else {
    live = tile_pixel(U, tile_id, lane_id, x, y) && owns_row(F, y);
  }
  live = live && tile_id < n_tiles;
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
      const float4* ts = S.tri_shade + 8 * (size_t)hit.tri;
      const float4 d0 = ts[0], d2 = ts[2], q4 = ts[4], q5 = ts[5], q6 = ts[6], q7 = ts[7];
      rt3 wn0 = rt_normalize(normal_to_world(m, xyz(q4)));
      rt3 wn1 = rt_normalize(normal_to_world(m, xyz(q5)));
      rt3 wn2 = rt_normalize(normal_to_world(m, xyz(q6)));
      rt3 n = rt_normalize(wn0 * b.w + wn1 * b.u + wn2 * b.v);
      rt2 pn = pack_normal(n);
      rt3 albedo = xyz(d0);
      if (d2.x > -0.5f) {
        rt2 tuv = rt2_make(q4.w, q5.w) * b.w + rt2_make(q6.w, q7.x) * b.u + rt2_make(q7.y, q7.z) * b.v;
        albedo = albedo * sample_tex(S, tuv, rt_f2i32_sat(d2.x));
      }
      F.albedo[p_idx] = rt_unorm8(albedo.x) | (rt_unorm8(albedo.y) << 8) | (rt_unorm8(albedo.z) << 16) | (255u << 24);
      F.normal_id[p_idx] = make_float4(pn.x, pn.y, rt_u2f((uint32_t)hit.tri), rt_u2f((uint32_t)hit.inst));
      float z_view = hit.t * focal_length;
      float z_clip = z_view * (z_far / (z_far - z_near)) - (z_far * z_near) / (z_far - z_near);
      F.depth[p_idx] = z_clip / z_view;
    }
  }
  flush_counters<DETAIL>(c, F.counters, tile_id + blockIdx.y * 977u);
}

}  // namespace rtk
#endif
