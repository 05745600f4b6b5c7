// k_common.hip.h — small helpers, texture fetch and the RNG (Raytracer.wgsl:178-199, textureSampleLevel).
// Part of the kernel set of csrc/kernels.hip.h (included from there, in order; not a stand-alone header).
#ifndef MI355RT_K_COMMON_HIP_H
#define MI355RT_K_COMMON_HIP_H

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

}  // namespace rtk
#endif
