// k_texture_post.hip.h — k_resize_texture (texture ingest) and k_postprocess (PostProcess.wgsl:103-176).
// Part of the kernel set of csrc/kernels.hip.h (included from there, in order; not a stand-alone header).
#ifndef MI355RT_K_TEXTURE_POST_HIP_H
#define MI355RT_K_TEXTURE_POST_HIP_H

namespace rtk {

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
  return rt_div3_plain(rt3_make(a.x, a.y, a.z), a.w);   // plain operators in this pass: zero numerators (black pixels) are the rule
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
  return rt_clamp3(rt3_make(num.x / den.x, num.y / den.y, num.z / den.z), rt3_splat(0.0f), rt3_splat(1.0f));
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
  rt3 denoised = rt_div3_plain(filtered_sum, rt_max(total_weight, 1e-4f));

  const size_t p_idx = (size_t)y * U.width + x;
  ushort4 hp = P.history_in[p_idx];
  rt3 hist = rt3_make(rt_f16_to_f32(hp.x), rt_f16_to_f32(hp.y), rt_f16_to_f32(hp.z));
  rt3 mean = rt_div3_plain(m1, 9.0f);
  rt3 var = rt_max3(rt_div3_plain(m2, 9.0f) - mean * mean, rt3_splat(0.0f));
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
