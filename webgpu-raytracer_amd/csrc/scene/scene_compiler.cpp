// scene_compiler.cpp — native scene compiler behind include/mi355scene.h.
//
// Produces the ten flat "bridge" arrays the renderer consumes,
// following the behaviour of the reference's Rust crate:
//   geometry assembly   rust-shader-tools/src/geometry.rs, scene/helpers.rs, mesh.rs
//   procedural scenes   rust-shader-tools/src/scene/procedural.rs
//   BLAS (binned SAH)   rust-shader-tools/src/bvh/blas.rs
//   TLAS (median split) rust-shader-tools/src/bvh/tlas.rs
//   packing             rust-shader-tools/src/rebuilder.rs, lib.rs:149-271
//   camera              rust-shader-tools/src/scene/camera.rs
// glam 0.30.9 (un-vendored dependency, Cargo.toml:10-18) supplies Mat4::inverse,
// from_rotation_y, transform_point3 and Vec3::normalize; their published scalar
// algorithms are restated in the `linalg` section.
//
// Compile with -ffp-contract=off: every expression below is plain IEEE f32.

#include "../../../include/mi355scene.h"
#include "../../../include/mi355rt_layout.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <type_traits>
#include <utility>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

// ---------------------------------------------------------------- linalg
struct V2 {
  float x = 0, y = 0;
};
struct V3 {
  float x = 0, y = 0, z = 0;
  float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
inline V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return v3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
// f32::min / f32::max: a NaN operand loses; for zeros of opposite sign Rust leaves the result unspecified — here
// -0 < +0 (the total order of the bit patterns), so that a chain of unions does not depend on the visiting order and
// the GPU builder (csrc/bvh_build.hip.h), which reduces in parallel, lands on the same bits
inline uint32_t fkey(float f) {
  uint32_t b;
  std::memcpy(&b, &f, 4);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
inline float fmin_nn(float a, float b) { return (b != b) ? a : ((a != a) ? b : (fkey(b) < fkey(a) ? b : a)); }
inline float fmax_nn(float a, float b) { return (b != b) ? a : ((a != a) ? b : (fkey(b) > fkey(a) ? b : a)); }
inline V3 vmin(V3 a, V3 b) { return v3(fmin_nn(a.x, b.x), fmin_nn(a.y, b.y), fmin_nn(a.z, b.z)); }
inline V3 vmax(V3 a, V3 b) { return v3(fmax_nn(a.x, b.x), fmax_nn(a.y, b.y), fmax_nn(a.z, b.z)); }
inline float length(V3 a) { return std::sqrt(dot(a, a)); }
// glam Vec3::normalize = self * (1 / length)
inline V3 normalize(V3 a) { return a * (1.0f / length(a)); }
inline V3 normalize_or_zero(V3 a) {
  float rcp = 1.0f / length(a);
  if (std::isfinite(rcp) && rcp > 0.0f) return a * rcp;
  return v3(0, 0, 0);
}
inline bool is_nan(V3 a) { return a.x != a.x || a.y != a.y || a.z != a.z; }
inline float to_radians(float deg) { return deg * (3.14159274101257324219f / 180.0f); }

struct M4 {  // column-major, c[col][row]
  float c[4][4];
};
inline M4 m4_identity() {
  M4 m{};
  for (int i = 0; i < 4; i++) m.c[i][i] = 1.0f;
  return m;
}
inline M4 m4_from_scale(V3 s) {
  M4 m{};
  m.c[0][0] = s.x;
  m.c[1][1] = s.y;
  m.c[2][2] = s.z;
  m.c[3][3] = 1.0f;
  return m;
}
inline M4 m4_from_translation(V3 t) {
  M4 m = m4_identity();
  m.c[3][0] = t.x;
  m.c[3][1] = t.y;
  m.c[3][2] = t.z;
  return m;
}
inline M4 m4_from_rotation_y(float angle) {
  float s = std::sin(angle), c = std::cos(angle);
  M4 m{};
  m.c[0][0] = c;
  m.c[0][2] = -s;
  m.c[1][1] = 1.0f;
  m.c[2][0] = s;
  m.c[2][2] = c;
  m.c[3][3] = 1.0f;
  return m;
}
inline M4 m4_mul(const M4& a, const M4& b) {
  M4 r{};
  for (int j = 0; j < 4; j++)
    for (int i = 0; i < 4; i++) {
      float acc = a.c[0][i] * b.c[j][0];
      acc = acc + a.c[1][i] * b.c[j][1];
      acc = acc + a.c[2][i] * b.c[j][2];
      acc = acc + a.c[3][i] * b.c[j][3];
      r.c[j][i] = acc;
    }
  return r;
}
inline V3 m4_transform_point3(const M4& m, V3 p) {
  float r[3];
  for (int i = 0; i < 3; i++) {
    float acc = m.c[0][i] * p.x;
    acc = m.c[1][i] * p.y + acc;
    acc = m.c[2][i] * p.z + acc;
    acc = m.c[3][i] + acc;
    r[i] = acc;
  }
  return v3(r[0], r[1], r[2]);
}
inline V3 m4_transform_vector3(const M4& m, V3 p) {  // glam Mat4::transform_vector3: w = 0
  float r[3];
  for (int i = 0; i < 3; i++) {
    float acc = m.c[0][i] * p.x;
    acc = m.c[1][i] * p.y + acc;
    acc = m.c[2][i] * p.z + acc;
    r[i] = acc;
  }
  return v3(r[0], r[1], r[2]);
}
// General 4x4 inverse by 2x2 sub-determinant factors (the GLM/glam scalar path).
M4 m4_inverse(const M4& m) {
  const float m00 = m.c[0][0], m01 = m.c[0][1], m02 = m.c[0][2], m03 = m.c[0][3];
  const float m10 = m.c[1][0], m11 = m.c[1][1], m12 = m.c[1][2], m13 = m.c[1][3];
  const float m20 = m.c[2][0], m21 = m.c[2][1], m22 = m.c[2][2], m23 = m.c[2][3];
  const float m30 = m.c[3][0], m31 = m.c[3][1], m32 = m.c[3][2], m33 = m.c[3][3];
  const float f[6][4] = {
      {m22 * m33 - m32 * m23, m22 * m33 - m32 * m23, m12 * m33 - m32 * m13, m12 * m23 - m22 * m13},
      {m21 * m33 - m31 * m23, m21 * m33 - m31 * m23, m11 * m33 - m31 * m13, m11 * m23 - m21 * m13},
      {m21 * m32 - m31 * m22, m21 * m32 - m31 * m22, m11 * m32 - m31 * m12, m11 * m22 - m21 * m12},
      {m20 * m33 - m30 * m23, m20 * m33 - m30 * m23, m10 * m33 - m30 * m13, m10 * m23 - m20 * m13},
      {m20 * m32 - m30 * m22, m20 * m32 - m30 * m22, m10 * m32 - m30 * m12, m10 * m22 - m20 * m12},
      {m20 * m31 - m30 * m21, m20 * m31 - m30 * m21, m10 * m31 - m30 * m11, m10 * m21 - m20 * m11}};
  const float v[4][4] = {{m10, m00, m00, m00}, {m11, m01, m01, m01}, {m12, m02, m02, m02}, {m13, m03, m03, m03}};
  const float sa[4] = {1.0f, -1.0f, 1.0f, -1.0f};
  const float sb[4] = {-1.0f, 1.0f, -1.0f, 1.0f};
  M4 inv{};
  for (int k = 0; k < 4; k++) {
    inv.c[0][k] = ((v[1][k] * f[0][k] - v[2][k] * f[1][k]) + v[3][k] * f[2][k]) * sa[k];
    inv.c[1][k] = ((v[0][k] * f[0][k] - v[2][k] * f[3][k]) + v[3][k] * f[4][k]) * sb[k];
    inv.c[2][k] = ((v[0][k] * f[1][k] - v[1][k] * f[3][k]) + v[3][k] * f[5][k]) * sa[k];
    inv.c[3][k] = ((v[0][k] * f[2][k] - v[1][k] * f[4][k]) + v[2][k] * f[5][k]) * sb[k];
  }
  const float d0 = m00 * inv.c[0][0], d1 = m01 * inv.c[1][0], d2 = m02 * inv.c[2][0], d3 = m03 * inv.c[3][0];
  const float det = d0 + d1 + d2 + d3;
  const float rcp = 1.0f / det;
  for (int j = 0; j < 4; j++)
    for (int i = 0; i < 4; i++) inv.c[j][i] = inv.c[j][i] * rcp;
  return inv;
}

// ------------------------------------------------------------------ AABB
struct Aabb {  // primitives.rs:6-76
  V3 mn = v3(INFINITY, INFINITY, INFINITY);
  V3 mx = v3(-INFINITY, -INFINITY, -INFINITY);
};
inline Aabb aabb_union(const Aabb& a, const Aabb& b) { return Aabb{vmin(a.mn, b.mn), vmax(a.mx, b.mx)}; }
inline float aabb_area(const Aabb& a) {
  V3 d = a.mx - a.mn;
  if (d.x < 0.0f || d.y < 0.0f || d.z < 0.0f) return 0.0f;
  return 2.0f * (d.x * d.y + d.y * d.z + d.z * d.x);
}
inline V3 aabb_center(const Aabb& a) { return (a.mn + a.mx) * 0.5f; }
Aabb aabb_transform(const Aabb& a, const M4& m) {
  Aabb out;
  for (int k = 0; k < 8; k++) {
    V3 p = v3((k & 1) ? a.mx.x : a.mn.x, (k & 2) ? a.mx.y : a.mn.y, (k & 4) ? a.mx.z : a.mn.z);
    V3 tp = m4_transform_point3(m, p);
    out.mn = vmin(out.mn, tp);
    out.mx = vmax(out.mx, tp);
  }
  return out;
}

// -------------------------------------------------------------- materials
enum : uint32_t { LAMBERTIAN = 0, METAL = 1, DIELECTRIC = 2, LIGHT = 3 };  // scene/material.rs

// ------------------------------------------------------------------ mesh
struct ObjMesh {  // mesh.rs:4-9
  std::vector<V3> vertices, normals;
  std::vector<V2> uvs;
  std::vector<uint32_t> indices;
};

std::vector<std::string> split_ws(const std::string& s) {
  std::vector<std::string> out;
  size_t i = 0;
  while (i < s.size()) {
    while (i < s.size() && isspace((unsigned char)s[i])) i++;
    size_t j = i;
    while (j < s.size() && !isspace((unsigned char)s[j])) j++;
    if (j > i) out.push_back(s.substr(i, j - i));
    i = j;
  }
  return out;
}
float parse_f32_or_zero(const std::string& s) {
  char* end = nullptr;
  float v = std::strtof(s.c_str(), &end);
  if (end == s.c_str() || *end != '\0') return 0.0f;
  return v;
}
// parse::<usize>().unwrap_or(0).saturating_sub(1)
size_t parse_index(const std::string& s) {
  size_t b = (!s.empty() && s[0] == '+') ? 1 : 0;
  if (b >= s.size()) return 0;
  for (size_t i = b; i < s.size(); i++)
    if (s[i] < '0' || s[i] > '9') return 0;
  unsigned long long v = std::strtoull(s.c_str() + b, nullptr, 10);
  return v == 0 ? 0 : (size_t)(v - 1);
}

// Wavefront OBJ subset: v / vt / vn / f with p, p/t, p//n, p/t/n; polygons are fan-
// triangulated; (p,t,n) triples are de-duplicated in first-seen order (mesh.rs:12-124).
ObjMesh parse_obj(const std::string& source) {
  ObjMesh mesh;
  std::vector<V3> raw_p, raw_n;
  std::vector<V2> raw_t;
  struct Key {
    size_t p;
    long t, n;  // -1 = absent
  };
  std::vector<Key> unique;
  size_t pos = 0;
  while (pos <= source.size()) {
    size_t eol = source.find('\n', pos);
    if (eol == std::string::npos) eol = source.size();
    std::string line = source.substr(pos, eol - pos);
    pos = eol + 1;
    std::vector<std::string> parts = split_ws(line);
    if (parts.empty()) {
      if (eol == source.size()) break;
      continue;
    }
    const std::string& tag = parts[0];
    if (tag == "v" && parts.size() >= 4) {
      raw_p.push_back(v3(parse_f32_or_zero(parts[1]), parse_f32_or_zero(parts[2]), parse_f32_or_zero(parts[3])));
    } else if (tag == "vt" && parts.size() >= 3) {
      raw_t.push_back(V2{parse_f32_or_zero(parts[1]), parse_f32_or_zero(parts[2])});
    } else if (tag == "vn" && parts.size() >= 4) {
      raw_n.push_back(v3(parse_f32_or_zero(parts[1]), parse_f32_or_zero(parts[2]), parse_f32_or_zero(parts[3])));
    } else if (tag == "f") {
      std::vector<uint32_t> face;
      for (size_t k = 1; k < parts.size(); k++) {
        std::vector<std::string> segs;
        {
          size_t a = 0;
          const std::string& s = parts[k];
          while (true) {
            size_t b = s.find('/', a);
            if (b == std::string::npos) {
              segs.push_back(s.substr(a));
              break;
            }
            segs.push_back(s.substr(a, b - a));
            a = b + 1;
          }
        }
        Key key;
        key.p = parse_index(segs[0]);
        key.t = (segs.size() > 1 && !segs[1].empty()) ? (long)parse_index(segs[1]) : -1;
        key.n = (segs.size() > 2 && !segs[2].empty()) ? (long)parse_index(segs[2]) : -1;
        long found = -1;
        for (size_t u = 0; u < unique.size(); u++)
          if (unique[u].p == key.p && unique[u].t == key.t && unique[u].n == key.n) {
            found = (long)u;
            break;
          }
        if (found >= 0) {
          face.push_back((uint32_t)found);
          continue;
        }
        uint32_t idx = (uint32_t)unique.size();
        unique.push_back(key);
        mesh.vertices.push_back(key.p < raw_p.size() ? raw_p[key.p] : v3(0, 0, 0));
        mesh.uvs.push_back((key.t >= 0 && (size_t)key.t < raw_t.size()) ? raw_t[(size_t)key.t] : V2{0, 0});
        mesh.normals.push_back((key.n >= 0 && (size_t)key.n < raw_n.size()) ? raw_n[(size_t)key.n] : v3(0, 1, 0));
        face.push_back(idx);
      }
      for (size_t i = 1; i + 1 < face.size(); i++) {
        mesh.indices.push_back(face[0]);
        mesh.indices.push_back(face[i]);
        mesh.indices.push_back(face[i + 1]);
      }
    }
    if (eol == source.size()) break;
  }
  return mesh;
}

// -------------------------------------------------------------- geometry
struct Geometry {  // geometry.rs:6-25
  std::vector<V3> positions, normals;  // base_positions / base_normals (the skinning input)
  std::vector<V2> uvs;
  std::vector<uint32_t> indices;
  std::vector<float> attributes;  // 16 f32 per triangle
  std::vector<uint32_t> joints;   // 4 per vertex
  std::vector<float> weights;     // 4 per vertex
  long skin_index = -1;           // Option<usize>

  uint32_t push_vertex(V3 p, V3 n, V2 uv) {  // geometry.rs:33-45
    const uint32_t j[4] = {0, 0, 0, 0};
    const float w[4] = {0, 0, 0, 0};
    return push_vertex_skinned(p, n, uv, j, w);
  }
  uint32_t push_vertex_skinned(V3 p, V3 n, V2 uv, const uint32_t j[4], const float w[4]) {  // geometry.rs:47-66
    positions.push_back(p);
    normals.push_back(n);
    uvs.push_back(uv);
    joints.insert(joints.end(), j, j + 4);
    weights.insert(weights.end(), w, w + 4);
    return (uint32_t)positions.size() - 1;
  }
  // geometry.rs:68-103
  void push_attributes(V3 base, uint32_t mat, float metallic, float roughness, float ior, V3 emissive,
                       const float tex[4], float occlusion_tex) {
    const float a[16] = {base.x, base.y, base.z, (float)mat, metallic, roughness, ior, 0.0f,
                         tex[0], tex[1], tex[2], tex[3], emissive.x, emissive.y, emissive.z, occlusion_tex};
    attributes.insert(attributes.end(), a, a + 16);
  }
  void push_simple_attributes(V3 color, uint32_t mat, float extra, float texture_index) {
    float metallic = 0.0f, roughness = 1.0f, ior = 1.5f;  // LAMBERTIAN / LIGHT
    if (mat == METAL) {
      metallic = 1.0f;
      roughness = extra;
    } else if (mat == DIELECTRIC) {
      roughness = 0.0f;
      ior = extra;
    }
    const float tex[4] = {texture_index, -1.0f, -1.0f, -1.0f};
    push_attributes(color, mat, metallic, roughness, ior, v3(0, 0, 0), tex, -1.0f);
  }
};

// scene/helpers.rs:6-54
void add_quad(Geometry& g, V3 a, V3 b, V3 c, V3 d, V3 color, uint32_t mat, float extra, float tex) {
  V3 n = normalize(cross(b - a, d - a));
  uint32_t i0 = g.push_vertex(a, n, V2{0, 0});
  uint32_t i1 = g.push_vertex(b, n, V2{1, 0});
  uint32_t i2 = g.push_vertex(c, n, V2{1, 1});
  uint32_t i3 = g.push_vertex(d, n, V2{0, 1});
  const uint32_t t0[3] = {i0, i1, i2}, t1[3] = {i0, i2, i3};
  g.indices.insert(g.indices.end(), t0, t0 + 3);
  g.push_simple_attributes(color, mat, extra, tex);
  g.indices.insert(g.indices.end(), t1, t1 + 3);
  g.push_simple_attributes(color, mat, extra, tex);
}

// scene/helpers.rs:56-151 — faces in the order front, back, top, bottom, right, left
void add_box(Geometry& g, V3 size, V3 center, float rot_y_deg, V3 color, uint32_t mat, float extra, float tex) {
  float rad = to_radians(rot_y_deg);
  float cr = std::cos(rad), sr = std::sin(rad);
  auto tf = [&](V3 p) {
    float x = p.x * cr + p.z * sr;
    float z = -p.x * sr + p.z * cr;
    return v3(x, p.y, z) + center;
  };
  V3 dx = v3(size.x / 2.0f, 0, 0), dy = v3(0, size.y / 2.0f, 0), dz = v3(0, 0, size.z / 2.0f);
  V3 nx = -dx;
  add_quad(g, tf(nx - dy + dz), tf(dx - dy + dz), tf(dx + dy + dz), tf(nx + dy + dz), color, mat, extra, tex);
  add_quad(g, tf(dx - dy - dz), tf(nx - dy - dz), tf(nx + dy - dz), tf(dx + dy - dz), color, mat, extra, tex);
  add_quad(g, tf(nx + dy + dz), tf(dx + dy + dz), tf(dx + dy - dz), tf(nx + dy - dz), color, mat, extra, tex);
  add_quad(g, tf(nx - dy - dz), tf(dx - dy - dz), tf(dx - dy + dz), tf(nx - dy + dz), color, mat, extra, tex);
  add_quad(g, tf(dx - dy + dz), tf(dx - dy - dz), tf(dx + dy - dz), tf(dx + dy + dz), color, mat, extra, tex);
  add_quad(g, tf(nx - dy - dz), tf(nx - dy + dz), tf(nx + dy + dz), tf(nx + dy - dz), color, mat, extra, tex);
}

// geometry.rs:204-275 — UV sphere, 24 sectors x 12 stacks
void add_sphere(Geometry& g, V3 center, float radius, V3 color, uint32_t mat, float extra, float tex) {
  const uint32_t sectors = 24, stacks = 12;
  const float PI = 3.14159274101257324219f;
  uint32_t start = (uint32_t)g.positions.size();
  for (uint32_t i = 0; i <= stacks; i++) {
    float vc = (float)i / (float)stacks;
    float stack_angle = PI / 2.0f - PI * vc;
    float xy = radius * std::cos(stack_angle);
    float z = radius * std::sin(stack_angle);
    for (uint32_t j = 0; j <= sectors; j++) {
      float uc = (float)j / (float)sectors;
      float sector_angle = 2.0f * PI * uc;
      float x = xy * std::cos(sector_angle);
      float y = xy * std::sin(sector_angle);
      g.push_vertex(v3(x, y, z) + center, normalize(v3(x, y, z)), V2{uc, vc});
    }
  }
  for (uint32_t i = 0; i < stacks; i++) {
    uint32_t k1 = start + i * (sectors + 1);
    uint32_t k2 = k1 + sectors + 1;
    for (uint32_t j = 0; j < sectors; j++) {
      if (i != 0) {
        const uint32_t t[3] = {k1 + j, k2 + j, k1 + j + 1};
        g.indices.insert(g.indices.end(), t, t + 3);
        g.push_simple_attributes(color, mat, extra, tex);
      }
      if (i != stacks - 1) {
        const uint32_t t[3] = {k1 + j + 1, k2 + j, k2 + j + 1};
        g.indices.insert(g.indices.end(), t, t + 3);
        g.push_simple_attributes(color, mat, extra, tex);
      }
    }
  }
}

// geometry.rs:277-327
void add_mesh_instance(Geometry& g, const ObjMesh& mesh, V3 pos, float scale, float rot_y_deg, V3 color,
                       uint32_t mat, float extra, float tex) {
  if (mesh.vertices.empty()) return;
  float rad = to_radians(rot_y_deg);
  float s = std::sin(rad), c = std::cos(rad);
  // Mat3::from_rotation_y: columns (c,0,-s), (0,1,0), (s,0,c); M*v = col0*v.x + col1*v.y + col2*v.z
  auto rot = [&](V3 v) {
    V3 r = v3(c * v.x, 0.0f * v.x, -s * v.x);
    r = r + v3(0.0f * v.y, 1.0f * v.y, 0.0f * v.y);
    r = r + v3(s * v.z, 0.0f * v.z, c * v.z);
    return r;
  };
  uint32_t start = (uint32_t)g.positions.size();
  for (size_t i = 0; i < mesh.vertices.size(); i++) {
    V3 tv = rot(mesh.vertices[i] * scale) + pos;
    V3 tn = i < mesh.normals.size() ? rot(mesh.normals[i]) : v3(0, 1, 0);
    V2 uv = i < mesh.uvs.size() ? mesh.uvs[i] : V2{0, 0};
    g.push_vertex(tv, tn, uv);
  }
  for (size_t k = 0; k + 2 < mesh.indices.size(); k += 3) {
    g.indices.push_back(mesh.indices[k] + start);
    g.indices.push_back(mesh.indices[k + 1] + start);
    g.indices.push_back(mesh.indices[k + 2] + start);
    g.push_simple_attributes(color, mat, extra, tex);
  }
}

// geometry.rs:105-131 / 133-166 (used by `cornell` + OBJ, kept for completeness)
Geometry geometry_from_mesh(const ObjMesh& mesh) {
  Geometry g;
  for (size_t i = 0; i < mesh.vertices.size(); i++)
    g.push_vertex(mesh.vertices[i], i < mesh.normals.size() ? mesh.normals[i] : v3(0, 1, 0),
                  i < mesh.uvs.size() ? mesh.uvs[i] : V2{0, 0});
  const float notex[4] = {-1, -1, -1, -1};
  for (size_t k = 0; k + 2 < mesh.indices.size(); k += 3) {
    g.indices.insert(g.indices.end(), mesh.indices.begin() + k, mesh.indices.begin() + k + 3);
    g.push_attributes(v3(1, 1, 1), LAMBERTIAN, 0.0f, 1.0f, 1.5f, v3(0, 0, 0), notex, -1.0f);
  }
  return g;
}

// ----------------------------------------------------------------- scene
struct CameraConfig {  // scene/camera.rs:3-11
  V3 lookfrom, lookat, vup;
  float vfov = 60, defocus_angle = 0, focus_dist = 1;
};
struct SceneInstance {
  M4 transform;
  size_t geometry_index;
};
struct SceneData {
  CameraConfig camera;
  std::vector<Geometry> geometries;
  std::vector<SceneInstance> instances;
  std::vector<std::vector<uint8_t>> textures_rgba;  // decoded 1024x1024 RGBA8 layers (synthetic scenes only)
  std::vector<std::vector<uint8_t>> textures;       // encoded image bytes, one per glTF texture (SceneData.textures)
  bool keep_instance_transforms = false;            // synthetic scenes bypass lib.rs:196-204
};

#include "gltf_loader.h"

// scene/camera.rs:14-56
void camera_buffer(const CameraConfig& c, float aspect, float out[24]) {
  float theta = to_radians(c.vfov);
  float h = std::tan(theta / 2.0f);
  float vh = 2.0f * h * c.focus_dist;
  float vw = vh * aspect;
  V3 w = normalize(c.lookfrom - c.lookat);
  V3 u = normalize(cross(c.vup, w));
  V3 v = cross(w, u);
  V3 horizontal = u * vw;
  V3 vertical = v * vh;
  V3 ll = c.lookfrom - horizontal * 0.5f - vertical * 0.5f - w * c.focus_dist;
  float lens_radius = c.focus_dist * std::tan(to_radians(c.defocus_angle) / 2.0f);
  const float buf[24] = {c.lookfrom.x, c.lookfrom.y, c.lookfrom.z, lens_radius, ll.x, ll.y, ll.z, 0.0f,
                         horizontal.x, horizontal.y, horizontal.z, 0.0f, vertical.x, vertical.y, vertical.z, 0.0f,
                         u.x, u.y, u.z, 0.0f, v.x, v.y, v.z, 0.0f};
  std::memcpy(out, buf, sizeof(buf));
}

// Cornell coordinate maps, procedural.rs:23-25
inline V3 cb_v(float x, float y, float z) {
  const float s = 555.0f;
  return v3(x / s * 2.0f - 1.0f, y / s * 2.0f, z / s * 2.0f - 1.0f);
}
inline V3 cb_sz(float x, float y, float z) {
  const float s = 555.0f;
  return v3(x / s * 2.0f, y / s * 2.0f, z / s * 2.0f);
}

struct CornellStyle {
  uint32_t floor_mat;
  float floor_extra;
  V3 light_color;
  float lx0, lz0, lx1, lz1;
};
// The five walls + ceiling light shared by cornell / special / viewer
// (procedural.rs:27-96, 388-459, 641-720).
void add_cornell_shell(Geometry& g, const CornellStyle& st) {
  V3 white = v3(0.73f, 0.73f, 0.73f), red = v3(0.65f, 0.05f, 0.05f), green = v3(0.12f, 0.45f, 0.15f);
  add_quad(g, cb_v(0, 0, 0), cb_v(555, 0, 0), cb_v(555, 0, 555), cb_v(0, 0, 555), white, st.floor_mat,
           st.floor_extra, -1.0f);
  add_quad(g, cb_v(0, 555, 0), cb_v(0, 555, 555), cb_v(555, 555, 555), cb_v(555, 555, 0), white, LAMBERTIAN, 0.0f,
           -1.0f);
  add_quad(g, cb_v(0, 0, 555), cb_v(555, 0, 555), cb_v(555, 555, 555), cb_v(0, 555, 555), white, LAMBERTIAN, 0.0f,
           -1.0f);
  add_quad(g, cb_v(0, 0, 0), cb_v(0, 555, 0), cb_v(0, 555, 555), cb_v(0, 0, 555), green, LAMBERTIAN, 0.0f, -1.0f);
  add_quad(g, cb_v(555, 0, 0), cb_v(555, 0, 555), cb_v(555, 555, 555), cb_v(555, 555, 0), red, LAMBERTIAN, 0.0f,
           -1.0f);
  add_quad(g, cb_v(st.lx0, 554, st.lz0), cb_v(st.lx1, 554, st.lz0), cb_v(st.lx1, 554, st.lz1),
           cb_v(st.lx0, 554, st.lz1), st.light_color, LIGHT, 0.0f, -1.0f);
}

SceneData one_geometry_scene(Geometry&& g, const CameraConfig& cam) {
  SceneData sd;
  sd.camera = cam;
  sd.geometries.push_back(std::move(g));
  sd.instances.push_back(SceneInstance{m4_identity(), 0});
  return sd;
}

// procedural.rs:16-171 (the OBJ variant is unreachable from factory.rs:12 — `cornell` passes None)
SceneData scene_cornell() {
  Geometry g;
  V3 white = v3(0.73f, 0.73f, 0.73f);
  add_cornell_shell(g, CornellStyle{LAMBERTIAN, 0.0f, v3(20, 20, 20), 213, 227, 343, 332});
  add_box(g, cb_sz(165, 330, 165), cb_v(297.5f, 165, 378.5f), -15.0f, white, LAMBERTIAN, 0.0f, -1.0f);
  add_box(g, cb_sz(165, 165, 165), cb_v(232.5f, 82.5f, 147.5f), 18.0f, white, LAMBERTIAN, 0.0f, -1.0f);
  CameraConfig cam{v3(0, 1, -2.4f), v3(0, 1, 0), v3(0, 1, 0), 60.0f, 0.0f, 2.4f};
  return one_geometry_scene(std::move(g), cam);
}

// procedural.rs:266-376
SceneData scene_mixed() {
  Geometry g;
  const float PI = 3.14159274101257324219f;
  add_box(g, v3(40, 2, 40), v3(0, -1.0f, 0), 0.0f, v3(0.1f, 0.1f, 0.1f), METAL, 0.05f, -1.0f);
  V3 la = v3(-4, 8, 4);
  add_quad(g, la, la + v3(2, 0, 0), la + v3(2, 0, 2), la + v3(0, 0, 2), v3(40, 30, 10), LIGHT, 0.0f, -1.0f);
  V3 lb = v3(4, 6, -4);
  add_quad(g, lb, lb + v3(3, 0, 0), lb + v3(3, -3, 0), lb + v3(0, -3, 0), v3(5, 10, 20), LIGHT, 0.0f, -1.0f);
  add_box(g, v3(2, 1, 2), v3(0, 0.5f, 0), 0.0f, v3(0.8f, 0.6f, 0.2f), METAL, 0.1f, -1.0f);
  add_sphere(g, v3(0, 1.8f, 0), 0.8f, v3(1, 1, 1), DIELECTRIC, 1.5f, -1.0f);
  add_sphere(g, v3(0, 1.8f, 0), -0.7f, v3(1, 1, 1), DIELECTRIC, 1.0f, -1.0f);
  add_box(g, v3(0.8f, 0.8f, 0.8f), v3(0, 3.2f, 0), 15.0f, v3(0.9f, 0.1f, 0.1f), METAL, 0.2f, -1.0f);
  for (int i = 0; i < 12; i++) {
    float fi = (float)i;
    float angle = fi / 12.0f * PI * 2.0f;
    V3 pos = v3(std::cos(angle) * 4.0f, 1.0f + std::sin(angle * 3.0f) * 0.5f, std::sin(angle) * 4.0f);
    if (i % 2 == 0) {
      add_sphere(g, pos, 0.4f, v3(0.8f, 0.8f, 0.8f), METAL, 0.0f, -1.0f);
    } else {
      V3 col = v3(0.5f + 0.5f * std::cos(fi), 0.5f + 0.5f * std::sin(fi), 0.8f);
      add_box(g, v3(0.6f, 0.6f, 0.6f), pos, fi * 20.0f, col, LAMBERTIAN, 0.0f, -1.0f);
    }
  }
  add_box(g, v3(1, 6, 1), v3(-4, 3, -6), 10.0f, v3(0.2f, 0.2f, 0.3f), LAMBERTIAN, 0.0f, -1.0f);
  add_box(g, v3(1, 4, 1), v3(4, 2, -5), -20.0f, v3(0.2f, 0.2f, 0.3f), LAMBERTIAN, 0.0f, -1.0f);
  CameraConfig cam{v3(0, 3.5f, 9), v3(0, 1.5f, 0), v3(0, 1, 0), 40.0f, 0.3f, 9.0f};
  return one_geometry_scene(std::move(g), cam);
}

// procedural.rs:378-508
SceneData scene_special() {
  Geometry g;
  V3 white = v3(0.73f, 0.73f, 0.73f);
  add_cornell_shell(g, CornellStyle{METAL, 0.1f, v3(10, 10, 10), 213, 227, 343, 332});
  V3 tall = cb_v(366, 165, 383);
  add_box(g, cb_sz(165, 330, 165), tall, 15.0f, v3(0.95f, 0.95f, 0.95f), DIELECTRIC, 1.5f, -1.0f);
  add_box(g, cb_sz(165, 165, 165), cb_v(183, 82.5f, 209), -18.0f, white, METAL, 0.2f, -1.0f);
  add_sphere(g, tall, (60.0f / 555.0f) * 1.0f, v3(0.1f, 0.1f, 10.0f), LIGHT, 0.0f, -1.0f);
  CameraConfig cam{v3(0, 1, -3.9f), v3(0, 1, 0), v3(0, 1, 0), 40.0f, 0.0f, 2.4f};
  return one_geometry_scene(std::move(g), cam);
}

// procedural.rs:510-595
SceneData scene_mesh() {
  static const char* kCube =
      "v -1 -1 1\nv 1 -1 1\nv -1 1 1\nv 1 1 1\nv -1 -1 -1\nv 1 -1 -1\nv -1 1 -1\nv 1 1 -1\n"
      "f 1 2 4 3\nf 3 4 8 7\nf 7 8 6 5\nf 5 6 2 1\nf 3 7 5 1\nf 8 4 2 6";
  Geometry g;
  ObjMesh cube = parse_obj(kCube);
  add_sphere(g, v3(0, -1000, 0), 1000.0f, v3(0.5f, 0.5f, 0.5f), LAMBERTIAN, 0.0f, -1.0f);
  add_mesh_instance(g, cube, v3(-2, 1, 0), 1.0f, 45.0f, v3(0.8f, 0.2f, 0.2f), METAL, 0.2f, -1.0f);
  add_mesh_instance(g, cube, v3(0, 1, 1.5f), 1.2f, 0.0f, v3(1, 1, 1), DIELECTRIC, 1.5f, -1.0f);
  for (int i = 0; i < 5; i++) {
    float fi = (float)i;
    add_mesh_instance(g, cube, v3(2.0f + fi * 0.5f, 0.5f + fi * 0.5f, -fi), 0.5f, fi * 30.0f, v3(0.2f, 0.4f, 0.8f),
                      LAMBERTIAN, 0.0f, -1.0f);
  }
  add_sphere(g, v3(0, 10, 0), 3.0f, v3(10, 10, 10), LIGHT, 0.0f, -1.0f);
  CameraConfig cam{v3(0, 3, 6), v3(0, 1, 0), v3(0, 1, 0), 40.0f, 0.0f, 6.0f};
  return one_geometry_scene(std::move(g), cam);
}

// procedural.rs:634-791
SceneData scene_viewer(const ObjMesh* mesh, bool has_glb) {
  Geometry env, model;
  add_cornell_shell(env, CornellStyle{METAL, 0.15f, v3(10, 10, 10), 150, 150, 405, 405});
  if (mesh) {
    add_mesh_instance(model, *mesh, v3(0, 1, 0), 1.0f, 0.0f, v3(0.8f, 0.8f, 0.8f), LAMBERTIAN, 0.0f, -1.0f);
  } else if (!has_glb) {
    add_sphere(model, v3(0, 1, 0), 0.5f, v3(1, 0, 1), LAMBERTIAN, 0.0f, -1.0f);  // placeholder (no OBJ, no GLB)
  }
  SceneData sd;
  sd.camera = CameraConfig{v3(0, 1, -3.9f), v3(0, 1, 0), v3(0, 1, 0), 40.0f, 0.0f, 3.9f};
  bool has_model = !model.positions.empty();
  sd.geometries.push_back(std::move(env));
  sd.geometries.push_back(std::move(model));
  sd.instances.push_back(SceneInstance{m4_identity(), 0});
  if (has_model) sd.instances.push_back(SceneInstance{m4_identity(), 1});
  return sd;
}

// ---- extensions: BASELINE.json configs 3-5 (not expressible by World::update) ----
ObjMesh octahedron_mesh() {  // same shape as public/diamond.obj (6 vertices, 8 faces, no vn/vt)
  return parse_obj(
      "v 0 1 0\nv 1 0 0\nv 0 0 1\nv -1 0 0\nv 0 0 -1\nv 0 -1 0\n"
      "f 1 3 2\nf 1 2 5\nf 1 5 4\nf 1 4 3\nf 6 2 3\nf 6 5 2\nf 6 4 5\nf 6 3 4\n");
}

// Config 3: Cornell shell (instance 0) + 10x10x10 lattice of octahedra sharing
// three BLASes (Lambert / metal 0.2 / dielectric 1.5), scale 0.06, rotY(0.37 i).
// The same with nx x ny x nz octahedra (minus `skip` at the end): instanced16384 = the Cornell shell + 16 383 of a
// 32 x 32 x 16 lattice = the 16 384 instances the device TLAS takes (a scale test of World::update's TLAS, not a BASELINE config).
SceneData scene_instanced_lattice(int nx, int ny, int nz, int skip, float scale);
SceneData scene_instanced1000() { return scene_instanced_lattice(10, 10, 10, 0, 0.06f); }
SceneData scene_instanced_lattice(int nx, int ny, int nz, int skip, float scale) {
  SceneData sd;
  Geometry env;
  add_cornell_shell(env, CornellStyle{LAMBERTIAN, 0.0f, v3(20, 20, 20), 213, 227, 343, 332});
  sd.geometries.push_back(std::move(env));
  ObjMesh oct = octahedron_mesh();
  const V3 colors[3] = {v3(0.8f, 0.3f, 0.3f), v3(0.9f, 0.9f, 0.9f), v3(1.0f, 1.0f, 1.0f)};
  const uint32_t mats[3] = {LAMBERTIAN, METAL, DIELECTRIC};
  const float extras[3] = {0.0f, 0.2f, 1.5f};
  for (int k = 0; k < 3; k++) {
    Geometry g;
    add_mesh_instance(g, oct, v3(0, 0, 0), 1.0f, 0.0f, colors[k], mats[k], extras[k], -1.0f);
    sd.geometries.push_back(std::move(g));
  }
  sd.instances.push_back(SceneInstance{m4_identity(), 0});
  // (10 x 10 x 10: the literal spacing 0.18 and origin of round 1, so that config 3's arrays do not change by a bit)
  const float dx = nx == 10 ? 0.18f : 1.8f / (float)nx, dy = ny == 10 ? 0.18f : 1.8f / (float)ny, dz = nz == 10 ? 0.18f : 1.8f / (float)nz;
  const float x0 = nx == 10 ? -0.81f : -0.9f + 0.5f * dx, y0 = ny == 10 ? 0.19f : 0.1f + 0.5f * dy, z0 = nz == 10 ? -0.81f : -0.9f + 0.5f * dz;
  const int total = nx * ny * nz - skip;
  int i = 0;
  for (int iz = 0; iz < nz; iz++)
    for (int iy = 0; iy < ny; iy++)
      for (int ix = 0; ix < nx && i < total; ix++, i++) {
        V3 p = v3(x0 + dx * (float)ix, y0 + dy * (float)iy, z0 + dz * (float)iz);
        M4 t = m4_mul(m4_from_translation(p),
                      m4_mul(m4_from_rotation_y(0.37f * (float)i), m4_from_scale(v3(scale, scale, scale))));
        sd.instances.push_back(SceneInstance{t, (size_t)(1 + i % 3)});
      }
  sd.camera = CameraConfig{v3(0, 1, -2.4f), v3(0, 1, 0), v3(0, 1, 0), 60.0f, 0.0f, 2.4f};
  sd.keep_instance_transforms = true;
  return sd;
}

// Deterministic LCG for procedural texture / geometry noise (scene generation only).
struct Lcg {
  uint32_t s;
  explicit Lcg(uint32_t seed) : s(seed) {}
  uint32_t next() {
    s = s * 1664525u + 1013904223u;
    return s;
  }
  float unit() { return (float)(next() >> 8) * (1.0f / 16777216.0f); }
};

void make_texture(std::vector<uint8_t>& out, int kind, uint32_t seed) {
  const int N = RT_TEX_SIZE;
  out.resize((size_t)N * N * 4);
  Lcg rng(seed);
  std::vector<uint8_t> noise((size_t)64 * 64);
  for (auto& v : noise) v = (uint8_t)(rng.next() >> 24);
  for (int y = 0; y < N; y++)
    for (int x = 0; x < N; x++) {
      uint8_t* px = &out[((size_t)y * N + x) * 4];
      int n = noise[(size_t)((y >> 4) & 63) * 64 + ((x >> 4) & 63)];
      int r, g, b;
      switch (kind & 3) {
        case 0: {  // checker
          int c = (((x >> 6) + (y >> 6)) & 1) ? 220 : 90;
          r = c; g = c - 10; b = c - 30;
        } break;
        case 1: {  // brick
          int row = y >> 5;
          int xx = x + ((row & 1) ? 32 : 0);
          bool mortar = ((y & 31) < 3) || ((xx & 63) < 3);
          r = mortar ? 200 : 150 + (n >> 3); g = mortar ? 200 : 70 + (n >> 4); b = mortar ? 190 : 50;
        } break;
        case 2: {  // blocky noise
          r = 100 + (n >> 1); g = 110 + (n >> 2); b = 120 + (n >> 3);
        } break;
        default: {  // metal-roughness map: g = roughness scale, b = metallic scale
          r = 255; g = 40 + (n >> 1); b = (((x >> 7) + (y >> 7)) & 1) ? 255 : 60;
        } break;
      }
      px[0] = (uint8_t)std::min(255, std::max(0, r));
      px[1] = (uint8_t)std::min(255, std::max(0, g));
      px[2] = (uint8_t)std::min(255, std::max(0, b));
      px[3] = 255;
    }
}

// Tessellated helpers for the large synthetic scenes: smooth normals, real UVs.
// jitter > 0: every interior grid vertex is moved ON the surface by up to +-jitter cells in u and v (LCG, seeded per
// patch).  A regular grid is the worst case for the reference's builder: blas.rs:106 picks axis y whenever extent.y >
// extent.x, a one-cell-high strip of a regular grid has all its centroids at ONE height, nothing can be split and the
// strip becomes a fallback leaf whose 3-bit count overflows (blas.rs:111-115) — a third of the regular sponza-like mesh
// was unreachable for every ray.  Real meshes are not regular grids; with jittered vertices the centroids differ and
// the builder (unchanged) splits them.
void add_grid_patch(Geometry& g, int nu, int nv, V3 (*fn)(float, float, const float*), const float* prm, V3 color,
                    uint32_t mat, float metallic, float roughness, float ior, const float tex[4], float uv_scale,
                    float jitter = 0.0f, uint32_t jitter_seed = 0u) {
  uint32_t start = (uint32_t)g.positions.size();
  const float eps = 1e-3f;
  Lcg jr(jitter_seed * 2654435761u + 12345u);
  for (int j = 0; j <= nv; j++)
    for (int i = 0; i <= nu; i++) {
      float u = (float)i / (float)nu, v = (float)j / (float)nv;
      if (jitter > 0.0f) {
        const float du = (jr.unit() * 2.0f - 1.0f) * jitter / (float)nu, dv = (jr.unit() * 2.0f - 1.0f) * jitter / (float)nv;
        if (i > 0 && i < nu) u += du;   // the border (and the seam of a closed patch) stays where the neighbours expect it
        if (j > 0 && j < nv) v += dv;
      }
      V3 p = fn(u, v, prm);
      V3 du = fn(u + eps, v, prm) - fn(u - eps, v, prm);
      V3 dv = fn(u, v + eps, prm) - fn(u, v - eps, prm);
      V3 n = normalize_or_zero(cross(du, dv));
      if (n.x == 0.0f && n.y == 0.0f && n.z == 0.0f) n = v3(0, 1, 0);
      g.push_vertex(p, n, V2{u * uv_scale, v * uv_scale});
    }
  for (int j = 0; j < nv; j++)
    for (int i = 0; i < nu; i++) {
      uint32_t a = start + (uint32_t)(j * (nu + 1) + i), b = a + 1, c = a + (uint32_t)(nu + 1), d = c + 1;
      const uint32_t t0[3] = {a, b, d}, t1[3] = {a, d, c};
      g.indices.insert(g.indices.end(), t0, t0 + 3);
      g.push_attributes(color, mat, metallic, roughness, ior, v3(0, 0, 0), tex, -1.0f);
      g.indices.insert(g.indices.end(), t1, t1 + 3);
      g.push_attributes(color, mat, metallic, roughness, ior, v3(0, 0, 0), tex, -1.0f);
    }
}
V3 fn_cylinder(float u, float v, const float* p) {  // p: cx, cz, radius, y0, y1
  float a = 6.28318548202514648438f * u;
  return v3(p[0] + p[2] * std::cos(a), p[3] + (p[4] - p[3]) * v, p[1] + p[2] * std::sin(a));
}
V3 fn_arch(float u, float v, const float* p) {  // half-torus arch: cx0, cx1, z, y, tube radius
  float a = 3.14159274101257324219f * u, b = 6.28318548202514648438f * v;
  float R = (p[1] - p[0]) * 0.5f, cx = (p[0] + p[1]) * 0.5f;
  float rr = R + p[4] * std::cos(b);
  return v3(cx - rr * std::cos(a), p[3] + rr * std::sin(a), p[2] + p[4] * std::sin(b));
}
V3 fn_plane_xz(float u, float v, const float* p) {  // x0,x1,z0,z1,y, flip
  float x = p[0] + (p[1] - p[0]) * (p[5] > 0 ? v : u), z = p[2] + (p[3] - p[2]) * (p[5] > 0 ? u : v);
  return v3(x, p[4], z);
}
V3 fn_plane_xy(float u, float v, const float* p) {  // x0,x1,y0,y1,z, flip
  float x = p[0] + (p[1] - p[0]) * (p[5] > 0 ? v : u), y = p[2] + (p[3] - p[2]) * (p[5] > 0 ? u : v);
  return v3(x, y, p[4]);
}
V3 fn_plane_zy(float u, float v, const float* p) {  // z0,z1,y0,y1,x, flip
  float z = p[0] + (p[1] - p[0]) * (p[5] > 0 ? v : u), y = p[2] + (p[3] - p[2]) * (p[5] > 0 ? u : v);
  return v3(p[4], y, z);
}
V3 fn_torus_knot(float u, float v, const float* p) {  // (2,3) torus knot tube: scale, tube r, cy
  const float TWO_PI = 6.28318548202514648438f;
  auto center = [&](float t) {
    float r = 0.5f * (2.0f + std::cos(3.0f * t));
    return v3(r * std::cos(2.0f * t), r * std::sin(3.0f * t) * 0.9f, r * std::sin(2.0f * t)) * p[0];
  };
  float t = TWO_PI * u, a = TWO_PI * v;
  V3 c = center(t);
  V3 tan = normalize(center(t + 0.01f) - center(t - 0.01f));
  V3 up = v3(0, 1, 0);
  V3 nrm = normalize(cross(tan, up));
  V3 bin = cross(tan, nrm);
  V3 q = c + nrm * (p[1] * std::cos(a)) + bin * (p[1] * std::sin(a));
  return v3(q.x, q.y + p[2], q.z);
}

// Config 4: "Sponza-like" hall, ~262k triangles, 8 procedural textures,
// Lambert + textured metal, 4 ceiling light quads.
SceneData scene_sponza_like() {
  SceneData sd;
  Geometry g;
  const float SPONZA_JITTER = 0.40f;   // cells; see add_grid_patch
  sd.textures_rgba.resize(8);
  for (int i = 0; i < 8; i++) make_texture(sd.textures_rgba[i], i, 1u + (uint32_t)i);
  const float t_floor[4] = {0, -1, -1, -1}, t_wall[4] = {1, -1, -1, -1}, t_col[4] = {2, 3, -1, -1},
              t_arch[4] = {5, -1, -1, -1}, t_ceil[4] = {6, -1, -1, -1}, t_metal[4] = {4, 7, -1, -1};
  // hall: x in [-6,6], y in [0,5], z in [-2.5,2.5]
  {
    const float fl[6] = {-6, 6, -2.5f, 2.5f, 0, 1};
    add_grid_patch(g, 128, 128, fn_plane_xz, fl, v3(0.8f, 0.8f, 0.8f), METAL, 0.6f, 0.35f, 1.5f, t_metal, 6.0f, SPONZA_JITTER, 1u);
    const float ce[6] = {-6, 6, -2.5f, 2.5f, 5, 0};
    add_grid_patch(g, 96, 96, fn_plane_xz, ce, v3(0.7f, 0.7f, 0.7f), LAMBERTIAN, 0, 1, 1.5f, t_ceil, 4.0f, SPONZA_JITTER, 2u);
    const float bk[6] = {-6, 6, 0, 5, 2.5f, 1};
    add_grid_patch(g, 128, 64, fn_plane_xy, bk, v3(0.75f, 0.7f, 0.65f), LAMBERTIAN, 0, 1, 1.5f, t_wall, 5.0f, SPONZA_JITTER, 3u);
    const float fr[6] = {-6, 6, 0, 5, -2.5f, 0};
    add_grid_patch(g, 128, 64, fn_plane_xy, fr, v3(0.75f, 0.7f, 0.65f), LAMBERTIAN, 0, 1, 1.5f, t_wall, 5.0f, SPONZA_JITTER, 4u);
    const float lf[6] = {-2.5f, 2.5f, 0, 5, -6, 0};
    add_grid_patch(g, 64, 64, fn_plane_zy, lf, v3(0.65f, 0.3f, 0.25f), LAMBERTIAN, 0, 1, 1.5f, t_floor, 3.0f, SPONZA_JITTER, 5u);
    const float rt[6] = {-2.5f, 2.5f, 0, 5, 6, 1};
    add_grid_patch(g, 64, 64, fn_plane_zy, rt, v3(0.3f, 0.45f, 0.65f), LAMBERTIAN, 0, 1, 1.5f, t_floor, 3.0f, SPONZA_JITTER, 6u);
  }
  // two colonnades of 8 columns + arches between neighbours
  for (int side = 0; side < 2; side++) {
    float z = side == 0 ? -1.4f : 1.4f;
    for (int k = 0; k < 8; k++) {
      float x = -5.25f + 1.5f * (float)k;
      const float cy[5] = {x, z, 0.18f, 0.0f, 3.0f};
      add_grid_patch(g, 32, 96, fn_cylinder, cy, v3(0.85f, 0.8f, 0.7f), METAL, 1.0f, 0.5f, 1.5f, t_col, 2.0f, SPONZA_JITTER, 16u + 2u * (uint32_t)(side * 8 + k));
      if (k < 7) {
        const float ar[5] = {x, x + 1.5f, z, 3.0f, 0.14f};
        add_grid_patch(g, 64, 36, fn_arch, ar, v3(0.8f, 0.75f, 0.7f), LAMBERTIAN, 0, 1, 1.5f, t_arch, 2.0f, SPONZA_JITTER, 17u + 2u * (uint32_t)(side * 8 + k));
      }
    }
  }
  // four ceiling lights
  for (int k = 0; k < 4; k++) {
    float x = -4.5f + 3.0f * (float)k;
    add_quad(g, v3(x - 0.5f, 4.98f, -0.5f), v3(x + 0.5f, 4.98f, -0.5f), v3(x + 0.5f, 4.98f, 0.5f),
             v3(x - 0.5f, 4.98f, 0.5f), v3(18, 17, 15), LIGHT, 0.0f, -1.0f);
  }
  sd.geometries.push_back(std::move(g));
  sd.instances.push_back(SceneInstance{m4_identity(), 0});
  sd.camera = CameraConfig{v3(-5.2f, 1.7f, 0.0f), v3(0, 1.9f, 0.0f), v3(0, 1, 0), 65.0f, 0.0f, 5.0f};
  sd.keep_instance_transforms = true;
  return sd;
}

// Config 5: Cornell box + ~200k-triangle dielectric (ior 1.5) torus knot.
SceneData scene_glass_blob() {
  SceneData sd;
  Geometry env;
  add_cornell_shell(env, CornellStyle{LAMBERTIAN, 0.0f, v3(20, 20, 20), 213, 227, 343, 332});
  Geometry blob;
  const float prm[3] = {0.42f, 0.12f, 0.95f};
  const float notex[4] = {-1, -1, -1, -1};
  add_grid_patch(blob, 1600, 64, fn_torus_knot, prm, v3(0.98f, 0.98f, 0.98f), DIELECTRIC, 0.0f, 0.0f, 1.5f, notex,
                 1.0f);
  sd.geometries.push_back(std::move(env));
  sd.geometries.push_back(std::move(blob));
  sd.instances.push_back(SceneInstance{m4_identity(), 0});
  sd.instances.push_back(SceneInstance{m4_identity(), 1});
  sd.camera = CameraConfig{v3(0, 1, -2.4f), v3(0, 1, 0), v3(0, 1, 0), 60.0f, 0.0f, 2.4f};
  sd.keep_instance_transforms = true;
  return sd;
}

// ------------------------------------------------------------------ BLAS
// Binned-SAH builder: 16 bins, leaf <= 4, DFS pre-order + skip pointers (bvh/blas.rs).
struct BuildNode {
  V3 mn, mx;
  uint32_t skip = 0, data = 0;
};
class BlasBuilder {
 public:
  BlasBuilder(const float* verts4, size_t /*n_verts*/, const std::vector<uint32_t>& indices) : indices_(indices) {
    size_t n = indices.size() / 3;
    boxes_.reserve(n);
    centers_.reserve(n);
    for (size_t i = 0; i < n; i++) {
      auto get = [&](uint32_t id) { return v3(verts4[id * 4], verts4[id * 4 + 1], verts4[id * 4 + 2]); };
      V3 a = get(indices[i * 3]), b = get(indices[i * 3 + 1]), c = get(indices[i * 3 + 2]);
      V3 mn = vmin(vmin(a, b), c), mx = vmax(vmax(a, b), c);
      V3 size = mx - mn;
      const float eps = 1e-5f;
      V3 pad = v3(size.x < eps ? eps : 0.0f, size.y < eps ? eps : 0.0f, size.z < eps ? eps : 0.0f);
      Aabb bx{mn - pad * 0.5f, mx + pad * 0.5f};
      boxes_.push_back(bx);
      centers_.push_back(aabb_center(bx));
    }
  }
  void build() {
    nodes.clear();
    order.resize(indices_.size() / 3);
    for (size_t i = 0; i < order.size(); i++) order[i] = i;
    if (!order.empty()) subdivide(0, order.size());
  }
  std::vector<BuildNode> nodes;
  std::vector<size_t> order;  // sorted position -> original triangle id

 private:
  void make_leaf(size_t node, size_t first, size_t count) {
    // NOTE: count > 7 overflows the 3-bit field exactly like the reference (blas.rs:111-115)
    nodes[node].data = ((uint32_t)first << 3) | (uint32_t)count;
    nodes[node].skip = (uint32_t)nodes.size();
  }
  void subdivide(size_t first, size_t count) {
    size_t node = nodes.size();
    nodes.emplace_back();
    Aabb bb;
    for (size_t i = 0; i < count; i++) bb = aabb_union(bb, boxes_[order[first + i]]);
    nodes[node].mn = bb.mn;
    nodes[node].mx = bb.mx;
    if (count <= 4) return make_leaf(node, first, count);

    V3 ext = bb.mx - bb.mn;
    int axis = ext.y > ext.x ? 1 : ((ext.z > ext.x && ext.z > ext.y) ? 2 : 0);
    float split_len = ext[axis], split_min = bb.mn[axis];
    if (split_len < 1e-6f) return make_leaf(node, first, count);

    const int BINS = 16;
    float scale = (float)BINS / split_len;
    auto bin_of = [&](float val) -> size_t {
      float f = (val - split_min) * scale;
      size_t idx = (f != f || f <= 0.0f) ? 0 : (f >= 1.8446744e19f ? (size_t)-1 : (size_t)f);  // `as usize`
      return idx < (size_t)(BINS - 1) ? idx : (size_t)(BINS - 1);
    };
    Aabb bin_box[BINS];
    uint32_t bin_cnt[BINS] = {0};
    for (size_t i = 0; i < count; i++) {
      size_t t = order[first + i];
      size_t b = bin_of(centers_[t][axis]);
      bin_cnt[b]++;
      bin_box[b] = aabb_union(bin_box[b], boxes_[t]);
    }
    float l_area[BINS], r_area[BINS];
    uint32_t l_cnt[BINS], r_cnt[BINS];
    {
      Aabb cur;
      uint32_t sum = 0;
      for (int i = 0; i < BINS; i++) {
        sum += bin_cnt[i];
        cur = aabb_union(cur, bin_box[i]);
        l_area[i] = aabb_area(cur);
        l_cnt[i] = sum;
      }
      cur = Aabb();
      sum = 0;
      for (int i = BINS - 1; i >= 0; i--) {
        sum += bin_cnt[i];
        cur = aabb_union(cur, bin_box[i]);
        r_area[i] = aabb_area(cur);
        r_cnt[i] = sum;
      }
    }
    float best = INFINITY;
    int best_split = -1;
    for (int i = 0; i < BINS - 1; i++) {
      if (l_cnt[i] == 0 || r_cnt[i + 1] == 0) continue;
      float cost = l_area[i] * (float)l_cnt[i] + r_area[i + 1] * (float)r_cnt[i + 1];
      if (cost < best) {
        best = cost;
        best_split = i;
      }
    }
    if (best_split < 0) return make_leaf(node, first, count);

    // two-pointer partition (blas.rs:179-199)
    size_t i = first, j = first + count - 1;
    while (i <= j) {
      if (bin_of(centers_[order[i]][axis]) <= (size_t)best_split) {
        i++;
      } else if (bin_of(centers_[order[j]][axis]) > (size_t)best_split) {
        if (j == 0) break;
        j--;
      } else {
        std::swap(order[i], order[j]);
        i++;
        if (j == 0) break;
        j--;
      }
    }
    size_t l_count = i - first, r_count = count - l_count;
    if (l_count == 0 || l_count == count) return make_leaf(node, first, count);

    // static front-to-back heuristic: the costlier child goes first (blas.rs:209-217)
    float l_cost = l_area[best_split] * (float)l_count;
    float r_cost = r_area[best_split + 1] * (float)r_count;
    if (r_cost > l_cost) {
      std::rotate(order.begin() + first, order.begin() + first + l_count, order.begin() + first + count);
      std::swap(l_count, r_count);
    }
    nodes[node].data = 0;
    subdivide(first, l_count);
    subdivide(first + l_count, r_count);
    nodes[node].skip = (uint32_t)nodes.size();
  }
  const std::vector<uint32_t>& indices_;
  std::vector<Aabb> boxes_;
  std::vector<V3> centers_;
};

void pack_nodes(const std::vector<BuildNode>& nodes, std::vector<float>& out) {
  for (const BuildNode& n : nodes) {
    float f[8];
    f[0] = n.mn.x; f[1] = n.mn.y; f[2] = n.mn.z;
    std::memcpy(&f[3], &n.skip, 4);
    f[4] = n.mx.x; f[5] = n.mx.y; f[6] = n.mx.z;
    std::memcpy(&f[7], &n.data, 4);
    out.insert(out.end(), f, f + 8);
  }
}

// ------------------------------------------------------------------ TLAS
// Median split on sorted centres, leaf = one instance (bvh/tlas.rs).
struct RawInstance {
  M4 transform, inverse;
  uint32_t blas_node_offset = 0, attr_offset = 0, instance_id = 0, pad = 0;
};
class TlasBuilder {
 public:
  TlasBuilder(const std::vector<RawInstance>& inst, const std::vector<Aabb>& blas_boxes) : inst_(inst) {
    for (size_t i = 0; i < inst.size(); i++) {
      Aabb wb = aabb_transform(blas_boxes[i], inst[i].transform);
      boxes_.push_back(wb);
      centers_.push_back(aabb_center(wb));
      order.push_back(i);
    }
  }
  void build() {
    nodes.clear();
    if (!inst_.empty()) subdivide(0, inst_.size());
  }
  std::vector<BuildNode> nodes;
  std::vector<size_t> order;

 private:
  void subdivide(size_t first, size_t count) {
    size_t node = nodes.size();
    nodes.emplace_back();
    Aabb bb;
    for (size_t i = 0; i < count; i++) bb = aabb_union(bb, boxes_[order[first + i]]);
    nodes[node].mn = bb.mn;
    nodes[node].mx = bb.mx;
    if (count == 1) {
      nodes[node].data = ((uint32_t)first << 3) | 1u;
      nodes[node].skip = (uint32_t)nodes.size();
      return;
    }
    V3 ext = bb.mx - bb.mn;
    int axis = ext.y > ext.x ? 1 : ((ext.z > ext.x && ext.z > ext.y) ? 2 : 0);
    std::stable_sort(order.begin() + first, order.begin() + first + count,
                     [&](size_t a, size_t b) { return centers_[a][axis] < centers_[b][axis]; });
    size_t mid = count / 2, l_count = mid, r_count = count - mid;
    Aabb lb, rb;
    for (size_t i = 0; i < l_count; i++) lb = aabb_union(lb, boxes_[order[first + i]]);
    for (size_t i = 0; i < r_count; i++) rb = aabb_union(rb, boxes_[order[first + mid + i]]);
    if (aabb_area(rb) * (float)r_count > aabb_area(lb) * (float)l_count) {
      std::rotate(order.begin() + first, order.begin() + first + l_count, order.begin() + first + count);
      std::swap(l_count, r_count);
    }
    nodes[node].data = 0;
    subdivide(first, l_count);
    subdivide(first + l_count, r_count);
    nodes[node].skip = (uint32_t)nodes.size();
  }
  const std::vector<RawInstance>& inst_;
  std::vector<Aabb> boxes_;
  std::vector<V3> centers_;
};

}  // namespace

// std::vector whose resize() leaves new elements uninitialised (default-initialisation instead of value-initialisation)
template <class T>
struct DefaultInitAllocator : std::allocator<T> {
  template <class U>
  struct rebind {
    using other = DefaultInitAllocator<U>;
  };
  using std::allocator<T>::allocator;
  template <class U>
  void construct(U* p) noexcept(std::is_nothrow_default_constructible<U>::value) {
    ::new (static_cast<void*>(p)) U;
  }
  template <class U, class... Args>
  void construct(U* p, Args&&... args) {
    ::new (static_cast<void*>(p)) U(std::forward<Args>(args)...);
  }
};
template <class T>
using RawVec = std::vector<T, DefaultInitAllocator<T>>;

// ------------------------------------------------------------------ World
struct ms_world {
  SceneData scene;
  std::vector<RawInstance> raw_instances;
  std::vector<Aabb> instance_blas_boxes;
  std::vector<uint32_t> blas_root_offsets;
  // bridge arrays (render_buffers.rs:6-17)
  // the big ones grow with resize() and are then written element by element: no value-initialisation (21 MB of zeros per
  // update for 262 k triangles otherwise)
  RawVec<float> vertices, normals, uvs, blas;
  RawVec<uint32_t> topology;
  std::vector<float> tlas, instances, camera;
  std::vector<uint32_t> lights, draw_commands;
  // glTF scene graph (SceneData.nodes / skins / animations) and World.active_anim_index
  GltfScene gltf;
  size_t active_anim = 0;
  std::vector<float> scratch_nodes;     // BLAS build output of the geometry being processed (world_update)
  std::vector<uint32_t> scratch_order;
  // optional BLAS builder hook (ms_world_set_blas_builder): the GPU builder of libmi355rt.so
  ms_blas_builder blas_hook = nullptr;
  void* blas_hook_user = nullptr;
  // optional device updater (ms_world_set_device_updater): rt_world_update of libmi355rt.so
  ms_device_updater device_hook = nullptr;
  void* device_hook_user = nullptr;
  uint64_t static_epoch = 0;            // names the static scene description handed to the updater
  bool device_resident = false;         // the last update ran on the device: the host bridge arrays were not refreshed
};

static thread_local std::string g_last_error;

// Index-parallel host loop for the per-vertex / per-triangle passes of world_update (the reference runs them on one
// thread; every index writes its own output slot, so the result does not depend on the split).  A small pool of
// worker threads is started on first use and kept: an animated scene calls this twice per frame.
class WorkerPool {
 public:
  static WorkerPool& get() {
    static WorkerPool pool;
    return pool;
  }
  size_t size() const { return workers_.size() + 1; }  // + the calling thread
  // run job(k) for k = 0..n_jobs-1, job 0 on the calling thread; returns when all are done
  void run(size_t n_jobs, const std::function<void(size_t)>& job) {
    if (n_jobs <= 1 || workers_.empty()) {
      for (size_t k = 0; k < n_jobs; k++) job(k);
      return;
    }
    std::unique_lock<std::mutex> call(call_mutex_);  // one parallel region at a time
    {
      std::lock_guard<std::mutex> lk(m_);
      job_ = &job;
      n_jobs_ = n_jobs;
      next_ = 1;
      pending_ = n_jobs - 1;
      generation_++;
    }
    cv_.notify_all();
    job(0);
    for (;;) {  // the caller helps with whatever is left
      size_t k;
      {
        std::lock_guard<std::mutex> lk(m_);
        if (next_ >= n_jobs_) break;
        k = next_++;
      }
      job(k);
      std::lock_guard<std::mutex> lk(m_);
      pending_--;
    }
    std::unique_lock<std::mutex> lk(m_);
    done_cv_.wait(lk, [&] { return pending_ == 0; });
    job_ = nullptr;
  }

 private:
  WorkerPool() {
    size_t n = std::thread::hardware_concurrency();
    if (n > 16) n = 16;
    if (const char* e = std::getenv("MS_THREADS")) {  // MS_THREADS=1: everything on the calling thread
      const long v = std::atol(e);
      if (v >= 1 && (size_t)v < n) n = (size_t)v;
    }
    for (size_t i = 1; i < n; i++) workers_.emplace_back([this] { loop(); });
  }
  ~WorkerPool() {
    {
      std::lock_guard<std::mutex> lk(m_);
      stop_ = true;
    }
    cv_.notify_all();
    for (std::thread& t : workers_) t.join();
  }
  void loop() {
    uint64_t seen = 0;
    std::unique_lock<std::mutex> lk(m_);
    for (;;) {
      cv_.wait(lk, [&] { return stop_ || (generation_ != seen && job_ && next_ < n_jobs_); });
      if (stop_) return;
      seen = generation_;
      while (job_ && next_ < n_jobs_) {
        const size_t k = next_++;
        const std::function<void(size_t)>* job = job_;
        lk.unlock();
        (*job)(k);
        lk.lock();
        if (--pending_ == 0) done_cv_.notify_all();
      }
    }
  }
  std::vector<std::thread> workers_;
  std::mutex m_, call_mutex_;
  std::condition_variable cv_, done_cv_;
  const std::function<void(size_t)>* job_ = nullptr;
  size_t n_jobs_ = 0, next_ = 0, pending_ = 0;
  uint64_t generation_ = 0;
  bool stop_ = false;
};

template <class F>
static void parallel_for(size_t n, F&& body) {
  const size_t kMinPerThread = 16384;
  size_t parts = WorkerPool::get().size();
  if (parts > n / kMinPerThread) parts = n / kMinPerThread;
  if (parts <= 1) {
    body((size_t)0, n);
    return;
  }
  const size_t step = (n + parts - 1) / parts;
  WorkerPool::get().run(parts, [&](size_t k) {
    const size_t lo = k * step, hi = std::min(n, lo + step);
    if (lo < hi) body(lo, hi);
  });
}

// lib.rs:383-491 apply_animation: sample every channel of one animation at `time` into the nodes' local TRS
static void apply_animation(ms_world& w, size_t anim_idx, float time_in) {
  const GAnimation& anim = w.gltf.animations[anim_idx];
  for (const GChannel& ch : anim.channels) {
    if (ch.target_node >= w.gltf.nodes.size()) continue;
    const float time = anim.duration > 0.0f ? std::fmod(time_in, anim.duration) : time_in;
    const std::vector<float>& in = ch.inputs;
    const size_t count = in.size();
    if (count == 0) continue;
    size_t next = 0;
    while (next < count && in[next] < time) next++;
    if (next == 0) next = 1;
    if (next >= count) next = 0;
    const size_t prev = next == 0 ? count - 1 : next - 1;
    const float t0 = in[prev], t1 = in[next];
    const float dt = t1 < t0 ? anim.duration - t0 + t1 : t1 - t0;
    const float current = t1 < t0 ? (time >= t0 ? time - t0 : (anim.duration - t0) + time) : time - t0;
    float factor = 0.0f;
    if (dt > 0.0001f) {
      factor = current / dt;
      factor = factor < 0.0f ? 0.0f : (factor > 1.0f ? 1.0f : factor);
    }
    const size_t stride = ch.interpolation == 2 ? 3 : 1, offset = ch.interpolation == 2 ? 1 : 0;  // CUBICSPLINE: the value of (in, value, out)
    const size_t i0 = prev * stride + offset, i1 = next * stride + offset;
    const float tf = ch.interpolation == 1 ? 0.0f : factor;  // STEP holds the previous key
    GNode& node = w.gltf.nodes[ch.target_node];
    const size_t comps = ch.kind == 1 ? 4 : 3, n_keys = ch.out.size() / comps;
    if (i0 >= n_keys || i1 >= n_keys) continue;
    const float* a = &ch.out[i0 * comps];
    const float* b = &ch.out[i1 * comps];
    if (ch.kind == 0) {
      node.translation = v3_lerp(v3(a[0], a[1], a[2]), v3(b[0], b[1], b[2]), tf);
    } else if (ch.kind == 1) {
      node.rotation = q_slerp(q_normalize(Quat{a[0], a[1], a[2], a[3]}), q_normalize(Quat{b[0], b[1], b[2], b[3]}), tf);
    } else {
      node.scale = v3_lerp(v3(a[0], a[1], a[2]), v3(b[0], b[1], b[2]), tf);
    }
  }
}

// lib.rs:372-381 update_node_global (iterative: assets may nest deeper than the C stack should)
static void update_globals(const ms_world& w, std::vector<M4>& globals) {
  const size_t n = w.gltf.nodes.size();
  globals.assign(n, m4_identity());
  std::vector<std::pair<size_t, M4>> stack;
  std::vector<uint8_t> seen(n, 0);
  for (size_t root = 0; root < n; root++) {
    if (w.gltf.nodes[root].parent >= 0) continue;
    stack.emplace_back(root, m4_identity());
    while (!stack.empty()) {
      const size_t i = stack.back().first;
      const M4 parent = stack.back().second;
      stack.pop_back();
      if (i >= n || seen[i]) continue;  // a malformed graph (cycle / shared child) is walked once
      seen[i] = 1;
      const GNode& node = w.gltf.nodes[i];
      const M4 global = m4_mul(parent, m4_from_srt(node.scale, node.rotation, node.translation));
      globals[i] = global;
      for (size_t k = node.children.size(); k-- > 0;) stack.emplace_back(node.children[k], global);
    }
  }
}

struct PhaseTimer {  // MS_PROFILE=1: per-phase wall time of world_update on stderr
  bool on = std::getenv("MS_PROFILE") != nullptr;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  double acc[6] = {0, 0, 0, 0, 0, 0};
  void lap(int k) {
    if (!on) return;
    auto t = std::chrono::steady_clock::now();
    acc[k] += std::chrono::duration<double, std::milli>(t - t0).count();
    t0 = t;
  }
  ~PhaseTimer() {
    if (on)
      std::fprintf(stderr, "world_update: animation+graph %.2f ms, skinning %.2f, BLAS build %.2f, topology %.2f, copies %.2f, instances+TLAS %.2f\n",
                   acc[0], acc[1], acc[2], acc[3], acc[4], acc[5]);
  }
};

// rebuilder.rs:40-47: the joint matrices of one skin, global(joint) * inverse_bind
static void skin_joint_mats(const GSkin& skin, const std::vector<M4>& globals, std::vector<M4>& out) {
  for (size_t k = 0; k < skin.joints.size() && k < skin.inverse_bind.size(); k++)
    out.push_back(m4_mul(skin.joints[k] < globals.size() ? globals[skin.joints[k]] : m4_identity(), skin.inverse_bind[k]));
}

// The device half of update(t) (ms_world_set_device_updater): describe the scene and this frame's joint matrices to the
// updater.  false = it refused or failed; the caller then runs the host path.
static bool device_update(ms_world& w, const std::vector<M4>& globals) {
  std::vector<rt_world_geometry> geos(w.scene.geometries.size());
  for (size_t gi = 0; gi < geos.size(); gi++) {
    const Geometry& g = w.scene.geometries[gi];
    rt_world_geometry& d = geos[gi];
    std::memset(&d, 0, sizeof(d));
    d.positions = g.positions.empty() ? nullptr : &g.positions[0].x;
    d.normals = g.normals.empty() ? nullptr : &g.normals[0].x;
    d.uvs = g.uvs.empty() ? nullptr : &g.uvs[0].x;
    d.joints = g.joints.data();
    d.weights = g.weights.data();
    d.indices = g.indices.data();
    d.attributes = g.attributes.data();
    d.n_verts = (uint32_t)g.positions.size();
    d.n_uvs = (uint32_t)std::min(g.uvs.size(), g.positions.size());
    d.n_tris = (uint32_t)(g.indices.size() / 3);
    d.skin = (g.skin_index >= 0 && (size_t)g.skin_index < w.gltf.skins.size()) ? (int32_t)g.skin_index : -1;
    if (g.normals.size() < g.positions.size() || g.joints.size() < g.positions.size() * 4 ||
        g.weights.size() < g.positions.size() * 4 || g.attributes.size() < (size_t)d.n_tris * 16) {
      g_last_error = "device updater: geometry " + std::to_string(gi) + " has short attribute arrays; host update used";
      return false;
    }
  }
  std::vector<rt_instance> inst(w.raw_instances.size());
  for (size_t i = 0; i < inst.size(); i++) {
    const RawInstance& r = w.raw_instances[i];
    std::memcpy(inst[i].transform, r.transform.c, 64);
    std::memcpy(inst[i].inverse, r.inverse.c, 64);
    inst[i].blas_node_offset = 0;
    inst[i].attr_offset = r.attr_offset;
    inst[i].instance_id = r.instance_id;
    inst[i].pad = r.pad;
  }
  std::vector<uint32_t> skin_first(w.gltf.skins.size() + 1, 0u);
  std::vector<M4> mats;
  for (size_t si = 0; si < w.gltf.skins.size(); si++) {
    skin_first[si] = (uint32_t)mats.size();
    skin_joint_mats(w.gltf.skins[si], globals, mats);
  }
  skin_first[w.gltf.skins.size()] = (uint32_t)mats.size();
  rt_world_frame f;
  std::memset(&f, 0, sizeof(f));
  f.static_epoch = w.static_epoch;
  f.n_geometries = (uint32_t)geos.size();
  f.n_instances = (uint32_t)inst.size();
  f.n_skins = (uint32_t)w.gltf.skins.size();
  f.geometries = geos.data();
  f.instances = inst.data();
  f.skin_first = skin_first.data();
  f.joint_mats = mats.empty() ? nullptr : &mats[0].c[0][0];
  const int rc = w.device_hook(w.device_hook_user, &f);
  if (rc < 0) {
    g_last_error = "device updater failed (" + std::to_string(rc) + "); this update ran on the host";
    return false;
  }
  return true;
}

static void world_update(ms_world& w, float time = 0.0f) {
  PhaseTimer pt;
  // --- lib.rs:149-184: animation, then global transforms of the scene graph ---
  if (!w.gltf.animations.empty()) {
    const size_t ai = w.active_anim < w.gltf.animations.size() ? w.active_anim : 0;
    const float duration = w.gltf.animations[ai].duration;
    apply_animation(w, ai, duration > 0.001f ? std::fmod(time, duration) : 0.0f);
  }
  std::vector<M4> globals;
  update_globals(w, globals);
  pt.lap(0);

  // lib.rs:196-204 (quirk kept): every instance after the first is overwritten with rotY(pi) * scale(0.7).  Done before
  // anything reads the transforms; the result does not depend on the frame.
  if (!w.scene.keep_instance_transforms)
    for (size_t i = 1; i < w.raw_instances.size(); i++) {
      M4 t = m4_mul(m4_from_rotation_y(3.14159274101257324219f), m4_from_scale(v3(0.7f, 0.7f, 0.7f)));
      w.raw_instances[i].transform = t;
      w.raw_instances[i].inverse = m4_inverse(t);
    }
  w.device_resident = false;
  if (w.device_hook) {
    if (device_update(w, globals)) {
      w.device_resident = true;
      return;
    }
  }

  // --- rebuilder.rs:9-190: per geometry, vertices + BLAS + topology ---
  w.vertices.clear();
  w.normals.clear();
  w.uvs.clear();
  w.topology.clear();
  w.blas.clear();
  w.lights.clear();
  w.draw_commands.clear();
  w.blas_root_offsets.clear();
  std::vector<std::vector<uint32_t>> emissive(w.scene.geometries.size());
  std::vector<std::pair<uint32_t, uint32_t>> geom_ranges(w.scene.geometries.size(), {0u, 0u});
  uint32_t node_offset = 0;
  for (size_t gi = 0; gi < w.scene.geometries.size(); gi++) {
    const Geometry& geo = w.scene.geometries[gi];
    if (geo.positions.empty()) {
      w.blas_root_offsets.push_back(0);
      continue;
    }
    // vertices / normals / uvs of this geometry are written in place at the end of the world arrays
    const size_t n_geo_verts = geo.positions.size();
    const uint32_t v_offset = (uint32_t)(w.vertices.size() / 4);
    w.vertices.resize(w.vertices.size() + n_geo_verts * 4);
    w.normals.resize(w.normals.size() + n_geo_verts * 4);
    w.uvs.resize(w.uvs.size() + n_geo_verts * 2);
    float* v4 = &w.vertices[(size_t)v_offset * 4];
    float* n4 = &w.normals[(size_t)v_offset * 4];
    float* uv2 = &w.uvs[(size_t)v_offset * 2];
    // rebuilder.rs:36-91: linear blend skinning with joint matrices global(joint) * inverse_bind
    const GSkin* skin = (geo.skin_index >= 0 && (size_t)geo.skin_index < w.gltf.skins.size()) ? &w.gltf.skins[(size_t)geo.skin_index] : nullptr;
    std::vector<M4> joint_mats;
    if (skin) skin_joint_mats(*skin, globals, joint_mats);
    parallel_for(n_geo_verts, [&](size_t lo_i, size_t hi_i) {
    for (size_t i = lo_i; i < hi_i; i++) {
      V3 p = geo.positions[i], n = geo.normals[i];
      V2 uv = i < geo.uvs.size() ? geo.uvs[i] : V2{0, 0};
      if (skin) {
        M4 mat;
        std::memset(mat.c, 0, sizeof(mat.c));
        for (int k = 0; k < 4; k++) {
          const float wk = geo.weights[i * 4 + (size_t)k];
          const uint32_t jk = geo.joints[i * 4 + (size_t)k];
          if (wk > 0.0f && jk < joint_mats.size())  // an out-of-range joint panics in the reference; skipped here
            for (int c = 0; c < 4; c++)
              for (int r = 0; r < 4; r++) mat.c[c][r] = mat.c[c][r] + joint_mats[jk].c[c][r] * wk;
        }
        bool any = false;
        for (int c = 0; c < 4; c++)
          for (int r = 0; r < 4; r++) any = any || mat.c[c][r] != 0.0f;
        if (!any) mat = m4_identity();  // "if mat == Mat4::ZERO"
        p = m4_transform_point3(mat, p);
        n = normalize_or_zero(m4_transform_vector3(mat, n));
      }
      if (is_nan(p)) p = v3(0, 0, 0);
      if (is_nan(n)) n = v3(0, 0, 1);
      v4[i * 4] = p.x; v4[i * 4 + 1] = p.y; v4[i * 4 + 2] = p.z; v4[i * 4 + 3] = 1.0f;
      n4[i * 4] = n.x; n4[i * 4 + 1] = n.y; n4[i * 4 + 2] = n.z; n4[i * 4 + 3] = 0.0f;
      uv2[i * 2] = uv.x;
      uv2[i * 2 + 1] = uv.y;
    }
    });
    pt.lap(1);
    // BLAS: nodes (8 f32 each, BLAS-local skips) + triangle order, from the CPU builder or from the hook
    std::vector<float>& packed = w.scratch_nodes;  // reused between updates: no 17 MB of zero-fill per frame
    std::vector<uint32_t>& order = w.scratch_order;
    const uint32_t n_tris = (uint32_t)(geo.indices.size() / 3);
    size_t n_packed = 0;                            // nodes in `packed`
    bool built = false;
    if (w.blas_hook && n_tris) {
      if (packed.size() < (size_t)2 * n_tris * 8) packed.resize((size_t)2 * n_tris * 8);
      if (order.size() < n_tris) order.resize(n_tris);
      uint32_t n_nodes = 0;
      const int rc = w.blas_hook(w.blas_hook_user, v4, (uint32_t)n_geo_verts, geo.indices.data(), n_tris, packed.data(),
                                 2 * n_tris, &n_nodes, order.data());
      if (rc >= 0) {
        n_packed = n_nodes;
        built = true;
      } else {
        g_last_error = "BLAS builder hook failed (" + std::to_string(rc) + "); the CPU builder was used for this update";
      }
    }
    if (!built) {
      BlasBuilder bb(v4, n_geo_verts, geo.indices);
      bb.build();
      packed.clear();
      pack_nodes(bb.nodes, packed);
      n_packed = bb.nodes.size();
      order.assign(bb.order.begin(), bb.order.end());
    }
    pt.lap(2);
    uint32_t topo_start = (uint32_t)(w.topology.size() / 20);
    for (size_t ni = 0; ni < n_packed; ni++) {  // leaf `first` becomes a global topology index (rebuilder.rs:123-134)
      uint32_t data;
      std::memcpy(&data, &packed[ni * 8 + 7], 4);
      if (data != 0) {
        data = (((data >> 3) + topo_start) << 3) | (data & 7u);
        std::memcpy(&packed[ni * 8 + 7], &data, 4);
      }
    }
    w.topology.resize(w.topology.size() + (size_t)n_tris * 20);
    uint32_t* topo_rows = &w.topology[(size_t)topo_start * 20];
    parallel_for(n_tris, [&](size_t lo_i, size_t hi_i) {
      for (size_t i = lo_i; i < hi_i; i++) {
        size_t old_id = order[i];
        uint32_t* row = topo_rows + i * 20;
        row[0] = geo.indices[old_id * 3] + v_offset;
        row[1] = geo.indices[old_id * 3 + 1] + v_offset;
        row[2] = geo.indices[old_id * 3 + 2] + v_offset;
        row[3] = (uint32_t)gi;
        std::memcpy(&row[4], &geo.attributes[old_id * 16], 64);
      }
    });
    for (size_t i = 0; i < n_tris; i++) {  // emissive triangles, in topology order (rebuilder.rs:163-168)
      float mat_val = geo.attributes[(size_t)order[i] * 16 + 3];
      if (std::fabs(mat_val - 3.0f) < 1e-6f) emissive[gi].push_back(topo_start + (uint32_t)i);
    }
    pt.lap(3);
    w.blas.insert(w.blas.end(), packed.begin(), packed.begin() + (long)n_packed * 8);
    w.blas_root_offsets.push_back(node_offset);
    node_offset += (uint32_t)n_packed;
    geom_ranges[gi] = {topo_start, (uint32_t)(w.topology.size() / 20) - topo_start};
    pt.lap(4);
  }

  // --- lib.rs:194-230: instance transforms, BLAS offsets, local boxes ---
  for (size_t i = 0; i < w.raw_instances.size(); i++) {
    RawInstance& inst = w.raw_instances[i];
    size_t gi = inst.instance_id;
    if (gi < w.blas_root_offsets.size()) {
      inst.blas_node_offset = w.blas_root_offsets[gi];
      size_t base = (size_t)inst.blas_node_offset * 8;
      if (base < w.blas.size())
        w.instance_blas_boxes[i] = Aabb{v3(w.blas[base], w.blas[base + 1], w.blas[base + 2]),
                                        v3(w.blas[base + 4], w.blas[base + 5], w.blas[base + 6])};
    }
  }

  // --- lib.rs:232-270: TLAS, lights, draw commands, instance packing ---
  TlasBuilder tb(w.raw_instances, w.instance_blas_boxes);
  tb.build();
  w.tlas.clear();
  pack_nodes(tb.nodes, w.tlas);
  w.instances.clear();
  for (size_t i = 0; i < tb.order.size(); i++) {
    const RawInstance& inst = w.raw_instances[tb.order[i]];
    size_t gi = inst.instance_id;
    uint32_t v_count = 0, v_start = 0;
    if (gi < w.blas_root_offsets.size()) {
      if (gi < geom_ranges.size()) {
        v_count = geom_ranges[gi].second * 3;
        v_start = geom_ranges[gi].first * 3;
      }
      if (gi < emissive.size())
        for (uint32_t tri : emissive[gi]) {
          w.lights.push_back((uint32_t)i);
          w.lights.push_back(tri);
        }
    }
    const uint32_t dc[4] = {v_count, 1u, v_start, (uint32_t)i};
    w.draw_commands.insert(w.draw_commands.end(), dc, dc + 4);
    rt_instance packed;
    std::memcpy(packed.transform, inst.transform.c, 64);
    std::memcpy(packed.inverse, inst.inverse.c, 64);
    packed.blas_node_offset = inst.blas_node_offset;
    packed.attr_offset = inst.attr_offset;
    packed.instance_id = inst.instance_id;
    packed.pad = inst.pad;
    const float* pf = reinterpret_cast<const float*>(&packed);
    w.instances.insert(w.instances.end(), pf, pf + 36);
  }
}

extern "C" {

const char* ms_last_error(void) { return g_last_error.c_str(); }

ms_world* ms_world_create(const char* scene_name, const char* obj_source) {
  return ms_world_create_glb(scene_name, obj_source, nullptr, 0);
}

ms_world* ms_world_create_glb(const char* scene_name, const char* obj_source, const uint8_t* glb, size_t glb_size) {
  const bool has_glb = glb != nullptr;
  std::string name = scene_name ? scene_name : "cornell";
  ObjMesh mesh;
  bool has_mesh = obj_source != nullptr;
  if (has_mesh) mesh = parse_obj(obj_source);
  ms_world* w = new ms_world();
  if (name == "spheres") {
    g_last_error = "scene 'spheres' is seeded from rand::rng() in the reference and has no reproducible output";
    delete w;
    return nullptr;
  } else if (name == "mixed") {
    w->scene = scene_mixed();
  } else if (name == "special") {
    w->scene = scene_special();
  } else if (name == "mesh") {
    w->scene = scene_mesh();
  } else if (name == "viewer") {
    w->scene = scene_viewer(has_mesh ? &mesh : nullptr, has_glb);
  } else if (name == "instanced1000") {
    w->scene = scene_instanced1000();
  } else if (name == "instanced16384") {
    w->scene = scene_instanced_lattice(32, 32, 16, 1, 0.02f);
  } else if (name == "sponza_like") {
    w->scene = scene_sponza_like();
  } else if (name == "glass_blob") {
    w->scene = scene_glass_blob();
  } else {
    w->scene = scene_cornell();
  }
  g_last_error.clear();
  if (has_glb) {
    // lib.rs:57-67: `let _ = loader::load_gltf(..)` — a GLB that fails to load leaves the procedural scene as it is;
    // the reason is kept for ms_last_error()
    std::string err;
    if (!load_gltf(w->scene, w->gltf, glb, glb_size, err)) g_last_error = err;
    w->scene.textures = w->gltf.textures;
  }
  for (const SceneInstance& si : w->scene.instances) {
    RawInstance ri;
    ri.transform = si.transform;
    ri.inverse = m4_inverse(si.transform);
    ri.instance_id = (uint32_t)si.geometry_index;
    w->raw_instances.push_back(ri);
    w->instance_blas_boxes.push_back(Aabb());
  }
  if (w->raw_instances.empty()) {
    RawInstance ri;
    ri.transform = m4_identity();
    ri.inverse = m4_identity();
    w->raw_instances.push_back(ri);
    w->instance_blas_boxes.push_back(Aabb());
  }
  w->camera.assign(24, 0.0f);
  static std::atomic<uint64_t> next_epoch{0};
  w->static_epoch = ++next_epoch;   // unique per world: a renderer fed by two worlds in turn re-reads the static part
  world_update(*w);
  return w;
}

void ms_world_destroy(ms_world* w) { delete w; }

void ms_world_update(ms_world* w, float time) {
  if (!w) return;
  g_last_error.clear();
  world_update(*w, time);
}

size_t ms_world_animation_count(const ms_world* w) { return w ? w->gltf.animations.size() : 0; }
const char* ms_world_animation_name(const ms_world* w, size_t index) {
  return (w && index < w->gltf.animations.size()) ? w->gltf.animations[index].name.c_str() : "";
}
void ms_world_set_animation(ms_world* w, size_t index) {
  if (w && index < w->gltf.animations.size()) w->active_anim = index;
}
int ms_world_load_animation_glb(ms_world* w, const uint8_t* glb, size_t glb_size) {
  if (!w || !glb) return -1;
  SceneData tmp_scene;
  GltfScene tmp;
  std::string err;
  if (!load_gltf(tmp_scene, tmp, glb, glb_size, err)) {
    g_last_error = err;
    return -1;
  }
  const int added = (int)tmp.animations.size();
  for (GAnimation& a : tmp.animations) w->gltf.animations.push_back(std::move(a));
  return added;
}
void ms_world_set_blas_builder(ms_world* w, ms_blas_builder fn, void* user) {
  if (!w) return;
  w->blas_hook = fn;
  w->blas_hook_user = user;
}
void ms_world_set_device_updater(ms_world* w, ms_device_updater fn, void* user) {
  if (!w) return;
  w->device_hook = fn;
  w->device_hook_user = user;
  if (!fn) w->device_resident = false;
}
int ms_world_device_resident(const ms_world* w) { return (w && w->device_resident) ? 1 : 0; }
int ms_build_blas(const float* verts4, uint32_t n_verts, const uint32_t* indices, uint32_t n_tris, float* nodes_out,
                  uint32_t nodes_cap, uint32_t* n_nodes_out, uint32_t* order_out) {
  if (!n_nodes_out) return -1;
  *n_nodes_out = 0;
  if (n_tris == 0) return 0;
  if (!verts4 || !indices || !nodes_out || !order_out) return -1;
  std::vector<uint32_t> idx(indices, indices + (size_t)n_tris * 3);
  for (uint32_t v : idx)
    if (v >= n_verts) return -1;
  BlasBuilder bb(verts4, n_verts, idx);
  bb.build();
  if (bb.nodes.size() > nodes_cap) return -1;
  std::vector<float> packed;
  pack_nodes(bb.nodes, packed);
  std::memcpy(nodes_out, packed.data(), packed.size() * 4);
  for (size_t i = 0; i < bb.order.size(); i++) order_out[i] = (uint32_t)bb.order[i];
  *n_nodes_out = (uint32_t)bb.nodes.size();
  return 0;
}
int ms_build_tlas(const float* boxes6, const float* transforms16, uint32_t n, float* nodes_out, uint32_t nodes_cap,
                  uint32_t* n_nodes_out, uint32_t* order_out) {
  if (!n_nodes_out) return -1;
  *n_nodes_out = 0;
  if (n == 0) return 0;
  if (!boxes6 || !nodes_out || !order_out) return -1;
  std::vector<RawInstance> inst(n);
  std::vector<Aabb> boxes(n);
  for (uint32_t i = 0; i < n; i++) {
    boxes[i].mn = v3(boxes6[6 * i], boxes6[6 * i + 1], boxes6[6 * i + 2]);
    boxes[i].mx = v3(boxes6[6 * i + 3], boxes6[6 * i + 4], boxes6[6 * i + 5]);
    if (transforms16)
      std::memcpy(&inst[i].transform, transforms16 + 16 * (size_t)i, 64);
    else
      inst[i].transform = m4_identity();
  }
  TlasBuilder tb(inst, boxes);
  tb.build();
  if (tb.nodes.size() > nodes_cap) return -1;
  std::vector<float> packed;
  pack_nodes(tb.nodes, packed);
  std::memcpy(nodes_out, packed.data(), packed.size() * 4);
  for (size_t i = 0; i < tb.order.size(); i++) order_out[i] = (uint32_t)tb.order[i];
  *n_nodes_out = (uint32_t)tb.nodes.size();
  return 0;
}
size_t ms_world_node_count(const ms_world* w) { return w ? w->gltf.nodes.size() : 0; }
size_t ms_world_encoded_texture_count(const ms_world* w) { return w ? w->scene.textures.size() : 0; }
const uint8_t* ms_world_encoded_texture(const ms_world* w, size_t index, size_t* size) {
  if (size) *size = 0;
  if (!w || index >= w->scene.textures.size()) return nullptr;
  if (size) *size = w->scene.textures[index].size();
  return w->scene.textures[index].data();
}

void ms_world_update_camera(ms_world* w, float width, float height) {
  if (!w || height == 0.0f) return;
  w->camera.resize(24);
  camera_buffer(w->scene.camera, width / height, w->camera.data());
}

#define MS_GETTER(NAME, TYPE, FIELD)                         \
  const TYPE* ms_world_##NAME(const ms_world* w, size_t* len) { \
    if (len) *len = w ? w->FIELD.size() : 0;                 \
    return w ? w->FIELD.data() : nullptr;                    \
  }
MS_GETTER(vertices, float, vertices)
MS_GETTER(normals, float, normals)
MS_GETTER(uvs, float, uvs)
MS_GETTER(mesh_topology, uint32_t, topology)
MS_GETTER(tlas, float, tlas)
MS_GETTER(blas, float, blas)
MS_GETTER(instances, float, instances)
MS_GETTER(lights, uint32_t, lights)
MS_GETTER(draw_commands, uint32_t, draw_commands)
MS_GETTER(camera, float, camera)

size_t ms_world_texture_count(const ms_world* w) { return w ? w->scene.textures_rgba.size() : 0; }
const uint8_t* ms_world_texture_rgba(const ms_world* w, size_t index) {
  if (!w || index >= w->scene.textures_rgba.size()) return nullptr;
  return w->scene.textures_rgba[index].data();
}

}  // extern "C"
