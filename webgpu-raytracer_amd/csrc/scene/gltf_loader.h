// gltf_loader.h — glTF 2.0 / GLB input of the scene compiler.  Included by scene_compiler.cpp
// inside its anonymous namespace, after the vector / matrix / Geometry / SceneData definitions.
//
// Restates rust-shader-tools/src/loader.rs:7-354 (`load_gltf`): textures (one blob per glTF *texture*, in texture order,
// empty for external images), nodes (local TRS + children), skins (joints + inverse bind matrices), one Geometry per mesh
// primitive (positions / normals / uv0 / joints0 / weights0, material -> 16-float attribute rows, material type from
// metallic / emissive), one SceneInstance per (node with mesh, primitive) and animations (channels with time keys).
// The reference delegates parsing to the `gltf` crate 1.4.1 (un-vendored, Cargo.lock); what that crate does on the way is
// restated from the glTF 2.0 specification: GLB container, JSON, buffers (GLB BIN chunk or base64 data URIs), buffer
// views with byteStride, accessors of every component type, `normalized` integer conversion, sparse accessors, default
// material (base colour 1, metallic 1, roughness 1), TRS defaults and matrix decomposition.  Nothing of the reference
// pins these (no asset, no test): parity unpinned, behaviour checked against a numpy restatement in tests/test_gltf.py.

// ------------------------------------------------------------------------------------------------------ JSON
struct JVal {
  enum Type { NUL, BOOL, NUM, STR, ARR, OBJ } t = NUL;
  double num = 0.0;
  bool b = false;
  std::string str;
  std::vector<JVal> arr;
  std::vector<std::pair<std::string, JVal>> obj;

  const JVal* get(const char* key) const {
    if (t != OBJ) return nullptr;
    for (const auto& kv : obj)
      if (kv.first == key) return &kv.second;
    return nullptr;
  }
  const JVal* at(size_t i) const { return (t == ARR && i < arr.size()) ? &arr[i] : nullptr; }
  size_t size() const { return t == ARR ? arr.size() : 0; }
  double number(const char* key, double def) const {
    const JVal* v = get(key);
    return (v && v->t == NUM) ? v->num : def;
  }
  long index(const char* key) const {  // non-negative integer member or -1
    const JVal* v = get(key);
    return (v && v->t == NUM && v->num >= 0.0) ? (long)v->num : -1;
  }
  std::string string(const char* key) const {
    const JVal* v = get(key);
    return (v && v->t == STR) ? v->str : std::string();
  }
};

struct JParser {
  const char* p;
  const char* end;
  std::string err;
  int depth = 0;

  void ws() {
    while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++;
  }
  bool fail(const char* m) {
    if (err.empty()) err = m;
    return false;
  }
  static void utf8(std::string& s, uint32_t c) {
    if (c < 0x80) {
      s.push_back((char)c);
    } else if (c < 0x800) {
      s.push_back((char)(0xc0 | (c >> 6)));
      s.push_back((char)(0x80 | (c & 63)));
    } else if (c < 0x10000) {
      s.push_back((char)(0xe0 | (c >> 12)));
      s.push_back((char)(0x80 | ((c >> 6) & 63)));
      s.push_back((char)(0x80 | (c & 63)));
    } else {
      s.push_back((char)(0xf0 | (c >> 18)));
      s.push_back((char)(0x80 | ((c >> 12) & 63)));
      s.push_back((char)(0x80 | ((c >> 6) & 63)));
      s.push_back((char)(0x80 | (c & 63)));
    }
  }
  bool hex4(uint32_t& v) {
    if (end - p < 4) return fail("json: short \\u escape");
    v = 0;
    for (int i = 0; i < 4; i++) {
      char c = *p++;
      v <<= 4;
      if (c >= '0' && c <= '9') v |= (uint32_t)(c - '0');
      else if (c >= 'a' && c <= 'f') v |= (uint32_t)(c - 'a' + 10);
      else if (c >= 'A' && c <= 'F') v |= (uint32_t)(c - 'A' + 10);
      else return fail("json: bad \\u escape");
    }
    return true;
  }
  bool string(std::string& out) {
    if (p >= end || *p != '"') return fail("json: expected string");
    p++;
    while (p < end && *p != '"') {
      char c = *p++;
      if (c == '\\') {
        if (p >= end) return fail("json: short escape");
        char e = *p++;
        switch (e) {
          case '"': out.push_back('"'); break;
          case '\\': out.push_back('\\'); break;
          case '/': out.push_back('/'); break;
          case 'b': out.push_back('\b'); break;
          case 'f': out.push_back('\f'); break;
          case 'n': out.push_back('\n'); break;
          case 'r': out.push_back('\r'); break;
          case 't': out.push_back('\t'); break;
          case 'u': {
            uint32_t v;
            if (!hex4(v)) return false;
            if (v >= 0xd800 && v < 0xdc00 && end - p >= 6 && p[0] == '\\' && p[1] == 'u') {
              p += 2;
              uint32_t lo;
              if (!hex4(lo)) return false;
              v = 0x10000 + ((v - 0xd800) << 10) + (lo - 0xdc00);
            }
            utf8(out, v);
            break;
          }
          default: return fail("json: bad escape");
        }
      } else {
        out.push_back(c);
      }
    }
    if (p >= end) return fail("json: unterminated string");
    p++;
    return true;
  }
  bool value(JVal& v) {
    if (++depth > 200) return fail("json: nesting too deep");
    ws();
    if (p >= end) return fail("json: unexpected end");
    bool ok = true;
    if (*p == '{') {
      p++;
      v.t = JVal::OBJ;
      ws();
      if (p < end && *p == '}') {
        p++;
      } else {
        for (;;) {
          ws();
          std::string k;
          if (!string(k)) { ok = false; break; }
          ws();
          if (p >= end || *p != ':') { ok = fail("json: expected ':'"); break; }
          p++;
          v.obj.emplace_back(std::move(k), JVal());
          if (!value(v.obj.back().second)) { ok = false; break; }
          ws();
          if (p < end && *p == ',') { p++; continue; }
          if (p < end && *p == '}') { p++; break; }
          ok = fail("json: expected ',' or '}'");
          break;
        }
      }
    } else if (*p == '[') {
      p++;
      v.t = JVal::ARR;
      ws();
      if (p < end && *p == ']') {
        p++;
      } else {
        for (;;) {
          v.arr.emplace_back();
          if (!value(v.arr.back())) { ok = false; break; }
          ws();
          if (p < end && *p == ',') { p++; continue; }
          if (p < end && *p == ']') { p++; break; }
          ok = fail("json: expected ',' or ']'");
          break;
        }
      }
    } else if (*p == '"') {
      v.t = JVal::STR;
      ok = string(v.str);
    } else if (end - p >= 4 && !std::memcmp(p, "true", 4)) {
      v.t = JVal::BOOL;
      v.b = true;
      p += 4;
    } else if (end - p >= 5 && !std::memcmp(p, "false", 5)) {
      v.t = JVal::BOOL;
      p += 5;
    } else if (end - p >= 4 && !std::memcmp(p, "null", 4)) {
      p += 4;
    } else {
      const char* s = p;
      if (p < end && (*p == '-' || *p == '+')) p++;
      while (p < end && ((*p >= '0' && *p <= '9') || *p == '.' || *p == 'e' || *p == 'E' || *p == '-' || *p == '+')) p++;
      if (p == s) return fail("json: unexpected character");
      std::string tmp(s, p);
      char* e = nullptr;
      v.num = std::strtod(tmp.c_str(), &e);
      if (!e || *e) return fail("json: bad number");
      v.t = JVal::NUM;
    }
    depth--;
    return ok;
  }
};

inline bool base64_decode(const std::string& s, size_t from, std::vector<uint8_t>& out) {
  uint32_t acc = 0;
  int bits = 0;
  for (size_t i = from; i < s.size(); i++) {
    const char c = s[i];
    int v;
    if (c >= 'A' && c <= 'Z') v = c - 'A';
    else if (c >= 'a' && c <= 'z') v = c - 'a' + 26;
    else if (c >= '0' && c <= '9') v = c - '0' + 52;
    else if (c == '+' || c == '-') v = 62;
    else if (c == '/' || c == '_') v = 63;
    else if (c == '=' || c == '\n' || c == '\r') continue;
    else return false;
    acc = (acc << 6) | (uint32_t)v;
    bits += 6;
    if (bits >= 8) {
      bits -= 8;
      out.push_back((uint8_t)((acc >> bits) & 255u));
    }
  }
  return true;
}

// ------------------------------------------------------------------------------------------------ quaternions
struct Quat {
  float x = 0, y = 0, z = 0, w = 1;
};
inline float q_dot(Quat a, Quat b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
inline Quat q_scale(Quat a, float s) { return Quat{a.x * s, a.y * s, a.z * s, a.w * s}; }
inline Quat q_add(Quat a, Quat b) { return Quat{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline Quat q_sub(Quat a, Quat b) { return Quat{a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
inline Quat q_normalize(Quat a) { return q_scale(a, 1.0f / std::sqrt(q_dot(a, a))); }  // glam Quat::normalize
// glam's acos_approx (the DirectXMath XMScalarAcos polynomial) as used by Quat::slerp
inline float acos_approx(float v) {
  const bool nonneg = v >= 0.0f;
  const float x = std::fabs(v);
  float omx = 1.0f - x;
  if (omx < 0.0f) omx = 0.0f;
  const float root = std::sqrt(omx);
  float r = ((((((-0.0012624911f * x + 0.0066700901f) * x - 0.0170881256f) * x + 0.0308918810f) * x - 0.0501743046f) * x +
              0.0889789874f) * x - 0.2145988016f) * x + 1.5707963050f;
  r *= root;
  return nonneg ? r : 3.14159265358979323846f - r;
}
// glam Quat::slerp: shortest arc, nlerp when the ends (almost) coincide
inline Quat q_slerp(Quat a, Quat b, float s) {
  float dot = q_dot(a, b);
  if (dot < 0.0f) {
    b = q_scale(b, -1.0f);
    dot = -dot;
  }
  if (dot > 1.0f - 1.1920929e-7f) return q_normalize(q_add(a, q_scale(q_sub(b, a), s)));
  const float theta = acos_approx(dot);
  const float s1 = std::sin(theta * (1.0f - s)), s2 = std::sin(theta * s), st = std::sin(theta);
  return q_scale(q_add(q_scale(a, s1), q_scale(b, s2)), 1.0f / st);
}
inline V3 v3_lerp(V3 a, V3 b, float s) { return a + (b - a) * s; }  // glam Vec3::lerp: self + (rhs - self) * s

// glam Mat4::from_scale_rotation_translation
inline M4 m4_from_srt(V3 s, Quat q, V3 t) {
  const float x2 = q.x + q.x, y2 = q.y + q.y, z2 = q.z + q.z;
  const float xx = q.x * x2, xy = q.x * y2, xz = q.x * z2, yy = q.y * y2, yz = q.y * z2, zz = q.z * z2;
  const float wx = q.w * x2, wy = q.w * y2, wz = q.w * z2;
  M4 m;
  m.c[0][0] = (1.0f - (yy + zz)) * s.x; m.c[0][1] = (xy + wz) * s.x; m.c[0][2] = (xz - wy) * s.x; m.c[0][3] = 0.0f;
  m.c[1][0] = (xy - wz) * s.y; m.c[1][1] = (1.0f - (xx + zz)) * s.y; m.c[1][2] = (yz + wx) * s.y; m.c[1][3] = 0.0f;
  m.c[2][0] = (xz + wy) * s.z; m.c[2][1] = (yz - wx) * s.z; m.c[2][2] = (1.0f - (xx + yy)) * s.z; m.c[2][3] = 0.0f;
  m.c[3][0] = t.x; m.c[3][1] = t.y; m.c[3][2] = t.z; m.c[3][3] = 1.0f;
  return m;
}

// ------------------------------------------------------------------------------------------------ scene graph
struct GNode {  // scene/node.rs:7-34
  std::string name;
  long parent = -1;
  std::vector<size_t> children;
  V3 translation = v3(0, 0, 0), scale = v3(1, 1, 1);
  Quat rotation;
};
struct GSkin {  // scene/node.rs:38-42
  std::vector<size_t> joints;
  std::vector<M4> inverse_bind;
};
struct GChannel {  // scene/animation.rs:11-25
  size_t target_node = 0;
  std::vector<float> inputs;
  int kind = 0;           // 0 translations, 1 rotations, 2 scales
  std::vector<float> out; // 3 or 4 floats per key (x3 keys for CUBICSPLINE)
  int interpolation = 0;  // 0 LINEAR, 1 STEP, 2 CUBICSPLINE
};
struct GAnimation {
  std::string name;
  std::vector<GChannel> channels;
  float duration = 0.0f;
};

// `node.transform().decomposed()` of the gltf crate: TRS members as given, or the matrix split into translation,
// per-axis scale (z carries the sign of the determinant) and the rotation of the normalised axes
inline void decompose_matrix(const float m[16], V3& t, Quat& r, V3& s) {
  t = v3(m[12], m[13], m[14]);
  V3 cx = v3(m[0], m[1], m[2]), cy = v3(m[4], m[5], m[6]), cz = v3(m[8], m[9], m[10]);
  const float det = dot(cx, cross(cy, cz));
  const float sx = length(cx), sy = length(cy), sz = (det < 0.0f ? -1.0f : 1.0f) * length(cz);
  s = v3(sx, sy, sz);
  cx = cx * (1.0f / sx);
  cy = cy * (1.0f / sy);
  cz = cz * (1.0f / sz);
  const float m00 = cx.x, m01 = cx.y, m02 = cx.z, m10 = cy.x, m11 = cy.y, m12 = cy.z, m20 = cz.x, m21 = cz.y, m22 = cz.z;
  const float trace = m00 + m11 + m22;
  if (trace >= 0.0f) {
    float q = std::sqrt(1.0f + trace);
    r.w = 0.5f * q;
    q = 0.5f / q;
    r.x = (m12 - m21) * q;
    r.y = (m20 - m02) * q;
    r.z = (m01 - m10) * q;
  } else if (m00 > m11 && m00 > m22) {
    float q = std::sqrt((m00 - m11 - m22) + 1.0f);
    r.x = 0.5f * q;
    q = 0.5f / q;
    r.y = (m10 + m01) * q;
    r.z = (m02 + m20) * q;
    r.w = (m12 - m21) * q;
  } else if (m11 > m22) {
    float q = std::sqrt((m11 - m00 - m22) + 1.0f);
    r.y = 0.5f * q;
    q = 0.5f / q;
    r.z = (m21 + m12) * q;
    r.x = (m10 + m01) * q;
    r.w = (m20 - m02) * q;
  } else {
    float q = std::sqrt((m22 - m00 - m11) + 1.0f);
    r.z = 0.5f * q;
    q = 0.5f / q;
    r.x = (m02 + m20) * q;
    r.y = (m21 + m12) * q;
    r.w = (m01 - m10) * q;
  }
}

// ---------------------------------------------------------------------------------------------------- document
struct GltfDoc {
  JVal root;
  std::vector<std::vector<uint8_t>> buffers;
  std::string err;

  bool fail(const std::string& m) {
    if (err.empty()) err = m;
    return false;
  }
  const JVal* item(const char* array, long i) const {
    const JVal* a = root.get(array);
    return (a && i >= 0) ? a->at((size_t)i) : nullptr;
  }
  // bytes of a buffer view (+ its stride, 0 = tightly packed)
  bool view(long index, const uint8_t*& ptr, size_t& len, size_t& stride) {
    const JVal* v = item("bufferViews", index);
    if (!v) return fail("gltf: bufferView index out of range");
    const long b = v->index("buffer");
    if (b < 0 || (size_t)b >= buffers.size()) return fail("gltf: buffer index out of range");
    const double off = v->number("byteOffset", 0.0), n = v->number("byteLength", -1.0);
    if (off < 0 || n < 0 || off + n > (double)buffers[(size_t)b].size()) return fail("gltf: bufferView outside its buffer");
    ptr = buffers[(size_t)b].data() + (size_t)off;
    len = (size_t)n;
    stride = (size_t)v->number("byteStride", 0.0);
    return true;
  }
  static int components(const std::string& type) {
    if (type == "SCALAR") return 1;
    if (type == "VEC2") return 2;
    if (type == "VEC3") return 3;
    if (type == "VEC4") return 4;
    if (type == "MAT2") return 4;
    if (type == "MAT3") return 9;
    if (type == "MAT4") return 16;
    return 0;
  }
  static size_t comp_size(long ct) { return (ct == 5120 || ct == 5121) ? 1 : (ct == 5122 || ct == 5123) ? 2 : (ct == 5125 || ct == 5126) ? 4 : 0; }
  static double load(const uint8_t* p, long ct) {
    switch (ct) {
      case 5120: return (double)(int8_t)p[0];
      case 5121: return (double)p[0];
      case 5122: { int16_t v; std::memcpy(&v, p, 2); return (double)v; }
      case 5123: { uint16_t v; std::memcpy(&v, p, 2); return (double)v; }
      case 5125: { uint32_t v; std::memcpy(&v, p, 4); return (double)v; }
      default: { float v; std::memcpy(&v, p, 4); return (double)v; }
    }
  }
  // Every element of an accessor as doubles (ncomp per element); integers are returned raw, `ct` reports their type.
  bool accessor(long index, int want_comp, std::vector<double>& out, size_t& count, long& ct) {
    const JVal* a = item("accessors", index);
    if (!a) return fail("gltf: accessor index out of range");
    const int nc = components(a->string("type"));
    ct = a->index("componentType");
    const size_t cs = comp_size(ct);
    const double cnt = a->number("count", -1.0);
    if (nc == 0 || cs == 0 || cnt < 0 || cnt > 1e9) return fail("gltf: bad accessor");
    if (want_comp && nc != want_comp) return fail("gltf: accessor has the wrong type");
    count = (size_t)cnt;
    out.assign(count * (size_t)nc, 0.0);
    const long bv = a->index("bufferView");
    if (bv >= 0) {
      const uint8_t* p;
      size_t len, stride;
      if (!view(bv, p, len, stride)) return false;
      const size_t off = (size_t)a->number("byteOffset", 0.0), elem = cs * (size_t)nc;
      if (stride == 0) stride = elem;
      if (count && off + (count - 1) * stride + elem > len) return fail("gltf: accessor outside its bufferView");
      for (size_t i = 0; i < count; i++)
        for (int c = 0; c < nc; c++) out[i * (size_t)nc + (size_t)c] = load(p + off + i * stride + (size_t)c * cs, ct);
    }
    if (const JVal* sp = a->get("sparse")) {  // glTF 2.0 §3.6.2.4: replace `count` elements at the given indices
      const size_t n = (size_t)sp->number("count", 0.0);
      const JVal* si = sp->get("indices");
      const JVal* sv = sp->get("values");
      if (!si || !sv) return fail("gltf: bad sparse accessor");
      const uint8_t *ip, *vp;
      size_t il, vl, st;
      if (!view(si->index("bufferView"), ip, il, st) || !view(sv->index("bufferView"), vp, vl, st)) return false;
      const long ict = si->index("componentType");
      const size_t ics = comp_size(ict), ioff = (size_t)si->number("byteOffset", 0.0), voff = (size_t)sv->number("byteOffset", 0.0);
      if (ics == 0 || ioff + n * ics > il || voff + n * cs * (size_t)nc > vl) return fail("gltf: sparse accessor outside its bufferView");
      for (size_t k = 0; k < n; k++) {
        const size_t idx = (size_t)load(ip + ioff + k * ics, ict);
        if (idx >= count) return fail("gltf: sparse index out of range");
        for (int c = 0; c < nc; c++)
          out[idx * (size_t)nc + (size_t)c] = load(vp + voff + (k * (size_t)nc + (size_t)c) * cs, ct);
      }
    }
    return true;
  }
  // `into_f32()` of the gltf crate's readers: floats as they are, normalised integers per glTF 2.0 §3.6.2.2
  static float to_f32(double v, long ct) {
    switch (ct) {
      case 5120: return std::max((float)v / 127.0f, -1.0f);
      case 5121: return (float)v / 255.0f;
      case 5122: return std::max((float)v / 32767.0f, -1.0f);
      case 5123: return (float)v / 65535.0f;
      default: return (float)v;
    }
  }
  bool floats(long index, int ncomp, std::vector<float>& out, size_t& count) {
    std::vector<double> d;
    long ct;
    if (!accessor(index, ncomp, d, count, ct)) return false;
    out.resize(d.size());
    for (size_t i = 0; i < d.size(); i++) out[i] = to_f32(d[i], ct);
    return true;
  }
  bool uints(long index, int ncomp, std::vector<uint32_t>& out, size_t& count) {
    std::vector<double> d;
    long ct;
    if (!accessor(index, ncomp, d, count, ct)) return false;
    if (ct == 5126) return fail("gltf: float accessor where integers are required");
    out.resize(d.size());
    for (size_t i = 0; i < d.size(); i++) out[i] = (uint32_t)d[i];
    return true;
  }

  bool parse(const uint8_t* data, size_t size) {
    const char* json = (const char*)data;
    size_t json_len = size;
    const uint8_t* bin = nullptr;
    size_t bin_len = 0;
    if (size >= 12 && !std::memcmp(data, "glTF", 4)) {  // GLB container (glTF 2.0 §4.4)
      uint32_t version, total;
      std::memcpy(&version, data + 4, 4);
      std::memcpy(&total, data + 8, 4);
      if (version != 2 || total > size) return fail("glb: unsupported version or truncated file");
      size_t pos = 12;
      json = nullptr;
      while (pos + 8 <= total) {
        uint32_t len, type;
        std::memcpy(&len, data + pos, 4);
        std::memcpy(&type, data + pos + 4, 4);
        if (pos + 8 + (size_t)len > total) return fail("glb: chunk outside the file");
        if (type == 0x4e4f534au && !json) {
          json = (const char*)data + pos + 8;
          json_len = len;
        } else if (type == 0x004e4942u && !bin) {
          bin = data + pos + 8;
          bin_len = len;
        }
        pos += 8 + (size_t)len;
        pos = (pos + 3) & ~(size_t)3;
      }
      if (!json) return fail("glb: no JSON chunk");
    }
    JParser jp{json, json + json_len, std::string(), 0};
    if (!jp.value(root) || root.t != JVal::OBJ) return fail(jp.err.empty() ? "gltf: the document is not a JSON object" : jp.err);
    const JVal* asset = root.get("asset");
    if (!asset || asset->string("version").substr(0, 1) != "2") return fail("gltf: asset.version 2.x required");
    const JVal* bufs = root.get("buffers");
    for (size_t i = 0; bufs && i < bufs->size(); i++) {
      const JVal& b = bufs->arr[i];
      const std::string uri = b.string("uri");
      std::vector<uint8_t> bytes;
      if (uri.empty()) {
        if (i != 0 || !bin) return fail("gltf: buffer without uri and without a GLB BIN chunk");
        bytes.assign(bin, bin + bin_len);
      } else if (uri.compare(0, 5, "data:") == 0) {
        const size_t comma = uri.find(',');
        if (comma == std::string::npos || uri.find(";base64") == std::string::npos || !base64_decode(uri, comma + 1, bytes))
          return fail("gltf: unsupported data URI");
      } else {
        return fail("gltf: external buffer files are not available (import from a slice)");
      }
      if ((double)bytes.size() < b.number("byteLength", 0.0)) return fail("gltf: buffer shorter than its byteLength");
      buffers.push_back(std::move(bytes));
    }
    return true;
  }
};

struct GltfScene {  // what load_gltf appends to (loader.rs:7-15)
  std::vector<GNode> nodes;
  std::vector<GSkin> skins;
  std::vector<GAnimation> animations;
  std::vector<std::vector<uint8_t>> textures;  // encoded image bytes per glTF texture
};

// loader.rs:7-354. On failure nothing of `scene` / `g` that the caller relies on is left half-built: the reference
// ignores the error (`let _ = load_gltf(..)`, lib.rs:57-67) and keeps whatever was appended; so does the caller here.
inline bool load_gltf(SceneData& scene, GltfScene& g, const uint8_t* data, size_t size, std::string& err) {
  GltfDoc doc;
  if (!doc.parse(data, size)) {
    err = doc.err;
    return false;
  }
  const JVal& root = doc.root;
  // 0. textures, in glTF texture order (material indices refer to textures, not images): loader.rs:20-35
  const JVal* textures = root.get("textures");
  for (size_t i = 0; textures && i < textures->size(); i++) {
    std::vector<uint8_t> blob;
    const JVal* img = doc.item("images", textures->arr[i].index("source"));
    if (img) {
      const long bv = img->index("bufferView");
      const uint8_t* p;
      size_t len, stride;
      if (bv >= 0 && doc.view(bv, p, len, stride)) blob.assign(p, p + len);
      // URI images: the reference keeps an empty blob ("External ref or empty") -> white fallback layer
    }
    g.textures.push_back(std::move(blob));
  }
  // 1. nodes: loader.rs:37-63
  const JVal* nodes = root.get("nodes");
  const size_t n_nodes = nodes ? nodes->size() : 0;
  g.nodes.assign(n_nodes, GNode());
  for (size_t i = 0; i < n_nodes; i++) {
    const JVal& n = nodes->arr[i];
    GNode& o = g.nodes[i];
    o.name = n.string("name");
    const JVal* m = n.get("matrix");
    if (m && m->size() == 16) {
      float mm[16];
      for (int k = 0; k < 16; k++) mm[k] = (float)m->arr[(size_t)k].num;
      decompose_matrix(mm, o.translation, o.rotation, o.scale);
    } else {
      if (const JVal* t = n.get("translation"))
        if (t->size() == 3) o.translation = v3((float)t->arr[0].num, (float)t->arr[1].num, (float)t->arr[2].num);
      if (const JVal* r = n.get("rotation"))
        if (r->size() == 4) o.rotation = Quat{(float)r->arr[0].num, (float)r->arr[1].num, (float)r->arr[2].num, (float)r->arr[3].num};
      if (const JVal* s = n.get("scale"))
        if (s->size() == 3) o.scale = v3((float)s->arr[0].num, (float)s->arr[1].num, (float)s->arr[2].num);
    }
    if (const JVal* ch = n.get("children"))
      for (const JVal& c : ch->arr)
        if (c.t == JVal::NUM && c.num >= 0) o.children.push_back((size_t)c.num);
  }
  for (size_t i = 0; i < n_nodes; i++)
    for (size_t c : g.nodes[i].children)
      if (c < n_nodes) g.nodes[c].parent = (long)i;
  // 2. skins: loader.rs:68-82
  const JVal* skins = root.get("skins");
  for (size_t i = 0; skins && i < skins->size(); i++) {
    const JVal& s = skins->arr[i];
    GSkin sk;
    if (const JVal* j = s.get("joints"))
      for (const JVal& v : j->arr) sk.joints.push_back((size_t)v.num);
    const long ibm = s.index("inverseBindMatrices");
    std::vector<float> f;
    size_t count = 0;
    if (ibm >= 0 && doc.floats(ibm, 16, f, count)) {
      for (size_t k = 0; k < count; k++) {
        M4 m;
        std::memcpy(m.c, &f[k * 16], 64);
        sk.inverse_bind.push_back(m);
      }
    } else {
      sk.inverse_bind.assign(sk.joints.size(), m4_identity());
    }
    g.skins.push_back(std::move(sk));
  }
  // 3. meshes -> geometries: loader.rs:84-235
  const JVal* meshes = root.get("meshes");
  std::vector<std::vector<size_t>> mesh_geos(meshes ? meshes->size() : 0);
  for (size_t mi = 0; meshes && mi < meshes->size(); mi++) {
    const JVal* prims = meshes->arr[mi].get("primitives");
    for (size_t pi = 0; prims && pi < prims->size(); pi++) {
      const JVal& prim = prims->arr[pi];
      const JVal* attrs = prim.get("attributes");
      if (!attrs) continue;
      const long mode = prim.index("mode");
      if (mode >= 0 && mode != 4) continue;  // the reference consumes index triples: triangle lists only
      std::vector<float> pos, nrm, uv, wts;
      std::vector<uint32_t> idx, jnt;
      size_t n_pos = 0, n = 0;
      const long a_pos = attrs->index("POSITION");
      if (a_pos < 0 || !doc.floats(a_pos, 3, pos, n_pos) || n_pos == 0) continue;  // "if positions.is_empty() continue"
      const long a_n = attrs->index("NORMAL"), a_uv = attrs->index("TEXCOORD_0"), a_j = attrs->index("JOINTS_0"),
                 a_w = attrs->index("WEIGHTS_0"), a_i = prim.index("indices");
      if (a_n < 0 || !doc.floats(a_n, 3, nrm, n) || n != n_pos) {
        nrm.assign(n_pos * 3, 0.0f);
        for (size_t k = 0; k < n_pos; k++) nrm[k * 3 + 1] = 1.0f;  // default normal (0, 1, 0)
      }
      if (a_uv < 0 || !doc.floats(a_uv, 2, uv, n) || n != n_pos) uv.assign(n_pos * 2, 0.0f);
      if (a_i < 0 || !doc.uints(a_i, 1, idx, n)) {
        idx.resize(n_pos);
        for (size_t k = 0; k < n_pos; k++) idx[k] = (uint32_t)k;
      }
      if (a_j < 0 || !doc.uints(a_j, 4, jnt, n) || n != n_pos) jnt.assign(n_pos * 4, 0u);
      if (a_w < 0 || !doc.floats(a_w, 4, wts, n) || n != n_pos) wts.assign(n_pos * 4, 0.0f);
      for (uint32_t& i : idx)
        if (i >= n_pos) i = 0;  // robustness: the crate's reader would hand the index through; never index out of range
      // material: loader.rs:136-178; a primitive without one gets glTF's default material
      V3 col = v3(1, 1, 1), emis = v3(0, 0, 0);
      float metallic = 1.0f, roughness = 1.0f, tex[4] = {-1.0f, -1.0f, -1.0f, -1.0f}, occl = -1.0f;
      if (const JVal* mat = doc.item("materials", prim.index("material"))) {
        auto tex_index = [](const JVal* info) -> float {
          const long t = info ? info->index("index") : -1;
          return t >= 0 ? (float)t : -1.0f;
        };
        if (const JVal* pbr = mat->get("pbrMetallicRoughness")) {
          if (const JVal* bc = pbr->get("baseColorFactor"))
            if (bc->size() >= 3) col = v3((float)bc->arr[0].num, (float)bc->arr[1].num, (float)bc->arr[2].num);
          metallic = (float)pbr->number("metallicFactor", 1.0);
          roughness = (float)pbr->number("roughnessFactor", 1.0);
          tex[0] = tex_index(pbr->get("baseColorTexture"));
          tex[1] = tex_index(pbr->get("metallicRoughnessTexture"));
        }
        if (const JVal* e = mat->get("emissiveFactor"))
          if (e->size() == 3) emis = v3((float)e->arr[0].num, (float)e->arr[1].num, (float)e->arr[2].num);
        tex[2] = tex_index(mat->get("normalTexture"));
        tex[3] = tex_index(mat->get("emissiveTexture"));
        occl = tex_index(mat->get("occlusionTexture"));
      }
      uint32_t mat_type = LAMBERTIAN;
      if (metallic > 0.0f) mat_type = METAL;
      if (dot(emis, emis) > 1e-4f) mat_type = LIGHT;
      Geometry geom;
      for (size_t k = 0; k < n_pos; k++) {
        geom.push_vertex_skinned(v3(pos[k * 3], pos[k * 3 + 1], pos[k * 3 + 2]), v3(nrm[k * 3], nrm[k * 3 + 1], nrm[k * 3 + 2]),
                                 V2{uv[k * 2], uv[k * 2 + 1]}, &jnt[k * 4], &wts[k * 4]);
      }
      for (size_t k = 0; k + 2 < idx.size(); k += 3) {
        geom.indices.insert(geom.indices.end(), idx.begin() + (long)k, idx.begin() + (long)k + 3);
        geom.push_attributes(col, mat_type, metallic, roughness, 1.5f, emis, tex, occl);
      }
      mesh_geos[mi].push_back(scene.geometries.size());
      scene.geometries.push_back(std::move(geom));
    }
  }
  // 4. instances: one per (node with a mesh, primitive); a skinned mesh sits at identity, a static one gets the
  //    node's LOCAL transform (the reference says so itself: "This is LOCAL. If there's a parent, it's wrong"): loader.rs:240-300
  for (size_t i = 0; i < n_nodes; i++) {
    const JVal& n = nodes->arr[i];
    const long mesh = n.index("mesh");
    if (mesh < 0 || (size_t)mesh >= mesh_geos.size()) continue;
    const long skin = n.index("skin");
    for (size_t gi : mesh_geos[(size_t)mesh]) {
      if (skin >= 0 && gi < scene.geometries.size()) scene.geometries[gi].skin_index = skin;
      SceneInstance si;
      si.transform = skin >= 0 ? m4_identity() : m4_from_srt(g.nodes[i].scale, g.nodes[i].rotation, g.nodes[i].translation);
      si.geometry_index = gi;
      scene.instances.push_back(si);
    }
  }
  // 5. animations: loader.rs:305-351
  const JVal* anims = root.get("animations");
  for (size_t ai = 0; anims && ai < anims->size(); ai++) {
    const JVal& a = anims->arr[ai];
    GAnimation out;
    out.name = a.get("name") ? a.string("name") : std::string("anim");
    const JVal* channels = a.get("channels");
    const JVal* samplers = a.get("samplers");
    for (size_t ci = 0; channels && ci < channels->size(); ci++) {
      const JVal& ch = channels->arr[ci];
      const JVal* target = ch.get("target");
      const JVal* smp = samplers ? samplers->at((size_t)std::max(0L, ch.index("sampler"))) : nullptr;
      if (!target || !smp || target->index("node") < 0) continue;
      GChannel c;
      c.target_node = (size_t)target->index("node");
      const std::string path = target->string("path"), interp = smp->string("interpolation");
      c.interpolation = interp == "STEP" ? 1 : interp == "CUBICSPLINE" ? 2 : 0;
      if (path == "translation") c.kind = 0;
      else if (path == "rotation") c.kind = 1;
      else if (path == "scale") c.kind = 2;
      else continue;  // morph target weights: `_ => continue`
      size_t n_in = 0, n_out = 0;
      if (!doc.floats(smp->index("input"), 1, c.inputs, n_in)) continue;
      if (!doc.floats(smp->index("output"), c.kind == 1 ? 4 : 3, c.out, n_out)) continue;
      out.channels.push_back(std::move(c));
    }
    float mx = 0.0f;  // fold(NaN, max) over the channels' last keys, then .max(0.0)
    for (const GChannel& c : out.channels)
      if (!c.inputs.empty() && c.inputs.back() > mx) mx = c.inputs.back();
    out.duration = mx;
    g.animations.push_back(std::move(out));
  }
  return true;
}
