/* addon.c — N-API (v8, plain C) shim over the C ABI of include/mi355rt.h and include/mi355scene.h.
 *
 * The reference's host code is TypeScript running in a browser; its WebGPURenderer talks to WebGPU.
 * Here a Node process loads this addon and index.js rebuilds the same class surface on top of it
 * (src/renderer/WebGPURenderer.ts:7-138, src/world-bridge.ts:172-205).  Every function is a thin
 * argument-unpacking wrapper: no rendering logic lives in this file.
 * Build: gcc -shared -fPIC -I/usr/include/node addon.c -L../lib -lmi355rt -lmi355scene -lmi355tex -Wl,-rpath,'$ORIGIN/../lib'
 */
#include <node_api.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "mi355rt.h"
#include "mi355scene.h"
#include "mi355tex.h"

#define NAPI_OK(env, call)                                              \
  do {                                                                  \
    if ((call) != napi_ok) {                                            \
      napi_throw_error((env), NULL, "N-API call failed: " #call);       \
      return NULL;                                                      \
    }                                                                   \
  } while (0)

static napi_value make_int(napi_env env, int v) {
  napi_value r;
  napi_create_int32(env, v, &r);
  return r;
}
static int get_args(napi_env env, napi_callback_info info, size_t want, napi_value* argv) {
  size_t argc = want;
  if (napi_get_cb_info(env, info, &argc, argv, NULL, NULL) != napi_ok || argc < want) {
    napi_throw_type_error(env, NULL, "wrong number of arguments");
    return 0;
  }
  return 1;
}
static void* get_ptr(napi_env env, napi_value v) {
  void* p = NULL;
  if (napi_get_value_external(env, v, &p) != napi_ok) napi_throw_type_error(env, NULL, "expected a native handle");
  return p;
}
static uint32_t get_u32(napi_env env, napi_value v) {
  uint32_t x = 0;
  napi_get_value_uint32(env, v, &x);
  return x;
}
/* TypedArray / ArrayBuffer view -> (pointer, byte length) */
static int get_bytes(napi_env env, napi_value v, void** data, size_t* bytes) {
  bool is_ta = false;
  napi_is_typedarray(env, v, &is_ta);
  if (is_ta) {
    napi_typedarray_type t;
    size_t len = 0, off = 0;
    napi_value ab;
    if (napi_get_typedarray_info(env, v, &t, &len, data, &ab, &off) != napi_ok) return 0;
    size_t esz = (t == napi_int8_array || t == napi_uint8_array || t == napi_uint8_clamped_array) ? 1
                 : (t == napi_int16_array || t == napi_uint16_array)                              ? 2
                 : (t == napi_float64_array)                                                      ? 8
                                                                                                   : 4;
    *bytes = len * esz;
    return 1;
  }
  napi_valuetype vt;
  napi_typeof(env, v, &vt);
  if (vt == napi_null || vt == napi_undefined) {
    *data = NULL;
    *bytes = 0;
    return 1;
  }
  napi_throw_type_error(env, NULL, "expected a TypedArray");
  return 0;
}

/* ------------------------------------------------------------------ renderer */
static napi_value rtCreate(napi_env env, napi_callback_info info) {
  napi_value a[1];
  if (!get_args(env, info, 1, a)) return NULL;
  rt_ctx* c = rt_create((int)get_u32(env, a[0]));
  if (!c) {
    napi_throw_error(env, NULL, rt_last_error(NULL));  /* init() throws in the reference (WebGPUContext.ts:15,19) */
    return NULL;
  }
  napi_value ext;
  NAPI_OK(env, napi_create_external(env, c, NULL, NULL, &ext));
  return ext;
}
static napi_value rtDestroy(napi_env env, napi_callback_info info) {
  napi_value a[1];
  if (!get_args(env, info, 1, a)) return NULL;
  rt_destroy((rt_ctx*)get_ptr(env, a[0]));
  return NULL;
}
static napi_value rtLastError(napi_env env, napi_callback_info info) {
  napi_value a[1], s;
  if (!get_args(env, info, 1, a)) return NULL;
  napi_create_string_utf8(env, rt_last_error((rt_ctx*)get_ptr(env, a[0])), NAPI_AUTO_LENGTH, &s);
  return s;
}
static napi_value rtSetPipeline(napi_env env, napi_callback_info info) {
  napi_value a[3];
  if (!get_args(env, info, 3, a)) return NULL;
  return make_int(env, rt_set_pipeline((rt_ctx*)get_ptr(env, a[0]), get_u32(env, a[1]), get_u32(env, a[2])));
}
static napi_value rtResize(napi_env env, napi_callback_info info) {
  napi_value a[3];
  if (!get_args(env, info, 3, a)) return NULL;
  return make_int(env, rt_resize((rt_ctx*)get_ptr(env, a[0]), get_u32(env, a[1]), get_u32(env, a[2])));
}
static napi_value rtResetAccum(napi_env env, napi_callback_info info) {
  napi_value a[1];
  if (!get_args(env, info, 1, a)) return NULL;
  return make_int(env, rt_reset_accum((rt_ctx*)get_ptr(env, a[0])));
}
static napi_value rtUploadTextures(napi_env env, napi_callback_info info) {
  napi_value a[3];
  void* p;
  size_t n;
  if (!get_args(env, info, 3, a) || !get_bytes(env, a[1], &p, &n)) return NULL;
  return make_int(env, rt_upload_textures((rt_ctx*)get_ptr(env, a[0]), (const uint8_t*)p, get_u32(env, a[2])));
}
static napi_value rtUpload(napi_env env, napi_callback_info info) {
  napi_value a[3];
  void* p;
  size_t n;
  if (!get_args(env, info, 3, a) || !get_bytes(env, a[2], &p, &n)) return NULL;
  return make_int(env, rt_upload((rt_ctx*)get_ptr(env, a[0]), (rt_kind)get_u32(env, a[1]), p, n));
}
static napi_value rtUploadGeometry(napi_env env, napi_callback_info info) {
  napi_value a[4];
  void *v, *nr, *uv;
  size_t nv, nn, nu;
  if (!get_args(env, info, 4, a) || !get_bytes(env, a[1], &v, &nv) || !get_bytes(env, a[2], &nr, &nn) ||
      !get_bytes(env, a[3], &uv, &nu))
    return NULL;
  uint32_t count = (uint32_t)(nv / 16);
  if (nn < (size_t)count * 16 || nu < (size_t)count * 8) {
    napi_throw_range_error(env, NULL, "normal / uv arrays shorter than the vertex array");
    return NULL;
  }
  return make_int(env, rt_upload_geometry((rt_ctx*)get_ptr(env, a[0]), (const float*)v, (const float*)nr,
                                          (const float*)uv, count));
}
static napi_value rtUploadBVH(napi_env env, napi_callback_info info) {
  napi_value a[3];
  void *t, *b;
  size_t nt, nb;
  if (!get_args(env, info, 3, a) || !get_bytes(env, a[1], &t, &nt) || !get_bytes(env, a[2], &b, &nb)) return NULL;
  return make_int(env, rt_upload_bvh((rt_ctx*)get_ptr(env, a[0]), (const float*)t, (uint32_t)(nt / 32),
                                     (const float*)b, (uint32_t)(nb / 32)));
}
static napi_value rtSetScene(napi_env env, napi_callback_info info) {
  napi_value a[4];
  void* cam;
  size_t n;
  if (!get_args(env, info, 4, a) || !get_bytes(env, a[1], &cam, &n)) return NULL;
  if (n < 96) {
    napi_throw_range_error(env, NULL, "cameraData must hold 24 floats");
    return NULL;
  }
  return make_int(env, rt_set_scene((rt_ctx*)get_ptr(env, a[0]), (const float*)cam, get_u32(env, a[2]), get_u32(env, a[3])));
}
static napi_value rtCompute(napi_env env, napi_callback_info info) {
  napi_value a[2];
  if (!get_args(env, info, 2, a)) return NULL;
  return make_int(env, rt_compute((rt_ctx*)get_ptr(env, a[0]), get_u32(env, a[1])));
}
static napi_value rtComputeBatch(napi_env env, napi_callback_info info) { /* Uint32Array of frame counts */
  napi_value a[2];
  void* p;
  size_t n;
  if (!get_args(env, info, 2, a) || !get_bytes(env, a[1], &p, &n)) return NULL;
  return make_int(env, rt_compute_batch((rt_ctx*)get_ptr(env, a[0]), (const uint32_t*)p, (uint32_t)(n / 4)));
}
static napi_value rtPresent(napi_env env, napi_callback_info info) {
  napi_value a[1];
  if (!get_args(env, info, 1, a)) return NULL;
  return make_int(env, rt_present((rt_ctx*)get_ptr(env, a[0])));
}
static napi_value rtSync(napi_env env, napi_callback_info info) {
  napi_value a[1];
  if (!get_args(env, info, 1, a)) return NULL;
  return make_int(env, rt_sync((rt_ctx*)get_ptr(env, a[0])));
}
static napi_value rtCapture(napi_env env, napi_callback_info info) {
  napi_value a[2];
  void* p;
  size_t n;
  if (!get_args(env, info, 2, a) || !get_bytes(env, a[1], &p, &n)) return NULL;
  return make_int(env, rt_capture((rt_ctx*)get_ptr(env, a[0]), (uint8_t*)p, n));
}
static napi_value rtReadAccum(napi_env env, napi_callback_info info) {
  napi_value a[2];
  void* p;
  size_t n;
  if (!get_args(env, info, 2, a) || !get_bytes(env, a[1], &p, &n)) return NULL;
  return make_int(env, rt_read_accum((rt_ctx*)get_ptr(env, a[0]), (float*)p, n));
}
static napi_value rtGetCounters(napi_env env, napi_callback_info info) {
  napi_value a[1], arr;
  if (!get_args(env, info, 1, a)) return NULL;
  rt_counters c;
  memset(&c, 0, sizeof(c));
  rt_get_counters((rt_ctx*)get_ptr(env, a[0]), &c);
  const uint64_t v[6] = {c.primary_rays, c.extension_rays, c.shadow_rays, c.nodes_visited, c.tris_tested, c.shaded_hits};
  NAPI_OK(env, napi_create_array_with_length(env, 6, &arr));
  for (uint32_t i = 0; i < 6; i++) {
    napi_value d;
    napi_create_double(env, (double)v[i], &d);
    napi_set_element(env, arr, i, d);
  }
  return arr;
}

/* --------------------------------------------------------------------- world */
/* (sceneName, objSource | null, glbData | null) — World::new, lib.rs:45-102 */
static napi_value msCreate(napi_env env, napi_callback_info info) {
  napi_value a[3];
  void* glb = NULL;
  size_t glb_size = 0;
  if (!get_args(env, info, 3, a) || !get_bytes(env, a[2], &glb, &glb_size)) return NULL;
  char name[64];
  size_t len = 0;
  napi_get_value_string_utf8(env, a[0], name, sizeof(name), &len);
  char* obj = NULL;
  napi_valuetype vt;
  napi_typeof(env, a[1], &vt);
  if (vt == napi_string) {
    size_t olen = 0;
    napi_get_value_string_utf8(env, a[1], NULL, 0, &olen);
    obj = (char*)malloc(olen + 1);
    napi_get_value_string_utf8(env, a[1], obj, olen + 1, &olen);
  }
  ms_world* w = glb ? ms_world_create_glb(name, obj, (const uint8_t*)glb, glb_size) : ms_world_create(name, obj);
  free(obj);
  if (!w) {
    napi_throw_error(env, NULL, ms_last_error());
    return NULL;
  }
  napi_value ext;
  NAPI_OK(env, napi_create_external(env, w, NULL, NULL, &ext));
  return ext;
}
static napi_value msDestroy(napi_env env, napi_callback_info info) {
  napi_value a[1];
  if (!get_args(env, info, 1, a)) return NULL;
  ms_world_destroy((ms_world*)get_ptr(env, a[0]));
  return NULL;
}
static napi_value msUpdate(napi_env env, napi_callback_info info) {
  napi_value a[2];
  double t = 0;
  if (!get_args(env, info, 2, a)) return NULL;
  napi_get_value_double(env, a[1], &t);
  ms_world_update((ms_world*)get_ptr(env, a[0]), (float)t);
  return NULL;
}
static napi_value msUpdateCamera(napi_env env, napi_callback_info info) {
  napi_value a[3];
  double w = 0, h = 0;
  if (!get_args(env, info, 3, a)) return NULL;
  napi_get_value_double(env, a[1], &w);
  napi_get_value_double(env, a[2], &h);
  ms_world_update_camera((ms_world*)get_ptr(env, a[0]), (float)w, (float)h);
  return NULL;
}
/* msGet(world, name) -> Float32Array | Uint32Array copy (the worker also copies out of WASM memory,
 * src/worker/wasm-worker.ts:21-91) */
static napi_value msGet(napi_env env, napi_callback_info info) {
  napi_value a[2];
  if (!get_args(env, info, 2, a)) return NULL;
  const ms_world* w = (const ms_world*)get_ptr(env, a[0]);
  char name[32];
  size_t len = 0, n = 0;
  napi_get_value_string_utf8(env, a[1], name, sizeof(name), &len);
  const void* src = NULL;
  int is_u32 = 0;
  if (!strcmp(name, "vertices")) src = ms_world_vertices(w, &n);
  else if (!strcmp(name, "normals")) src = ms_world_normals(w, &n);
  else if (!strcmp(name, "uvs")) src = ms_world_uvs(w, &n);
  else if (!strcmp(name, "tlas")) src = ms_world_tlas(w, &n);
  else if (!strcmp(name, "blas")) src = ms_world_blas(w, &n);
  else if (!strcmp(name, "instances")) src = ms_world_instances(w, &n);
  else if (!strcmp(name, "camera")) src = ms_world_camera(w, &n);
  else if (!strcmp(name, "mesh_topology")) { src = ms_world_mesh_topology(w, &n); is_u32 = 1; }
  else if (!strcmp(name, "lights")) { src = ms_world_lights(w, &n); is_u32 = 1; }
  else if (!strcmp(name, "draw_commands")) { src = ms_world_draw_commands(w, &n); is_u32 = 1; }
  else {
    napi_throw_error(env, NULL, "unknown world array");
    return NULL;
  }
  void* dst = NULL;
  napi_value ab, ta;
  NAPI_OK(env, napi_create_arraybuffer(env, n * 4, &dst, &ab));
  if (n) memcpy(dst, src, n * 4);
  NAPI_OK(env, napi_create_typedarray(env, is_u32 ? napi_uint32_array : napi_float32_array, n, ab, 0, &ta));
  return ta;
}
static napi_value rtAllocTextureLayers(napi_env env, napi_callback_info info) {
  napi_value a[2];
  if (!get_args(env, info, 2, a)) return NULL;
  return make_int(env, rt_alloc_texture_layers((rt_ctx*)get_ptr(env, a[0]), get_u32(env, a[1])));
}
/* (ctx, layer, rgba | null, width, height) */
static napi_value rtUploadTextureImage(napi_env env, napi_callback_info info) {
  napi_value a[5];
  void* p = NULL;
  size_t n = 0;
  if (!get_args(env, info, 5, a) || !get_bytes(env, a[2], &p, &n)) return NULL;
  const uint32_t w = get_u32(env, a[3]), h = get_u32(env, a[4]);
  if (p && n < (size_t)w * h * 4) {
    napi_throw_error(env, NULL, "rtUploadTextureImage: buffer smaller than width * height * 4");
    return NULL;
  }
  return make_int(env, rt_upload_texture_image((rt_ctx*)get_ptr(env, a[0]), get_u32(env, a[1]), (const uint8_t*)p, w, h));
}
/* (encoded bytes) -> {data: Uint8Array, width, height}; throws with mt_last_error() when the blob does not decode */
static napi_value mtDecode(napi_env env, napi_callback_info info) {
  napi_value a[1];
  void* p = NULL;
  size_t n = 0;
  if (!get_args(env, info, 1, a) || !get_bytes(env, a[0], &p, &n)) return NULL;
  mt_image img;
  if (mt_decode((const uint8_t*)p, n, &img) != MT_OK) {
    napi_throw_error(env, NULL, mt_last_error());
    return NULL;
  }
  const size_t bytes = (size_t)img.width * img.height * 4;
  void* dst = NULL;
  napi_value ab, ta, obj, w, h;
  if (napi_create_arraybuffer(env, bytes, &dst, &ab) != napi_ok) {
    mt_free(&img);
    napi_throw_error(env, NULL, "mtDecode: allocation failed");
    return NULL;
  }
  memcpy(dst, img.rgba, bytes);
  const uint32_t iw = img.width, ih = img.height;
  mt_free(&img);
  NAPI_OK(env, napi_create_typedarray(env, napi_uint8_array, bytes, ab, 0, &ta));
  NAPI_OK(env, napi_create_object(env, &obj));
  NAPI_OK(env, napi_create_uint32(env, iw, &w));
  NAPI_OK(env, napi_create_uint32(env, ih, &h));
  NAPI_OK(env, napi_set_named_property(env, obj, "data", ta));
  NAPI_OK(env, napi_set_named_property(env, obj, "width", w));
  NAPI_OK(env, napi_set_named_property(env, obj, "height", h));
  return obj;
}
/* animations: get_animation_count/name, set_animation, load_animation_glb (lib.rs:106-147) */
static napi_value msAnimationNames(napi_env env, napi_callback_info info) {
  napi_value a[1], arr;
  if (!get_args(env, info, 1, a)) return NULL;
  const ms_world* w = (const ms_world*)get_ptr(env, a[0]);
  const size_t n = ms_world_animation_count(w);
  NAPI_OK(env, napi_create_array_with_length(env, n, &arr));
  for (size_t i = 0; i < n; i++) {
    napi_value s;
    NAPI_OK(env, napi_create_string_utf8(env, ms_world_animation_name(w, i), NAPI_AUTO_LENGTH, &s));
    NAPI_OK(env, napi_set_element(env, arr, (uint32_t)i, s));
  }
  return arr;
}
static napi_value msSetAnimation(napi_env env, napi_callback_info info) {
  napi_value a[2];
  if (!get_args(env, info, 2, a)) return NULL;
  ms_world_set_animation((ms_world*)get_ptr(env, a[0]), get_u32(env, a[1]));
  return NULL;
}
static napi_value msLoadAnimation(napi_env env, napi_callback_info info) {
  napi_value a[2];
  void* p = NULL;
  size_t n = 0;
  if (!get_args(env, info, 2, a) || !get_bytes(env, a[1], &p, &n)) return NULL;
  return make_int(env, ms_world_load_animation_glb((ms_world*)get_ptr(env, a[0]), (const uint8_t*)p, n));
}
/* (world, ctx | null): World::update(t) builds its BLASes with rt_build_blas (GPU) instead of the CPU builder */
static napi_value msSetBlasBuilder(napi_env env, napi_callback_info info) {
  napi_value a[2];
  if (!get_args(env, info, 2, a)) return NULL;
  napi_valuetype vt;
  napi_typeof(env, a[1], &vt);
  void* ctx = vt == napi_external ? get_ptr(env, a[1]) : NULL;
  ms_world_set_blas_builder((ms_world*)get_ptr(env, a[0]), ctx ? (ms_blas_builder)rt_build_blas : NULL, ctx);
  return NULL;
}
/* (world, ctx | null): the whole per-frame half of World::update(t) runs on the GPU inside the renderer's buffers
 * (rt_world_update); the host arrays are then not refreshed and nothing needs uploading */
static napi_value msSetDeviceUpdater(napi_env env, napi_callback_info info) {
  napi_value a[2];
  if (!get_args(env, info, 2, a)) return NULL;
  napi_valuetype vt;
  napi_typeof(env, a[1], &vt);
  void* ctx = vt == napi_external ? get_ptr(env, a[1]) : NULL;
  ms_world_set_device_updater((ms_world*)get_ptr(env, a[0]), ctx ? (ms_device_updater)rt_world_update : NULL, ctx);
  return NULL;
}
static napi_value msDeviceResident(napi_env env, napi_callback_info info) {
  napi_value a[1];
  if (!get_args(env, info, 1, a)) return NULL;
  return make_int(env, ms_world_device_resident((const ms_world*)get_ptr(env, a[0])));
}
static napi_value msLastError(napi_env env, napi_callback_info info) {
  napi_value s;
  (void)info;
  NAPI_OK(env, napi_create_string_utf8(env, ms_last_error(), NAPI_AUTO_LENGTH, &s));
  return s;
}
static napi_value msEncodedTextureCount(napi_env env, napi_callback_info info) {
  napi_value a[1];
  if (!get_args(env, info, 1, a)) return NULL;
  return make_int(env, (int)ms_world_encoded_texture_count((const ms_world*)get_ptr(env, a[0])));
}
/* (world, index) -> Uint8Array of the encoded image, or undefined for an external / missing image */
static napi_value msEncodedTexture(napi_env env, napi_callback_info info) {
  napi_value a[2];
  if (!get_args(env, info, 2, a)) return NULL;
  size_t n = 0;
  const uint8_t* src = ms_world_encoded_texture((const ms_world*)get_ptr(env, a[0]), get_u32(env, a[1]), &n);
  if (!src || n == 0) return NULL;
  void* dst = NULL;
  napi_value ab, ta;
  NAPI_OK(env, napi_create_arraybuffer(env, n, &dst, &ab));
  memcpy(dst, src, n);
  NAPI_OK(env, napi_create_typedarray(env, napi_uint8_array, n, ab, 0, &ta));
  return ta;
}
static napi_value msTextureCount(napi_env env, napi_callback_info info) {
  napi_value a[1];
  if (!get_args(env, info, 1, a)) return NULL;
  return make_int(env, (int)ms_world_texture_count((const ms_world*)get_ptr(env, a[0])));
}
static napi_value msTexture(napi_env env, napi_callback_info info) {
  napi_value a[2];
  if (!get_args(env, info, 2, a)) return NULL;
  const uint8_t* src = ms_world_texture_rgba((const ms_world*)get_ptr(env, a[0]), get_u32(env, a[1]));
  if (!src) return NULL;
  void* dst = NULL;
  napi_value ab, ta;
  const size_t n = (size_t)1024 * 1024 * 4;
  NAPI_OK(env, napi_create_arraybuffer(env, n, &dst, &ab));
  memcpy(dst, src, n);
  NAPI_OK(env, napi_create_typedarray(env, napi_uint8_array, n, ab, 0, &ta));
  return ta;
}

static napi_value Init(napi_env env, napi_value exports) {
  static const struct {
    const char* name;
    napi_callback fn;
  } table[] = {{"rtCreate", rtCreate}, {"rtDestroy", rtDestroy}, {"rtLastError", rtLastError},
               {"rtSetPipeline", rtSetPipeline}, {"rtResize", rtResize}, {"rtResetAccum", rtResetAccum},
               {"rtUploadTextures", rtUploadTextures}, {"rtAllocTextureLayers", rtAllocTextureLayers},
               {"rtUploadTextureImage", rtUploadTextureImage}, {"mtDecode", mtDecode}, {"rtUpload", rtUpload}, {"rtUploadGeometry", rtUploadGeometry},
               {"rtUploadBVH", rtUploadBVH}, {"rtSetScene", rtSetScene}, {"rtCompute", rtCompute},
               {"rtComputeBatch", rtComputeBatch},
               {"rtPresent", rtPresent}, {"rtSync", rtSync}, {"rtCapture", rtCapture}, {"rtReadAccum", rtReadAccum},
               {"rtGetCounters", rtGetCounters}, {"msCreate", msCreate}, {"msDestroy", msDestroy},
               {"msUpdate", msUpdate}, {"msUpdateCamera", msUpdateCamera}, {"msGet", msGet},
               {"msTextureCount", msTextureCount}, {"msTexture", msTexture},
               {"msAnimationNames", msAnimationNames}, {"msSetAnimation", msSetAnimation},
               {"msLoadAnimation", msLoadAnimation}, {"msLastError", msLastError}, {"msSetBlasBuilder", msSetBlasBuilder}, {"msSetDeviceUpdater", msSetDeviceUpdater},
               {"msDeviceResident", msDeviceResident},
               {"msEncodedTextureCount", msEncodedTextureCount}, {"msEncodedTexture", msEncodedTexture}};
  for (size_t i = 0; i < sizeof(table) / sizeof(table[0]); i++) {
    napi_value fn;
    if (napi_create_function(env, table[i].name, NAPI_AUTO_LENGTH, table[i].fn, NULL, &fn) != napi_ok) return NULL;
    napi_set_named_property(env, exports, table[i].name, fn);
  }
  return exports;
}

NAPI_MODULE(NODE_GYP_MODULE_NAME, Init)
