/* mi355rt_layout.h — byte-exact layouts of the buffers that cross the drop-in
 * boundary.  Produced by the scene compiler
 * (rust-shader-tools/src/rebuilder.rs, bvh/blas.rs, bvh/tlas.rs, lib.rs in the reference),
 * consumed unchanged by the renderer.  Little-endian f32/u32 throughout.
 */
#ifndef MI355RT_LAYOUT_H
#define MI355RT_LAYOUT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Raytracer.wgsl:40-49 `MeshTopology`; rebuilder.rs:140-161. 80 bytes. */
typedef struct rt_topology {
  uint32_t v0, v1, v2; /* global vertex ids */
  uint32_t pad;        /* = geometry index (rebuilder.rs:155) */
  float data0[4];      /* base colour rgb, material type as float */
  float data1[4];      /* metallic, roughness, ior, 0 */
  float data2[4];      /* baseTex, metRoughTex, normalTex, emissiveTex (-1 = none) */
  float data3[4];      /* emissive rgb, occlusionTex */
} rt_topology;

/* Raytracer.wgsl:56-59 `BVHNode`; bvh/mod.rs:9-16. 32 bytes.
 * data == 0: internal (first child = curr+1); else leaf: first = data>>3, count = data&7.
 * skip: TLAS absolute, BLAS relative to the BLAS root. */
typedef struct rt_node {
  float min_b[3];
  uint32_t skip;
  float max_b[3];
  uint32_t data;
} rt_node;

/* Raytracer.wgsl:61-74 `Instance`; bvh/mod.rs:18-27. 144 bytes. */
typedef struct rt_instance {
  float transform[16]; /* column-major */
  float inverse[16];   /* column-major */
  uint32_t blas_node_offset; /* node units, relative to blas_base_idx */
  uint32_t attr_offset;
  uint32_t instance_id; /* geometry index */
  uint32_t pad;
} rt_instance;

/* Raytracer.wgsl:51-54 `LightRef`. 8 bytes. */
typedef struct rt_light_ref {
  uint32_t inst_idx;
  uint32_t tri_idx;
} rt_light_ref;

/* Raytracer.wgsl:16-38 `SceneUniforms`; ResourceManager.ts:63-67,374-403.
 * 240 bytes used of a 256-byte buffer. */
typedef struct rt_camera {
  float origin[4]; /* w = lens_radius */
  float lower_left[4];
  float horizontal[4];
  float vertical[4];
  float u[4];
  float v[4];
} rt_camera;

typedef struct rt_scene_uniforms {
  rt_camera camera;      /* @0   */
  rt_camera prev_camera; /* @96  */
  uint32_t frame_count;  /* @192 */
  uint32_t blas_base_idx;
  uint32_t vertex_count;
  uint32_t rand_seed; /* uploaded, never read by any shader */
  uint32_t light_count;
  uint32_t width;
  uint32_t height;
  uint32_t pad;
  float jitter[2];
  float average_jitter[2]; /* @232, ends @240 */
  uint32_t tail_pad[4];    /* buffer is 256 bytes (ResourceManager.ts:65) */
} rt_scene_uniforms;


/* ---- device-resident World::update(t): what the scene compiler hands the renderer so that the
 * per-frame half of World::update (lib.rs:149-270) runs on the GPU and the bridge arrays never leave HBM.
 *
 * STATIC part (changes only when `static_epoch` does: scene load / animation file load): per geometry the skinning
 * input of rebuilder.rs:36-91 (base positions / normals / uvs, joints, weights), its index list and its per-triangle
 * attribute rows (geometry.rs:6-25), and the instance list of lib.rs:194-230 in DECLARATION order with the transforms
 * the update would leave in place.  PER-FRAME part: the joint matrices global(joint) * inverse_bind of every skin
 * (rebuilder.rs:40-47) after the animation was sampled at t and the scene graph re-evaluated (lib.rs:149-184) - a few
 * hundred bytes.  Everything else of update(t) - skinning, BLAS build (bvh/blas.rs), topology / light / draw-command
 * packing (rebuilder.rs:121-168, lib.rs:237-270), TLAS (bvh/tlas.rs:58-111), instance packing - is derived from these
 * on the device by rt_world_update (include/mi355rt.h). */
typedef struct rt_world_geometry {
  const float* positions;     /* 3 f32 per vertex (base pose) */
  const float* normals;       /* 3 f32 per vertex */
  const float* uvs;           /* 2 f32 per vertex, n_uvs of them (vertices beyond get 0, 0) */
  const uint32_t* joints;     /* 4 per vertex */
  const float* weights;       /* 4 per vertex */
  const uint32_t* indices;    /* 3 per triangle, geometry-local vertex ids */
  const float* attributes;    /* 16 f32 per triangle (the 64 bytes after rt_topology.pad) */
  uint32_t n_verts, n_uvs, n_tris;
  int32_t skin;               /* index into the frame's skins, -1 = not skinned */
} rt_world_geometry;

typedef struct rt_world_frame {
  uint64_t static_epoch;               /* the static part below is unchanged while this is */
  uint32_t n_geometries, n_instances, n_skins, pad;
  const rt_world_geometry* geometries; /* n_geometries */
  const rt_instance* instances;        /* n_instances, declaration order; blas_node_offset is filled in on the device */
  const uint32_t* skin_first;          /* n_skins + 1: first joint matrix of each skin in joint_mats */
  const float* joint_mats;             /* 16 f32 each (column-major), this frame */
} rt_world_frame;

#ifdef __cplusplus
}
static_assert(sizeof(rt_topology) == 80, "MeshTopology is 80 bytes");
static_assert(sizeof(rt_node) == 32, "BVHNode is 32 bytes");
static_assert(sizeof(rt_instance) == 144, "Instance is 144 bytes");
static_assert(sizeof(rt_light_ref) == 8, "LightRef is 8 bytes");
static_assert(sizeof(rt_camera) == 96, "Camera is 96 bytes");
static_assert(sizeof(rt_scene_uniforms) == 256, "uniform buffer is 256 bytes");
static_assert(__builtin_offsetof(rt_scene_uniforms, frame_count) == 192, "mixed block at 192");
static_assert(__builtin_offsetof(rt_scene_uniforms, jitter) == 224, "jitter at 224");
#endif

#define RT_TEX_SIZE 1024 /* ResourceManager.ts:181-196: every layer is 1024x1024 rgba8unorm */

#endif
