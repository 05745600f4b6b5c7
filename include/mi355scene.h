/* mi355scene.h — C ABI of the native scene compiler.
 *
 * Replaces, for the path-tracing hot path only, the Rust->WASM `World` object of
 * the reference (rust-shader-tools/src/lib.rs:27-381) whose flat arrays the
 * TypeScript `WorldBridge` getters hand to the renderer
 * (src/world-bridge.ts:172-205, shapes in src/worker/protocol.ts:14-44).
 * Every getter returns a pointer into memory owned by the world, valid until
 * the next ms_world_update()/ms_world_update_camera()/ms_world_destroy(), plus
 * the element count in f32/u32 units exactly like World::*_ptr / *_len
 * (lib.rs:274-345).
 *
 * Scene names accepted by ms_world_create (scene/factory.rs:5-13):
 *   "cornell" (default for unknown names), "mixed", "special", "mesh", "viewer".
 *   "spheres" is refused: the reference seeds it from rand::rng() (helpers.rs:142-151)
 *   so it has no reproducible output.
 * Extensions (not in the reference; BASELINE.json configs 3-5, emitted in the
 * same bridge layout because World::update cannot express them, lib.rs:196-204):
 *   "instanced1000", "sponza_like", "glass_blob".
 */
#ifndef MI355SCENE_H
#define MI355SCENE_H

#include <stddef.h>
#include <stdint.h>

#include "mi355rt_layout.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ms_world ms_world;

/* World::new(scene_name, mesh_obj_source, glb_data) lib.rs:45-102.  obj_source may be
 * NULL.  glTF/GLB input is out of scope.  Returns NULL on error
 * (ms_last_error() holds the reason). */
ms_world* ms_world_create(const char* scene_name, const char* obj_source);
void ms_world_destroy(ms_world* w);
const char* ms_last_error(void);

/* World::update(time) lib.rs:149-271: rebuild BLAS + vertices, instances, TLAS,
 * lights and draw commands. (No animations exist without glTF; time is accepted
 * for signature parity.) */
void ms_world_update(ms_world* w, float time);

/* World::new(scene_name, mesh_obj_source, glb_data) with a glTF 2.0 binary (or JSON with base64 data URIs) — lib.rs:45-102
 * + loader.rs:7-354: geometries (one per mesh primitive), instances, nodes, skins, animations and
 * encoded textures are appended to the named procedural scene ("viewer": the Cornell-style room without the dummy
 * sphere).  A GLB that does not parse leaves the procedural scene alone, like the reference's `let _ = load_gltf(..)`;
 * ms_last_error() then says why.  ms_world_update(t) samples the active animation at t (looping), recomputes the
 * scene graph's global transforms, skins every skinned geometry on the CPU and rebuilds BLAS / TLAS (lib.rs:149-270,
 * rebuilder.rs:36-91). */
ms_world* ms_world_create_glb(const char* scene_name, const char* obj_source, const uint8_t* glb, size_t glb_size);
/* get_animation_count / get_animation_name / set_animation / load_animation_glb — lib.rs:106-147.
 * load_animation_glb appends the animations of another GLB (returns how many, or -1). */
size_t ms_world_animation_count(const ms_world* w);
const char* ms_world_animation_name(const ms_world* w, size_t index);
void ms_world_set_animation(ms_world* w, size_t index);
int ms_world_load_animation_glb(ms_world* w, const uint8_t* glb, size_t glb_size);
size_t ms_world_node_count(const ms_world* w);

/* Replace the BLAS builder World::update runs per geometry (BVHBuilder::build_with_ids, rebuilder.rs:93-98) with one
 * that returns the SAME nodes and triangle order — libmi355rt.so's rt_build_blas has this signature with its rt_ctx*
 * as `user` (the per-frame rebuild of an animated scene moves to the GPU).  fn == NULL restores the
 * CPU builder.  When the hook fails (< 0) the CPU builder does that geometry and ms_last_error() reports it after
 * ms_world_update. */
typedef int (*ms_blas_builder)(void* user, const float* verts4, uint32_t n_verts, const uint32_t* indices, uint32_t n_tris,
                               float* nodes_out, uint32_t nodes_cap, uint32_t* n_nodes_out, uint32_t* order_out);
void ms_world_set_blas_builder(ms_world* w, ms_blas_builder fn, void* user);
/* Move the whole per-frame half of World::update(t) to the device: with an updater set,
 * ms_world_update(t) samples the animation, re-evaluates the scene graph and computes the joint matrices (lib.rs:149-184,
 * rebuilder.rs:40-47) and hands them - with the static scene description, see rt_world_frame in mi355rt_layout.h - to
 * `fn`; libmi355rt.so's rt_world_update has this signature with its rt_ctx* as `user`.  The updater derives every
 * bridge array on the GPU, inside the renderer's own buffers: the host arrays behind the getters below are then NOT
 * refreshed (ms_world_device_resident() = 1; they keep the sizes and contents of the last host update) and nothing
 * needs re-uploading.  When the updater fails (< 0) this update runs on the host like without one
 * (ms_world_device_resident() = 0, ms_last_error() says why).  fn == NULL removes the updater. */
typedef int (*ms_device_updater)(void* user, const rt_world_frame* frame);
void ms_world_set_device_updater(ms_world* w, ms_device_updater fn, void* user);
int ms_world_device_resident(const ms_world* w);
/* The CPU builder itself, with the hook's signature minus `user`: BVHBuilder::new + build_with_ids of bvh/blas.rs:20-85
 * on one mesh (4 f32 per vertex, 3 u32 per triangle) -> 8 f32 per node {min.xyz, bits(skip)} {max.xyz, bits(data)} and
 * the triangle order.  0 on success, -1 on a bad argument.  tests/test_bvh_independent.py pins it with node arrays
 * worked out by hand from the Rust source. */
int ms_build_blas(const float* verts4, uint32_t n_verts, const uint32_t* indices, uint32_t n_tris, float* nodes_out,
                  uint32_t nodes_cap, uint32_t* n_nodes_out, uint32_t* order_out);
/* The CPU TLAS builder alone: TLASBuilder::new + build of bvh/tlas.rs:17-111 on n instances given by their BLAS root
 * boxes (6 f32: min.xyz, max.xyz, object space) and transforms (16 f32 column-major each; NULL = identity) -> 8 f32 per
 * node and the instance order (sorted_instances[k] = instance order_out[k]).  0 on success, -1 on a bad argument.
 * tests/test_bvh_independent.py pins it with node arrays worked out by hand from the Rust source. */
int ms_build_tlas(const float* boxes6, const float* transforms16, uint32_t n, float* nodes_out, uint32_t nodes_cap,
                  uint32_t* n_nodes_out, uint32_t* order_out);
/* World::update_camera(width, height) lib.rs:347-352. */
void ms_world_update_camera(ms_world* w, float width, float height);

/* World::*_ptr()/_len() pairs, lib.rs:274-345. *len = number of f32/u32 elements. */
const float* ms_world_vertices(const ms_world* w, size_t* len);         /* 4 f32 / vertex  */
const float* ms_world_normals(const ms_world* w, size_t* len);          /* 4 f32 / vertex  */
const float* ms_world_uvs(const ms_world* w, size_t* len);              /* 2 f32 / vertex  */
const uint32_t* ms_world_mesh_topology(const ms_world* w, size_t* len); /* 20 u32 / tri    */
const float* ms_world_tlas(const ms_world* w, size_t* len);             /* 8 f32 / node    */
const float* ms_world_blas(const ms_world* w, size_t* len);             /* 8 f32 / node    */
const float* ms_world_instances(const ms_world* w, size_t* len);        /* 36 f32 / inst   */
const uint32_t* ms_world_lights(const ms_world* w, size_t* len);        /* 2 u32 / light   */
const uint32_t* ms_world_draw_commands(const ms_world* w, size_t* len); /* 4 u32 / inst    */
const float* ms_world_camera(const ms_world* w, size_t* len);           /* 24 f32          */

/* World::get_texture_count lib.rs:355-357.  The reference hands out *encoded*
 * image bytes that the browser decodes and resizes to 1024x1024
 * (ResourceManager.ts:153-198).  Decoding is out of scope here; synthetic scenes
 * provide already-decoded RGBA8 1024x1024 layers instead. */
size_t ms_world_texture_count(const ms_world* w);
const uint8_t* ms_world_texture_rgba(const ms_world* w, size_t index); /* 1024*1024*4 bytes */
/* get_texture_count / get_texture_ptr / get_texture_size for a glTF input (lib.rs:350-369): the ENCODED image bytes of
 * each glTF texture, in texture order; size 0 = external or missing image (the renderer substitutes the white layer). */
size_t ms_world_encoded_texture_count(const ms_world* w);
const uint8_t* ms_world_encoded_texture(const ms_world* w, size_t index, size_t* size);

#ifdef __cplusplus
}
#endif
#endif
