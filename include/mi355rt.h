/* mi355rt.h — C ABI of the MI355X path-tracing renderer (libmi355rt.so).
 *
 * Drop-in boundary: these entry points are what an FFI binding of the reference's
 * `class WebGPURenderer` (src/renderer/WebGPURenderer.ts:7-138) would call; each one
 * cites the TypeScript method it replaces.  Plain pointers and sizes only.
 *
 * Ownership: the caller owns every input array; the callee copies during the call
 * (queue.writeBuffer semantics, ResourceManager.ts:282,318-320,340-341), so inputs may
 * be freed or mutated right after return.  The callee owns all device memory.
 * Threading: one HIP stream per context; calls on one context are not re-entrant;
 * rt_compute / rt_present only enqueue, rt_sync fences (device.queue.onSubmittedWorkDone).
 * Errors: int status, 0 = ok, > 0 = informational (e.g. "buffer was reallocated"),
 * < 0 = failure with rt_last_error(ctx) holding the message.  Like the reference's
 * passes (RaytracePass.ts:38-46,94; RasterizerPass.ts:55,97; PostProcessPass.ts:32-38,60),
 * rt_compute / rt_present silently skip (return RT_SKIPPED) while resources are missing.
 */
#ifndef MI355RT_H
#define MI355RT_H

#include <stddef.h>
#include <stdint.h>

#include "mi355rt_layout.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_ctx rt_ctx;

enum {
  RT_OK = 0,
  RT_REALLOCATED = 1, /* updateBuffer & co. return `needsRebind` */
  RT_SKIPPED = 2,     /* pass skipped: resources not ready */
  RT_ERR_INVALID = -1,
  RT_ERR_HIP = -2,
  RT_ERR_NOT_READY = -3,
  RT_ERR_NO_DEVICE = -4,
  RT_ERR_INTERNAL = -5
};

/* updateBuffer(type, data) kinds — WebGPURenderer.ts:55-60 */
typedef enum rt_kind {
  RT_KIND_TOPOLOGY = 0,     /* Uint32Array, 20 u32 per triangle  */
  RT_KIND_INSTANCE = 1,     /* Float32Array, 36 f32 per instance */
  RT_KIND_LIGHTS = 2,       /* Uint32Array, 2 u32 per light      */
  RT_KIND_DRAW_COMMANDS = 3 /* Uint32Array, 4 u32 per instance   */
} rt_kind;

/* device counters (SURVEY.md §8d "Ray definition") */
typedef struct rt_counters {
  uint64_t primary_rays;   /* primary-visibility casts                        */
  uint64_t extension_rays; /* intersect_tlas calls,        Raytracer.wgsl:732 */
  uint64_t shadow_rays;    /* intersect_tlas_shadow calls, Raytracer.wgsl:688 */
  uint64_t nodes_visited;  /* BVH node fetches (TLAS + BLAS)                  */
  uint64_t tris_tested;    /* hit_triangle_raw calls                          */
  uint64_t shaded_hits;    /* bounce-loop iterations (one shaded hit each)    */
} rt_counters;

/* constructor(canvas) + init() — WebGPURenderer.ts:17-32, WebGPUContext.ts:14-36.
 * Returns NULL when no HIP device / the ordinal is invalid (init() throws there). */
rt_ctx* rt_create(int device_ordinal);
void rt_destroy(rt_ctx* ctx);
const char* rt_last_error(const rt_ctx* ctx); /* ctx may be NULL: error of the last failed rt_create */

/* buildPipeline(depth, spp) — WebGPURenderer.ts:34-39, RaytracePass.ts:18-36
 * (pipeline-override constants MAX_DEPTH, SPP). */
int rt_set_pipeline(rt_ctx* ctx, uint32_t max_depth, uint32_t spp);

/* updateScreenSize(w, h) — WebGPURenderer.ts:41-45, ResourceManager.ts:97-142:
 * (re)allocates render target, G-buffer, accumulation buffer and both history textures. */
int rt_resize(rt_ctx* ctx, uint32_t width, uint32_t height);

/* resetAccumulation() — WebGPURenderer.ts:47-49, ResourceManager.ts:144-151 */
int rt_reset_accum(rt_ctx* ctx);

/* loadTexturesFromWorld(bridge) — WebGPURenderer.ts:51-53, ResourceManager.ts:153-198.
 * Takes the already decoded + resized 1024x1024 RGBA8 layers (decode/resize is the
 * browser's job in the reference).  layers == 0 binds the 1x1 white default texture. */
int rt_upload_textures(rt_ctx* ctx, const uint8_t* rgba, uint32_t layers);
/* The same method fed with images at their own size: the host decodes each blob
 * (include/mi355tex.h), the GPU resizes it into its layer — createImageBitmap(blob, {resizeWidth: 1024,
 * resizeHeight: 1024}) + copyExternalImageToTexture, ResourceManager.ts:162-196; resize rule: mi355rt_math.h
 * rt_resize_coord / rt_bilinear_u8.
 *   rt_alloc_texture_layers   createTexture({size: [1024, 1024, layers]}); every layer starts as the white fallback
 *   rt_upload_texture_image   layer <- width x height straight-alpha RGBA8; rgba == NULL writes the white fallback
 *                             bitmap the reference substitutes when decoding fails (:169-175, 200-208)
 *   rt_read_texture_layer     1024*1024*4 bytes back to the host (tests) */
int rt_alloc_texture_layers(rt_ctx* ctx, uint32_t layers);
int rt_upload_texture_image(rt_ctx* ctx, uint32_t layer, const uint8_t* rgba, uint32_t width, uint32_t height);
int rt_read_texture_layer(rt_ctx* ctx, uint32_t layer, uint8_t* out_rgba, size_t cap);

/* BLAS build for World::update(t) on the GPU: the binned-SAH builder of bvh/blas.rs
 * (BVHBuilder::build_with_ids, called per geometry per frame at rebuilder.rs:93-98) — same tree, same triangle order
 * as the scene compiler's CPU restatement, byte for byte.
 *   verts4      4 f32 per vertex (the skinned positions rebuilder.rs hands to the builder), n_verts of them
 *   indices     3 u32 per triangle
 *   nodes_out   8 f32 per node: {min.xyz, bits(skip)} {max.xyz, bits(data)}, BLAS-local skip pointers, leaf data =
 *               (first << 3) | count with `first` an index into order_out; room for nodes_cap nodes (2 * n_tris is enough)
 *   order_out   n_tris u32: position in the BLAS's triangle order -> original triangle id
 * The signature (ctx first) is the one ms_world_set_blas_builder (mi355scene.h) takes as its hook. */
int rt_build_blas(rt_ctx* ctx, const float* verts4, uint32_t n_verts, const uint32_t* indices, uint32_t n_tris,
                  float* nodes_out, uint32_t nodes_cap, uint32_t* n_nodes_out, uint32_t* order_out);

/* updateBuffer(type, data) -> needsRebind — WebGPURenderer.ts:55-60, ResourceManager.ts:230-284 */
int rt_upload(rt_ctx* ctx, rt_kind kind, const void* data, size_t bytes);

/* updateCombinedGeometry(v, n, uv) -> needsRebind — WebGPURenderer.ts:62-68, ResourceManager.ts:286-323 */
int rt_upload_geometry(rt_ctx* ctx, const float* pos4, const float* nrm4, const float* uv2, uint32_t vertex_count);

/* updateCombinedBVH(tlas, blas) -> needsRebind — WebGPURenderer.ts:70-72, ResourceManager.ts:325-346 */
int rt_upload_bvh(rt_ctx* ctx, const float* tlas, uint32_t n_tlas_nodes, const float* blas, uint32_t n_blas_nodes);

/* updateSceneUniforms(cameraData, frameCount, lightCount) — WebGPURenderer.ts:74-80, ResourceManager.ts:359-405 */
int rt_set_scene(rt_ctx* ctx, const float camera[24], uint32_t frame_count, uint32_t light_count);

/* recreateBindGroup() — WebGPURenderer.ts:82-86.  Nothing to rebind on HIP; kept for call parity. */
int rt_recreate_bind_group(rt_ctx* ctx);

/* The first compute() after an upload checks every index the kernels follow (vertex ids, skip pointers, leaf ranges,
 * instance BLAS offsets, light references) on the GPU and returns RT_ERR_INVALID with the reason for a malformed scene —
 * the stand-in for WebGPU's robust buffer access, which a HIP kernel does not have.
 * compute(frameCount) — WebGPURenderer.ts:88-102: totalFrames++, frame uniforms (Halton jitter),
 * primary-visibility pass, path-trace pass.  Enqueues on the context stream. */
int rt_compute(rt_ctx* ctx, uint32_t frame_count);

/* n consecutive compute() calls as ONE dispatch of each kernel — the recorder's batch loop
 * `for (k < batch) renderer.compute(samplesDone + k)` (VideoRecorder.ts:278-280, batch <= 50 there, <= 64 here).
 * Host state (totalFrames, jitter) and every output are bit-identical to calling rt_compute n times; each frame keeps
 * its own G-buffer (24 B/px extra per frame but the last). Needs the persistent kernel form (the default). */
int rt_compute_batch(rt_ctx* ctx, const uint32_t* frame_counts, uint32_t n);

/* present() — WebGPURenderer.ts:104-129: post pass -> render target, swap history. */
int rt_present(rt_ctx* ctx);

/* captureFrame() — WebGPURenderer.ts:131-137, WebGPUContext.ts:38-107: tight RGBA8 rows of the
 * render target (blocking). cap = capacity of out_rgba in bytes (>= width*height*4). */
int rt_capture(rt_ctx* ctx, uint8_t* out_rgba, size_t cap);

/* device.queue.onSubmittedWorkDone() — main.ts:115, VideoRecorder.ts:167,293 */
int rt_sync(rt_ctx* ctx);

/* ---- additions for parity tests, checkpointing and multi-GPU sharding (no reference counterpart) ---- */
int rt_read_accum(rt_ctx* ctx, float* out_rgba32f, size_t cap_bytes);
int rt_write_accum(rt_ctx* ctx, const float* in_rgba32f, size_t bytes);
/* any of the three outputs may be NULL */
int rt_read_gbuffer(rt_ctx* ctx, uint8_t* albedo_rgba8, float* normal_id_rgba32f, float* depth_f32);
int rt_read_history(rt_ctx* ctx, uint16_t* out_rgba16f, size_t cap_bytes); /* last written history texture */
int rt_read_uniforms(rt_ctx* ctx, void* out256);                            /* the 256-byte scene uniform block */
int rt_get_counters(rt_ctx* ctx, rt_counters* out);                         /* blocking; both kernels */
/* counters of one kernel only: 0 = primary-visibility kernel, 1 = path-trace kernel */
int rt_get_kernel_counters(rt_ctx* ctx, int kernel, rt_counters* out);
int rt_reset_counters(rt_ctx* ctx);
/* count nodes/tris/shaded hits too (slower kernel variant); rays are always counted */
int rt_set_counting(rt_ctx* ctx, int detailed);
/* Restrict compute() to rows y with (y / stripe_rows) % count == rank (interleaved row stripes).
 * count <= 1 renders everything.  present() is never sharded. */
int rt_set_stripes(rt_ctx* ctx, uint32_t stripe_rows, uint32_t rank, uint32_t count);
/* Raw device pointer of the float4 accumulation buffer (width*height*16 bytes) so the caller
 * can hand it to a collective (RCCL) without a host round trip. */
void* rt_accum_device_ptr(rt_ctx* ctx);
/* Use caller-owned device memory (width*height*16 bytes, same device) as the accumulation buffer,
 * e.g. a tensor the caller will hand to RCCL; NULL returns to the context's own buffer.
 * Call after rt_resize; a later rt_resize drops the binding and compute() / present() fail until it is renewed. */
int rt_bind_accum(rt_ctx* ctx, void* device_ptr);
/* present() reads this caller-owned float4 buffer (width*height*16 bytes, same device) instead of the accumulation
 * buffer; NULL returns to the accumulation buffer.  The sharded renderer reduces the ranks' stripe accumulators into
 * such a display buffer, so the per-rank accumulators stay disjoint and a progressive render can go on after a
 * gather.  Like rt_bind_accum, the binding is dropped by rt_resize; staleness is tracked PER BINDING: compute() and
 * present() fail with RT_ERR_INVALID until rt_bind_accum is called again (if an accumulator was bound), and present()
 * also until rt_bind_present_source is called again (if a present source was bound) — NULL included; re-binding one
 * does not acknowledge the other. */
int rt_bind_present_source(rt_ctx* ctx, void* device_ptr);
/* Run every subsequent enqueue on a caller-provided hipStream_t (NULL = context's own stream). */
int rt_set_stream(rt_ctx* ctx, void* hip_stream);
/* Average duration in ms of the path-trace kernel over the launches since the last call
 * (HIP events recorded on the context stream around each launch); also returns the launch count. */
int rt_kernel_time_ms(rt_ctx* ctx, double* avg_pathtrace_ms, double* avg_primary_ms, uint32_t* launches);
/* Speculative lookahead for the live loop (src/main.ts:168-173: compute(frameCount); present() per displayed frame).  With
 * max_frames > 1, a compute(f) that continues a run of consecutive frame counts traces the frames f .. f+L-1 as ONE batched
 * dispatch (L doubles along the run, up to max_frames <= 64) and accumulates only frame f; the compute(f+1) ... that follow
 * find their frame ready and only add it.  Images, G-buffer read-backs and the accumulation buffer are bit for bit those of
 * one dispatch per frame; any call that changes what a frame looks like (uploads, rt_set_scene, rt_resize, rt_set_pipeline,
 * ...) drops what was traced ahead.  While it is on, the ray counters count a frame when it is TRACED (ahead of its compute()
 * call), and the detailed-counter build does not trace ahead.  0 (default) / 1 = off. */
int rt_set_lookahead(rt_ctx* ctx, uint32_t max_frames);
/* A caller that knows when its run of consecutive frames ends - renderFrame advances the world every updateInterval frames
 * (main.ts:127-131) - says so: the next rt_compute traces at most `frames_left` frames (itself included), so nothing is
 * traced ahead in vain across the end of the run.  0 = unknown (the default).  Changes no result and discards nothing. */
int rt_set_lookahead_limit(rt_ctx* ctx, uint32_t frames_left);
/* Traversal of the wavefront trace kernels: 1 = child-pair records (csrc/k_pairwalk.hip.h: one 64-byte record per inner
 * node, both children tested per fetch, quad-cooperative LDS-DMA fetch, short per-lane stack); 0 = one 32-byte node per step
 * with skip pointers only (rounds 1-2); 2 (default) = auto: pairs for a scene of one instance, nodes otherwise (where each
 * was measured faster).  Same results bit for bit, same counters.  MI355RT_WALK=node|pairs|auto sets the
 * default of new contexts. */
int rt_set_walk(rt_ctx* ctx, int walk);
int rt_set_kernel_timing(rt_ctx* ctx, int enabled);
/* Per-kernel timers (HIP events on the context stream around every launch while timing is enabled): sum of the
 * durations in ms and launch count per RT_TIMER_* since the last read; n = number of entries the arrays hold. */
enum {
  RT_TIMER_PRIMARY = 0,         /* k_primary_visibility                                                    */
  RT_TIMER_PATHTRACE = 1,       /* k_pathtrace_persistent, or one whole wavefront dispatch (all depths)    */
  RT_TIMER_WF_SHADE = 2,        /* k_wf_shade, one entry per depth                                         */
  RT_TIMER_WF_TRACE_SHADOW = 3, /* k_wf_trace<any hit>                                                     */
  RT_TIMER_WF_TRACE_EXT = 4,    /* k_wf_trace<closest hit>                                                 */
  RT_TIMER_POST = 5,            /* k_postprocess                                                           */
  RT_TIMER_COUNT = 6
};
int rt_kernel_times(rt_ctx* ctx, double* sum_ms, uint32_t* launches, uint32_t n);
/* Diagnostic build (-DRT_CLOCK_STAMP) only: {delta s_memtime, delta s_memrealtime} of each workgroup of the last
 * k_pathtrace_persistent launch (in-kernel clock = ratio x 100 MHz).  Returns the number of pairs written, 0 in the
 * product build, where no stamp executes. */
int rt_debug_clock_stamps(rt_ctx* ctx, uint64_t* out_pairs, uint32_t cap_pairs);
/* The derived traversal array of the uploaded scene (csrc/k_treelet.hip.h) for tests: 8 f32 per node in the new order,
 * the original-index -> new-index table, and the per-instance BLAS roots (any pointer may be NULL).  Returns the node count. */
int rt_debug_read_traversal_nodes(rt_ctx* ctx, float* tnodes_out, uint32_t* new_index_out, uint32_t* inst_root_out,
                                  uint32_t cap_nodes);
/* Diagnostics of the last rt_build_blas: tree levels the breadth-first build went through (nodes of at most 64 triangles
 * are finished inside one wave and do not count) | levels that held a node above 4 096 triangles << 16. */
int rt_build_blas_levels(const rt_ctx* ctx);
/* Device-resident World::update(t): derive EVERY bridge array of this frame on the GPU, inside the
 * renderer's own scene buffers, from the static scene description and the frame's joint matrices (rt_world_frame,
 * mi355rt_layout.h): linear-blend skinning (rebuilder.rs:36-91), the binned-SAH BLAS of every geometry (bvh/blas.rs,
 * without host synchronisation between tree levels), topology / light / draw-command packing (rebuilder.rs:121-168,
 * lib.rs:237-270), the median-split TLAS (bvh/tlas.rs:58-111) and the packed instances - byte for byte the arrays
 * World::update would have produced, which therefore need no upload: it replaces update(t) + updateCombinedGeometry +
 * updateBuffer(topology / instance / lights / draw commands) + updateCombinedBVH of the live loop (src/main.ts:133-163).
 * The signature (with the rt_ctx* as `user`) is ms_device_updater's of mi355scene.h.  The static description is copied
 * to the device when frame->static_epoch differs from the last call's; the pointers need only live during the call.
 * One stream synchronisation at the end (node counts, the TLAS root).  Returns RT_OK / RT_REALLOCATED, or < 0: a
 * description this path does not take (an instance of an empty geometry, a NaN instance box; the caller then runs the
 * host update and uploads as before) or an error.  The update writes into the live scene buffers: after a failure that
 * came when its kernels had started (a NaN instance box, a tree that did not settle) those hold a mix of two scenes and
 * rt_compute refuses with RT_ERR_INVALID until the scene has been uploaded again (rt_upload* / rt_upload_geometry /
 * rt_upload_bvh); a refusal of the arguments (null arrays, a skin table that does not match the static description) is made
 * before anything is written and leaves the previous scene renderable. */
int rt_world_update(rt_ctx* ctx, const rt_world_frame* frame);
/* A geometry without a skin has the same vertices in every frame of a static description, hence the same BLAS, topology
 * rows and emissive list: after the first update rt_world_update leaves its rows in place and copies its node block from a
 * cache (World::update rebuilds it every frame - the same bytes).  Any host upload into the scene buffers (rt_upload*,
 * a new static_epoch) drops the cache.  enabled = 0 rebuilds everything every frame (measurements); default 1. */
int rt_world_set_static_cache(rt_ctx* ctx, int enabled);
/* Stream time (ms, HIP events) of the last rt_world_update: kernels and the small copies, without host work. */
double rt_world_last_ms(const rt_ctx* ctx);
/* ... and of its TLAS kernel alone (k_tlas: instance boxes, median-split tree, packed instances). */
double rt_world_last_tlas_ms(const rt_ctx* ctx);
/* Read a bridge array back from the device-resident world (tests; a host that wants the arrays after all).  out == NULL:
 * only *bytes_out is set. */
typedef enum rt_world_array {
  RT_WORLD_VERTICES = 0, RT_WORLD_NORMALS, RT_WORLD_UVS, RT_WORLD_TOPOLOGY, RT_WORLD_TLAS, RT_WORLD_BLAS, RT_WORLD_INSTANCES,
  RT_WORLD_LIGHTS, RT_WORLD_DRAW_COMMANDS
} rt_world_array;
int rt_world_read(rt_ctx* ctx, int which, void* out, size_t cap_bytes, size_t* bytes_out);
/* Debug / test read-back of the child-pair records the trace kernels walk (csrc/k_pairs.hip.h): pairs_out receives
 * 16 floats per inner node, root_rec_out 8 floats per instance plus 8 for the TLAS root (either may be NULL).  Returns the
 * number of pair records, or < 0 (cap_pairs too small, no scene). */
int rt_debug_read_pairs(rt_ctx* ctx, float* pairs_out, float* root_rec_out, uint32_t cap_pairs);
/* Diagnostic build (-DRT_TRACE_STAMPS) only: s_memtime cycles the waves of k_wf_trace spent in its three sections,
 * summed over all launches since the last reset: out16[queue][k], queue 0 = closest hit, 1 = any hit; k = 0..2 cycles in
 * {retire / pull, node step, triangle flush}, 3..5 how often each did work, 6 waves, 7 loop trips.  Returns 1 in the
 * diagnostic build, 0 in the product build (where no stamp executes and the array stays zero). */
int rt_debug_trace_sections(rt_ctx* ctx, uint64_t* out16, int reset);
/* Diagnostic build (-DRT_PT_STAMPS) only: the same for k_pathtrace_persistent: out8[0..4] = cycles in {regenerate + start,
 * shade, shadow traversal, extension traversal + surface frame, finish}, [5] trips, [6] waves. */
int rt_debug_pt_sections(rt_ctx* ctx, uint64_t* out8, int reset);
/* Diagnostic build (-DRT_LANE_STATS) only: lane utilisation of the parts of a trip of k_pathtrace_persistent:
 * out32[2 k] = times part k ran (wave level), out32[2 k + 1] = active lanes summed; k = 0 shade, 1 / 2 node step / triangle
 * chunk of the shadow walk, 3 / 4 of the extension walk, 5 surface frame of a new hit, 6 start of a sample, 7 end of a sample.
 * Returns 1 in the diagnostic build, 0 in the product build (no counter executes, the array stays zero). */
int rt_debug_lane_stats(rt_ctx* ctx, uint64_t* out32, int reset);
/* Proof obligation of the device build's short division / reciprocal / square-root sequences (csrc/k_ieee.hip.h): run
 * them on the GPU against the compiler's correctly rounded IEEE expansions over inputs [first, first + count) of the
 * input set of `op` (csrc/k_ieee_inputs.h: RT_IEEE_OP_*; one-operand ops: index = bit pattern, 2^32 of them; divisions:
 * 2^24 mantissa samples x 2^8 exponent classes).  n = inputs checked; guard_pass = inputs the guard sends down the short
 * sequence, wrong_fast = of those, results that differ from IEEE (must be 0); wrong_fn = results of the composed function
 * (guard + wave-uniform fallback, what the kernels call) that differ (must be 0); checksum = sum of
 * rt_ieee_mix(IEEE result, index), to be compared with the host CPU's own IEEE results (tests/model/ieee_ref.cpp);
 * bad = operand a, operand b, got, ieee of the first mismatches.  No renderer state is touched.  Returns RT_OK, or
 * RT_ERR_INVALID for an unknown op / a build with -DRT_IEEE_PLAIN (nothing to check). */
typedef struct rt_ieee_report {
  uint64_t n, guard_pass, wrong_fast, wrong_fn, checksum;
  uint32_t n_bad;
  uint32_t bad[32];
} rt_ieee_report;
int rt_debug_ieee_check(rt_ctx* ctx, int op, uint64_t first, uint64_t count, rt_ieee_report* out);
/* Path-trace kernel form (all four are bit-identical; tests/test_gpu_parity.py::test_kernel_forms_agree_bitwise):
 *   3 = auto (default): wavefront form when the scene's records do not fit LDS, SPP == 1 and the dispatch carries
 *       >= 4 frames (rt_compute_batch); the persistent kernel otherwise
 *   2 = wavefront: shade / trace stages per depth, path state in HBM, ray-level regeneration in the trace kernels
 *       (SPP != 1 falls back to the persistent kernel)
 *   1 = persistent waves with per-lane path regeneration
 *   0 = one pixel per lane, one 8x8 tile per wave (the reference's dispatch shape; kept for A/B timing; no batches) */
int rt_set_kernel_variant(rt_ctx* ctx, int variant);
int rt_device_count(void);

#ifdef __cplusplus
}
#endif
#endif
