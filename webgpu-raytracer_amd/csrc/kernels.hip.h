// kernels.hip.h — hand-written gfx950 kernels of the path-tracing hot path.
//
//   k_common.hip.h            helpers, texture fetch, RNG
//   k_intersect.hip.h         rays, slab / triangle tests, per-lane stackless walk
//   k_shading.hip.h           surface frame, BSDFs, light sampling, counters
//   k_prepare_primary.hip.h   k_prepare_tris / _instances / _lights (upload-time re-layout, device_scene.h) and
//                             k_primary_visibility: the hardware-raster G-buffer pass (Rasterizer.wgsl:81-173,
//                             RasterizerPass.ts:97-140) as one closest-hit cast per pixel
//   k_treelet.hip.h           upload-time re-layout of the node array: explicit successors, most-visited nodes first
//   k_pairs.hip.h             upload-time re-layout into CHILD-PAIR records (one 64-byte record per inner node)
//   k_traverse.hip.h          the wave-level TLAS / BLAS walk (node step + LDS triangle queue) of the persistent kernel
//   k_pairwalk.hip.h          per-ray state machine of the walk over pair records (plain C++: also run on the host by the tests)
//   k_pairtrav.hip.h          its wave-level side: quad-cooperative record fetch, LDS stack, batched entry, triangle flush
//   k_pathtrace.hip.h         Raytracer.wgsl `main` + ray_color (:607-819): k_pathtrace, k_pathtrace_persistent
//   k_wavefront.hip.h         the same bounce as shade / trace stages over device queues (large scenes)
//   k_texture_post.hip.h      k_resize_texture; k_postprocess = PostProcess.wgsl `main` (:103-176)
//   k_validate.hip.h          k_validate_scene: every index the kernels follow, checked once per upload
//
// All arithmetic is unfused IEEE f32 (-ffp-contract=off) with the builtin semantics of
// include/mi355rt_math.h, in the evaluation order of the WGSL source, so that every path
// takes the same branches as the CPU oracle and results agree bit for bit.
#ifndef MI355RT_KERNELS_HIP_H
#define MI355RT_KERNELS_HIP_H

#include "device_scene.h"

#define RT_T_MIN 0.001f
#define RT_T_MAX 1e30f
#define RT_COUNTER_SHARDS 1024

#include "k_common.hip.h"
#include "k_intersect.hip.h"
#include "k_shading.hip.h"
#include "k_prepare_primary.hip.h"
#include "k_treelet.hip.h"
#include "k_pairwalk.hip.h"
#include "k_pairs.hip.h"
#include "k_traverse.hip.h"
#include "k_pairtrav.hip.h"
#include "k_pathtrace.hip.h"
#include "k_wavefront.hip.h"
#include "k_texture_post.hip.h"
#include "k_validate.hip.h"

#endif
