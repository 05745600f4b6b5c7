#!/usr/bin/env python3
"""World::update(t) rebuild cost: scene compiler's CPU BLAS builder vs the GPU builder (rt_build_blas) as its hook, and
the GPU build alone (upload + kernels + download)."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import webgpu_raytracer_amd as W  # noqa: E402

r = W.WebGPURenderer(0)
for scene in ("sponza_like", "glass_blob"):
    b = W.WorldBridge()
    b.loadScene(scene)
    n_tris = len(b.mesh_topology) // 20

    def upd(n=3):
        t0 = time.perf_counter()
        for _ in range(n):
            b._lib.ms_world_update(b._world, 0.0)
        return (time.perf_counter() - t0) / n * 1e3
    cpu_ms = upd()
    b.setBlasBuilder(r)
    upd(1)
    gpu_ms = upd()
    verts = np.asarray(b.vertices, dtype=np.float32).reshape(-1, 4)
    idx = np.asarray(b.mesh_topology, dtype=np.uint32).reshape(-1, 20)[:, :3].copy().reshape(-1)
    r.buildBlas(verts, idx)
    t0 = time.perf_counter()
    for _ in range(5):
        nodes, order = r.buildBlas(verts, idx)
    build_ms = (time.perf_counter() - t0) / 5 * 1e3
    print("%s: %d triangles, %d nodes | update(t): CPU builder %.1f ms, GPU builder hook %.1f ms | rt_build_blas alone %.2f ms "
          "(%.1f Mtris/s)" % (scene, n_tris, len(nodes), cpu_ms, gpu_ms, build_ms, n_tris / build_ms / 1e3))
