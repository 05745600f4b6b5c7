#!/usr/bin/env python3
"""rt_build_blas on the whole triangle soup of a scene (all geometries as one mesh): time per call and tree levels."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import numpy as np
import webgpu_raytracer_amd as W
r = W.WebGPURenderer(0)
for scene in sys.argv[1:] or ["sponza_like", "glass_blob"]:
    b = W.WorldBridge(); b.loadScene(scene)
    verts = np.asarray(b.vertices, dtype=np.float32).reshape(-1, 4)
    topo = np.asarray(b.mesh_topology, dtype=np.uint32).reshape(-1, 20)
    idx = topo[:, :3].copy().reshape(-1)
    print(scene, "geometries:", sorted(set(topo[:, 3].tolist()))[:8], "tris", len(topo))
    for k in range(4):
        t0 = time.perf_counter(); nodes, order = r.buildBlas(verts, idx); dt = time.perf_counter() - t0
        lv = r.L.rt_build_blas_levels(r.ctx)
        print("  call %d: %.2f ms, %d nodes, levels %d, large-node levels %d" % (k, dt * 1e3, len(nodes), lv & 0xffff, lv >> 16))
