#!/usr/bin/env python3
"""Host-side cost of one live-loop frame (compute(f); present()): the same loop on a 64x64 image, where the GPU work is
negligible, so the wall time per frame is what the host needs to enqueue it."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import webgpu_raytracer_amd as W  # noqa: E402
b = W.WorldBridge(); b.loadScene("cornell")
r = W.WebGPURenderer(0); r.buildPipeline(8, 1); W.upload_scene(r, b, 64, 64)
for f in range(1, 50): r.compute(f); r.present()
r.sync()
t0 = time.perf_counter()
n = 2000
for f in range(50, 50 + n): r.compute(f); r.present()
t1 = time.perf_counter()
r.sync()
t2 = time.perf_counter()
print("host enqueue ms per frame %.4f; with final sync %.4f" % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
