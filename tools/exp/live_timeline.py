#!/usr/bin/env python3
"""A short live loop (compute(f); present() per frame) for a rocprofv3 --kernel-trace timeline: do consecutive frames overlap?
usage: live_timeline.py [scene] [frames]"""
import os
import sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import webgpu_raytracer_amd as W  # noqa: E402
scene = sys.argv[1] if len(sys.argv) > 1 else "cornell"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 12
b = W.WorldBridge()
b.loadScene(scene)
r = W.WebGPURenderer(0)
r.buildPipeline(8, 1)
W.upload_scene(r, b, 1920, 1080)
for f in range(1, frames + 1):
    r.compute(f)
    r.present()
r.sync()
r.setKernelTiming(True)
r.kernelTimes()
import time
t0 = time.perf_counter()
for f in range(frames + 1, 2 * frames + 1):
    r.compute(f)
    r.present()
r.sync()
dt = time.perf_counter() - t0
print("wall ms per frame", dt / frames * 1e3, {k: round(v["ms"] / max(1, v["launches"]), 4) for k, v in r.kernelTimes().items() if v["launches"]})
