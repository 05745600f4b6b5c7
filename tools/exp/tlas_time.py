#!/usr/bin/env python3
"""k_tlas alone and the whole device update at 1 001 and 16 384 instances (rt_world_last_tlas_ms); bench.py world_update.tlas has the same."""

import sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
import webgpu_raytracer_amd as pkg
for sc in ("instanced1000", "instanced16384"):
    r = pkg.WebGPURenderer(0); r.buildPipeline(8, 1); r.setWorldStaticCache(False)
    b = pkg.WorldBridge(zero_copy=True); b.setDeviceUpdater(r); b.loadScene(sc); pkg.upload_scene(r, b, 640, 360)
    for _ in range(3): b.update(0.0)
    r.sync(); ta = tt = 0.0
    for _ in range(10):
        b.update(0.0); assert b.deviceResident, b.deviceWarning
        ta += r.worldLastMs(); tt += r.worldLastTlasMs()
    print(sc, len(b.instances)//36, "instances: k_tlas %.4f ms, whole device update %.4f ms" % (tt/10, ta/10))
    b.close(); r.destroy()
