#!/usr/bin/env python3
"""Device-memory leak check: 40 create / upload / render / GPU-BLAS-build / destroy cycles, free memory must not drift."""
import sys, ctypes
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import webgpu_raytracer_amd as W
hip = ctypes.CDLL("libamdhip64.so")
def free_mb():
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t))
    return f.value / 2**20
b = W.WorldBridge(); b.loadScene("sponza_like")
r0 = W.WebGPURenderer(0); base = free_mb(); r0.destroy()
vals = []
for i in range(40):
    r = W.WebGPURenderer(0)
    r.buildPipeline(6, 1)
    W.upload_scene(r, b, 320, 180)
    r.computeBatch([1, 2, 3, 4]); r.compute(5); r.present(); r.captureFrame()
    b.setBlasBuilder(r); b.update(0.0); b.setBlasBuilder(None)
    r.destroy()
    if i % 10 == 9: vals.append(free_mb())
print("free MB after every 10 create/destroy cycles:", [round(v) for v in vals], "baseline", round(base))
assert abs(vals[-1] - vals[0]) < 64, "device memory is leaking"
print("no leak")
