#!/usr/bin/env python3
"""In-kernel clock of k_pathtrace_persistent under the bench workload (MI355X_MICROARCH.md "DVFS give-back" item 6):
a DIAGNOSTIC build (-DRT_CLOCK_STAMP) stamps s_memtime / s_memrealtime around each workgroup's work loop; after
~2 s of back-to-back launches the median ratio x 100 MHz is the clock the chip holds under this kernel.
Rebuilds the product library afterwards.  usage: clock_check.py [out.json]"""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import webgpu_raytracer_amd as W  # noqa: E402

W._build.build_rt(force=True, extra_flags=["-DRT_CLOCK_STAMP"])
try:
    b = W.WorldBridge()
    b.loadScene("cornell")
    r = W.WebGPURenderer(0)
    r.buildPipeline(8, 1)
    W.upload_scene(r, b, 1920, 1080)
    frames = list(range(1, 65))
    t0 = time.perf_counter()
    images = 0
    while time.perf_counter() - t0 < 2.5:
        r.resetAccumulation()
        r.computeBatch(frames[:32])
        r.computeBatch(frames[32:])
        r.sync()
        images += 1
    import ctypes
    buf = np.zeros((4096, 2), dtype=np.uint64)
    n = r.L.rt_debug_clock_stamps(r.ctx, buf.ctypes.data_as(ctypes.c_void_p), 4096)
    pairs = buf[:n]
    pairs = pairs[pairs[:, 1] > 0]
    ghz = pairs[:, 0].astype(np.float64) / pairs[:, 1].astype(np.float64) * 0.1
    out = {"kernel": "k_pathtrace_persistent", "workload": "cornell 1920x1080 depth 8, 32-frame batches, %d images back to back" % images,
           "workgroups_stamped": int(len(ghz)), "in_kernel_clock_GHz_median": round(float(np.median(ghz)), 4),
           "in_kernel_clock_GHz_min": round(float(ghz.min()), 4), "in_kernel_clock_GHz_max": round(float(ghz.max()), 4),
           "cycles_per_workgroup_median": int(np.median(pairs[:, 0]))}
    print(json.dumps(out))
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)
    r.destroy()
finally:
    W._build.build_rt(force=True)
