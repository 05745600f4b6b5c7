#!/usr/bin/env python3
"""Where do the waves of k_pathtrace_persistent spend their cycles (Cornell, bench workload)?  DIAGNOSTIC build
(-DRT_PT_STAMPS): s_memtime around the five sections of a trip, each closed by s_waitcnt 0.  Rebuilds the product library."""
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import webgpu_raytracer_amd as W  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else "cornell"
W._build.build_rt(force=True, extra_flags=["-DRT_PT_STAMPS"])
try:
    b = W.WorldBridge()
    b.loadScene(scene)
    r = W.WebGPURenderer(0)
    r.buildPipeline(8, 1)
    W.upload_scene(r, b, 1920, 1080)
    r.setKernelVariant(1)
    fl = list(range(1, 33))
    r.computeBatch(fl)
    r.sync()
    buf = np.zeros(8, dtype=np.uint64)
    r.L.rt_debug_pt_sections(r.ctx, buf.ctypes.data_as(ctypes.c_void_p), 1)
    r.setKernelTiming(True)
    r.kernelTimes()
    r.computeBatch(fl)
    r.sync()
    kt = r.kernelTimes()
    r.L.rt_debug_pt_sections(r.ctx, buf.ctypes.data_as(ctypes.c_void_p), 1)
    cyc, trips, waves = buf[:5].astype(float), float(buf[5]), float(buf[6])
    print("scene=%s: pathtrace %.2f ms; %d waves, %.0f trips/wave, %.0f cycles/trip" % (
        scene, kt["pathtrace"]["ms"], waves, trips / waves, cyc.sum() / trips))
    for k, name in enumerate(("regenerate + start", "shade", "shadow traversal", "extension traversal + surface", "finish")):
        print("   %-30s %5.1f %%   %7.0f cycles per trip" % (name, 100 * cyc[k] / cyc.sum(), cyc[k] / trips))
    r.destroy()
finally:
    W._build.build_rt(force=True)
