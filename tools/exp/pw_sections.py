#!/usr/bin/env python3
"""Where does a round of the child-pair walk (k_pairtrav.hip.h pw_trip) spend a wave's cycles?  DIAGNOSTIC build
(-DRT_PW_STAMPS).  usage: pw_sections.py [scene] [frames] [depth] [w] [h]"""
import ctypes
import os
import sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import webgpu_raytracer_amd as W  # noqa: E402
scene = sys.argv[1] if len(sys.argv) > 1 else "sponza_like"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 32
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 8
w = int(sys.argv[4]) if len(sys.argv) > 4 else 1920
h = int(sys.argv[5]) if len(sys.argv) > 5 else 1080
W._build.build_rt(force=True, extra_flags=["-DRT_PW_STAMPS"])
try:
    b = W.WorldBridge()
    b.loadScene(scene)
    r = W.WebGPURenderer(0)
    r.buildPipeline(depth, 1)
    W.upload_scene(r, b, w, h)
    fl = list(range(1, frames + 1))
    r.computeBatch(fl)
    r.sync()
    buf = np.zeros((2, 8), dtype=np.uint64)
    r.L.rt_debug_trace_sections(r.ctx, buf.ctypes.data_as(ctypes.c_void_p), 1)
    r.computeBatch(fl)
    r.sync()
    r.L.rt_debug_trace_sections(r.ctx, buf.ctypes.data_as(ctypes.c_void_p), 1)
    names = ("ask for records", "pops", "instance entry", "wait + read records", "slab tests + decision")
    for q, name in ((0, "closest-hit"), (1, "any-hit")):
        cyc, rounds, waves = buf[q, 0:5].astype(float), float(buf[q, 5]), float(buf[q, 6])
        print("%s %s: %d waves, %.0f rounds per wave, %.0f cycles per round" % (scene, name, waves, rounds / waves, cyc.sum() / rounds))
        for k, n in enumerate(names):
            print("   %-24s %5.1f %%  %7.0f cycles per round" % (n, 100 * cyc[k] / cyc.sum(), cyc[k] / rounds))
    r.destroy()
finally:
    W._build.build_rt(force=True)
