#!/usr/bin/env python3
"""Where do the waves of k_wf_trace spend their cycles?  DIAGNOSTIC build (-DRT_TRACE_STAMPS): s_memtime around the three
sections of the trace loop (retire / pull rays, node step, triangle flush), each closed by s_waitcnt 0 so that a
section is charged for the memory it waits on.  Rebuilds the product library afterwards.
usage: trace_sections.py [scene] [frames] [depth] [width] [height]"""
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import webgpu_raytracer_amd as W  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else "sponza_like"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 32
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 8
width = int(sys.argv[4]) if len(sys.argv) > 4 else 1920
height = int(sys.argv[5]) if len(sys.argv) > 5 else 1080
W._build.build_rt(force=True, extra_flags=["-DRT_TRACE_STAMPS"])
try:
    b = W.WorldBridge()
    b.loadScene(scene)
    r = W.WebGPURenderer(0)
    r.buildPipeline(depth, 1)
    W.upload_scene(r, b, width, height)
    fl = list(range(1, frames + 1))
    r.computeBatch(fl)
    r.sync()
    buf = np.zeros((2, 8), dtype=np.uint64)
    r.L.rt_debug_trace_sections(r.ctx, buf.ctypes.data_as(ctypes.c_void_p), 1)
    r.setKernelTiming(True)
    r.kernelTimes()
    r.computeBatch(fl)
    r.sync()
    kt = r.kernelTimes()
    r.L.rt_debug_trace_sections(r.ctx, buf.ctypes.data_as(ctypes.c_void_p), 1)
    print("scene=%s frames=%d depth=%d kernel ms:" % (scene, frames, depth), {k: round(v["ms"], 2) for k, v in kt.items() if v["launches"]})
    for q, name in ((0, "closest-hit (extension rays)"), (1, "any-hit (shadow rays)")):
        cyc, cnt, waves, trips = buf[q, 0:3].astype(float), buf[q, 3:6].astype(float), float(buf[q, 6]), float(buf[q, 7])
        tot = cyc.sum()
        print("%s: %d waves, %.0f trips/wave, %.0f cycles/wave" % (name, waves, trips / waves, tot / waves))
        for k, sec in enumerate(("retire / pull", "node step", "triangle flush")):
            print("   %-14s %5.1f %% of the cycles; did work in %5.1f %% of the trips; %7.0f cycles per working trip"
                  % (sec, 100 * cyc[k] / tot, 100 * cnt[k] / trips, cyc[k] / max(cnt[k], 1)))
    r.destroy()
finally:
    W._build.build_rt(force=True)
