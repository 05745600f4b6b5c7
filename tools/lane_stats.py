#!/usr/bin/env python3
"""Lane utilisation of k_pathtrace_persistent by part of a trip (Cornell, one 32-frame batch of the bench workload).
DIAGNOSTIC build -DRT_LANE_STATS: a ballot + two atomics where a wave enters a part (counts, no timing; the share of a wave's
cycles each section takes comes from tools/pt_sections.py, its own process and build).  Rebuilds the product library at the end.
usage: lane_stats.py [scene]"""
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import webgpu_raytracer_amd as W  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else "cornell"
PARTS = ("shade", "shadow walk: node step", "shadow walk: triangle chunk", "extension walk: node step",
         "extension walk: triangle chunk", "surface frame of the new hit", "start of a sample", "end of a sample")


def render(flags):
    W._build.build_rt(force=True, extra_flags=flags)
    b = W.WorldBridge()
    b.loadScene(scene)
    r = W.WebGPURenderer(0)
    r.buildPipeline(8, 1)
    W.upload_scene(r, b, 1920, 1080)
    r.setKernelVariant(1)
    fl = list(range(1, 33))
    r.computeBatch(fl)
    r.sync()
    return r, fl


try:
    r, fl = render(["-DRT_LANE_STATS"])
    buf = np.zeros(32, dtype=np.uint64)
    r.L.rt_debug_lane_stats(r.ctx, buf.ctypes.data_as(ctypes.c_void_p), 1)
    r.computeBatch(fl)
    r.sync()
    r.L.rt_debug_lane_stats(r.ctx, buf.ctypes.data_as(ctypes.c_void_p), 1)
    n, lanes = buf[0:16:2].astype(float), buf[1:16:2].astype(float)
    trips = n[0]
    print("scene=%s, one 32-frame batch: %d wave-level trips that shade" % (scene, trips))
    print("   %-34s %14s %12s %12s" % ("part", "runs per trip", "lanes / run", "utilisation"))
    for k, name in enumerate(PARTS):
        if n[k]:
            print("   %-34s %14.2f %12.1f %12.3f" % (name, n[k] / trips, lanes[k] / n[k], lanes[k] / n[k] / 64.0))
    r.destroy()
finally:
    W._build.build_rt(force=True)
