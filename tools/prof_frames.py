#!/usr/bin/env python3
"""Small driver for rocprofv3 runs: N compute() dispatches of one scene at a given size, no torch.
usage: prof_frames.py [scene] [width] [height] [frames] [depth] [variant] [detailed] [ranks] [batch]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import webgpu_raytracer_amd as W  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else "cornell"
w = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
h = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 8
depth = int(sys.argv[5]) if len(sys.argv) > 5 else 8
variant = int(sys.argv[6]) if len(sys.argv) > 6 else 1
detailed = int(sys.argv[7]) if len(sys.argv) > 7 else 0
ranks = int(sys.argv[8]) if len(sys.argv) > 8 else 1   # render only rank 0 of N interleaved 16-row stripes
batch = int(sys.argv[9]) if len(sys.argv) > 9 else 1   # frames per batched dispatch

sys.path.insert(0, os.path.join(REPO, "tests"))
import parity_util as pu  # noqa: E402  (scene names as in the tests: viewer_diamond, viewer_diamond_1k, ...)

b = pu.bridge_for(W, scene)
r = W.WebGPURenderer(0)
r.buildPipeline(depth, 1)
W.upload_scene(r, b, w, h)
r.setKernelVariant(variant)
if ranks > 1:
    r.setStripes(16, 0, ranks)
r.setCounting(bool(detailed))
r.compute(1)
r.sync()
r.resetCounters()
r.setKernelTiming(True)
r.kernelTimes()
t0 = time.perf_counter()
fl = list(range(2, frames + 2))
if batch > 1:
    for i in range(0, len(fl), batch):
        r.computeBatch(fl[i:i + batch])
else:
    for f in fl:
        r.compute(f)
r.sync()
dt = time.perf_counter() - t0
kt = r.kernelTimes()
k = {"pathtrace_ms": kt["pathtrace"]["ms"] / max(1, kt["pathtrace"]["launches"]),
     "primary_ms": kt["primary"]["ms"] / max(1, kt["primary"]["launches"])}
c = r.getCounters()
rays = c["primary_rays"] + c["extension_rays"] + c["shadow_rays"]
print("scene=%s %dx%d frames=%d depth=%d variant=%d: %.3f ms/frame wall, pathtrace %.4f ms, primary %.4f ms, "
      "%.1f Mrays/frame, %.1f Mrays/s" % (scene, w, h, frames, depth, variant, dt / frames * 1e3, k["pathtrace_ms"],
                                          k["primary_ms"], rays / frames / 1e6, rays / dt / 1e6))
print("kernel ms (sum over the run):", {kk: round(vv["ms"], 2) for kk, vv in kt.items() if vv["launches"]})
print({kk: vv // frames for kk, vv in c.items()})
if detailed:
    print("pathtrace kernel only:", {kk: vv // frames for kk, vv in r.getKernelCounters(1).items()})
