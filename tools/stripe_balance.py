#!/usr/bin/env python3
"""Strong-scaling balance of the interleaved row stripes: time of the slowest of N ranks' shards (one GPU, one rank at a
time) for several stripe heights.  usage: stripe_balance.py [ranks]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import webgpu_raytracer_amd as W  # noqa: E402

ranks = int(sys.argv[1]) if len(sys.argv) > 1 else 8
b = W.WorldBridge()
b.loadScene("cornell")
r = W.WebGPURenderer(0)
r.buildPipeline(8, 1)
W.upload_scene(r, b, 1920, 1080)
frames = list(range(1, 65))


def image():
    r.resetAccumulation()
    t0 = time.perf_counter()
    for i in range(0, 64, 32):
        r.computeBatch(frames[i:i + 32])
    r.sync()
    return (time.perf_counter() - t0) * 1e3


r.setStripes(0, 0, 1)
image()
full = min(image() for _ in range(3))
print("whole image: %.2f ms" % full)
for rows in (8, 16, 32, 64):
    times = []
    for k in range(ranks):
        r.setStripes(rows, k, ranks)
        image()
        times.append(min(image() for _ in range(2)))
    print("stripes of %2d rows x %d ranks: slowest %.2f ms, mean %.2f ms -> efficiency %.1f%% (balance %.1f%%)" % (
        rows, ranks, max(times), sum(times) / ranks, 100 * full / ranks / max(times), 100 * sum(times) / ranks / max(times)))
