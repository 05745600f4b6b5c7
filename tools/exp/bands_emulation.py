#!/usr/bin/env python3
"""Would tracing ONE frame of a big scene as S interleaved stripe sets on S streams pay (the tail of every depth's trace launch
of one set overlapping the bulk of another's)?  Emulation with S renderer contexts in one process, each owning the stripes of
"rank k of S" (the multi-GPU sharding on one GPU).  usage: bands_emulation.py [scene] [variant]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import webgpu_raytracer_amd as W  # noqa: E402
import parity_util as pu  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else "sponza_like"
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 2
w, h, depth, frames = 1920, 1080, 8, 24
b = pu.bridge_for(W, scene)


def run(S, variant):
    rs = []
    for k in range(S):
        r = W.WebGPURenderer(0)
        r.setKernelVariant(variant)
        r.buildPipeline(depth, 1)
        W.upload_scene(r, b, w, h)
        if S > 1:
            r.setStripes(8, k, S)
        rs.append(r)
    for f in (1, 2):
        for r in rs:
            r.compute(f)
    for r in rs:
        r.sync()
    t0 = time.perf_counter()
    for f in range(3, 3 + frames):
        for r in rs:
            r.compute(f)
    for r in rs:
        r.sync()
    ms = (time.perf_counter() - t0) * 1e3 / frames
    for r in rs:
        r.destroy()
    return ms


for S in (1, 2, 3, 4):
    print("%s variant %d: %d stripe set(s) on %d stream(s): %.2f ms per frame" % (scene, variant, S, S, run(S, variant)), flush=True)
