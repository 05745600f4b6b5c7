#!/usr/bin/env python3
"""Mrays/s of every BASELINE.json config on one MI355X + the CPU oracle on a row-band sample of the same frames.
Prints a markdown table (the results table of BASELINE.md)."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import webgpu_raytracer_amd as W  # noqa: E402
import oracle_lib  # noqa: E402
import parity_util as pu  # noqa: E402

CONFIGS = [
    ("1 cornell", "cornell", 512, 512, 4, 4, 1),                 # name, scene, w, h, frames, depth, cpu stripe divisor
    ("2 viewer+diamond", "viewer_diamond", 1280, 720, 16, 8, 1),
    ("2b viewer+diamond, 968 tris", "viewer_diamond_1k", 1280, 720, 16, 8, 3),
    ("3 instanced x1000", "instanced1000", 1920, 1080, 64, 8, 27),
    ("4 sponza-like 263k tris", "sponza_like", 1920, 1080, 64, 8, 27),
    ("5 glass blob 205k tris", "glass_blob", 3840, 2160, 256, 16, 270),
    ("headline cornell 1080p", "cornell", 1920, 1080, 64, 8, 3),
]


def rays(c):
    return c["primary_rays"] + c["extension_rays"] + c["shadow_rays"]


print("| config | WxH, frames x depth | GPU Mrays/s | ms/image | CPU oracle Mrays/s (threads) | ratio | band parity |")
print("|---|---|---|---|---|---|---|")
for name, scene, w, h, frames, depth, div in CONFIGS:
    b = pu.bridge_for(W, scene)
    g = W.WebGPURenderer(0)
    g.buildPipeline(depth, 1)
    W.upload_scene(g, b, w, h)
    fr = list(range(1, frames + 1))
    B = 32      # frames per batched dispatch (the recorder's cap is 50; at 4K the path state + queues + G-buffers of 32 frames
                # are 63 GB of the 288: a depth-16 batch is 49 launches, and their drain tails are paid per batch)
    g.computeBatch(fr[:B])                     # first-use allocations and module load stay out of the timed image
    g.sync()
    g.resetAccumulation()
    g.resetCounters()
    t0 = time.perf_counter()
    for i in range(0, len(fr), B):
        g.computeBatch(fr[i:i + B])
    g.present()
    g.sync()
    dt = time.perf_counter() - t0
    gr = rays(g.getCounters())
    g2 = W.WebGPURenderer(0)         # fresh context so that totalFrames == frame_count like the oracle run below
    g2.buildPipeline(depth, 1)
    W.upload_scene(g2, b, w, h)
    for i in range(0, len(fr), B):
        g2.computeBatch(fr[i:i + B])
    g2.sync()
    acc = g2.readAccum()
    cpu = oracle_lib.OracleRenderer()
    cpu.buildPipeline(depth, 1)
    W.upload_scene(cpu, b, w, h)
    stripes = max(1, (h // 8))
    cpu.setStripes(8, 0, div)
    cpu.resetCounters()
    t1 = time.perf_counter()
    for f in fr:
        cpu.compute(f)
    ct = time.perf_counter() - t1
    cr = rays(cpu.getCounters())
    rows = (np.arange(h) // 8) % div == 0
    ok = np.array_equal(cpu.readAccum()[rows].view(np.uint32), acc[rows].view(np.uint32))
    print("| %s | %dx%d, %d x d%d | %.0f | %.1f | %.1f (%d) | %.0fx | %s (%d rows) |" % (
        name, w, h, frames, depth, gr / dt / 1e6, dt * 1e3, cr / ct / 1e6, oracle_lib.lib().oracle_hardware_threads(),
        (gr / dt) / (cr / ct), "bit-exact" if ok else "MISMATCH", int(rows.sum())), flush=True)
    g.destroy()
    g2.destroy()
