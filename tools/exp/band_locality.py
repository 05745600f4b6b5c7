#!/usr/bin/env python3
"""Would XCD-affine ray queues pay?  Upper bound without building them: trace ONE contiguous band of the image (1/8 of the
rows) on the whole chip and compare the trace kernels' time per ray with the full image's.  A band's rays start on 1/8 of the
visible surfaces, so every XCD's L2 then serves what an XCD-affine assignment would give it.  usage: band_locality.py [scene]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import webgpu_raytracer_amd as W  # noqa: E402
import parity_util as pu  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else "sponza_like"
w, h, depth, frames = 1920, 1080, 8, 32
b = pu.bridge_for(W, scene)


def run(stripes):
    r = W.WebGPURenderer(0)
    r.buildPipeline(depth, 1)
    W.upload_scene(r, b, w, h)
    if stripes:
        r.setStripes(*stripes)
    fl = list(range(1, frames + 1))
    r.computeBatch(fl)
    r.sync()
    r.resetCounters()
    r.setKernelTiming(True)
    r.kernelTimes()
    r.computeBatch(fl)
    r.sync()
    kt = r.kernelTimes()
    c = r.getCounters()
    r.destroy()
    rays = c["extension_rays"] + c["shadow_rays"]
    trace = kt["pathtrace"]["ms"] - kt["wf_shade"]["ms"]
    return rays, trace, kt["wf_shade"]["ms"], kt["primary"]["ms"], c["primary_rays"]


rays, trace, shade, prim, pr = run(None)
print("%s full image: %.1f M secondary rays, trace %.1f ms = %.2f ns/ray, shade %.1f ms, primary %.2f ms (%.2f ns/ray)" % (
    scene, rays / 1e6, trace, trace * 1e6 / rays, shade, prim, prim * 1e6 / pr))
for k in range(8):
    rays, trace, shade, prim, pr = run((h // 8, k, 8))
    print("   band %d (rows %4d..%4d): %.1f M secondary rays, trace %.1f ms = %.2f ns/ray, shade %.1f ms, primary %.2f ms (%.2f ns/ray)" % (
        k, k * (h // 8), (k + 1) * (h // 8) - 1, rays / 1e6, trace, trace * 1e6 / rays, shade, prim, prim * 1e6 / pr))
