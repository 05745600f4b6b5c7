#!/usr/bin/env python3
"""Driver for rocprofv3 passes: N images of one scene exactly as bench.py issues them (batched dispatches only, so every
path-trace launch of the run carries the same work).  No torch.
usage: prof_workload.py [scene] [frames] [depth] [batch] [images] [width] [height]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import webgpu_raytracer_amd as W  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else "cornell"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 64
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 8
batch = int(sys.argv[4]) if len(sys.argv) > 4 else 32
images = int(sys.argv[5]) if len(sys.argv) > 5 else 2
w = int(sys.argv[6]) if len(sys.argv) > 6 else 1920
h = int(sys.argv[7]) if len(sys.argv) > 7 else 1080
b = W.WorldBridge()
b.loadScene(scene)
r = W.WebGPURenderer(0)
r.buildPipeline(depth, 1)
W.upload_scene(r, b, w, h)
fl = list(range(1, frames + 1))
for _ in range(images):
    r.resetAccumulation()
    for i in range(0, frames, batch):
        r.computeBatch(fl[i:i + batch])
    r.present()
r.sync()
c = r.getCounters()
print("scene=%s %dx%d images=%d frames=%d batch=%d rays/image=%d" % (
    scene, w, h, images, frames, batch, (c["primary_rays"] + c["extension_rays"] + c["shadow_rays"]) // images))
