#!/usr/bin/env python3
"""An animated, skinned glTF through the live loop (src/main.ts:119-181): per displayed frame the world advances
(animation, skinning, BLAS + TLAS rebuild), the scene is re-uploaded, one 1-spp frame is traced and presented.
Compares the scene compiler's CPU BLAS builder, the GPU builder hook (host arrays, re-uploaded) and the device-resident
update (rt_world_update: nothing leaves HBM).  usage: animate_bench.py [nu] [nv] [frames]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import webgpu_raytracer_amd as W  # noqa: E402
import test_gltf  # noqa: E402

nu = int(sys.argv[1]) if len(sys.argv) > 1 else 512
nv = int(sys.argv[2]) if len(sys.argv) > 2 else 256
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 20
what = "%d triangles skinned + animated"
if os.environ.get("SCENE") == "hall":      # a small animated character inside a large static mesh (the static cache's case)
    glb, n_static, n_skinned = test_gltf.character_in_hall_glb(W, (nu, nv))
    n_tris = n_static + n_skinned
    what = "%d triangles (" + "%d static + %d skinned, animated)" % (n_static, n_skinned)
else:
    glb, n_tris = test_gltf.big_skinned_glb(W, nu, nv)
for mode in os.environ.get("MODES", "cpu gpu-blas device").split():
    use_gpu = mode != "cpu"
    r = W.WebGPURenderer(0)
    r.buildPipeline(8, 1)
    if "VARIANT" in os.environ:
        r.setKernelVariant(int(os.environ["VARIANT"]))     # 1 persistent, 2 wavefront, 3 auto
    b = W.WorldBridge(zero_copy=True)
    if mode == "gpu-blas":
        b.setBlasBuilder(r)
    elif mode == "device":
        b.setDeviceUpdater(r)
    b.loadScene("viewer", glbData=glb)
    W.upload_scene(r, b, 1920, 1080)
    loop = W.LiveLoop(r, b, 1920, 1080, update_interval=1)
    loop.render_frame()
    r.sync()
    t_upd = t_sync = t_trace = t_dev = 0.0
    for _ in range(3):                       # the device path learns the trees' depths on its first updates
        loop.totalFrameCount += 1
        b.update(loop.totalFrameCount / 60)
        W.sync_world(r, b, 1920, 1080)
        r.compute(1)
        r.sync()
    for _ in range(frames):
        t0 = time.perf_counter()
        b.update(loop.totalFrameCount / 60)
        t1 = time.perf_counter()
        if mode == "device":
            assert b.deviceResident, b.deviceWarning
            t_dev += r.worldLastMs()
        W.sync_world(r, b, 1920, 1080)
        t2 = time.perf_counter()
        loop.frameCount = 1
        loop.totalFrameCount += 1
        r.compute(1)
        r.present()
        r.sync()
        t3 = time.perf_counter()
        t_upd += t1 - t0
        t_sync += t2 - t1
        t_trace += t3 - t2
    f = frames / 1e3
    print((what + ", 1920x1080, %s: update(t) %.2f ms%s, re-upload / sync %.2f ms, trace+present %.2f ms "
           "-> %.1f frames/s") % (n_tris, {"cpu": "CPU BLAS builder", "gpu-blas": "GPU BLAS builder hook (rt_build_blas)",
                                         "device": "device-resident update (rt_world_update)"}[mode], t_upd / f,
                                " (%.2f ms of it on the GPU stream)" % (t_dev / frames) if mode == "device" else "", t_sync / f, t_trace / f,
                                frames / (t_upd + t_sync + t_trace)))
    r.destroy()
