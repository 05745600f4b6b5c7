import sys, time; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import webgpu_raytracer_amd as W
b = W.WorldBridge(); b.loadScene("cornell")
r = W.WebGPURenderer(0); r.buildPipeline(8, 1); W.upload_scene(r, b, 1920, 1080)
for f in range(1, 9):
    r.compute(f)
r.sync()
for fc in (8, 40):
    r.updateSceneUniforms(b.cameraData, fc, b.lightCount)
    r.present(); r.sync()
    t0 = time.perf_counter()
    for _ in range(50):
        r.present()
    r.sync()
    print("present() at frame_count=%d: %.4f ms" % (fc, (time.perf_counter() - t0) / 50 * 1e3))
