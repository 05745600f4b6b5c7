set -o pipefail
mkdir -p gpurun_out/r04i
O=gpurun_out/r04i
timeout -k 10 900 python -m pytest tests/test_gpu_world_update.py tests/test_gpu_sharded.py -q -x -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 15 $O/pytest.log
timeout -k 10 300 python - > $O/tlas_time.log 2>&1 <<'PY'
import sys, time
sys.path.insert(0, '.')
import webgpu_raytracer_amd as pkg
for sc in ("cornell", "instanced1000", "instanced16384"):
    r = pkg.WebGPURenderer(0); r.buildPipeline(8, 1); r.setWorldStaticCache(False)
    b = pkg.WorldBridge(zero_copy=True); b.setDeviceUpdater(r); b.loadScene(sc); pkg.upload_scene(r, b, 640, 360)
    for _ in range(3): b.update(0.0)
    r.sync(); ta = tt = 0.0
    for _ in range(10):
        b.update(0.0); assert b.deviceResident, b.deviceWarning
        ta += r.worldLastMs(); tt += r.worldLastTlasMs()
    print(sc, len(b.instances)//36, "instances: k_tlas %.4f ms, whole device update %.4f ms" % (tt/10, ta/10))
    b.close(); r.destroy()
PY
cat $O/tlas_time.log | grep -v amdgpu.ids
