set -o pipefail
mkdir -p gpurun_out/r04c
O=gpurun_out/r04c
timeout -k 10 900 python -m pytest tests/test_gpu_product_build.py -q > $O/pytest_nowait.log 2>&1; echo "pytest rc=$?" >> $O/pytest_nowait.log
grep -E "FAILED|passed|failed" $O/pytest_nowait.log | tail -40
python - <<PY
import webgpu_raytracer_amd as W
W._build.build_rt(force=True, extra_flags=["-DRT_ANY_NOWAIT=0"])
PY
timeout -k 10 900 python -m pytest tests/test_gpu_product_build.py -q > $O/pytest_wait.log 2>&1; echo "pytest rc=$?" >> $O/pytest_wait.log
grep -E "FAILED|passed|failed" $O/pytest_wait.log | tail -40
for i in 1 2; do timeout -k 10 120 python tools/prof_frames.py cornell 1920 1080 64 8 1 0 1 32 >> $O/time_cornell_wait.log 2>&1; done
MI355RT_WALK=node timeout -k 10 200 python tools/prof_frames.py sponza_like 1920 1080 32 8 3 0 1 32 >> $O/time_cornell_wait.log 2>&1
grep -E "scene=|kernel ms" $O/time_cornell_wait.log
