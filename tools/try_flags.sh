#!/bin/bash
# A/B compiler flags for the renderer library
for f in "" "-mllvm -amdgpu-sched-strategy=max-ilp" "-mllvm -amdgpu-sched-strategy=max-memory-clause" "-fno-unroll-loops" "-mllvm -amdgpu-use-divergent-register-indexing" ; do
python - <<PY
import webgpu_raytracer_amd as W
try:
    W._build.build_rt(force=True, extra_flags="$f".split())
except Exception as e:
    print("build failed", e)
PY
echo "flags: $f"; python tools/prof_frames.py cornell 1920 1080 16 8 1 0 2>&1 | tail -2 | head -1
done
python - <<PY
import webgpu_raytracer_amd as W
W._build.build_rt(force=True)
PY
