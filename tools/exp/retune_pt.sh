#!/bin/bash
# persistent kernel (Cornell): node steps per trip x flush threshold, re-swept on the final kernels (ms per 32-frame launch)
for st in 3 4 5 6; do for fl in 16 24 32 48; do
python - <<PY
import webgpu_raytracer_amd as W
W._build.build_rt(force=True, extra_flags=["-DRT_STEPS_PER_TRIP=$st", "-DRT_FLUSH_ITEMS=${fl}u"])
PY
echo -n "steps $st flush $fl: "; timeout -k 10 120 python tools/prof_frames.py cornell 1920 1080 128 8 3 0 1 32 2>&1 | tail -3 | head -1 | sed 's/.*pathtrace \([0-9.]*\) ms.*/\1/'
done; done
python - <<PY
import webgpu_raytracer_amd as W
W._build.build_rt(force=True)
PY
