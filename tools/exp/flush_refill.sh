#!/bin/bash
# flush threshold / refill threshold re-sweep with 4 node steps per trip
for cfg in "24 16" "32 16" "48 16" "24 24" "24 32" "32 24"; do
set -- $cfg
python - <<PY
import webgpu_raytracer_amd as W
W._build.build_rt(force=True, extra_flags=["-DRT_FLUSH_ITEMS=$1u", "-DRT_WF_REFILL=$2"])
PY
echo "== RT_FLUSH_ITEMS=$1 RT_WF_REFILL=$2"
python tools/prof_frames.py cornell 1920 1080 128 8 3 0 1 32 2>&1 | tail -3 | head -1 | cut -c1-120
for s in sponza_like instanced1000; do python tools/prof_frames.py $s 1920 1080 32 8 3 0 1 32 2>&1 | tail -2 | head -1; done
done
python - <<PY
import webgpu_raytracer_amd as W
W._build.build_rt(force=True)
PY
