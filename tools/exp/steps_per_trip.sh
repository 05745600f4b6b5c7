#!/bin/bash
for k in 8 12; do
python - <<PY
import webgpu_raytracer_amd as W
W._build.build_rt(force=True, extra_flags=["-DRT_STEPS_PER_TRIP=$k"])
PY
echo "== RT_STEPS_PER_TRIP=$k"
python tools/prof_frames.py cornell 1920 1080 128 8 3 0 1 32 2>&1 | tail -3 | head -1
done
for k in 4 6 8; do
python - <<PY
import webgpu_raytracer_amd as W
W._build.build_rt(force=True, extra_flags=["-DRT_WF_STEPS_PER_TRIP=$k", "-DRT_STEPS_PER_TRIP=4"])
PY
echo "== RT_WF_STEPS_PER_TRIP=$k"
for s in sponza_like instanced1000; do python tools/prof_frames.py $s 1920 1080 32 8 3 0 1 32 2>&1 | tail -3 | head -2; done
done
python - <<PY
import webgpu_raytracer_amd as W
W._build.build_rt(force=True)
PY
