#!/bin/bash
for k in 3 4; do
python - <<PY
import webgpu_raytracer_amd as W
W._build.build_rt(force=True, extra_flags=["-DRT_TRI_STRIDE=$k"])
PY
echo "== RT_TRI_STRIDE=$k"
python tools/prof_frames.py cornell 1920 1080 128 8 3 0 1 32 2>&1 | tail -3 | head -1 | cut -c1-130
for s in sponza_like instanced1000; do python tools/prof_frames.py $s 1920 1080 32 8 3 0 1 32 2>&1 | tail -2 | head -1; done
done
python - <<PY
import webgpu_raytracer_amd as W
W._build.build_rt(force=True)
PY
