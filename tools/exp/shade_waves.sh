#!/bin/bash
for w in 2 3 4 5 6 8; do
python - <<PY
import webgpu_raytracer_amd as W
W._build.build_rt(force=True, extra_flags=["-DRT_SHADE_WAVES=$w"])
PY
echo "== RT_SHADE_WAVES=$w"
for s in sponza_like instanced1000; do python tools/prof_frames.py $s 1920 1080 32 8 3 0 1 32 2>&1 | tail -2 | head -1; done
done
python - <<PY
import webgpu_raytracer_amd as W
W._build.build_rt(force=True)
PY
