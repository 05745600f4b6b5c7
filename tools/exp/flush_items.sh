#!/bin/bash
for k in 8 16 24 40; do
python - <<PY
import webgpu_raytracer_amd as W
W._build.build_rt(force=True, extra_flags=["-DRT_FLUSH_ITEMS=${k}u"])
PY
echo "== RT_FLUSH_ITEMS=$k"
python tools/prof_frames.py cornell 1920 1080 128 8 3 0 1 32 2>&1 | tail -3 | head -1 | cut -c1-130
python tools/prof_frames.py sponza_like 1920 1080 32 8 3 0 1 32 2>&1 | tail -2 | head -1
done
python - <<PY
import webgpu_raytracer_amd as W
W._build.build_rt(force=True)
PY
