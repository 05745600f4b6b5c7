#!/bin/bash
for n in 1u 8u 16u 24u; do
python - <<PY
import webgpu_raytracer_amd as W
W._build.build_rt(force=True, extra_flags=["-DRT_FLUSH_ITEMS=$n"])
PY
echo "flush at $n"; python tools/prof_frames.py cornell 1920 1080 16 8 1 0 2>&1 | tail -2 | head -1
python tools/prof_frames.py instanced1000 1920 1080 4 8 1 0 2>&1 | tail -2 | head -1
done
python - <<PY
import webgpu_raytracer_amd as W
W._build.build_rt(force=True)
PY
