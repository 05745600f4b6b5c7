#!/bin/bash
for n in 16u 24u 32u 48u 64u; do
python - <<PY
import webgpu_raytracer_amd as W
W._build.build_rt(force=True, extra_flags=["-DRT_FLUSH_ITEMS=$n"])
PY
echo "flush at $n"; python tools/prof_frames.py cornell 1920 1080 64 8 1 0 1 32 2>&1 | tail -2 | head -1 | cut -d: -f2
done
python - <<PY
import webgpu_raytracer_amd as W
W._build.build_rt(force=True)
PY
