#!/bin/bash
# timing-only experiments (results are wrong on purpose): which phase costs what
for f in "" "-DRT_EXP_NOSHADOW" "-DRT_EXP_NOEXT" "-DRT_EXP_NOSHADOW -DRT_EXP_NOEXT"; do
python - <<PY
import webgpu_raytracer_amd as W
W._build.build_rt(force=True, extra_flags="$f".split())
PY
echo "flags: $f"; python tools/prof_frames.py cornell 1920 1080 64 8 1 0 1 32 2>&1 | tail -2 | head -1 | cut -d: -f2
done
python - <<PY
import webgpu_raytracer_amd as W
W._build.build_rt(force=True)
PY
