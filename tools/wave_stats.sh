#!/bin/bash
# diagnostic build: counters report wave-level events (node steps, 64-item chunks, outer trips)
python - <<'PY'
import webgpu_raytracer_amd as W
W._build.build_rt(force=True, extra_flags=["-DRT_WAVE_STATS"])
PY
python tools/prof_frames.py ${1:-cornell} 1920 1080 32 8 1 1 1 ${2:-32} 2>&1 | tail -2
python - <<'PY'
import webgpu_raytracer_amd as W
W._build.build_rt(force=True)
PY
