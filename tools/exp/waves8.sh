#!/bin/bash
python - <<'PY'
import webgpu_raytracer_amd as W
W._build.build_rt(force=True, extra_flags=["-DRT_WF_WAVES=8"])
PY
for s in sponza_like instanced1000; do MI355RT_WF_BLOCKS_PER_CU=8 python tools/prof_frames.py $s 1920 1080 32 8 3 0 1 32 | tail -3 | head -2; done
python - <<'PY'
import webgpu_raytracer_amd as W
W._build.build_rt(force=True)
PY
for s in sponza_like instanced1000; do python tools/prof_frames.py $s 1920 1080 32 8 3 0 1 32 | tail -3 | head -2; done
