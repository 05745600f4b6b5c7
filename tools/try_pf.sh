#!/bin/bash
# A/B: DFS-next node prefetch (RT_PF) and forced waves/SIMD of the wavefront trace kernel (RT_WF_WAVES)
for f in "-DRT_PF=0 -DRT_WF_WAVES=6" "-DRT_PF=1 -DRT_WF_WAVES=6" "-DRT_PF=0 -DRT_WF_WAVES=8" "-DRT_PF=0 -DRT_WF_WAVES=7"; do
python - <<PY
import webgpu_raytracer_amd as W
W._build.build_rt(force=True, extra_flags="$f".split())
PY
echo "flags: $f"
for v in 1 2; do for s in sponza_like instanced1000; do python tools/prof_frames.py $s 1920 1080 16 8 $v 0 1 8 2>&1 | tail -2 | head -1 | cut -d: -f1,2 | cut -c1-120; done; done
done
python - <<PY
import webgpu_raytracer_amd as W
W._build.build_rt(force=True)
PY
