#!/bin/bash
# timing-only: how much of the persistent kernel's time is the IEEE-exact arithmetic contract?  (results differ from the oracle)
for f in "" "-ffp-contract=fast" "-ffast-math -ffp-contract=fast" "-ffast-math -ffp-contract=fast -fslp-vectorize"; do
python - <<PY
import webgpu_raytracer_amd as W
W._build.build_rt(force=True, extra_flags="$f".split())
PY
echo "flags: $f"; python tools/prof_frames.py cornell 1920 1080 64 8 1 0 1 32 2>&1 | tail -2 | head -1 | cut -d: -f2 | cut -c1-110
done
python - <<PY
import webgpu_raytracer_amd as W
W._build.build_rt(force=True)
PY
