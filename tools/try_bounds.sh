#!/bin/bash
# time the persistent kernel at several forced occupancies (waves/SIMD)
for w in 3 4 5; do
python - <<PY
import re, webgpu_raytracer_amd as W
p="webgpu-raytracer_amd/csrc/kernels.hip.h"
s=open(p).read()
s2=re.sub(r"__launch_bounds__\(256, \d+\) void k_pathtrace_persistent","__launch_bounds__(256, $w) void k_pathtrace_persistent",s)
open(p,"w").write(s2)
W._build.build_rt(force=True)
PY
echo "waves/SIMD=$w"; python tools/prof_frames.py cornell 1920 1080 16 8 1 0 2>&1 | tail -2 | head -1
python tools/prof_frames.py instanced1000 1920 1080 4 8 1 0 2>&1 | tail -2 | head -1
done
