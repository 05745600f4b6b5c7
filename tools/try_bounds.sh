#!/bin/bash
# time the persistent kernel at several forced occupancies (waves/SIMD)
for w in 3 4 5 6; do
python - <<PY
import re, webgpu_raytracer_amd as W
p="webgpu-raytracer_amd/csrc/k_pathtrace.hip.h"
s=open(p).read()
s2=re.sub(r"__launch_bounds__\(256, \d+\) void k_pathtrace_persistent","__launch_bounds__(256, $w) void k_pathtrace_persistent",s)
open(p,"w").write(s2)
W._build.build_rt(force=True)
PY
echo "waves/SIMD=$w"
for s in sponza_like instanced1000 glass_blob; do python tools/prof_frames.py $s 1920 1080 3 8 1 0 2>&1 | tail -2 | head -1; done
done
