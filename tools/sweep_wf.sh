#!/bin/bash
# sweep the workgroup shape of the wavefront trace kernels on a scene: bash tools/sweep_wf.sh <scene> [frames] [depth]
S=${1:-sponza_like}; F=${2:-32}; D=${3:-8}
for cfg in "256 6" "256 4" "512 3" "512 2" "512 1" "1024 1"; do
  set -- $cfg
  echo "== block $1 x $2 per CU"
  MI355RT_WF_BLOCK=$1 MI355RT_WF_BLOCKS_PER_CU=$2 python tools/prof_frames.py $S 1920 1080 $F $D 3 0 1 $F 2>&1 | tail -2 | head -1
done
