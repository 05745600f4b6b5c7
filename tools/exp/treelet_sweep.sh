#!/bin/bash
S=${1:-sponza_like}
for cfg in "256 6 0" "256 6 100000" "512 3 0" "512 3 100000" "1024 1 0" "1024 1 1024" "1024 1 100000"; do
  set -- $cfg
  echo "== block $1 x $2 per CU, treelet cap $3"
  MI355RT_WF_BLOCK=$1 MI355RT_WF_BLOCKS_PER_CU=$2 MI355RT_TREELET_MAX=$3 python tools/prof_frames.py $S 1920 1080 32 8 3 0 1 32 2>&1 | tail -3 | head -2
done
