#!/bin/bash
# ray-queue binning on / off for one scene: bash tools/sweep_sort.sh <scene> [frames] [depth] [w] [h]
S=${1:-sponza_like}; F=${2:-32}; D=${3:-8}; W=${4:-1920}; H=${5:-1080}
for sort in 0 1; do
  echo "== MI355RT_WF_SORT=$sort"
  MI355RT_WF_SORT=$sort python tools/prof_frames.py $S $W $H $F $D 3 0 1 $F 2>&1 | tail -3 | head -2
done
