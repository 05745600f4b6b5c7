#!/bin/bash
# A/B on ONE box: the files tools/exp/ab/<name>.A replace webgpu-raytracer_amd/csrc/<name> for variant A, the tree's files are variant B.
# usage (on the GPU box only: it overwrites sources of the scratch copy): ab_run.sh out_dir "scene w h frames depth variant;..."
O=$1; RUNS=$2; mkdir -p $O
C=webgpu-raytracer_amd/csrc
build() { python -c "import webgpu_raytracer_amd._build as b; b.build_rt(force=True)" > /dev/null 2>&1 || echo BUILD FAILED; }
run() { echo "$RUNS" | tr ';' '\n' | while read s w h f d v; do timeout -k 10 300 python tools/prof_frames.py $s $w $h $f $d $v 0 1 32 2>&1 | grep -E "kernel ms" | sed "s/.*kernel ms (sum over the run): /$1 $s /"; done; }
mkdir -p /tmp/ab_keep; for f in tools/exp/ab/*.A; do n=$(basename $f .A); cp $C/$n /tmp/ab_keep/$n; done
for rep in 1 2; do
  for f in tools/exp/ab/*.A; do n=$(basename $f .A); cp $f $C/$n; done; build; run A >> $O/ab.txt
  for f in tools/exp/ab/*.A; do n=$(basename $f .A); cp /tmp/ab_keep/$n $C/$n; done; build; run B >> $O/ab.txt
done
cat $O/ab.txt
