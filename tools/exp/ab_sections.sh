#!/bin/bash
# A/B of the trace-loop section stamps on ONE box (see ab_run.sh).  usage: ab_sections.sh out_dir scene frames depth
O=$1; mkdir -p $O
C=webgpu-raytracer_amd/csrc
mkdir -p /tmp/ab_keep; for f in tools/exp/ab/*.A; do n=$(basename $f .A); cp $C/$n /tmp/ab_keep/$n; done
for f in tools/exp/ab/*.A; do n=$(basename $f .A); cp $f $C/$n; done
echo "== A" >> $O/sections.txt; timeout -k 10 400 python tools/trace_sections.py $2 $3 $4 $5 $6 2>&1 | grep -v "warning\|amdgpu.ids\|\^\|^ *[0-9]* |" >> $O/sections.txt
for f in tools/exp/ab/*.A; do n=$(basename $f .A); cp /tmp/ab_keep/$n $C/$n; done
echo "== B" >> $O/sections.txt; timeout -k 10 400 python tools/trace_sections.py $2 $3 $4 $5 $6 2>&1 | grep -v "warning\|amdgpu.ids\|\^\|^ *[0-9]* |" >> $O/sections.txt
cat $O/sections.txt
