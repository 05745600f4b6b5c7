#!/bin/bash
# HBM traffic of the path-trace kernel from the TCC PMC counters, separate passes
# (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2: they do not fit one pass; MI355X_MICROARCH.md §rocprofv3 PMC slots).
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  d=$R/gpurun_out/hbm_$(echo $c | tr ' ' '_')
  rm -rf $d
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- python3 $R/tools/prof_frames.py cornell 1920 1080 8 8 1 0 > $d.log 2>&1 || echo "pass $c failed"
  python3 $R/tools/pmc_summary.py $d | grep -v prepare
done
