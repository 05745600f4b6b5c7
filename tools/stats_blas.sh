#!/bin/bash
# rocprofv3 kernel stats of the GPU BLAS build (tools/blas_build_time.py)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/st_blas
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/st_blas -- python3 $R/tools/blas_build_time.py > $R/gpurun_out/st_blas.log 2>&1 || tail -3 $R/gpurun_out/st_blas.log | cut -c1-300
grep triangles $R/gpurun_out/st_blas.log
f=$(find $R/gpurun_out/st_blas -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-40s calls=%s total_ms=%.3f avg_us=%.2f max_us=%.1f pct=%s" % (r["Name"].split("(")[0][-40:], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, float(r["MaxNs"])/1e3, r["Percentage"]))
PY
