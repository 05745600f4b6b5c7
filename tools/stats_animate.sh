#!/bin/bash
# rocprofv3 kernel stats of the animated live loop (tools/animate_bench.py)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/st_anim
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/st_anim -- python3 $R/tools/animate_bench.py 512 256 10 > $R/gpurun_out/st_anim.log 2>&1 || tail -3 $R/gpurun_out/st_anim.log | cut -c1-300
grep triangles $R/gpurun_out/st_anim.log
f=$(find $R/gpurun_out/st_anim -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print("%-44s calls=%s total_ms=%.3f avg_us=%.2f max_us=%.1f pct=%s" % (r["Name"].split("(")[0][-44:], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, float(r["MaxNs"])/1e3, r["Percentage"]))
PY
