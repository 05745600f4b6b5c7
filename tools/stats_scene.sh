#!/bin/bash
# rocprofv3 kernel stats for one scene/variant: bash tools/stats_scene.sh <scene> <variant> [frames] [batch]
R=$GRAFT_REPO_ROOT; S=${1:-cornell}; V=${2:-2}; N=${3:-32}; B=${4:-32}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/st
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/st -- python3 $R/tools/prof_frames.py $S 1920 1080 $N 8 $V 0 1 $B > $R/gpurun_out/st.log 2>&1 || tail -3 $R/gpurun_out/st.log | cut -c1-300
tail -2 $R/gpurun_out/st.log | head -1 | cut -c1-200
f=$(find $R/gpurun_out/st -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-60s calls=%s total_ms=%.3f avg_ms=%.4f pct=%s" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e6, r["Percentage"]))
PY
