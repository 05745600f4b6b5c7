#!/bin/bash
# L2 / fabric PMC pass for one scene/variant: bash tools/pmc_l2.sh <scene> <variant>
R=$GRAFT_REPO_ROOT; S=${1:-sponza_like}; V=${2:-2}
cd /tmp && export TMPDIR=/tmp
for set in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "FETCH_SIZE TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_LEVEL_sum"; do
rm -rf $R/gpurun_out/pmc_m
timeout -k 10 90 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_m -- python3 $R/tools/prof_frames.py $S 1920 1080 8 8 $V 0 1 8 > $R/gpurun_out/pmc_m.log 2>&1 || tail -3 $R/gpurun_out/pmc_m.log | cut -c1-300
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_m | grep -v prepare
done
