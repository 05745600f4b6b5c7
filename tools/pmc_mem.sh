#!/bin/bash
# memory-pipe PMC pass for one scene/variant: bash tools/pmc_mem.sh <scene> <variant>
R=$GRAFT_REPO_ROOT; S=${1:-sponza_like}; V=${2:-2}
cd /tmp && export TMPDIR=/tmp
for set in "TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCP_GATE_EN1_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum"; do
rm -rf $R/gpurun_out/pmc_m
timeout -k 10 90 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_m -- python3 $R/tools/prof_frames.py $S 1920 1080 8 8 $V 0 1 8 > $R/gpurun_out/pmc_m.log 2>&1 || tail -3 $R/gpurun_out/pmc_m.log
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_m | grep -v prepare
done
