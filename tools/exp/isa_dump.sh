#!/bin/bash
# gfx950 assembly of the library's kernels -> gpurun_out/isa/rt_api.s, and the body of one kernel (default: the headline
# k_pathtrace_persistent<false, true>) -> gpurun_out/isa/<tag>.s with an instruction census.  usage: isa_dump.sh [tag] [mangled-substring] [extra flags]
cd "$(dirname "$0")/../.." || exit 1
TAG=${1:-pt_lds}; SUB=${2:-k_pathtrace_persistentILb0ELb1E}; [ $# -ge 2 ] && shift 2 || shift $#
mkdir -p gpurun_out/isa
(cd webgpu-raytracer_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-gpu-rdc \
  -fno-slp-vectorize -I ../../include --cuda-device-only -S "$@" -o ../../gpurun_out/isa/rt_api.s rt_api.hip 2>&1 | grep -E "error" )
awk -v sub_="$SUB" '$0 ~ "^_Z.*"sub_".*:" {on=1} on {print} on && /^\.Lfunc_end/ {exit}' gpurun_out/isa/rt_api.s > gpurun_out/isa/$TAG.s
echo "$TAG: $(grep -c '^\s*v_' gpurun_out/isa/$TAG.s) VALU, $(grep -c '^\s*s_' gpurun_out/isa/$TAG.s) SALU (incl. waitcnt/branches), $(grep -c '^\s*ds_' gpurun_out/isa/$TAG.s) LDS, $(grep -c 'scratch_' gpurun_out/isa/$TAG.s) scratch instructions (static)"
grep -E "NumVgprs|ScratchSize|Occupancy" gpurun_out/isa/$TAG.s | head -4
