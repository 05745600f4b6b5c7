#!/bin/bash
# VGPRs / SGPRs / spills / occupancy of every kernel of the HIP library (hipcc -Rpass-analysis=kernel-resource-usage).
cd "$(dirname "$0")/../webgpu-raytracer_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -fno-gpu-rdc -fno-slp-vectorize -I ../../include "$@" -Rpass-analysis=kernel-resource-usage -o /tmp/rt_resources.so rt_api.hip 2>&1 | python3 -c '
import re, sys, subprocess
rows = []; cur = None
for line in sys.stdin:
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        name = t.split(":", 1)[1].strip()
        try: name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        except Exception: pass
        cur = {"name": re.sub(r"\(.*", "", name).replace("void ", "")}; rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
print("%-62s %5s %5s %6s %6s %4s %8s" % ("kernel", "VGPR", "AGPR", "SGPR", "spill", "occ", "scratch"))
for r in rows:
    print("%-62s %5s %5s %6s %6s %4s %8s" % (r["name"][:62], r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("VGPRs Spill", r.get("VGPR Spill")), r.get("Occupancy [waves/SIMD]"), r.get("ScratchSize [bytes/lane]")))
'
