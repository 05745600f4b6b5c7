#!/usr/bin/env python3
"""Per-kernel averages of the rocprofv3 --pmc passes under a directory -> one JSON object
{kernel: {counter: average per dispatch, "dispatches": n, "vgpr": .., "lds": .., "grid": ..}}.
usage: pmc_collect.py <dir with pass sub-directories> [out.json]"""
import collections
import csv
import glob
import json
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for f in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "rtk::" not in name and "bvhb::" not in name:
            continue
        k = name.split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[k] = {"vgpr": r.get("VGPR_Count"), "lds": r.get("LDS_Block_Size"), "grid": r.get("Grid_Size")}
out = {}
for k, v in agg.items():
    out[k] = {c: sum(x) / len(x) for c, x in v.items()}
    out[k]["dispatches"] = max(len(x) for x in v.values())
    out[k].update(meta[k])
print(json.dumps(out, indent=1))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
