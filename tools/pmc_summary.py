#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per kernel from a counter_collection.csv directory."""
import collections
import csv
import glob
import sys

for f in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-48:]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[k] = (r["VGPR_Count"], r["LDS_Block_Size"], r["Grid_Size"])
    for k, v in agg.items():
        if "rtk::" not in k:
            continue
        avg = {c: sum(x) / len(x) for c, x in v.items()}
        line = "%s vgpr=%s lds=%s grid=%s " % ((k,) + meta[k])
        line += " ".join("%s=%.3g" % (c, a) for c, a in sorted(avg.items()))
        if "SQ_THREAD_CYCLES_VALU" in avg and avg.get("SQ_ACTIVE_INST_VALU"):
            line += " | VALU lane util=%.1f%%" % (100 * avg["SQ_THREAD_CYCLES_VALU"] / (avg["SQ_ACTIVE_INST_VALU"] * 64))
        if "SQ_WAVE_CYCLES" in avg and avg["SQ_WAVE_CYCLES"]:
            for c in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"):
                if c in avg:
                    line += " %s=%.0f%%" % (c[3:], 100 * avg[c] / avg["SQ_WAVE_CYCLES"])
        print(line)
