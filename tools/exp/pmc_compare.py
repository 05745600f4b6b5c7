#!/usr/bin/env python3
"""Side-by-side of two per-kernel PMC JSON files (tools/pmc_collect.py): usage pmc_compare.py new.json old.json [kernel substring]"""
import json
import sys
new, old = json.load(open(sys.argv[1])), json.load(open(sys.argv[2]))
sub = sys.argv[3] if len(sys.argv) > 3 else "wf_trace"
for k, v in new.items():
    if sub not in k:
        continue
    o = old.get(k, {})
    print(k, "vgpr", v.get("vgpr"), "lds", v.get("lds"), "grid", v.get("grid"), "dispatches", v.get("dispatches"))
    for c in ("GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU",
              "SQ_THREAD_CYCLES_VALU", "TA_BUSY_avr", "TA_FLAT_READ_WAVEFRONTS_sum", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum",
              "TCP_PENDING_STALL_CYCLES_sum", "TCP_TOTAL_ACCESSES_sum"):
        if c in v:
            print("   %-32s new %.4g  old %.4g  ratio %.2f" % (c, v[c], o.get(c, 0), v[c] / max(o.get(c, 1), 1)))

    def f(d, a, b, scale=1.0):
        return d[a] / d[b] * scale if d.get(a) and d.get(b) else float("nan")
    for name, a, b, sc in (("TA busy fraction", "TA_BUSY_avr", "GRBM_GUI_ACTIVE", 8.0), ("VALU lane utilisation", "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU", 1 / 64.0),
                           ("wave cycles waiting", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES", 1.0), ("wave cycles with a VALU instruction", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", 1.0)):
        print("   %-36s new %.3f  old %.3f" % (name, f(v, a, b, sc), f(o, a, b, sc)))
