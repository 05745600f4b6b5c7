#!/usr/bin/env python3
"""Static instruction census of one kernel's assembly (tools/exp/isa_dump.sh) by innermost loop.  usage: isa_loops.py gpurun_out/isa/pt_lds.s"""
import collections
import re
import sys

cur = None
cnt = collections.OrderedDict()
kinds = {}
for line in open(sys.argv[1]):
    m = re.match(r"^(\.LBB\d+_\d+):\s*;\s*(.*)$", line) or re.match(r"^; %bb\.(\d+):\s*;\s*(.*)$", line)
    if m:
        c = m.group(2)
        h = re.search(r"Header=(BB\d+_\d+) Depth=(\d+)", c)
        if "Loop Header" in c:
            name = m.group(1).lstrip(".L")
            d = re.search(r"Depth=(\d+)", c)
            cur = (name, int(d.group(1)) if d else 0)
        elif h:
            cur = (h.group(1), int(h.group(2)))
        else:
            cur = ("(outside loops)", 0)
        continue
    if re.match(r"^\.LBB\d+_\d+:", line) or re.match(r"^; %bb\.\d+:", line):
        cur = ("(outside loops)", 0)
        continue
    m = re.match(r"\s+(v_|s_|ds_|global_|scratch_|buffer_|flat_)(\w+)", line)
    if m and cur:
        k = cnt.setdefault(cur, collections.Counter())
        t = {"v_": "valu", "s_": "salu", "ds_": "lds"}.get(m.group(1), "mem")
        if m.group(1) == "s_" and m.group(2).startswith(("waitcnt", "nop")):
            t = "wait/nop"
        k[t] += 1
for (name, d), k in cnt.items():
    print("%-20s depth %d: %5d valu %5d salu %4d lds %4d mem %4d wait/nop" % (name, d, k["valu"], k["salu"], k["lds"], k["mem"], k["wait/nop"]))
