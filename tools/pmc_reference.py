#!/usr/bin/env python3
"""profiles/pmc_reference.json from the per-kernel PMC averages of tools/make_profiles_round.sh (profiles/<RTAG>_*_pmc.json),
the in-kernel clock check and the VALU peak microbenchmark.  bench.py reads it for the fields it cannot measure from
inside its own process; every such field is reported with this file's "source".
usage: pmc_reference.py <dir with <RTAG>_*_pmc.json etc.> <commit> [out.json]"""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import webgpu_raytracer_amd as _W  # noqa: E402  (the hash of the kernel sources the counters were collected from)

d, commit = sys.argv[1], sys.argv[2]
TAG = os.environ.get("RTAG", "r04")
out_path = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_reference.json")


def load(name):
    p = os.path.join(d, name)
    return json.load(open(p)) if os.path.exists(p) else {}


def kernel(tab, prefix):
    for k, v in tab.items():
        if k.replace("rtk::", "").startswith(prefix):
            return v
    return {}


ref = {"source": "rocprofv3 --pmc passes of tools/make_profiles_round.sh (one counter set per run), collected at commit %s; "
                 "per-kernel averages in profiles/%s_*_pmc.json%s" % (commit, TAG, os.environ.get("PMC_SOURCE_NOTE", "")),
       "csrc_sha16": _W._build.kernel_source_hash(),
       "notes": {"FETCH_SIZE": "KB; doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B) - calibrated for wide "
                               "coalesced reads only, so both the raw and the doubled figure are given",
                 "WRITE_SIZE": "KB, exact for 16-B-per-lane stores",
                 "lane_instr": "SQ_THREAD_CYCLES_VALU = active lanes summed over the VALU instructions of the launch"}}
c = load(TAG + "_cornell_pmc.json")
pt = kernel(c, "k_pathtrace_persistent")
if pt:
    lane = pt.get("SQ_THREAD_CYCLES_VALU")
    entry = {"frames_per_launch": 32, "dispatches_averaged": pt.get("dispatches"),
             "insts_valu_per_launch": pt.get("SQ_INSTS_VALU"), "insts_salu_per_launch": pt.get("SQ_INSTS_SALU"),
             "insts_lds_per_launch": pt.get("SQ_INSTS_LDS"), "lane_instr_per_launch": lane}
    if pt.get("SQ_ACTIVE_INST_VALU"):
        entry["valu_lane_utilization"] = round(lane / (pt["SQ_ACTIVE_INST_VALU"] * 64.0), 4)
    if pt.get("SQ_WAVE_CYCLES"):
        for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU"):
            if pt.get(k):
                entry["frac_of_wave_cycles_" + k[3:].lower()] = round(pt[k] / pt["SQ_WAVE_CYCLES"], 4)
    if pt.get("SQ_WAVE_CYCLES") and pt.get("SQ_ACTIVE_INST_VALU") and pt.get("SQ_WAVES") and pt.get("SQ_BUSY_CYCLES"):
        # VALU-pipe busy: wave-cycles with a vector instruction in flight per wave-cycle, times the waves resident on a SIMD
        # (the persistent kernel keeps 4 there from start to end) = vector instructions in flight per SIMD cycle
        entry["valu_busy_per_simd"] = round(4.0 * pt["SQ_ACTIVE_INST_VALU"] / pt["SQ_WAVE_CYCLES"], 4)
        entry["valu_busy_note"] = ("4 resident waves x SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES: vector instructions in flight per SIMD "
                                   "cycle; >= 1 means the VALU pipe of a SIMD is never idle for lack of a ready instruction")
    if pt.get("SQ_LDS_BANK_CONFLICT") and pt.get("SQ_ACTIVE_INST_LDS"):
        entry["lds_bank_conflict_frac_of_lds_cycles"] = round(pt["SQ_LDS_BANK_CONFLICT"] / pt["SQ_ACTIVE_INST_LDS"], 4)
    if pt.get("FETCH_SIZE") is not None and pt.get("WRITE_SIZE") is not None:
        entry["fetch_bytes_raw"] = pt["FETCH_SIZE"] * 1024.0
        entry["fetch_bytes_doubled"] = pt["FETCH_SIZE"] * 2048.0
        entry["write_bytes"] = pt["WRITE_SIZE"] * 1024.0
        entry["hbm_bytes_per_launch"] = pt["FETCH_SIZE"] * 2048.0 + pt["WRITE_SIZE"] * 1024.0
    clk = load(TAG + "_clock_check.json")
    if clk:
        entry["effective_clock_GHz"] = clk.get("in_kernel_clock_GHz_median")
        entry["cycles_per_workgroup"] = clk.get("cycles_per_workgroup_median")
    vp = os.path.join(d, TAG + "_valu_peak.txt")
    if os.path.exists(vp):
        best = max(float(m) for m in re.findall(r"([0-9.]+) T lane-instr/s", open(vp).read()))
        entry["measured_issue_peak_Tlane"] = best
    ref["k_pathtrace_persistent"] = entry
tr = {}
# scene -> (frames in the profiled batch, frames of one image of the BASELINE config)
PROFILED = {"sponza_like": (32, 64), "instanced1000": (32, 64), "glass_blob": (8, 256)}
for scene in ("sponza_like", "instanced1000", "glass_blob"):
    t = load(TAG + "_%s_pmc.json" % scene)
    rows = {k: v for k, v in t.items() if "k_wf_trace" in k}
    if not rows:
        continue
    e = {"frames_per_image_profiled": PROFILED[scene][0], "frames_per_image": PROFILED[scene][1], "kernels": {}}
    hbm = 0.0
    for k, v in rows.items():
        name = "any_hit" if re.search(r"k_wf_trace(_pairs)?<true", k) else "closest_hit"
        if "k_wf_trace_pairs" in k:
            e["walk"] = "pairs"
        n = v.get("dispatches", 0)
        ke = {"dispatches": n}
        if v.get("FETCH_SIZE") is not None and v.get("WRITE_SIZE") is not None:
            ke["hbm_bytes_per_dispatch"] = v["FETCH_SIZE"] * 2048.0 + v["WRITE_SIZE"] * 1024.0
            hbm += ke["hbm_bytes_per_dispatch"] * n
        if v.get("TCC_HIT_sum") is not None:
            ke["l2_hit_rate"] = round(v["TCC_HIT_sum"] / max(1.0, v["TCC_HIT_sum"] + v["TCC_MISS_sum"]), 4)
        if v.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
            ke["l1_accesses_per_dispatch"] = v["TCP_TOTAL_CACHE_ACCESSES_sum"]
            ke["l1_miss_rate"] = round(v.get("TCP_TCC_READ_REQ_sum", 0.0) / v["TCP_TOTAL_CACHE_ACCESSES_sum"], 4)
        if v.get("TCP_TCC_READ_REQ_sum") and v.get("TCP_TCC_READ_REQ_LATENCY_sum"):
            ke["l2_read_latency_cycles"] = round(v["TCP_TCC_READ_REQ_LATENCY_sum"] / v["TCP_TCC_READ_REQ_sum"], 1)
        if v.get("GRBM_GUI_ACTIVE") and v.get("TCP_PENDING_STALL_CYCLES_sum"):
            cyc = v["GRBM_GUI_ACTIVE"] / 8.0
            ke["l1_pending_stall_frac"] = round(v["TCP_PENDING_STALL_CYCLES_sum"] / 256.0 / cyc, 4)
            ke["ta_busy_frac"] = round(v.get("TA_BUSY_avr", 0.0) / cyc, 4)
        if v.get("SQ_WAVE_CYCLES"):
            ke["wave_cycles_waiting_frac"] = round(v.get("SQ_WAIT_ANY", 0.0) / v["SQ_WAVE_CYCLES"], 4)
            if v.get("SQ_ACTIVE_INST_VALU"):
                ke["valu_lane_utilization"] = round(v["SQ_THREAD_CYCLES_VALU"] / (v["SQ_ACTIVE_INST_VALU"] * 64.0), 4)
        e["kernels"][name] = ke
    if hbm:
        e["hbm_bytes_per_image"] = hbm * PROFILED[scene][1] / PROFILED[scene][0]   # one profiled batch scaled to the image's frames
    tr[scene] = e
if tr:
    ref["k_wf_trace"] = tr
json.dump(ref, open(out_path, "w"), indent=1)
print(json.dumps(ref, indent=1)[:3000])
