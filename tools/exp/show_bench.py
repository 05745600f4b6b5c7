import json, sys
d = json.load(open(sys.argv[1]))
print("value", d["value"], "ms/step", d["ms_per_step"], "launch ms", d["roofline"]["avg_launch_ms"], "primary", d["roofline"]["avg_primary_ms"])
for c in d.get("configs", []):
    print("  ", c["config"][:20], c["ms_per_image"], c["Mrays_s"], c["kernel_ms_per_image"])
if "cpu_baseline" in d: print("  cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["one_thread_value"])
