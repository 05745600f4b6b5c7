#!/usr/bin/env python3
"""Which use of a torch side stream aborts with std::bad_variant_access (seen in bench.py when kernel timing was on)?
Each case runs in its own process."""
import subprocess
import sys

CASES = {
    "A_torch_timing_event_on_side_stream": """
import torch
s = torch.cuda.Stream(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
with torch.cuda.stream(s):
    e0.record(s); x = torch.zeros(1<<20, device='cuda'); e1.record(s)
s.synchronize(); print('ms', e0.elapsed_time(e1))
""",
    "B_ctypes_timing_event_on_side_stream": """
import ctypes, torch, os
s = torch.cuda.Stream()
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so'))
ev = ctypes.c_void_p(); ev2 = ctypes.c_void_p()
print(hip.hipEventCreate(ctypes.byref(ev)), hip.hipEventCreate(ctypes.byref(ev2)))
print(hip.hipEventRecord(ev, ctypes.c_void_p(s.cuda_stream)))
with torch.cuda.stream(s):
    x = torch.zeros(1<<20, device='cuda')
print(hip.hipEventRecord(ev2, ctypes.c_void_p(s.cuda_stream)))
print(hip.hipEventSynchronize(ev2)); ms = ctypes.c_float(); print(hip.hipEventElapsedTime(ctypes.byref(ms), ev, ev2), ms.value)
""",
    "C_renderer_on_side_stream_with_timing": """
import sys; sys.path.insert(0, '.')
import torch
import webgpu_raytracer_amd as W
s = torch.cuda.Stream()
b = W.WorldBridge(); b.loadScene('cornell')
r = W.WebGPURenderer(0); r.buildPipeline(4, 1); W.upload_scene(r, b, 128, 96)
r.setStream(s.cuda_stream)
r.computeBatch([1, 2, 3, 4]); r.sync(); print('no timing ok')
r.setKernelTiming(True)
r.computeBatch([5, 6, 7, 8]); r.sync(); print('timing ok', r.kernelTimes()['pathtrace'])
""",
    "D_renderer_on_side_stream_timing_single_compute": """
import sys; sys.path.insert(0, '.')
import torch
import webgpu_raytracer_amd as W
s = torch.cuda.Stream()
b = W.WorldBridge(); b.loadScene('cornell')
r = W.WebGPURenderer(0); r.buildPipeline(4, 1); W.upload_scene(r, b, 128, 96)
r.setStream(s.cuda_stream)
r.setKernelTiming(True)
r.compute(1); r.sync(); print('timing ok', r.kernelTimes()['pathtrace'])
""",
    "E_renderer_own_stream_with_timing_torch_loaded": """
import sys; sys.path.insert(0, '.')
import torch
import webgpu_raytracer_amd as W
x = torch.zeros(4, device='cuda')
b = W.WorldBridge(); b.loadScene('cornell')
r = W.WebGPURenderer(0); r.buildPipeline(4, 1); W.upload_scene(r, b, 128, 96)
r.setKernelTiming(True)
r.computeBatch([1, 2, 3, 4]); r.sync(); print('timing ok', r.kernelTimes()['pathtrace'])
""",
}
for name, code in CASES.items():
    p = subprocess.run([sys.executable, "-X", "faulthandler", "-c", code], capture_output=True, text=True, timeout=300)
    print("==== %s rc=%d" % (name, p.returncode))
    print(p.stdout[-600:])
    if p.returncode:
        print("\n".join(l for l in p.stderr.splitlines() if l.strip())[-1200:])
