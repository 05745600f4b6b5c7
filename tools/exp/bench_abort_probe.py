#!/usr/bin/env python3
import subprocess, sys
BASE = """
import sys, os; sys.path.insert(0, '.')
import torch
import webgpu_raytracer_amd as W
from webgpu_raytracer_amd import distributed as D
torch.cuda.set_device(0); device = torch.device('cuda', 0)
b = W.WorldBridge(); b.loadScene('cornell')
r = W.WebGPURenderer(0); r.buildPipeline(8, 1); W.upload_scene(r, b, WIDTH, HEIGHT)
shard = D.ShardedImage(r, 0, 1, device=device) if SHARD else None
frames = list(range(1, 65))
def step():
    r.resetAccumulation()
    for i in range(0, 64, BATCH): r.computeBatch(frames[i:i+BATCH])
    r.present()
step(); torch.cuda.synchronize(); print('warm ok', flush=True)
if RESETC: r.resetCounters()
r.setKernelTiming(True)
if KT: r.kernelTimes()
print('timing on', flush=True)
r.resetAccumulation(); print('reset ok', flush=True)
r.computeBatch(frames[:BATCH]); print('batch1 ok', flush=True)
r.computeBatch(frames[BATCH:2*BATCH]); print('batch2 ok', flush=True)
r.present(); r.sync(); print('timed ok', r.kernelTimes()['pathtrace'], flush=True)
"""
CASES = {
  "full_bench_flow":        dict(WIDTH=1920, HEIGHT=1080, SHARD=True, BATCH=32, RESETC=True, KT=True),
  "no_shard_own_stream":    dict(WIDTH=1920, HEIGHT=1080, SHARD=False, BATCH=32, RESETC=True, KT=True),
  "small_size":             dict(WIDTH=256, HEIGHT=128, SHARD=True, BATCH=32, RESETC=True, KT=True),
  "no_kernelTimes_call":    dict(WIDTH=1920, HEIGHT=1080, SHARD=True, BATCH=32, RESETC=True, KT=False),
  "batch8":                 dict(WIDTH=1920, HEIGHT=1080, SHARD=True, BATCH=8, RESETC=False, KT=False),
}
import os
env = dict(os.environ, MI355RT_DEBUG_TERMINATE="1")
for name, kv in CASES.items():
    code = "\n".join("%s = %r" % i for i in kv.items()) + BASE
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    print("==== %s rc=%d" % (name, p.returncode)); print(p.stdout[-400:])
    if p.returncode: print("\n".join(l for l in p.stderr.splitlines() if l.strip())[-3000:])
