#!/usr/bin/env python3
"""bench.py — headline metric of BASELINE.json on MI355X.

Metric   : Mrays/s (primary + secondary) at 1920x1080, SPP = 64, depth = 8 (Cornell box).
Step     : one full image = resetAccumulation + 64 x compute(frame_count = 1..64) with shader
           SPP = 1 (the canonical decomposition) + one present().
Rays     : primary-visibility casts + extension rays + shadow rays actually traced, from the
           device counters (deterministic; cross-checked against the oracle in tests/).
N > 1    : the image is split into interleaved 8-row stripes across ranks (strong scaling of one
           image), one RCCL sum-reduce of the float4 accumulation buffer to rank 0 per image.
Extra    : "roofline" for the dominant kernel (k_pathtrace; HIP events inside the C library, on the
           stream the kernel runs on) and "cpu_baseline" (the CPU oracle timed on a bounded
           1/3 row-interleaved sample of the same workload, rank 0 at N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
for _p in (REPO, os.path.join(REPO, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

WIDTH, HEIGHT, SPP_TOTAL, DEPTH = 1920, 1080, 64, 8
SCENE = "cornell"
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def cpu_baseline(pkg, bridge, frames):
    """Time the CPU oracle (C++ scalar restatement, all host threads) on rows
    {y : (y // 8) % 3 == 0} of the same 1080p frames: a 1/3 row-interleaved sample."""
    import oracle_lib
    cpu = oracle_lib.OracleRenderer()
    cores = oracle_lib.lib().oracle_hardware_threads()
    cpu.buildPipeline(DEPTH, 1)
    pkg.upload_scene(cpu, bridge, WIDTH, HEIGHT)
    cpu.setStripes(8, 0, 3)
    sample_px = int(((np.arange(HEIGHT) // 8) % 3 == 0).sum()) * WIDTH
    cpu.resetCounters()
    t0 = time.perf_counter()
    for f in frames:
        cpu.compute(f)
    dt = time.perf_counter() - t0
    c = cpu.getCounters()
    rays = c["primary_rays"] + c["extension_rays"] + c["shadow_rays"]
    return {"value": round(rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": int(cores), "kind": "port",
            "sample": "rows (y//8)%%3==0 (1/3 of 1080p, %d px) x %d frames, depth %d: %.1f Mrays in %.1f s"
                      % (sample_px, len(frames), DEPTH, rays / 1e6, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batch", type=int, default=32,
                    help="frames per batched dispatch (the recorder batches up to 50 compute() calls; 1 = one dispatch per frame)")
    args = ap.parse_args()

    import torch
    import webgpu_raytracer_amd as pkg
    from webgpu_raytracer_amd import distributed as rtdist


    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP renderer has no CPU fallback")
    # Rehearsal switches for a 1-GPU box (never used by the driver): BENCH_ONE_DEVICE=1 puts every rank on
    # cuda:0 and BENCH_BACKEND=gloo reduces through host memory (RCCL refuses two ranks on one GPU).
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    if os.environ.get("BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    force_dist = os.environ.get("BENCH_FORCE_DIST") == "1"   # rehearsal: exercise the RCCL calls with one rank
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend=backend)

    # in-tree libraries: built by local rank 0 if missing or stale (a no-op otherwise), the others wait
    if int(os.environ.get("LOCAL_RANK", "0")) == 0:
        pkg._build.build_scene()
        pkg._build.build_rt()
    if world > 1 or force_dist:
        dist.barrier()
    bridge = pkg.WorldBridge()
    bridge.loadScene(SCENE)
    r = pkg.WebGPURenderer(local_rank)
    r.buildPipeline(DEPTH, 1)
    pkg.upload_scene(r, bridge, WIDTH, HEIGHT)
    accum_t = rtdist.bind_torch_accum(r, device)
    shard = rtdist.ShardedImage(r, rank, world,
                                device_tensor=accum_t if ((world > 1 or force_dist) and backend == "nccl") else None)
    shard.force_collective = force_dist
    frames = list(range(1, SPP_TOTAL + 1))

    def step():
        r.resetAccumulation()
        shard.render(frames, batch=args.batch)
        shard.gather(present=True)

    def fence():
        if world > 1 or force_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    r.resetCounters()
    r.setKernelTiming(True)
    r.kernelTimeMs()  # drop anything recorded so far
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    ktime = r.kernelTimeMs()
    r.setKernelTiming(False)

    counts = r.getCounters()
    rays_local = counts["primary_rays"] + counts["extension_rays"] + counts["shadow_rays"]
    stats = torch.tensor([elapsed, float(rays_local)], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
    if world > 1 or force_dist:
        tmax = stats[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        rsum = stats[1:].clone()
        dist.all_reduce(rsum, op=dist.ReduceOp.SUM)
        elapsed, rays_total = float(tmax.item()), float(rsum.item())
    else:
        rays_total = float(rays_local)

    if rank == 0:
        # one extra untimed image with the detailed-counter kernel variant: per-launch algorithmic bytes
        # of the dominant kernel (counts are deterministic, so they equal the timed launches')
        r.setCounting(True)
        r.resetCounters()
        shard.render(frames, batch=args.batch)
        r.sync()
        kc = r.getKernelCounters(1)
        r.setCounting(False)
        n_launch = (len(frames) + args.batch - 1) // args.batch   # path-trace launches per image
        owned_px = int(shard.owned_rows(HEIGHT).sum()) * WIDTH
        alg_bytes = (32.0 * kc["nodes_visited"] + 64.0 * kc["tris_tested"] + 344.0 * kc["shaded_hits"]) / n_launch \
            + (32.0 + 24.0 * min(args.batch, len(frames))) * owned_px  # accumulation read+write once per launch, G-buffer read per frame
        achieved = alg_bytes / (ktime["pathtrace_ms"] * 1e-3) / 1e9 if ktime["pathtrace_ms"] > 0 else 0.0
        # HBM bytes per launch from the PMC passes committed under profiles/ (collected by tools/make_profiles.sh,
        # N = 1 only; counters cannot be read from inside this process)
        traffic, pmc = None, {}
        tpath = os.path.join(REPO, "profiles", "hbm_traffic.json")
        if world == 1 and os.path.exists(tpath):
            try:
                pmc = json.load(open(tpath))
                # the PMC passes were taken at 32 frames per launch; scale to this run's batch size
                traffic = pmc.get("k_pathtrace_bytes_per_launch")
                if traffic is not None and pmc.get("frames_per_launch"):
                    traffic = traffic * min(args.batch, len(frames)) / pmc["frames_per_launch"]
            except Exception:
                traffic, pmc = None, {}
        out = {
            "metric": "Mrays/s (primary+secondary) at 1920x1080 SPP=64 depth=8",
            "value": round(rays_total / elapsed / 1e6, 2),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "cornell box 1920x1080, 64 spp as 1 spp x 64 compute() frames (frame_count 1..64, "
                                   "issued as batched dispatches of %d frames), depth 8, one present() per image" % args.batch,
                       "scene": SCENE, "width": WIDTH, "height": HEIGHT, "spp": SPP_TOTAL, "max_depth": DEPTH,
                       "parallelism": "%d-row stripes x %d ranks + 1 RCCL reduce/image" % (rtdist.STRIPE_ROWS, world) if world > 1 else "1 GPU",
                       "frames_per_dispatch": args.batch, "rays_per_image": int(rays_total / args.steps)},
            "roofline": {"bound": "hbm", "kernel": "k_pathtrace_persistent", "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "avg_launch_ms": round(ktime["pathtrace_ms"], 4),
                         "avg_primary_ms": round(ktime["primary_ms"], 4), "launches": ktime["launches"],
                         "alg_bytes_per_launch": int(alg_bytes),
                         "note": "algorithmic gather bytes (SURVEY 8d) are served by LDS/L1, so achieved can exceed the HBM "
                                 "peak; the kernel is VALU-issue bound",
                         "valu_busy_frac": pmc.get("valu_busy_frac"),
                         "valu_lane_utilization": pmc.get("valu_lane_utilization"),
                         "valu": pmc.get("valu")},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pkg, bridge, frames)
        print(json.dumps(out), flush=True)
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
