#!/usr/bin/env python3
"""bench.py — headline metric of BASELINE.json on MI355X.

Metric   : Mrays/s (primary + secondary) at 1920x1080, SPP = 64, depth = 8 (Cornell box).
Step     : one full image = resetAccumulation + 64 x compute(frame_count = 1..64) with shader
           SPP = 1 (the canonical decomposition), issued as batched dispatches, + one present().
Rays     : primary-visibility casts + extension rays + shadow rays actually traced, from the
           device counters (deterministic; cross-checked against the oracle in tests/).
N > 1    : `python bench.py --gpus N` starts its own N ranks (one process per GPU, torch.distributed.run, before
           anything in this process touches the GPU); under a launcher (WORLD_SIZE set) it is one of the ranks.
           The image is split into interleaved 8-row stripes across ranks (strong scaling of one image), one RCCL
           gather of the ranks' compact stripes (1/N of the float4 accumulation buffer each) to rank 0 per image.
Extra    : "roofline" for the dominant kernel (k_pathtrace_persistent; launch time from HIP events inside the C
           library, on the stream the kernel runs on), "configs" (BASELINE configs 3-5, timed the same way,
           with the roofline of their dominant kernel k_wf_trace) and "cpu_baseline" (the CPU oracle timed on a
           bounded row-interleaved sample of the same workload, rank 0 at N = 1 only).
Fields that cannot be measured from inside this process (PMC counters) are read from profiles/ and every such field
carries its "source".
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for _p in (REPO, os.path.join(REPO, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

WIDTH, HEIGHT, SPP_TOTAL, DEPTH = 1920, 1080, 64, 8
SCENE = "cornell"
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
L2_PEAK_GBS = 34500.0        # aggregate L2 bandwidth, 8 XCDs (MI355X_MICROARCH.md §L2)
# f32 VALU: 157.3 TFLOP/s spec = 256 CUs x 4 SIMDs x 32 lanes/clk x 2 flop (FMA) x 2.4 GHz, i.e. 78.6 T lane-instructions/s
VALU_PEAK_TLANE = 256 * 4 * 32 * 2.4e9 / 1e12
DIAMOND_OBJ = ("v 0 1 0\nv 1 0 0\nv 0 0 1\nv -1 0 0\nv 0 0 -1\nv 0 -1 0\n"
               "f 1 3 2\nf 1 2 5\nf 1 5 4\nf 1 4 3\nf 6 2 3\nf 6 5 2\nf 6 4 5\nf 6 3 4\n")   # the octahedron of public/diamond.obj
# Config 2 (8-triangle diamond, 1280x720) is not here: it runs the headline's kernel, and the rocprofv3 summary of this
# command is meant to show ONE workload per kernel name (its figure is in BASELINE.md, from tools/config_table.py).
EXTRA_CONFIGS = [  # (BASELINE.json config, scene, frames, depth, images timed, width, height)
    ("3 instanced diamond x1000", "instanced1000", 64, 8, 3, 1920, 1080),
    ("4 sponza-like 263k tris, 8 textures", "sponza_like", 64, 8, 3, 1920, 1080),
    ("5 glass blob 205k tris", "glass_blob", 256, 16, 1, 3840, 2160),
]


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip the lines of BASELINE configs 3-5")
    ap.add_argument("--no-live-loop", action="store_true", help="skip the per-frame compute(); present() block")
    ap.add_argument("--no-world-update", action="store_true", help="skip the World::update(t) block (host / GPU builder / device-resident)")
    ap.add_argument("--batch", type=int, default=32,
                    help="frames per batched dispatch (the recorder batches up to 50 compute() calls; 1 = one dispatch per frame)")
    return ap.parse_args()


def launch_ranks(args):
    """`python bench.py --gpus N` from a plain shell: start N ranks as fresh child processes (this process has not
    imported torch or touched the GPU) and pass their exit code on. Rank 0 prints the JSON line to our stdout."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this pool
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def gather_calibration():
    """Rows of profiles/r03_gather_peak.txt (tools/gather_peak.hip on one MI355X): G lane-steps/s of dependent divergent
    gathers of one 32-byte record (2 x dwordx4) per step with ALL 64 lanes of every wave active, 6 workgroups per CU — the
    full-wave rate; the lanes a trace kernel leaves idle are reported as a loss term of their own (lane_utilization)."""
    for name in ("r04_gather_peak.txt", "r03_gather_peak.txt", "r02_gather_peak.txt"):
        path = os.path.join(REPO, "profiles", name)
        out = {}
        try:
            for line in open(path):
                f = line.split()
                if len(f) >= 6 and f[1:4] == ["2", "64", "6"]:
                    if f[0] == "16KB":
                        out["l1_resident_Gps"] = float(f[4])
                    elif f[0] == "4MB":
                        out["l2_resident_Gps"] = float(f[4])
                # the child-pair walk: one 64-byte record (TWO node tests) per step, quad-cooperative LDS-DMA fetch (rows "4d")
                if len(f) >= 6 and f[1:4] == ["4d", "64", "6"]:
                    if f[0] == "16KB":
                        out["pairs_l1_resident_Gnodes"] = 2.0 * float(f[4])
                    elif f[0] == "4MB":
                        out["pairs_l2_resident_Gnodes"] = 2.0 * float(f[4])
        except OSError:
            continue
        if "l1_resident_Gps" in out and "l2_resident_Gps" in out:
            out["source"] = ("profiles/%s (tools/gather_peak.hip: 64 of 64 lanes, 16 KB / 4 MB table; 32-B records with 2 loads per "
                             "lane, and 64-B records fetched quad-cooperatively by LDS-DMA = two node tests per step)" % name)
            return out
    return None


def cpu_baseline(pkg, bridge, frames):
    """Time the CPU oracle (C++ scalar restatement) on a bounded row-interleaved sample of the same 1080p frames:
    all usable host threads on rows (y//8)%3==0, and ONE thread on rows (y//8)%54==0, so that the scaling of the
    baseline itself is visible."""
    import numpy as np
    import oracle_lib

    def run(threads, div):
        cpu = oracle_lib.OracleRenderer(threads=threads)
        cpu.buildPipeline(DEPTH, 1)
        pkg.upload_scene(cpu, bridge, WIDTH, HEIGHT)
        cpu.setStripes(8, 0, div)
        px = int(((np.arange(HEIGHT) // 8) % div == 0).sum()) * WIDTH
        cpu.resetCounters()
        t0 = time.perf_counter()
        for f in frames:
            cpu.compute(f)
        dt = time.perf_counter() - t0
        c = cpu.getCounters()
        return (c["primary_rays"] + c["extension_rays"] + c["shadow_rays"]), dt, px

    cores = int(oracle_lib.lib().oracle_hardware_threads())   # affinity mask and cgroup quota respected
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = None
    r1, t1, px1 = run(1, 54)
    rn, tn, pxn = run(cores, 3)
    return {"value": round(rn / tn / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "one_thread_value": round(r1 / t1 / 1e6, 3),
            "per_thread_at_full_width": round(rn / tn / 1e6 / cores, 4),
            "sched_affinity": affinity, "os_cpu_count": os.cpu_count(),
            "sample": "rows (y//8)%%3==0 (1/3 of 1080p, %d px) x %d frames, depth %d: %.1f Mrays in %.1f s on %d threads; "
                      "one thread: rows (y//8)%%54==0 (%d px), %.1f Mrays in %.1f s"
                      % (pxn, len(frames), DEPTH, rn / 1e6, tn, cores, px1, r1 / 1e6, t1)}


def rehearsal(args, rank, world):
    """BENCH_REHEARSAL=launch: exercise the launcher, the rendezvous and the max-over-ranks reductions without a
    renderer (the build container has no GPU). Prints a line that cannot be mistaken for a measurement."""
    import torch
    import torch.distributed as dist
    dist.init_process_group(backend="gloo")
    dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    dist.barrier()
    stats = torch.tensor([time.perf_counter() - t0, float(rank + 1)], dtype=torch.float64)
    tmax, rsum = stats[:1].clone(), stats[1:].clone()
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dist.all_reduce(rsum, op=dist.ReduceOp.SUM)
    if rank == 0:
        print(json.dumps({"rehearsal": "launch", "metric": None, "value": None, "n_gpus": world,
                          "ranks_seen": int(rsum.item() * 2 / (world + 1)), "steps": args.steps, "warmup": args.warmup}),
              flush=True)
    dist.barrier()
    dist.destroy_process_group()


def issue_calibration():
    """profiles/r04_issue_peak.txt (tools/issue_peak.hip): wave-instructions per cycle per SIMD of independent VALU, SALU and
    interleaved streams at W waves per SIMD — the WALL-CLOCK rate of >= 10 ms launches (round 3 quoted the rate derived from
    per-workgroup s_memtime spans of 1 ms launches, which over-states it 1.4-2.0 x: profiles/r04_issue_peak.txt has both
    columns).  Returns {stream name: {waves: rate}}."""
    import re
    path = os.path.join(REPO, "profiles", "r04_issue_peak.txt")
    out = {}
    try:
        for line in open(path):
            f = re.split(r"\s{2,}", line.strip())
            if len(f) >= 4 and f[1].isdigit():
                try:
                    out.setdefault(f[0], {})[int(f[1])] = float(f[3])      # name, waves, span-derived, WALL-CLOCK, ...
                except ValueError:
                    pass
    except OSError:
        return None
    return out or None


def pmc_reference():
    """Counter-derived figures committed under profiles/ (PMC passes cannot run inside this process)."""
    path = os.path.join(REPO, "profiles", "pmc_reference.json")
    if not os.path.exists(path):
        return {}
    try:
        ref = json.load(open(path))
    except Exception:
        return {}
    # counters cannot be read from inside this process: the file names the kernel sources it was collected from, and
    # figures from other sources are flagged (and the fractions derived from them withheld) instead of quoted
    import webgpu_raytracer_amd as pkg
    ref["_stale"] = ref.get("csrc_sha16") != pkg._build.kernel_source_hash()
    return ref


def main():
    args = parse_args()
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        sys.exit(launch_ranks(args))      # nothing GPU-related has been imported yet

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(world_env or "1")
    if world != args.gpus:
        raise SystemExit("--gpus %d but the launcher started %d ranks" % (args.gpus, world))
    if os.environ.get("BENCH_REHEARSAL") == "launch":
        return rehearsal(args, rank, world)

    import numpy as np
    import torch
    import webgpu_raytracer_amd as pkg
    from webgpu_raytracer_amd import distributed as rtdist

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
    distributed = world > 1 or force_dist
    if distributed:
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
        pkg._build.build_tex()
        pkg._build.build_rt()
    if distributed:
        dist.barrier()

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce_stats(elapsed, rays):
        if not distributed:
            return elapsed, float(rays)
        stats = torch.tensor([elapsed, float(rays)], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        tmax, rsum = stats[:1].clone(), stats[1:].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(rsum, op=dist.ReduceOp.SUM)
        return float(tmax.item()), float(rsum.item())

    def run_workload(scene, frames, depth, steps, warmup, batch, width=WIDTH, height=HEIGHT):
        """Time `steps` images of one scene on all ranks; returns the renderer, the sharding and the measurements."""
        bridge = pkg.WorldBridge()
        if scene == "viewer_diamond":
            bridge.loadScene("viewer", DIAMOND_OBJ)
        else:
            bridge.loadScene(scene)
        r = pkg.WebGPURenderer(local_rank)
        r.buildPipeline(depth, 1)
        pkg.upload_scene(r, bridge, width, height)
        shard = rtdist.ShardedImage(r, rank, world, device=device, collective_on_device=(backend == "nccl"),
                                    force_collective=force_dist)

        def step(timed=False):
            r.resetAccumulation()
            shard.render(frames, batch=batch, timed=timed)
            shard.gather(present=True, timed=timed)

        for _ in range(warmup):
            step()
        fence()
        r.resetCounters()
        r.setKernelTiming(True)
        r.kernelTimes()  # drop anything recorded so far
        t0 = time.perf_counter()
        for _ in range(steps):
            step(timed=distributed)
        fence()
        elapsed = time.perf_counter() - t0
        coll_ms, coll_n = shard.collective_time_ms() if distributed else (0.0, 0)
        unpack_ms, unpack_n = shard.unpack_time_ms() if distributed else (0.0, 0)
        rend_ms, rend_n = shard.render_time_ms() if distributed else (0.0, 0)
        render_ms = None
        if distributed and rend_n:
            # this rank's stream time in its own dispatches per image: min / max over the ranks = the load balance of the stripes
            mine = torch.tensor([rend_ms / rend_n], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
            lo, hi = mine.clone(), mine.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            render_ms = {"min": round(float(lo.item()), 4), "max": round(float(hi.item()), 4)}
        ktimes = r.kernelTimes()
        r.setKernelTiming(False)
        counts = r.getCounters()
        rays_local = counts["primary_rays"] + counts["extension_rays"] + counts["shadow_rays"]
        elapsed, rays_total = reduce_stats(elapsed, rays_local)
        # one extra untimed image with the detailed-counter kernel variant: algorithmic bytes of the path-trace stage
        # (the counts are deterministic, so they equal the timed launches')
        r.setCounting(True)
        r.resetCounters()
        shard.render(frames, batch=batch)
        r.sync()
        kc = r.getKernelCounters(1)
        r.setCounting(False)
        return {"bridge": bridge, "renderer": r, "shard": shard, "elapsed": elapsed, "rays": rays_total,
                "ktimes": ktimes, "kc": kc, "steps": steps, "collective_ms": coll_ms / coll_n if coll_n else None,
                "unpack_ms": unpack_ms / unpack_n if unpack_n else None, "render_ms": render_ms,
                "wire_bytes_per_rank": shard.wire_bytes_per_rank() if distributed else 0}

    def live_loop(scene, nframes, depth, width=WIDTH, height=HEIGHT, passes=3, lookahead=0):
        """The reference's live loop (src/main.ts:168-173): compute(frameCount); present() per displayed frame, one call per
        frame; fire-and-forget like the reference (one fence at the end of a pass).  lookahead = 0: one dispatch per frame,
        nothing batched; > 1: rt_set_lookahead — the library traces consecutive frames ahead as batches (same images)."""
        bridge = pkg.WorldBridge()
        bridge.loadScene(scene)
        r = pkg.WebGPURenderer(local_rank)
        r.buildPipeline(depth, 1)
        pkg.upload_scene(r, bridge, width, height)
        r.setLookahead(lookahead)

        def one_pass():
            r.resetAccumulation()
            for f in range(1, nframes + 1):
                r.compute(f)
                r.present()
            r.sync()

        one_pass()
        r.resetCounters()
        t0 = time.perf_counter()
        for _ in range(passes):
            one_pass()
        dt = time.perf_counter() - t0
        c = r.getCounters()
        r.destroy()
        rays = c["primary_rays"] + c["extension_rays"] + c["shadow_rays"]
        e = {"scene": scene, "workload": "%s %dx%d depth %d: %d x { compute(f); present() }, %s" % (
                 scene, width, height, depth, nframes, "frames traced ahead (rt_set_lookahead %d)" % lookahead if lookahead > 1 else "one dispatch per frame"),
             "lookahead": lookahead, "ms_per_frame": round(dt / (passes * nframes) * 1e3, 4), "frames": passes * nframes}
        if lookahead > 1:
            # with lookahead the device counters count a frame when it is TRACED, the <= lookahead - 1 frames traced in vain
            # at the end of every pass included: that is not the rate of the frames shown
            e["Mrays_s_traced_incl_frames_traced_in_vain"] = round(rays / dt / 1e6, 1)
        else:
            e["Mrays_s"] = round(rays / dt / 1e6, 1)
            e["rays_per_displayed_frame"] = rays / float(passes * nframes)
        return e

    def animated_live_loop(kind, frames=24, width=WIDTH, height=HEIGHT):
        """renderFrame of src/main.ts:119-181 on an ANIMATED world, the world advanced before every displayed frame (so no
        frame can be traced ahead): update(t) -> sync -> compute(1) -> present(), device-resident update (rt_world_update).
        kind "tube": a fully skinned 262 144-triangle glTF; "hall": a skinned 9 216-triangle character inside a static
        262 144-triangle mesh (tests/test_gltf.py builds both).  ms per displayed frame, split by phase."""
        import test_gltf
        if kind == "hall":
            glb, n_static, n_skinned = test_gltf.character_in_hall_glb(pkg, (512, 256))
            what = "%d static + %d skinned triangles" % (n_static, n_skinned)
        else:
            glb, n_tris = test_gltf.big_skinned_glb(pkg, 512, 256)
            what = "%d skinned triangles" % n_tris
        res = {"scene": "animated glTF (%s): %s, %dx%d depth %d, world advanced every displayed frame" % (kind, what, width, height, DEPTH)}
        for mode in ("device_resident", "host_cpu_builder"):
            r = pkg.WebGPURenderer(local_rank)
            r.buildPipeline(DEPTH, 1)
            b = pkg.WorldBridge(zero_copy=True)
            if mode == "device_resident":
                b.setDeviceUpdater(r)
            b.loadScene("viewer", glbData=glb)
            pkg.upload_scene(r, b, width, height)
            n = frames if mode == "device_resident" else 2      # the reference's way costs 0.1 s per update
            tick = 0
            for _ in range(3 if mode == "device_resident" else 1):   # the device path learns the trees' depths on its first updates
                tick += 1
                b.update(tick / 60)
                pkg.sync_world(r, b, width, height)
                r.compute(1)
                r.present()
            r.sync()
            t_upd = t_sync = t_trace = t_gpu = 0.0
            for _ in range(n):
                tick += 1
                t0 = time.perf_counter()
                b.update(tick / 60)
                t1 = time.perf_counter()
                if mode == "device_resident":
                    if not b.deviceResident:
                        raise SystemExit("bench.py: the device-resident update fell back to the host: " + b.deviceWarning)
                    t_gpu += r.worldLastMs()
                pkg.sync_world(r, b, width, height)
                t2 = time.perf_counter()
                r.compute(1)
                r.present()
                r.sync()
                t3 = time.perf_counter()
                t_upd += t1 - t0
                t_sync += t2 - t1
                t_trace += t3 - t2
            e = {"update_ms": round(t_upd / n * 1e3, 3), "upload_ms": round(t_sync / n * 1e3, 3),
                 "trace_present_ms": round(t_trace / n * 1e3, 3), "frames_per_s": round(n / (t_upd + t_sync + t_trace), 1), "frames": n}
            if mode == "device_resident":
                e["update_gpu_stream_ms"] = round(t_gpu / n, 3)
            res[mode] = e
            b.close()
            r.destroy()
        return res

    def world_update_block(scene="sponza_like", width=WIDTH, height=HEIGHT):
        """World::update(t) + the scene sync of the live loop (src/main.ts:133-163) on config 4's scene, three ways: the
        scene compiler on the host (the reference's way: lib.rs:149-270, one CPU thread for the BLAS), the host path
        with the GPU BLAS builder as its hook (round 2), and the device-resident update (rt_world_update: skinning,
        BLAS, topology / lights / draw commands, TLAS and instances on the GPU, nothing uploaded).  ms per update."""
        r = pkg.WebGPURenderer(local_rank)
        r.buildPipeline(DEPTH, 1)
        res = {"scene": scene}
        # (an all-static world with the static-geometry cache on is left untouched by an update: nothing to time.  The case
        # the cache is for - a skinned character inside a large static mesh - is timed by animated_live_loop("hall").)
        for mode, reps in (("host_cpu_builder", 2), ("host_gpu_blas_hook", 5), ("device_resident", 20)):
            b = pkg.WorldBridge(zero_copy=True)
            if mode == "host_gpu_blas_hook":
                b.setBlasBuilder(r)
            elif mode.startswith("device_resident"):
                b.setDeviceUpdater(r)
                # the scene is static: with the cache (the default) only TLAS, instances and lights are redone per frame;
                # without it every frame skins, builds and packs everything - the cost of a fully animated world
                r.setWorldStaticCache(False)
            b.loadScene(scene)
            pkg.upload_scene(r, b, width, height)
            res["triangles"] = len(b.mesh_topology) // 20
            for _ in range(3 if mode != "host_cpu_builder" else 1):
                b.update(0.0)
                pkg.sync_world(r, b, width, height)
            r.sync()
            t_upd = t_sync = t_gpu = 0.0
            for _ in range(reps):
                t0 = time.perf_counter()
                b.update(0.0)
                t1 = time.perf_counter()
                pkg.sync_world(r, b, width, height)
                r.sync()
                t2 = time.perf_counter()
                t_upd += t1 - t0
                t_sync += t2 - t1
                if mode.startswith("device_resident"):
                    if not b.deviceResident:
                        raise SystemExit("bench.py: the device-resident update fell back to the host: " + b.deviceWarning)
                    t_gpu += r.worldLastMs()
            e = {"update_ms": round(t_upd / reps * 1e3, 3), "upload_ms": round(t_sync / reps * 1e3, 3)}
            if mode.startswith("device_resident"):
                e["gpu_stream_ms"] = round(t_gpu / reps, 3)
            res[mode] = e
            b.close()
        r.destroy()
        res["device_over_host"] = round((res["host_cpu_builder"]["update_ms"] + res["host_cpu_builder"]["upload_ms"]) /
                                        (res["device_resident"]["update_ms"] + res["device_resident"]["upload_ms"]), 1)
        # the TLAS of World::update (bvh/tlas.rs:58-111) where it matters: config 3's 1 001 instances and the 16 384 the device
        # path takes (k_tlas: LDS bitonic sort per depth); stream time of the kernel alone beside the whole device update and
        # the host's TLAS builder on the same instances
        tl = []
        for sc in ("instanced1000", "instanced16384"):
            r = pkg.WebGPURenderer(local_rank)
            r.buildPipeline(DEPTH, 1)
            r.setWorldStaticCache(False)
            b = pkg.WorldBridge(zero_copy=True)
            b.setDeviceUpdater(r)
            b.loadScene(sc)
            pkg.upload_scene(r, b, width, height)
            for _ in range(3):
                b.update(0.0)
            r.sync()
            t_all = t_tlas = 0.0
            reps = 10
            for _ in range(reps):
                b.update(0.0)
                if not b.deviceResident:
                    raise SystemExit("bench.py: the device-resident update fell back to the host: " + b.deviceWarning)
                t_all += r.worldLastMs()
                t_tlas += r.worldLastTlasMs()
            hb = pkg.WorldBridge(zero_copy=True)
            hb.loadScene(sc)
            t0 = time.perf_counter()
            hb.update(0.0)
            t_host = time.perf_counter() - t0
            tl.append({"scene": sc, "instances": len(b.instances) // 36, "k_tlas_ms": round(t_tlas / reps, 4),
                       "device_update_gpu_stream_ms": round(t_all / reps, 4), "host_update_ms": round(t_host * 1e3, 3)})
            hb.close()
            b.close()
            r.destroy()
        res["tlas"] = tl
        return res

    frames = list(range(1, SPP_TOTAL + 1))
    head = run_workload(SCENE, frames, DEPTH, args.steps, args.warmup, args.batch)
    extra = []
    if not args.no_extra_configs:
        for name, scene, nframes, depth, images, cw, ch in EXTRA_CONFIGS:
            m = run_workload(scene, list(range(1, nframes + 1)), depth, images, 1, args.batch, cw, ch)
            m["renderer"].destroy()      # the 4K batch holds 63 GB of path state, queues and G-buffers
            m["renderer"] = None
            extra.append((name, scene, nframes, depth, m, cw, ch))

    if rank == 0:
        ref = pmc_reference()
        n_launch = (len(frames) + args.batch - 1) // args.batch   # path-trace launches per image
        owned_px = int(head["shard"].owned_rows(HEIGHT).sum()) * WIDTH
        kt, kc = head["ktimes"], head["kc"]
        launch_ms = kt["pathtrace"]["ms"] / max(1, kt["pathtrace"]["launches"])
        gather_bytes = (32.0 * kc["nodes_visited"] + 64.0 * kc["tris_tested"] + 344.0 * kc["shaded_hits"]) / n_launch
        stream_bytes = (32.0 + 24.0 * min(args.batch, len(frames))) * owned_px  # accumulation read+write per launch, G-buffer read per frame
        alg_gather = gather_bytes / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
        pt = ref.get("k_pathtrace_persistent", {}) if world == 1 else {}
        src = ref.get("source")
        roof = {"kernel": "k_pathtrace_persistent", "bound": "valu", "unit": "T lane-instr/s",
                "peak": round(VALU_PEAK_TLANE, 2),
                "peak_note": "f32 VALU spec: 256 CUs x 4 SIMDs x 32 lanes/clk x 2.4 GHz (157.3 TFLOP/s / 2 flop per FMA)",
                "avg_launch_ms": round(launch_ms, 4), "launches": kt["pathtrace"]["launches"],
                "avg_primary_ms": round(kt["primary"]["ms"] / max(1, kt["primary"]["launches"]), 4),
                "frames_per_launch": min(args.batch, len(frames)),
                "alg_gather_GBps": round(alg_gather, 1), "alg_gather_bytes_per_launch": int(gather_bytes),
                "alg_gather_served_by": "lds (the 8.4 KB scene is staged per workgroup; SURVEY 8d gather bytes never reach HBM)",
                "alg_hbm_bytes_per_launch": int(stream_bytes),
                "achieved": None, "frac": None, "traffic": None}
        stale = bool(ref.get("_stale"))
        if ref and world == 1:
            roof["pmc_stale"] = stale
            roof["pmc_csrc_sha16"] = {"collected_from": ref.get("csrc_sha16"), "this_build": pkg._build.kernel_source_hash()}
        if stale:
            roof["pmc_stale_note"] = ("profiles/pmc_reference.json was collected from other kernel sources than this build's: "
                                      "achieved / frac / traffic (counter-derived) are withheld; re-run tools/make_profiles_round.sh")
        if pt.get("lane_instr_per_launch") and launch_ms > 0 and not stale:
            scale = min(args.batch, len(frames)) / float(pt.get("frames_per_launch", 32))
            lane = pt["lane_instr_per_launch"] * scale
            roof["achieved"] = round(lane / (launch_ms * 1e-3) / 1e12, 3)
            roof["frac"] = round(roof["achieved"] / VALU_PEAK_TLANE, 4)
            roof["lane_instr_per_launch"] = {"value": lane, "source": src}
            for k in ("valu_lane_utilization", "effective_clock_GHz", "measured_issue_peak_Tlane", "insts_valu_per_launch",
                      "insts_salu_per_launch", "frac_of_wave_cycles_active_inst_any", "frac_of_wave_cycles_wait_any",
                      "frac_of_wave_cycles_wait_inst_any", "lds_bank_conflict_frac_of_lds_cycles"):
                if k in pt:
                    roof[k] = {"value": pt[k], "source": src}
            if pt.get("measured_issue_peak_Tlane"):
                roof["frac_of_measured_issue_peak"] = round(roof["achieved"] / pt["measured_issue_peak_Tlane"], 4)
            if "valu_busy_per_simd" in pt:
                roof["valu_busy_per_simd"] = {"value": pt["valu_busy_per_simd"], "note": pt.get("valu_busy_note"), "source": src}
            # `achieved` counts the vector lane-instructions the kernel EXECUTES for its rays, so a round that removes
            # instructions lowers it while the image gets faster.  The same rays cost round 3's build 656.4 G lane-instructions
            # per 32-frame launch (profiles/r03_cornell_pmc.json); that stream over this build's launch time is the
            # like-for-like figure.
            try:
                r3 = json.load(open(os.path.join(REPO, "profiles", "r03_cornell_pmc.json")))
                r3k = [v for k, v in r3.items() if "k_pathtrace_persistent" in k][0]
                r3_lane = float(r3k["SQ_THREAD_CYCLES_VALU"]) * scale
                roof["same_rays_comparison"] = {
                    "r03_lane_instr_per_launch": r3_lane, "r03_insts_valu_per_launch": float(r3k["SQ_INSTS_VALU"]) * scale,
                    "r03_insts_salu_per_launch": float(r3k["SQ_INSTS_SALU"]) * scale,
                    "this_build_over_r03_lane_instr": round(lane / r3_lane, 4),
                    "frac_at_r03_instruction_stream": round(r3_lane / (launch_ms * 1e-3) / 1e12 / VALU_PEAK_TLANE, 4),
                    "note": "round 3: 0.369 at 22.63 ms per launch; the rays, their hits and every counter of the image are unchanged",
                    "source": "profiles/r03_cornell_pmc.json (SQ_THREAD_CYCLES_VALU) / this run's launch time"}
            except (OSError, ValueError, KeyError, IndexError):
                pass
            ic = issue_calibration()
            if ic and pt.get("insts_valu_per_launch") and pt.get("insts_salu_per_launch") and pt.get("effective_clock_GHz"):
                # Round 3 called the kernel "issue-bound at 0.79" against span-derived rates that wall clock does not confirm.
                # By wall clock (tools/issue_peak.hip, >= 10 ms launches) a SIMD with 4 resident waves issues 0.40 independent
                # v_add_f32 per cycle, 0.23 s_add_u32 (the scalar unit is shared by the CU's four SIMDs: 0.25), and of a
                # 3 : 1 vector : scalar stream - this kernel's mix - what the "3 v_add : 1 s_add" rows say.
                scale_i = scale
                valu = pt["insts_valu_per_launch"] * scale_i
                salu = pt["insts_salu_per_launch"] * scale_i
                lds = pt.get("insts_lds_per_launch", 0.0) * scale_i
                simd_cycles = launch_ms * 1e-3 * pt["effective_clock_GHz"] * 1e9 * 256 * 4
                mix = ic.get("3 v_add : 1 s_add (mixed)", {}).get(4)
                roof["issue"] = {"waves_per_simd": 4,
                                 "valu_instr_per_cycle_per_simd": round(valu / simd_cycles, 4),
                                 "salu_instr_per_cycle_per_simd": round(salu / simd_cycles, 4),
                                 "all_instr_per_cycle_per_simd": round((valu + salu + lds) / simd_cycles, 4),
                                 "wall_clock_peak_valu_only": ic.get("v_add_f32", {}).get(4),
                                 "wall_clock_peak_salu_only": ic.get("s_add_u32", {}).get(4),
                                 "wall_clock_peak_3to1_mix": mix,
                                 "valu_frac_of_wall_clock_valu_peak": round(valu / simd_cycles / ic["v_add_f32"][4], 4) if ic.get("v_add_f32", {}).get(4) else None,
                                 "all_frac_of_wall_clock_3to1_peak": round((valu + salu + lds) / simd_cycles / mix, 4) if mix else None,
                                 "source": "instruction counts: " + str(src) + "; issue rates: profiles/r04_issue_peak.txt (tools/issue_peak.hip, wall-clock column)"}
            if pt.get("hbm_bytes_per_launch"):
                tb = pt["hbm_bytes_per_launch"] * scale
                roof["traffic"] = tb
                roof["traffic_source"] = src
                roof["hbm_GBps"] = round(tb / (launch_ms * 1e-3) / 1e9, 1)
                roof["hbm_frac"] = round(roof["hbm_GBps"] / HBM_PEAK_GBS, 4)
        out = {
            "metric": "Mrays/s (primary+secondary) at 1920x1080 SPP=64 depth=8",
            "value": round(head["rays"] / head["elapsed"] / 1e6, 2),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(head["elapsed"] / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "cornell box 1920x1080, 64 spp as 1 spp x 64 compute() frames (frame_count 1..64, "
                                   "issued as batched dispatches of %d frames), depth 8, one present() per image" % args.batch,
                       "scene": SCENE, "width": WIDTH, "height": HEIGHT, "spp": SPP_TOTAL, "max_depth": DEPTH,
                       "parallelism": "%d-row stripes x %d ranks + 1 RCCL gather of compact stripes/image" % (rtdist.STRIPE_ROWS, world) if world > 1 else "1 GPU",
                       "frames_per_dispatch": args.batch, "rays_per_image": int(head["rays"] / args.steps)},
            "roofline": roof,
        }
        if distributed:
            # pack + gather + unpack on rank 0's stream, per image (events around the collective section)
            out["collective_ms"] = round(head["collective_ms"], 4) if head["collective_ms"] is not None else None
            out["collective_bytes_per_rank"] = head["wire_bytes_per_rank"]
            # of it: rank 0's unpacking copy (one kernel from the receive blocks into the display buffer); and what each rank
            # spends in its own dispatches per image (stream time, min / max over the ranks): the balance of the stripes
            out["unpack_ms"] = round(head["unpack_ms"], 4) if head["unpack_ms"] is not None else None
            out["rank_render_ms_per_image"] = head["render_ms"]
        cfgs = []
        for name, scene, nframes, depth, m, cw, ch in extra:
            kt, kc = m["ktimes"], m["kc"]
            # the any-hit and the closest-hit trace of a depth run side by side on two streams (their event times overlap):
            # the trace time of an image is the span of the path-trace stage minus the shade kernels, which run alone
            trace_ms = kt["pathtrace"]["ms"] - kt["wf_shade"]["ms"] if kt["wf_trace_ext"]["launches"] else 0.0
            trace_bytes = 32.0 * kc["nodes_visited"] + 64.0 * kc["tris_tested"]    # per image (detailed pass = one image)
            per_image_trace_ms = trace_ms / m["steps"]
            n_inst = int(np.asarray(m["bridge"].instances).size // 36)
            entry = {"config": name, "walk": "child-pair records (k_wf_trace_pairs)" if n_inst == 1 else "single nodes (k_wf_trace)", "workload": "%s %dx%d, %d frames x depth %d, batches of %d" % (scene, cw, ch, nframes, depth, min(args.batch, nframes)),
                     "ms_per_image": round(m["elapsed"] / m["steps"] * 1e3, 2),
                     "Mrays_s": round(m["rays"] / m["elapsed"] / 1e6, 1), "images": m["steps"],
                     "kernel_ms_per_image": {k: round(v["ms"] / m["steps"], 3) for k, v in kt.items() if v["launches"]},
                     "kernel_ms_note": "pathtrace = span of the whole path-trace stage; wf_trace_shadow runs beside wf_trace_ext "
                                       "on a second stream, so those two overlap and do not add up"}
            if distributed:
                entry["collective_ms"] = round(m["collective_ms"], 4) if m["collective_ms"] is not None else None
                entry["collective_bytes_per_rank"] = m["wire_bytes_per_rank"]
                entry["unpack_ms"] = round(m["unpack_ms"], 4) if m["unpack_ms"] is not None else None
                entry["rank_render_ms_per_image"] = m["render_ms"]
            if per_image_trace_ms > 0:
                gbps = trace_bytes / (per_image_trace_ms * 1e-3) / 1e9
                rk = ref.get("k_wf_trace", {}).get(scene, {}) if (world == 1 and not ref.get("_stale")) else {}
                # What the walk is up against (round 2, measured): not HBM and not L2 bandwidth, but the rate at which a CU
                # takes DEPENDENT lane-divergent 16-byte gathers — tools/gather_peak.hip, 32-byte records (2 x dwordx4 per
                # step, as a node step), 40 of 64 lanes active as in the trace kernels: every record from the L1 / from L2.
                gp = gather_calibration()
                if gp and n_inst == 1 and "pairs_l1_resident_Gnodes" in gp:   # priced against the fetch this walk uses
                    gp = dict(gp, l1_resident_Gps=gp["pairs_l1_resident_Gnodes"], l2_resident_Gps=gp["pairs_l2_resident_Gnodes"])
                steps_g = kc["nodes_visited"] / (per_image_trace_ms * 1e-3) / 1e9
                roof_c = {"kernel": "%s (any-hit + closest-hit launches)" % ("k_wf_trace_pairs" if n_inst == 1 else "k_wf_trace"), "bound": "gather",
                          "achieved": round(steps_g, 1), "peak": gp["l1_resident_Gps"] if gp else None,
                          "unit": "G node-steps/s",
                          "frac": round(steps_g / gp["l1_resident_Gps"], 4) if gp else None,
                          "frac_of_l2_resident_rate": round(steps_g / gp["l2_resident_Gps"], 4) if gp else None,
                          "peak_source": gp,
                          "peak_note": "full-wave rate (64 of 64 lanes active) of dependent divergent 32-byte gathers; the lanes the "
                                       "kernel leaves idle are a loss term of their own: pmc.*.valu_lane_utilization",
                          "pmc_stale": bool(ref.get("_stale")) if ref else None,
                          "alg_GBps": round(gbps, 1), "hbm_peak": HBM_PEAK_GBS, "alg_frac_of_hbm_peak": round(gbps / HBM_PEAK_GBS, 4),
                          "l2_peak": L2_PEAK_GBS, "alg_frac_of_l2_peak": round(gbps / L2_PEAK_GBS, 4),
                          "alg_bytes_per_image": int(trace_bytes), "trace_ms_per_image": round(per_image_trace_ms, 3),
                          "served_by": "L1 / L2 / Infinity Cache: the scene (<= 40 MB) stays on die, so the algorithmic bytes "
                                       "(32 B per node visit + 64 B per triangle test) are NOT HBM traffic; `traffic` is the "
                                       "measured fabric-side figure",
                          "traffic": rk.get("hbm_bytes_per_image"), "traffic_source": ref.get("source") if rk else None}
                if rk.get("hbm_bytes_per_image"):
                    roof_c["traffic_GBps"] = round(rk["hbm_bytes_per_image"] / (per_image_trace_ms * 1e-3) / 1e9, 1)
                    roof_c["traffic_frac_of_hbm_peak"] = round(roof_c["traffic_GBps"] / HBM_PEAK_GBS, 4)
                if rk.get("kernels"):
                    roof_c["pmc"] = {"value": rk["kernels"], "source": ref.get("source")}
                entry["roofline"] = roof_c
            cfgs.append(entry)
        if cfgs:
            out["configs"] = cfgs
        if world == 1 and not args.no_live_loop:
            # the reference's hot loop as it is written: per-frame dispatch + present, beside the batched cadence
            ll = []
            batched = {SCENE: out["ms_per_step"] / SPP_TOTAL}
            for name, scene, nframes, depth, m, cw, ch in extra:
                batched[scene] = m["elapsed"] / m["steps"] * 1e3 / nframes
            for scene, depth in ((SCENE, DEPTH), ("sponza_like", 8)):
                if scene != SCENE and args.no_extra_configs:
                    continue
                rays_per_frame = None
                for look in (0, 32):
                    # 256 frames per pass: a live view accumulates until the camera moves; the run must be long against the
                    # lookahead's ramp (1, 2, 4, ... frames) and against what is traced ahead in vain when it ends
                    e = live_loop(scene, 256, depth, lookahead=look, passes=2)
                    if look == 0:
                        rays_per_frame = e.pop("rays_per_displayed_frame")
                    elif rays_per_frame:
                        # rays of the frames DISPLAYED (frames x the per-frame count of the one-dispatch-per-frame run of the
                        # same 256 frames: the counts are deterministic) over the same wall time
                        e["Mrays_s"] = round(rays_per_frame / (e["ms_per_frame"] * 1e-3) / 1e6, 1)
                    if scene in batched:
                        e["batched_ms_per_frame"] = round(batched[scene], 4)
                        e["live_over_batched"] = round(e["ms_per_frame"] / batched[scene], 3)
                    ll.append(e)
                # the same with a still view four times as long: the ramp and the (at most 31) frames traced in vain when the
                # run ends are a fixed cost per run, so the ratio depends on how long the camera rests
                e = live_loop(scene, 1024, depth, lookahead=32, passes=1)
                if rays_per_frame:   # frames 257..1024 count within a fraction of a percent of the first 256: an estimate, said so
                    e["Mrays_s_displayed_est"] = round(rays_per_frame / (e["ms_per_frame"] * 1e-3) / 1e6, 1)
                if scene in batched:
                    e["batched_ms_per_frame"] = round(batched[scene], 4)
                    e["live_over_batched"] = round(e["ms_per_frame"] / batched[scene], 3)
                ll.append(e)
            out["live_loop"] = ll
            if not args.no_world_update:
                out["live_loop_animated"] = [animated_live_loop("tube"), animated_live_loop("hall")]
        if world == 1 and not args.no_world_update:
            out["world_update"] = world_update_block()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pkg, head["bridge"], frames)
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
