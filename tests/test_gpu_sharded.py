"""The sharded HIP path on real hardware: rt_bind_accum + rt_bind_present_source + the renderer on a torch
side stream + torch.distributed reduce, against the single-GPU image, bit for bit; and `bench.py --gpus N` started from
a plain shell.  A one-GPU box rehearses two ranks on cuda:0 with the reduce through gloo (RCCL refuses two ranks on one
device); with >= 2 GPUs the real RCCL reduce runs."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import parity_util as pu

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W_, H_, DEPTH, FRAMES_A, FRAMES_B, BATCH = 160, 96, 6, tuple(range(1, 9)), tuple(range(9, 13)), 4


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _single_gpu_reference(W):
    b = pu.bridge_for(W, "cornell")
    r = W.WebGPURenderer(0)
    r.buildPipeline(DEPTH, 1)
    W.upload_scene(r, b, W_, H_)
    out = []
    for frames in (FRAMES_A, FRAMES_B):
        for i in range(0, len(frames), BATCH):
            r.computeBatch(frames[i:i + BATCH])
        r.present()
        r.sync()
        out.append((r.readAccum().copy(), r.captureFrame()["data"].copy()))
    r.destroy()
    return out


def _rank(rank, world, port, backend, one_device, out_path):
    for p in (REPO, os.path.join(REPO, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    import webgpu_raytracer_amd as pkg
    from webgpu_raytracer_amd.distributed import ShardedImage

    dev_index = 0 if one_device else rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        bridge = pkg.WorldBridge()
        bridge.loadScene("cornell")
        r = pkg.WebGPURenderer(dev_index)
        r.buildPipeline(DEPTH, 1)
        pkg.upload_scene(r, bridge, W_, H_)
        shard = ShardedImage(r, rank, world, device=device, collective_on_device=(backend == "nccl"),
                             force_collective=(world == 1))
        rows = shard.owned_rows(H_)
        for k, frames in enumerate((FRAMES_A, FRAMES_B)):
            shard.render(frames, batch=BATCH)
            shard.gather(present=True)
            shard.synchronize()
            local = r.readAccum()
            assert not local[~rows].any(), "rank wrote outside its stripes (or the gather touched the stripe accumulator)"
            if rank == 0:
                np.save("%s.acc%d.npy" % (out_path, k), shard.read_image())
                np.save("%s.rgba%d.npy" % (out_path, k), r.captureFrame()["data"])
        dist.barrier()
        r.destroy()
    finally:
        dist.destroy_process_group()


def _run_ranks(tmp_path, world, backend, one_device):
    import torch.multiprocessing as mp
    out = str(tmp_path / "img")
    mp.spawn(_rank, args=(world, _free_port(), backend, one_device, out), nprocs=world, join=True)
    return [(np.load("%s.acc%d.npy" % (out, k)), np.load("%s.rgba%d.npy" % (out, k))) for k in range(2)]


def _assert_same(got, ref):
    for k in range(2):
        assert np.array_equal(got[k][0].view(np.uint32), ref[k][0].view(np.uint32)), "accumulation image %d differs" % k
        assert np.array_equal(got[k][1], ref[k][1]), "RGBA8 output %d differs" % k


def test_rccl_branch_with_one_rank_equals_plain_render(W, tmp_path):
    """The device path (side stream, out-of-place copy, dist.reduce on the nccl backend, present from the display buffer)
    with a single rank: render -> gather -> render -> gather equals the plain renderer at both moments."""
    W._build.build_rt()
    _assert_same(_run_ranks(tmp_path, 1, "nccl", True), _single_gpu_reference(W))


def test_two_hip_ranks_on_one_gpu_reduce_to_the_single_gpu_image(W, tmp_path):
    """Two processes, each with the real HIP renderer on cuda:0 and its stripes; reduce through gloo (host path)."""
    W._build.build_rt()
    _assert_same(_run_ranks(tmp_path, 2, "gloo", True), _single_gpu_reference(W))


def test_two_gpu_rccl_reduce_equals_the_single_gpu_image(W, tmp_path):
    """The real thing: one rank per GPU, RCCL sum-reduce of the float4 display buffer over xGMI."""
    W._build.build_rt()
    from webgpu_raytracer_amd import renderer
    if renderer.load_library().rt_device_count() < 2:
        pytest.skip("needs two GPUs")
    _assert_same(_run_ranks(tmp_path, 2, "nccl", False), _single_gpu_reference(W))


def test_bench_py_gpus_2_from_a_plain_shell(W):
    """`python bench.py --gpus 2` starts its own ranks and prints one JSON line with n_gpus = 2. On a one-GPU box both
    ranks share cuda:0 and the reduce goes through gloo (BENCH_ONE_DEVICE / BENCH_BACKEND rehearsal switches)."""
    W._build.build_rt()
    from webgpu_raytracer_amd import renderer
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    if renderer.load_library().rt_device_count() < 2:
        env.update(BENCH_ONE_DEVICE="1", BENCH_BACKEND="gloo")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--no-extra-configs", "--no-cpu-baseline", "--no-live-loop", "--no-world-update"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["unit"] == "Mrays/s" and rec["value"] > 0
    # every ray of the image is traced exactly once across the ranks: same count as one GPU (deterministic)
    one = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0",
                          "--no-extra-configs", "--no-cpu-baseline", "--no-live-loop", "--no-world-update"], env=env, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-3000:]
    rec1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][0])
    assert rec1["config"]["rays_per_image"] == rec["config"]["rays_per_image"]
    assert rec1["roofline"]["bound"] == "valu" and rec1["roofline"]["avg_launch_ms"] > 0


def test_resize_drops_a_bound_accumulator_loudly(W):
    """rt_resize drops rt_bind_accum / rt_bind_present_source; compute() and present() then fail until the caller binds
    again, instead of rendering into the internal buffer while the caller keeps reducing a stale tensor."""
    import torch
    W._build.build_rt()
    b = pu.bridge_for(W, "cornell")
    r = W.WebGPURenderer(0)
    r.buildPipeline(4, 1)
    W.upload_scene(r, b, 64, 48)
    t = torch.zeros((48, 64, 4), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    r.bindAccum(t.data_ptr())
    r.compute(1)
    r.sync()
    assert float(t[..., 3].min().item()) == 1.0     # the renderer wrote into the bound tensor
    r.updateScreenSize(80, 48)
    with pytest.raises(W.RendererError, match="rt_bind_accum again"):
        r.compute(1)
    with pytest.raises(W.RendererError, match="rt_bind_accum again"):
        r.present()
    t2 = torch.zeros((48, 80, 4), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    r.bindAccum(t2.data_ptr())
    b.updateCamera(80, 48)
    r.updateSceneUniforms(b.cameraData, 0, b.lightCount)
    assert r.compute(1) == 0
    r.sync()
    assert float(t2[..., 3].min().item()) == 1.0
    # staleness is per binding: with both bound, renewing only the accumulator leaves present() refusing to read
    # the dropped display buffer (it used to fall back to the internal accumulator silently)
    d2 = torch.zeros((48, 80, 4), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    r.bindPresentSource(d2.data_ptr())
    r.updateScreenSize(64, 48)
    t3 = torch.zeros((48, 64, 4), dtype=torch.float32, device="cuda:0")
    d3 = torch.zeros((48, 64, 4), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    r.bindAccum(t3.data_ptr())
    b.updateCamera(64, 48)
    r.updateSceneUniforms(b.cameraData, 0, b.lightCount)
    assert r.compute(1) == 0
    with pytest.raises(W.RendererError, match="rt_bind_present_source again"):
        r.present()
    r.bindPresentSource(d3.data_ptr())
    assert r.present() == 0
    r.sync()
    r.bindPresentSource(0)
    r.bindAccum(0)
    r.destroy()
