"""N > 1 path on CPU: two gloo ranks each render their interleaved row stripes (with the CPU oracle standing
in for the GPU renderer — same method surface), one sum-reduce of the accumulation buffer lands on rank 0,
and the result must be bitwise the single-process image."""
import os
import socket
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_path, w, h, frames, more_frames, stripe_rows):
    for p in (REPO, os.path.join(REPO, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    import webgpu_raytracer_amd as pkg
    from webgpu_raytracer_amd.distributed import ShardedImage
    import oracle_lib

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        bridge = pkg.WorldBridge()
        bridge.loadScene("cornell")
        r = oracle_lib.OracleRenderer(threads=2)
        r.buildPipeline(4, 1)
        pkg.upload_scene(r, bridge, w, h)
        shard = ShardedImage(r, rank, world, stripe_rows=stripe_rows)
        rows = shard.owned_rows(h)
        shard.render(frames)
        local = r.readAccum()
        assert not local[~rows].any(), "rank wrote outside its stripes"
        assert local[rows][..., 3].min() == len(frames)
        shard.gather(present=True)
        # the gather leaves the stripe accumulators alone (out-of-place reduce) ...
        assert np.array_equal(r.readAccum().view(np.uint32), local.view(np.uint32))
        if rank == 0:
            np.save(out_path, shard.read_image())
            np.save(out_path + ".rgba.npy", r.captureFrame()["data"])
        # ... so the progressive render goes on and a second gather is the single-process image of that moment
        shard.render(more_frames)
        shard.gather(present=True)
        if rank == 0:
            np.save(out_path + ".more.npy", shard.read_image())
            np.save(out_path + ".more.rgba.npy", r.captureFrame()["data"])
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,stripe_rows,h", [(2, 16, 72), (3, 8, 50)])
def test_two_rank_stripes_reduce_to_the_single_process_image(W, oracle_lib, tmp_path, world, stripe_rows, h):
    import torch.multiprocessing as mp
    w, frames, more_frames = 64, (1, 2, 3), (4, 5)
    out = str(tmp_path / "acc.npy")
    mp.spawn(_worker, args=(world, _free_port(), out, w, h, frames, more_frames, stripe_rows), nprocs=world, join=True)
    sharded = np.load(out)
    rgba = np.load(out + ".rgba.npy")

    b = W.WorldBridge()
    b.loadScene("cornell")
    ref = oracle_lib.OracleRenderer(threads=2)
    ref.buildPipeline(4, 1)
    W.upload_scene(ref, b, w, h)
    for f in frames:
        ref.compute(f)
    assert np.array_equal(sharded.view(np.uint32), ref.readAccum().view(np.uint32))
    ref.present()
    assert np.array_equal(rgba, ref.captureFrame()["data"])
    # render -> gather -> render -> gather (the live loop's present-every-frame pattern across ranks)
    for f in more_frames:
        ref.compute(f)
    assert np.array_equal(np.load(out + ".more.npy").view(np.uint32), ref.readAccum().view(np.uint32))
    ref.present()
    assert np.array_equal(np.load(out + ".more.rgba.npy"), ref.captureFrame()["data"])


def test_owned_rows_partition_the_image():
    sys.path.insert(0, REPO)
    from webgpu_raytracer_amd.distributed import ShardedImage

    class Dummy:
        def setStripes(self, *a):
            self.args = a

    for world in (1, 2, 4, 8):
        masks = [ShardedImage(Dummy(), r, world).owned_rows(1080) for r in range(world)]
        total = np.sum(masks, axis=0)
        assert (total == 1).all()
        # interleaving balances the load: no rank owns more than one stripe more than another
        counts = [int(m.sum()) for m in masks]
        assert max(counts) - min(counts) <= 16
