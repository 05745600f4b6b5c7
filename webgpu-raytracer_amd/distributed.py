"""Image-space sharding of one progressive render across ranks.

The reference distributes *video frame ranges* across browsers over WebRTC
(src/distributed/DistributedHost.ts:90-140) and reduces nothing.  On one MI355X node the
natural unit is the pixel: every pixel is independent given (pixel_idx, frame_count)
(Raytracer.wgsl:794-798), so each rank path-traces an interleaved set of 8-row stripes of
the same image into a zero-initialised full-size accumulation buffer and ONE sum-reduce of
the float4 buffer to rank 0 (RCCL over xGMI: 33 MB at 1080p) reassembles it.  Disjoint
stripes + zeros => bitwise identical to the single-GPU image.  The post pass needs a 2-pixel
halo and the history texture, so it runs on rank 0 only.

`ShardedImage` works with any object exposing the WebGPURenderer surface: on GPUs the reduce
runs on the device buffer through torch.distributed (backend "nccl" = RCCL); the host path
(used by the gloo CPU tests) goes through readAccum()/writeAccum().
"""
import numpy as np

STRIPE_ROWS = 8   # one tile row: at 1080p over 8 ranks 16-row stripes leave a 9:8 stripe imbalance (87.6 % vs 91.1 % efficiency)


class ShardedImage:
    def __init__(self, renderer, rank, world_size, stripe_rows=STRIPE_ROWS, device_tensor=None):
        self.r = renderer
        self.rank = rank
        self.world = world_size
        self.stripe_rows = stripe_rows
        self.device_tensor = device_tensor  # torch CUDA tensor aliasing the accumulation buffer, or None
        self.force_collective = False       # rehearsal: run the collective even with one rank
        renderer.setStripes(stripe_rows, rank, world_size)

    def owned_rows(self, height):
        return (np.arange(height) // self.stripe_rows) % self.world == self.rank

    def render(self, frames, batch=1):
        """Same per-frame call as the live loop, but present() is deferred to gather(). batch > 1 issues the
        frames as batched dispatches (rt_compute_batch, like the recorder's batch loop): bit-identical, and each
        launch carries `batch` times the work, which is what keeps a GPU busy on 1/N of an image."""
        frames = list(frames)
        if batch > 1 and hasattr(self.r, "computeBatch"):
            for i in range(0, len(frames), batch):
                self.r.computeBatch(frames[i:i + batch])
        else:
            for f in frames:
                self.r.compute(f)

    def gather(self, present=True):
        """Sum the per-rank accumulation buffers onto rank 0; rank 0 then runs the post pass."""
        if self.world > 1 or self.force_collective:
            import torch
            import torch.distributed as dist
            if self.device_tensor is not None:
                # Explicit fences on both sides of the collective: the renderer may run on its own HIP stream
                # (torch's default stream has handle 0, which rt_set_stream reads as "own stream"), and RCCL orders
                # itself only against torch's current stream. Two host syncs per image, ~20 us each.
                self.r.sync()
                dist.reduce(self.device_tensor, dst=0, op=dist.ReduceOp.SUM)
                torch.cuda.synchronize(self.device_tensor.device)
            else:
                self.r.sync()
                t = torch.from_numpy(self.r.readAccum())
                dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)
                if self.rank == 0:
                    self.r.writeAccum(t.numpy())
        if present and self.rank == 0:
            self.r.present()


def bind_torch_accum(renderer, device):
    """Allocate the accumulation buffer as a torch tensor (device memory + RCCL plumbing), bind it to
    the renderer and run the renderer on torch's current stream. Returns the (H, W, 4) f32 tensor."""
    import torch
    t = torch.zeros((renderer.height, renderer.width, 4), dtype=torch.float32, device=device)
    renderer.setStream(torch.cuda.current_stream(device).cuda_stream)
    renderer.bindAccum(t.data_ptr())
    return t
