"""Image-space sharding of one progressive render across ranks.

The reference distributes *video frame ranges* across browsers over WebRTC
(src/distributed/DistributedHost.ts:90-140) and reduces nothing (that unit is `recorder.FrameLoop`'s
`frame_range`).  On one MI355X node the natural unit is the pixel: every pixel is independent given
(pixel_idx, frame_count) (Raytracer.wgsl:794-798), so each rank path-traces an interleaved set of 8-row
stripes of the same image into its own zero-initialised full-size STRIPE accumulator, and ONE sum-reduce of
the float4 buffer to rank 0 (RCCL over xGMI: 33 MB at 1080p) assembles the image.  Disjoint stripes + zeros
=> bitwise identical to the single-GPU image.  The post pass needs a 2-pixel halo and the history texture,
so it runs on rank 0 only.

The reduce is OUT OF PLACE: every rank copies its stripe accumulator into a display buffer and the display
buffers are reduced; rank 0 presents from its display buffer (rt_bind_present_source).  The stripe accumulators
are never touched by a gather, so a progressive render may go on after it (render -> gather -> render -> gather,
the live loop's present-every-frame pattern) and every gather yields the single-GPU image of that moment.

`ShardedImage` works with any object exposing the WebGPURenderer surface.  On GPUs (`device` given) both buffers
are torch tensors and the renderer, the copy and the collective all run on ONE torch side stream, so they are
stream-ordered without host synchronisation (torch.distributed backend "nccl" = RCCL).  The host path (gloo CPU
tests, or BENCH_BACKEND=gloo rehearsals on a one-GPU box) goes through readAccum() / writeAccum().
"""
import numpy as np

STRIPE_ROWS = 8   # one tile row: at 1080p over 8 ranks 16-row stripes leave a 9:8 stripe imbalance (87.6 % vs 91.1 % efficiency)


class ShardedImage:
    def __init__(self, renderer, rank, world_size, stripe_rows=STRIPE_ROWS, device=None, collective_on_device=True,
                 force_collective=False):
        self.r = renderer
        self.rank = rank
        self.world = world_size
        self.stripe_rows = stripe_rows
        self.force_collective = force_collective   # rehearsal: run the collective even with one rank
        self.device = device
        self.on_device = device is not None and collective_on_device
        self.stripe = self.display = self.stream = None
        self._host_image = None
        renderer.setStripes(stripe_rows, rank, world_size)
        if device is not None:
            self.bind()

    # ------------------------------------------------------------------ device buffers
    def bind(self):
        """(Re)allocate the stripe accumulator and the display buffer for the renderer's current size and bind them.
        Call again after updateScreenSize(): rt_resize drops the bindings and compute() refuses to run until then."""
        import torch
        r = self.r
        if self.stream is None:
            self.stream = torch.cuda.Stream(device=self.device)   # a side stream: its handle is never 0 ("own stream")
        with torch.cuda.stream(self.stream):
            self.stripe = torch.zeros((r.height, r.width, 4), dtype=torch.float32, device=self.device)
            self.display = torch.zeros((r.height, r.width, 4), dtype=torch.float32, device=self.device)
        self.stream.synchronize()              # the zero fills are done before the renderer is pointed at the memory
        r.setStream(self.stream.cuda_stream)
        r.bindAccum(self.stripe.data_ptr())
        r.bindPresentSource(self.display.data_ptr() if (self._sharded() and self.on_device) else 0)

    def _sharded(self):
        return self.world > 1 or self.force_collective

    def owned_rows(self, height):
        return (np.arange(height) // self.stripe_rows) % self.world == self.rank

    # ------------------------------------------------------------------ render / gather
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
        """Sum the per-rank stripe accumulators into rank 0's display buffer; rank 0 then runs the post pass on it.
        The stripe accumulators are left as they are."""
        if self._sharded():
            import torch
            import torch.distributed as dist
            if self.on_device:
                with torch.cuda.stream(self.stream):
                    # renderer kernels, this copy and the collective are ordered by the one side stream
                    self.display.copy_(self.stripe, non_blocking=True)
                    dist.reduce(self.display, dst=0, op=dist.ReduceOp.SUM)
            else:
                self.r.sync()
                own = self.r.readAccum()
                t = torch.from_numpy(own.copy())   # gloo may use a non-root input as scratch: never hand it the accumulator
                dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)
                if self.rank == 0:
                    self._host_image = t.numpy()
                    if present:
                        # the host path has no second device buffer: show the sum, then put the stripes back
                        self.r.writeAccum(self._host_image)
                        self.r.present()
                        self.r.sync()
                        self.r.writeAccum(own)
                return
        if present and self.rank == 0:
            self.r.present()

    def read_image(self):
        """Rank 0: the assembled float4 accumulation image of the last gather() as (H, W, 4) float32."""
        if not self._sharded():
            return self.r.readAccum()
        if self.on_device:
            self.stream.synchronize()
            return self.display.cpu().numpy()
        return self._host_image

    def synchronize(self):
        if self.stream is not None:
            self.stream.synchronize()
        else:
            self.r.sync()
