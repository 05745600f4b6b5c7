"""Image-space sharding of one progressive render across ranks.

The reference distributes *video frame ranges* across browsers over WebRTC
(src/distributed/DistributedHost.ts:90-140) and reduces nothing (that unit is `recorder.FrameLoop`'s
`frame_range`).  On one MI355X node the natural unit is the pixel: every pixel is independent given
(pixel_idx, frame_count) (Raytracer.wgsl:794-798), so each rank path-traces an interleaved set of 8-row
stripes of the same image into its own full-size STRIPE accumulator (it only ever writes the rows it owns), and ONE
collective per image assembles the picture on rank 0: a GATHER of COMPACT stripes — every rank packs the rows it owns
(1/N of the image: 16.6 MB per rank at 4K over 8 ranks instead of the 133 MB a full-frame sum-reduce moved) and rank 0
copies each rank's rows to their place in its display buffer.  Pure copies, no floating-point addition anywhere
=> bitwise identical to the single-GPU image.  The post pass needs a 2-pixel halo and the history texture, so it runs
on rank 0 only.

The gather is OUT OF PLACE: the stripe accumulators are only read (packed into a send buffer); rank 0 presents from
its display buffer (rt_bind_present_source).  A progressive render may therefore go on after a gather (render ->
gather -> render -> gather, the live loop's present-every-frame pattern) and every gather yields the single-GPU image
of that moment.

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
            # the display buffer has world x max_rows rows: the image in the first `height`, then one scratch row per
            # padding row of the gathered blocks, so that rank 0 unpacks ALL blocks with ONE index_copy_ (below)
            self._display_ext = torch.zeros((max(r.height, self.world * self.max_rows(r.height)), r.width, 4),
                                            dtype=torch.float32, device=self.device)
            self.display = self._display_ext[:r.height]
        self._plan_key = None
        self.stream.synchronize()              # the zero fills are done before the renderer is pointed at the memory
        r.setStream(self.stream.cuda_stream)
        r.bindAccum(self.stripe.data_ptr())
        r.bindPresentSource(self.display.data_ptr() if (self._sharded() and self.on_device) else 0)

    def _sharded(self):
        return self.world > 1 or self.force_collective

    def owned_rows(self, height):
        return (np.arange(height) // self.stripe_rows) % self.world == self.rank

    def rows_of(self, rank, height):
        """Row indices (ascending) of the image that `rank` owns."""
        return np.nonzero((np.arange(height) // self.stripe_rows) % self.world == rank)[0]

    def max_rows(self, height):
        """Rows of the largest share: the gather moves equal-sized (padded) blocks."""
        return max(len(self.rows_of(k, height)) for k in range(self.world))

    def wire_bytes_per_rank(self):
        """Bytes one rank contributes to one gather (the padded compact block)."""
        return self.max_rows(self.r.height) * self.r.width * 16

    def _plan(self):
        """Index tensors and transfer buffers of the compact gather for the renderer's current size (device path)."""
        import torch
        h, w = self.r.height, self.r.width
        key = (h, w, self.world, self.rank)
        if getattr(self, "_plan_key", None) == key:
            return
        mr = self.max_rows(h)
        with torch.cuda.stream(self.stream):
            self._own_idx = torch.from_numpy(self.rows_of(self.rank, h)).to(self.device)
            self._send = torch.zeros((mr, w, 4), dtype=torch.float32, device=self.device)
            if self.rank == 0:
                # one receive buffer, rank k's block at [k]; row j of block k goes to image row rows_of(k)[j], a padding
                # row (j >= the rank's row count) to a scratch row of its own behind the image
                self._recv_all = torch.zeros((self.world, mr, w, 4), dtype=torch.float32, device=self.device)
                self._recv = [self._recv_all[k] for k in range(self.world)]
                dst, scratch = [], h
                for k in range(self.world):
                    rk = self.rows_of(k, h)
                    dst += list(rk) + list(range(scratch, scratch + mr - len(rk)))
                    scratch += mr - len(rk)
                self._dst_rows = torch.tensor(dst, dtype=torch.int64, device=self.device)
            else:
                self._recv_all = self._recv = self._dst_rows = None
        self._plan_key = key
        self.collective_ms = 0.0
        self._coll_events = []

    # ------------------------------------------------------------------ render / gather
    def render(self, frames, batch=1, timed=False):
        """Same per-frame call as the live loop, but present() is deferred to gather(). batch > 1 issues the
        frames as batched dispatches (rt_compute_batch, like the recorder's batch loop): bit-identical, and each
        launch carries `batch` times the work, which is what keeps a GPU busy on 1/N of an image.  timed (device path):
        events around this rank's dispatches on the stream (render_time_ms() reads them: the per-rank load balance)."""
        frames = list(frames)
        ev = None
        if timed and self.stream is not None:
            import torch
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record(self.stream)
        if batch > 1 and hasattr(self.r, "computeBatch"):
            for i in range(0, len(frames), batch):
                self.r.computeBatch(frames[i:i + batch])
        else:
            for f in frames:
                self.r.compute(f)
        if ev is not None:
            ev[1].record(self.stream)
            self.__dict__.setdefault("_render_events", []).append(ev)

    def render_time_ms(self):
        """Sum of the timed render() calls' stream durations on THIS rank since the last call, and their count."""
        ev = self.__dict__.get("_render_events", [])
        if not ev:
            return 0.0, 0
        self.stream.synchronize()
        ms, n = sum(a.elapsed_time(b) for a, b in ev), len(ev)
        self._render_events = []
        return ms, n

    def gather(self, present=True, timed=False):
        """Assemble the image on rank 0 from the ranks' compact stripes (one gather); rank 0 then runs the post pass on
        its display buffer.  The stripe accumulators are left as they are.  timed: bracket the pack + collective +
        unpack with events on the stream (collective_time_ms() reads them)."""
        if self._sharded():
            import torch
            import torch.distributed as dist
            if self.on_device:
                self._plan()
                with torch.cuda.stream(self.stream):
                    # renderer kernels, the packing copy, the collective and the unpacking copies are ordered by the one
                    # side stream: no host synchronisation anywhere
                    if timed:
                        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                        e0.record(self.stream)
                    n_own = self._own_idx.numel()
                    torch.index_select(self.stripe, 0, self._own_idx, out=self._send[:n_own])
                    dist.gather(self._send, gather_list=self._recv, dst=0)
                    if timed:
                        e1.record(self.stream)
                    if self.rank == 0:
                        # ONE copy kernel straight from the receive blocks (round 3: one index_copy_ launch per rank)
                        self._display_ext.index_copy_(0, self._dst_rows, self._recv_all.view(-1, self.r.width, 4))
                    if timed:
                        e2.record(self.stream)
                        self._coll_events.append((e0, e2))
                        self.__dict__.setdefault("_unpack_events", []).append((e1, e2))
            else:
                self.r.sync()
                h = self.r.height
                own_full = self.r.readAccum()
                rows = self.rows_of(self.rank, h)
                block = np.zeros((self.max_rows(h),) + own_full.shape[1:], dtype=np.float32)
                block[:len(rows)] = own_full[rows]
                t = torch.from_numpy(block)
                recv = [torch.zeros_like(t) for _ in range(self.world)] if self.rank == 0 else None
                dist.gather(t, gather_list=recv, dst=0)
                if self.rank == 0:
                    img = np.zeros_like(own_full)
                    for k in range(self.world):
                        rk = self.rows_of(k, h)
                        img[rk] = recv[k].numpy()[:len(rk)]
                    self._host_image = img
                    if present:
                        # the host path has no second device buffer: show the assembled image, then put the stripes back
                        self.r.writeAccum(self._host_image)
                        self.r.present()
                        self.r.sync()
                        self.r.writeAccum(own_full)
                return
        if present and self.rank == 0:
            self.r.present()

    def collective_time_ms(self):
        """Sum of the timed gathers' stream durations since the last call, and their count."""
        ev = getattr(self, "_coll_events", [])
        if not ev:
            return 0.0, 0
        self.stream.synchronize()
        ms = sum(a.elapsed_time(b) for a, b in ev)
        n = len(ev)
        self._coll_events = []
        return ms, n

    def unpack_time_ms(self):
        """Rank 0: stream time of the unpacking copy of the timed gathers since the last call, and their count."""
        ev = self.__dict__.get("_unpack_events", [])
        if not ev:
            return 0.0, 0
        self.stream.synchronize()
        ms, n = sum(a.elapsed_time(b) for a, b in ev), len(ev)
        self._unpack_events = []
        return ms, n

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
