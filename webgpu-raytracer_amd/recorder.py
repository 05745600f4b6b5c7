"""Headless offline frame loop — the call cadence of the reference's VideoRecorder
(src/recorder/VideoRecorder.ts:145-317) without WebCodecs: warm-up x5, adaptive batches of compute(),
throttled present(), fence, captureFrame().  This is what defines "N spp per video frame" in the reference
(SURVEY.md §3.3): frame_count runs 0..N-1, so frames 0 and 1 both overwrite the accumulation buffer.

Works with any object exposing the WebGPURenderer surface.  `clock` is injectable (seconds, monotonic) so tests
can make the 100 ms present throttle deterministic.
"""
import json
import os
import struct
import time
import zlib

import numpy as np


class FrameLoop:
    def __init__(self, renderer, bridge, width, height, batch=20, clock=time.perf_counter):
        self.renderer = renderer
        self.bridge = bridge
        self.width, self.height = width, height
        self.current_batch_size = batch      # VideoRecorder.currentBatchSize, initialised from config.batch
        self.clock = clock

    # VideoRecorder.updateSceneBuffers (:231-268) — note the upload order differs from main.ts
    def update_scene_buffers(self):
        r, b = self.renderer, self.bridge
        rebind = False
        rebind |= bool(r.updateCombinedBVH(b.tlas, b.blas))
        rebind |= bool(r.updateBuffer("instance", b.instances))
        rebind |= bool(r.updateCombinedGeometry(b.vertices, b.normals, b.uvs))
        rebind |= bool(r.updateBuffer("topology", b.mesh_topology))
        rebind |= bool(r.updateBuffer("lights", b.lights))
        rebind |= bool(r.updateBuffer("draw_commands", b.draw_commands))
        b.updateCamera(self.width, self.height)
        r.updateSceneUniforms(b.cameraData, 0, b.lightCount)
        if rebind:
            r.recreateBindGroup()
        r.resetAccumulation()

    # warm-up phase (:160-169): primes the TAA history
    def warm_up(self):
        self.update_scene_buffers()
        for k in range(5):
            self.renderer.compute(k)
            self.renderer.present()
            self.renderer.sync()
            self.renderer.resetAccumulation()

    # VideoRecorder.renderFrame (:270-317)
    def render_frame(self, total_spp):
        r = self.renderer
        samples_done = 0
        last_present = self.clock()
        presents = 0
        while samples_done < total_spp:
            batch = min(self.current_batch_size, total_spp - samples_done)
            t0 = self.clock()
            if hasattr(r, "computeBatch"):   # one dispatch per kernel for the whole batch (bit-identical)
                r.computeBatch(range(samples_done, samples_done + batch))
            else:
                for k in range(batch):
                    r.compute(samples_done + k)
            samples_done += batch
            now = self.clock()
            finished = samples_done >= total_spp
            if finished or (now - last_present) * 1000.0 > 100.0:
                r.present()
                presents += 1
                last_present = now
            r.sync()
            elapsed_ms = (self.clock() - t0) * 1000.0
            raw = 100.0 / elapsed_ms if elapsed_ms > 0 else 1.5
            raw = min(raw, 1.5)
            next_batch = int(np.floor(self.current_batch_size * (0.8 + 0.2 * raw) + 0.5))  # Math.round
            self.current_batch_size = max(1, min(total_spp, min(next_batch, 50)))
        return presents

    # renderAndEncode (:145-229) minus the encoder: one RGBA8 array per video frame
    def render_frames(self, total_frames, fps, spp, start_frame=0, on_frame=None, keep=True):
        self.bridge.update(start_frame / fps)
        self.warm_up()
        frames = []
        for i in range(total_frames):
            self.update_scene_buffers()
            if i < total_frames - 1:
                self.bridge.update((start_frame + i + 1) / fps)   # next frame's scene (synchronous here)
            self.render_frame(spp)
            img = self.renderer.captureFrame()["data"].copy()
            if on_frame:
                on_frame(i, img)
            if keep:
                frames.append(img)
        return frames


# ---------------------------------------------------------------------------------------------------------------
# Frames to disk.  The reference hands each captured frame to a WebCodecs VideoEncoder (VideoRecorder.ts:201-217);
# video encoding is out of scope here, so the frames are written as they come out of captureFrame(): PNG (RGBA8,
# zlib-deflated scanlines, filter 0 — readable by any decoder, including this repo's mt_decode) or raw RGBA bytes.
def write_png(path, rgba):
    a = np.ascontiguousarray(rgba, dtype=np.uint8)
    h, w = a.shape[0], a.shape[1]
    raw = np.empty((h, 1 + w * 4), dtype=np.uint8)
    raw[:, 0] = 0                                   # filter type 0 (None) on every scanline
    raw[:, 1:] = a.reshape(h, w * 4)

    def chunk(tag, data):
        body = tag + data
        return struct.pack(">I", len(data)) + body + struct.pack(">I", zlib.crc32(body) & 0xffffffff)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)))
        f.write(chunk(b"IDAT", zlib.compress(raw.tobytes(), 1)))
        f.write(chunk(b"IEND", b""))


def write_frame(out_dir, index, rgba, fmt):
    name = "frame_%06d.%s" % (index, "png" if fmt == "png" else "rgba")
    path = os.path.join(out_dir, name)
    if fmt == "png":
        write_png(path, rgba)
    else:
        np.ascontiguousarray(rgba, dtype=np.uint8).tofile(path)
    return name


def job_list(total_frames, job_batch=20):
    """The host's job queue (src/main.ts:278-290): frame ranges {start, count} of at most `jobBatch` (default 20) frames."""
    job_batch = max(1, int(job_batch))
    return [(f, min(job_batch, total_frames - f)) for f in range(0, total_frames, job_batch)]


class FrameJobRunner:
    """The reference's own distribution unit: VIDEO FRAME RANGES (src/distributed/DistributedHost.ts:90-140 hands each
    idle worker the next {start, count}; the worker runs renderAndEncode(count, config, .., startFrameOffset = start),
    DistributedWorker.ts).  Here the ranks of one job share nothing and exchange nothing: rank r renders jobs
    r, r + world, r + 2 world, ... of the queue and writes its frames; no collective is involved.

    In the reference a frame's pixels depend on what its worker happened to render before (WebGPURenderer.totalFrames
    drives the Halton jitter and lives as long as the renderer).  So that a frame is the same whichever rank renders it,
    every job starts from a fresh renderer (`make_renderer()`), exactly like a worker that has just loaded the scene;
    the present() throttle reads `clock`, which a test replaces by a deterministic one.
    """

    def __init__(self, make_renderer, make_bridge, width, height, fps, spp, depth=10, batch=20, out_dir=None, fmt="png",
                 clock=time.perf_counter):
        self.make_renderer, self.make_bridge = make_renderer, make_bridge
        self.width, self.height, self.fps, self.spp, self.depth, self.batch = width, height, fps, spp, depth, batch
        self.out_dir, self.fmt, self.clock = out_dir, fmt, clock

    def run_job(self, start, count, collect=None):
        r = self.make_renderer()
        bridge = self.make_bridge()
        r.buildPipeline(self.depth, 1)
        r.updateScreenSize(self.width, self.height)
        if hasattr(r, "loadTexturesFromWorld"):
            r.loadTexturesFromWorld(bridge)
        loop = FrameLoop(r, bridge, self.width, self.height, batch=self.batch,
                         clock=self.clock() if isinstance(self.clock, type) else self.clock)
        written = []

        def on_frame(i, img):
            index = start + i
            name = write_frame(self.out_dir, index, img, self.fmt) if self.out_dir else None
            written.append({"frame": index, "file": name})
            if collect is not None:
                collect[index] = img.copy()

        loop.render_frames(count, self.fps, self.spp, start_frame=start, on_frame=on_frame, keep=False)
        if hasattr(r, "destroy"):
            r.destroy()
        return written

    def run(self, total_frames, rank=0, world=1, job_batch=20, collect=None):
        """Render this rank's share of the job queue; returns the manifest entries of the frames it wrote."""
        if self.out_dir:
            os.makedirs(self.out_dir, exist_ok=True)
        jobs = job_list(total_frames, job_batch)
        mine = jobs[rank::world]
        manifest = []
        for start, count in mine:
            manifest += self.run_job(start, count, collect)
        if self.out_dir:
            with open(os.path.join(self.out_dir, "manifest_rank%d.json" % rank), "w") as f:
                json.dump({"rank": rank, "world": world, "width": self.width, "height": self.height, "fps": self.fps,
                           "spp": self.spp, "jobs": mine, "frames": manifest}, f, indent=1)
        return manifest
