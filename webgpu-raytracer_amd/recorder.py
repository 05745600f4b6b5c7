"""Headless offline frame loop — the call cadence of the reference's VideoRecorder
(src/recorder/VideoRecorder.ts:145-317) without WebCodecs: warm-up x5, adaptive batches of compute(),
throttled present(), fence, captureFrame().  This is what defines "N spp per video frame" in the reference
(SURVEY.md §3.3): frame_count runs 0..N-1, so frames 0 and 1 both overwrite the accumulation buffer.

Works with any object exposing the WebGPURenderer surface.  `clock` is injectable (seconds, monotonic) so tests
can make the 100 ms present throttle deterministic.
"""
import time

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
    def render_frames(self, total_frames, fps, spp, start_frame=0, on_frame=None):
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
            frames.append(img)
        return frames
