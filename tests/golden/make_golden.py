#!/usr/bin/env python3
"""Generates tests/golden/golden.json and cornell_bridge.npz with the CPU oracle + scene compiler.

The reference cannot run here (WGSL needs a WebGPU device, the Rust crate cannot be built: SURVEY.md §8c),
so these vectors come from this repository's own oracle; they pin it against regressions and give the
GPU tests a second, oracle-free comparison.  Re-run only when the numeric contract changes on purpose:
    python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
for p in (REPO, os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import webgpu_raytracer_amd as pkg  # noqa: E402
import oracle_lib  # noqa: E402
import parity_util as pu  # noqa: E402

CASES = [
    # name, scene, w, h, depth, spp, frames
    ("cornell_cfg1_small", "cornell", 128, 128, 4, 1, (1, 2, 3, 4)),
    ("viewer_diamond_cfg2_small", "viewer_diamond", 160, 90, 8, 1, (1, 2)),
    ("special_glass_metal", "special", 96, 72, 8, 1, (1, 2)),
    ("mixed_lens", "mixed", 96, 64, 10, 1, (1, 2)),
    ("instanced1000_cfg3_small", "instanced1000", 96, 54, 8, 1, (1,)),
    ("sponza_like_cfg4_small", "sponza_like", 64, 36, 8, 1, (1,)),
]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def render_case(renderer, scene, w, h, depth, spp, frames):
    b = pu.bridge_for(pkg, scene)
    pu.drive(renderer, pkg, b, w, h, depth, spp, frames, present=True)
    acc = renderer.readAccum()
    out = renderer.captureFrame()["data"]
    centre = acc[h // 2, w // 2]
    return {"accum_sha256": sha(acc), "rgba_sha256": sha(out), "counters": renderer.getCounters(),
            "centre_pixel_accum_bits": [int(x) for x in centre.view(np.uint32)],
            "mean_rgb": [float(x) for x in acc[..., :3].mean(axis=(0, 1), dtype=np.float64)]}


def main():
    out = {}
    for name, scene, w, h, depth, spp, frames in CASES:
        r = oracle_lib.OracleRenderer()
        out[name] = {"scene": scene, "width": w, "height": h, "depth": depth, "spp": spp, "frames": list(frames)}
        out[name].update(render_case(r, scene, w, h, depth, spp, frames))
        print(name, out[name]["accum_sha256"][:16], out[name]["counters"])
    json.dump(out, open(os.path.join(HERE, "golden.json"), "w"), indent=1, sort_keys=True)
    b = pkg.WorldBridge()
    b.loadScene("cornell")
    b.updateCamera(512, 512)
    np.savez_compressed(os.path.join(HERE, "cornell_bridge.npz"), vertices=b.vertices, normals=b.normals, uvs=b.uvs,
                        mesh_topology=b.mesh_topology, tlas=b.tlas, blas=b.blas, instances=b.instances,
                        lights=b.lights, draw_commands=b.draw_commands, camera_512=b.cameraData)


if __name__ == "__main__":
    main()
