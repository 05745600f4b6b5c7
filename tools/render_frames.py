#!/usr/bin/env python3
"""Offline frame loop from the shell: the reference recorder's cadence (VideoRecorder.ts:145-317) writing PNG / raw
frames, video frame RANGES sharded over the GPUs of the node the way the reference shards them over browsers
(main.ts:278-290, DistributedHost.ts:90-140): rank r renders jobs r, r + N, ... on GPU r; nothing is exchanged.

usage: render_frames.py --scene cornell --frames 60 --fps 30 --spp 64 --size 1280x720 --out out_dir [--gpus N] [--raw]
"""
import argparse
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="cornell")
    ap.add_argument("--glb", default=None, help="glTF binary appended to the scene (animated input)")
    ap.add_argument("--frames", type=int, default=30)
    ap.add_argument("--fps", type=int, default=30)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--depth", type=int, default=10)
    ap.add_argument("--batch", type=int, default=20)
    ap.add_argument("--job-batch", type=int, default=20)
    ap.add_argument("--size", default="720x480")
    ap.add_argument("--out", required=True)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--raw", action="store_true")
    ap.add_argument("--rank", type=int, default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.rank is None and args.gpus > 1:
        # build once here (compile only: no GPU is touched), so that the ranks never write the libraries concurrently;
        # then one fresh process per GPU, started before anything here touches the GPU
        import webgpu_raytracer_amd as W
        W._build.build_all()
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:] + ["--rank", str(r)])
                 for r in range(args.gpus)]
        sys.exit(max(p.wait() for p in procs))
    rank = args.rank or 0
    import webgpu_raytracer_amd as W
    W._build.build_all()
    w, h = (int(x) for x in args.size.lower().split("x"))
    glb = open(args.glb, "rb").read() if args.glb else None

    def make_bridge():
        b = W.WorldBridge()
        b.loadScene(args.scene, None, glb)
        return b

    runner = W.FrameJobRunner(lambda: W.WebGPURenderer(rank), make_bridge, w, h, args.fps, args.spp, depth=args.depth,
                              batch=args.batch, out_dir=args.out, fmt="raw" if args.raw else "png")
    m = runner.run(args.frames, rank=rank, world=max(1, args.gpus), job_batch=args.job_batch)
    print("rank %d wrote %d frames to %s" % (rank, len(m), args.out))


if __name__ == "__main__":
    main()
