"""The child-pair walk of the HIP kernels (csrc/k_pairwalk.hip.h), compiled for the host (tests/model/pairwalk_model.cpp)
and driven ray by ray, against the oracle's literal traversal loop: closest hit (t, triangle, instance), occlusion bit and
BOTH counters of every ray, on every scene, for closest-hit and shadow rays, with stacks from 1 entry (nearly always
overflowing into the stackless fall-back) to 64 (never).  No GPU: this pins the traversal LOGIC the GPU runs."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import pair_layout
import parity_util as pu

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
SRC = os.path.join(HERE, "model", "pairwalk_model.cpp")
LIB = os.path.join(HERE, "model", "_build", "libpairwalk_model.so")


def model_lib():
    deps = [SRC, os.path.join(REPO, "webgpu-raytracer_amd", "csrc", "k_pairwalk.hip.h"), os.path.join(REPO, "include", "mi355rt_math.h")]
    if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        subprocess.run(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-shared", "-fPIC", "-o", LIB, SRC], check=True)
    L = ctypes.CDLL(LIB)
    vp = ctypes.c_void_p
    L.pwm_trace.argtypes = [vp, vp, vp, vp, vp, vp, ctypes.c_uint32, ctypes.c_int, ctypes.c_uint32, ctypes.c_int, vp, vp, vp]
    L.pwm_trace.restype = None
    return L


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def rays_for(bridge, n, seed, shadow):
    rng = np.random.default_rng(seed)
    root = np.asarray(bridge.tlas, np.float32).reshape(-1, 8)[0]
    ext = root[4:7] - root[0:3]
    lo, hi = root[0:3] - 0.05 * ext, root[4:7] + 0.05 * ext
    o = rng.uniform(lo, hi, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[: n // 16, rng.integers(0, 3)] = 0.0          # some axis-parallel components (inf / NaN slabs)
    rays = np.zeros((n, 8), np.float32)
    rays[:, 0:3] = o
    rays[:, 3] = 0.001
    rays[:, 4:7] = d
    rays[:, 7] = rng.uniform(0.2, 1.5, size=n).astype(np.float32) * float(np.linalg.norm(ext)) if shadow else 1e30
    return rays


SCENES = ["cornell", "viewer_diamond", "viewer_diamond_1k", "special", "mixed", "mesh", "instanced1000", "sponza_like", "glass_blob"]


@pytest.mark.parametrize("scene", SCENES)
def test_pair_walk_equals_the_reference_loop(W, oracle_lib, scene):
    b = pu.bridge_for(W, scene)
    pairs, troot, inst_root = pair_layout.build(b.tlas, b.blas, b.instances)
    tri, inst_trav = pair_layout.traversal_records(b)
    L = model_lib()
    cpu = oracle_lib.OracleRenderer()
    cpu.buildPipeline(4, 1)
    W.upload_scene(cpu, b, 16, 16)
    big = scene in ("sponza_like", "glass_blob", "instanced1000")
    n = 1500 if big else 3000
    for shadow in (False, True):
        rays = rays_for(b, n, 7 + int(shadow), shadow)
        ref, ref_counts = cpu.traceRays(rays, any_hit=shadow)
        for k, count in ((64, 1), (8, 1), (4, 0), (2, 1), (1, 1), (1, 0), (6, 0)):
            out = np.zeros((n, 4), np.float32)
            counts = np.zeros((n, 2), np.uint64)
            stats = np.zeros(4, np.uint64)
            L.pwm_trace(_p(pairs), _p(troot), _p(inst_trav), _p(inst_root), _p(tri), _p(rays), n, int(shadow), k, count,
                        _p(out), _p(counts), _p(stats))
            if shadow:
                assert np.array_equal(out[:, 3], ref[:, 3]), (scene, k, count)
            else:
                assert np.array_equal(out[:, :3].view(np.uint32), ref[:, :3].view(np.uint32)), (scene, k, count)
            if count:
                assert np.array_equal(counts, ref_counts), (scene, "shadow" if shadow else "closest", k,
                                                             int((counts != ref_counts).any(axis=1).sum()))
            if k == 64:
                assert stats[1] == 0 and stats[2] == 0          # a 64-entry stack never falls back
            if k == 1 and scene not in ("cornell", "viewer_diamond"):
                assert stats[2] > 0                              # a 1-entry stack does, and still gets everything right
