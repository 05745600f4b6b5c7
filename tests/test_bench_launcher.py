"""`python bench.py --gpus N` from a plain shell must start its own N ranks. The build container has
no GPU, so the launch path is exercised with BENCH_REHEARSAL=launch: real child processes under torch.distributed.run,
a gloo rendezvous on 127.0.0.1, the barrier + max-over-ranks reductions, ONE JSON line from rank 0 — and no renderer."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_starts_its_own_ranks():
    env = dict(os.environ, BENCH_REHEARSAL="launch")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["rehearsal"] == "launch" and rec["value"] is None   # cannot be mistaken for a measurement
    assert rec["n_gpus"] == 2 and rec["ranks_seen"] == 2 and rec["steps"] == 2 and rec["warmup"] == 1


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0", BENCH_REHEARSAL="launch")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                         text=True, timeout=120)
    assert out.returncode != 0 and "launcher started 3 ranks" in out.stderr
