"""The device build's short reciprocal / division / square-root sequences (csrc/k_ieee.hip.h) are the IEEE operations.

`include/mi355rt_math.h` promises single correctly rounded binary32 operations; since round 4 a gfx950 compilation
computes them with v_rcp_f32 / v_rsq_f32 and one fma-residual correction behind a range guard instead of the compiler's
11- and 16-instruction expansions.  That is only legitimate if the results are the same for every input the guard lets
through (and the guard + fallback composition for every input at all).  Through the C ABI (`rt_debug_ieee_check`):

  * reciprocal, square root, 1 / sqrt (two roundings), x / pi: ALL 2^32 inputs, lane by lane behind the guard;
  * a / b, (a0, a1, a2) / b and its zero-tolerant form: 2^24 hashed mantissa samples x 2^8 exponent classes over the
    whole range (zeros, denormals, 2^-126 .. 2^127, infinities, NaN);
  * n / 255: all 256 inputs;
  * the GPU's reference results (hipcc's IEEE expansions) equal the HOST CPU's IEEE results over the same inputs, by
    checksum (tests/model/ieee_ref.cpp) — so "equal to the expansion" is "equal to IEEE", not "equal to another GPU recipe".
The oracle is not involved: this pins arithmetic, not rendering.
"""
import ctypes
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
SRC = os.path.join(HERE, "model", "ieee_ref.cpp")
LIB = os.path.join(HERE, "model", "_build", "libieee_ref.so")
OPS = {"rcp": 0, "sqrt": 1, "rsqrt": 2, "div": 3, "div3": 4, "div3z": 5, "div_pi": 6, "unorm8": 7}


class Report(ctypes.Structure):
    _fields_ = [("n", ctypes.c_uint64), ("guard_pass", ctypes.c_uint64), ("wrong_fast", ctypes.c_uint64),
                ("wrong_fn", ctypes.c_uint64), ("checksum", ctypes.c_uint64), ("n_bad", ctypes.c_uint32),
                ("bad", ctypes.c_uint32 * 32)]


def ref_lib():
    deps = [SRC, os.path.join(REPO, "webgpu-raytracer_amd", "csrc", "k_ieee_inputs.h")]
    if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        subprocess.run(["g++", "-O2", "-ffp-contract=off", "-fno-fast-math", "-std=c++17", "-shared", "-fPIC", "-pthread", "-o", LIB, SRC],
                       check=True)
    L = ctypes.CDLL(LIB)
    L.ieee_ref_checksum.argtypes = [ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int]
    L.ieee_ref_checksum.restype = ctypes.c_uint64
    return L


def check(r, op, first, count):
    rep = Report()
    rc = r.L.rt_debug_ieee_check(r.ctx, OPS[op], first, count, ctypes.byref(rep))
    assert rc == 0, r.L.rt_last_error(r.ctx)
    return rep


def describe(rep):
    rows = ["a=%08x b=%08x got=%08x ieee=%08x" % tuple(rep.bad[4 * k:4 * k + 4]) for k in range(min(rep.n_bad, 8))]
    return "wrong behind the guard %d, wrong composed %d of %d; first: %s" % (rep.wrong_fast, rep.wrong_fn, rep.n, "; ".join(rows))


@pytest.mark.parametrize("op,count,min_fast", [
    ("rcp", 1 << 32, 4_200_000_000),       # every input but zeros / denormals / |x| > 2^126 / inf / NaN takes the sequence
    ("sqrt", 1 << 32, 1_900_000_000),      # positive x >= 2^-100 (and the two zeros)
    ("rsqrt", 1 << 32, 1_900_000_000),
    ("div_pi", 1 << 32, 3_400_000_000),
    ("div", 1 << 32, 1_000_000_000),
    ("div3", 1 << 32, 1_000_000_000),
    ("div3z", 1 << 32, 300_000_000),
    ("unorm8", 256, 256),
])
def test_sequences_are_correctly_rounded(W, gpu_renderer, op, count, min_fast):
    """Behind its guard every sequence returns the bits of the IEEE operation; the composed functions do for all inputs;
    and enough inputs really go down the sequence (a guard that rejects everything would pass vacuously)."""
    rep = check(gpu_renderer, op, 0, count)
    assert rep.n == count
    assert rep.wrong_fast == 0 and rep.wrong_fn == 0, describe(rep)
    assert rep.guard_pass >= min_fast, "only %d of %d inputs take the short sequence" % (rep.guard_pass, count)


@pytest.mark.parametrize("op,count", [("rcp", 1 << 32), ("sqrt", 1 << 32), ("rsqrt", 1 << 32), ("div_pi", 1 << 32), ("div", 1 << 32),
                                      ("div3z", 1 << 30), ("unorm8", 256)])
def test_gpu_ieee_reference_equals_the_host_cpu(W, gpu_renderer, op, count):
    """What the sequences are compared WITH on the GPU — hipcc's IEEE expansions — equals the host CPU's IEEE division and
    square root over the same inputs (checksum of all results; NaN results count as one canonical NaN)."""
    L = ref_lib()
    threads = min(16, len(os.sched_getaffinity(0)))
    want = L.ieee_ref_checksum(OPS[op], 0, count, threads)
    rep = check(gpu_renderer, op, 0, count)
    assert rep.checksum == want, "GPU checksum %016x, host %016x" % (rep.checksum, want)


def test_input_ranges_can_be_split(W, gpu_renderer):
    """The check is additive over index ranges (how a longer sweep would be chunked)."""
    whole = check(gpu_renderer, "div", 5 << 24, 1 << 22)
    a = check(gpu_renderer, "div", 5 << 24, 1 << 21)
    b = check(gpu_renderer, "div", (5 << 24) + (1 << 21), 1 << 21)
    assert (a.checksum + b.checksum) % (1 << 64) == whole.checksum
    assert a.guard_pass + b.guard_pass == whole.guard_pass
