"""Build recipes for the native pieces (explicit compiler invocations, in-tree outputs).

  lib/libmi355scene.so   g++    csrc/scene/scene_compiler.cpp      (host, scene compiler)
  lib/libmi355rt.so      hipcc  csrc/rt_api.hip (+ kernels)        (gfx950, the C-ABI renderer)
  lib/libmi355tex.so     g++    csrc/texture/image_decode.cpp      (host, PNG / JPEG decode for texture ingest)
  node/mi355rt.node      gcc    node/addon.c                       (N-API shim over libmi355rt.so)

Outputs are git-ignored (*.so) but travel to the GPU box with the gpurun snapshot.
"""
import contextlib
import fcntl
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_DIR = os.path.dirname(PKG_DIR)
LIB_DIR = os.path.join(PKG_DIR, "lib")
CSRC = os.path.join(PKG_DIR, "csrc")
INCLUDE = os.path.join(REPO_DIR, "include")

SCENE_LIB = os.path.join(LIB_DIR, "libmi355scene.so")
RT_LIB = os.path.join(LIB_DIR, "libmi355rt.so")
TEX_LIB = os.path.join(LIB_DIR, "libmi355tex.so")
NODE_ADDON = os.path.join(PKG_DIR, "node", "mi355rt.node")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: the kernels must evaluate the same unfused f32 arithmetic as the oracle
# (SURVEY.md Appendix A.1 "FMA"); HIP's default would contract a*b+c.
HIP_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off", "-fno-fast-math", "-fno-gpu-rdc",
    # packed-f32 VALU ops (v_pk_mul_f32 / v_pk_add_f32) issue at half rate on gfx950 and cost extra v_mov
    # shuffles: the SLP vectoriser that creates them makes the path tracer 9% slower (measured), so it is off.
    "-fno-slp-vectorize",
    "-Wall", "-Wno-unused-function",
]


@contextlib.contextmanager
def _build_lock():
    """One builder at a time per checkout (ranks of one job may all find a library stale at once)."""
    os.makedirs(LIB_DIR, exist_ok=True)
    with open(os.path.join(LIB_DIR, ".build.lock"), "w") as f:
        fcntl.flock(f, fcntl.LOCK_EX)
        try:
            yield
        finally:
            fcntl.flock(f, fcntl.LOCK_UN)


def _compile(cmd, target):
    """Run a compiler command whose output (`-o <target>`) goes to a temporary name first and is renamed into place:
    a process that dlopens the library meanwhile sees the old file or the new one, never a partial one."""
    tmp = "%s.tmp.%d" % (target, os.getpid())
    cmd = [tmp if a == target else a for a in cmd]
    try:
        subprocess.run(cmd, check=True)
        os.replace(tmp, target)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)


def _newer(target, sources):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def _glob_sources(root, exts):
    out = []
    for d, _, files in os.walk(root):
        for f in files:
            if f.endswith(exts):
                out.append(os.path.join(d, f))
    return sorted(out)


def _headers():
    return _glob_sources(INCLUDE, (".h",))


def kernel_source_hash():
    """sha256 (first 16 hex digits) over every source the HIP library is built from (csrc/** and include/*.h, path-sorted,
    names included).  profiles/pmc_reference.json records it at collection time; bench.py recomputes it and flags counter
    figures collected from other kernels as stale."""
    import hashlib
    h = hashlib.sha256()
    files = _glob_sources(CSRC, (".hip", ".h", ".hpp", ".cpp")) + _headers()
    for f in sorted(files):
        h.update(os.path.relpath(f, REPO_DIR).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
        h.update(b"\0")
    return h.hexdigest()[:16]


def build_scene(force=False):
    src = os.path.join(CSRC, "scene", "scene_compiler.cpp")
    deps = [src] + _headers()
    if not force and _newer(SCENE_LIB, deps):
        return SCENE_LIB
    with _build_lock():
        if not force and _newer(SCENE_LIB, deps):   # another process built it while we waited
            return SCENE_LIB
        cmd = ["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-pthread", "-o", SCENE_LIB, src]
        _compile(cmd, SCENE_LIB)
    return SCENE_LIB


def build_tex(force=False):
    src = os.path.join(CSRC, "texture", "image_decode.cpp")
    deps = [src] + _headers()
    if not force and _newer(TEX_LIB, deps):
        return TEX_LIB
    with _build_lock():
        if not force and _newer(TEX_LIB, deps):
            return TEX_LIB
        cmd = ["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-Wall", "-o", TEX_LIB, src]
        _compile(cmd, TEX_LIB)
    return TEX_LIB


RT_FLAGS_SIDECAR = RT_LIB + ".flags"   # the extra compiler flags the library in the tree was built with ("" = the product build)


def _rt_flags_on_disk():
    try:
        return open(RT_FLAGS_SIDECAR).read()
    except OSError:
        return ""


def build_rt(force=False, extra_flags=()):
    """The HIP renderer.  extra_flags: a VARIANT build (sweeps, diagnostic stamps).  The flags of the library on disk are kept
    in a sidecar file, so a variant library is never taken for the product one: a call that asks for other flags than the
    library was built with rebuilds it, however new the file is."""
    src = os.path.join(CSRC, "rt_api.hip")
    deps = _glob_sources(CSRC, (".hip", ".h", ".hpp")) + _headers()
    want = " ".join(extra_flags)

    def fresh():
        return _newer(RT_LIB, deps) and _rt_flags_on_disk() == want

    if not force and fresh():
        return RT_LIB
    if not os.path.exists(HIPCC):
        raise RuntimeError("hipcc not found at %s: the HIP renderer cannot be built" % HIPCC)
    with _build_lock():
        if not force and fresh():
            return RT_LIB
        cmd = [HIPCC] + HIP_FLAGS + list(extra_flags) + ["-I", INCLUDE, "-o", RT_LIB, src]
        _compile(cmd, RT_LIB)
        with open(RT_FLAGS_SIDECAR, "w") as f:
            f.write(want)
    return RT_LIB


def build_node_addon(force=False):
    """N-API addon (plain C against /usr/include/node/node_api.h). Optional: skipped when the
    Node headers are absent."""
    src = os.path.join(PKG_DIR, "node", "addon.c")
    hdr = "/usr/include/node/node_api.h"
    if not os.path.exists(src) or not os.path.exists(hdr):
        return None
    if not force and _newer(NODE_ADDON, [src, RT_LIB, SCENE_LIB, TEX_LIB] + _headers()):
        return NODE_ADDON
    cmd = ["gcc", "-O2", "-fPIC", "-shared", "-I", "/usr/include/node", "-I", INCLUDE,
           "-o", NODE_ADDON, src, "-L", LIB_DIR, "-lmi355rt", "-lmi355scene", "-lmi355tex",
           "-Wl,-rpath,$ORIGIN/../lib"]
    with _build_lock():
        _compile(cmd, NODE_ADDON)
    return NODE_ADDON


def build_all(force=False):
    out = {"scene": build_scene(force), "tex": build_tex(force), "rt": build_rt(force)}
    try:
        out["node"] = build_node_addon(force)
    except subprocess.CalledProcessError as e:  # the addon is a convenience, not the product path
        out["node"] = None
        out["node_error"] = str(e)
    return out
