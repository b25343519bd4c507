"""Build libhenbun_hip.so (gfx950) in-tree with hipcc.

`python -m henbun_amd._build` or `henbun_amd._build.build()`.  hipcc
cross-compiles without a GPU, so this also serves as the driver's
"does it build" check (see __graft_entry__.build).
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJDIR = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libhenbun_hip.so")
SOURCES = ["runtime", "elementwise", "rng", "variational", "gram", "linalg", "sgp", "adam", "comm"]
ARCH = "gfx950"
FLAGS = ["-O3", "-fPIC", "-std=c++17", "--offload-arch=" + ARCH, "-Wno-unused-result"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP backend cannot be built")


def _digest() -> str:
    h = hashlib.sha256()
    names = sorted(os.listdir(CSRC))
    for n in names:
        p = os.path.join(CSRC, n)
        if os.path.isfile(p) and n.endswith((".hip", ".cuh", ".h")):
            h.update(n.encode())
            with open(p, "rb") as f:
                h.update(f.read())
    with open(os.path.join(HERE, "..", "include", "henbun_hip.h"), "rb") as f:
        h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def is_fresh() -> bool:
    stamp = LIB + ".stamp"
    if not (os.path.exists(LIB) and os.path.exists(stamp)):
        return False
    with open(stamp) as f:
        return f.read().strip() == _digest()


def build(force: bool = False, verbose: bool = True) -> str:
    """Compile every HIP source for gfx950 and link the shared library."""
    if not force and is_fresh():
        return LIB
    hipcc = _hipcc()
    os.makedirs(OBJDIR, exist_ok=True)

    def compile_one(name: str) -> str:
        src = os.path.join(CSRC, name + ".hip")
        obj = os.path.join(OBJDIR, name + ".o")
        cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (name, r.stdout, r.stderr))
        return obj

    with ThreadPoolExecutor(max_workers=min(4, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB] + objs + ["-ldl"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    with open(LIB + ".stamp", "w") as f:
        f.write(_digest())
    if verbose:
        print("built", LIB, file=sys.stderr)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
