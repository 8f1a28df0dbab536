"""Build libboxdreamer_hip.so for gfx950 with hipcc (in-tree, so it travels to the GPU box).

    python -m boxdreamer_amd.build [--force]

hipcc cross-compiles without a GPU.  One object per .hip file (compiled in parallel), then one
shared library exposing the C ABI declared in include/boxdreamer_hip.h.
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
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
OBJDIR = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libboxdreamer_hip.so")
SOURCES = ["gemm.hip", "attention.hip", "norm.hip", "layout.hip", "decode.hip", "match.hip", "pnp.hip", "forward.hip", "trace.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm's hipcc to build the gfx950 kernels)")


def _stamp() -> str:
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".h")):
            h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(open(os.path.join(INCLUDE, "boxdreamer_hip.h"), "rb").read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    stamp_file = os.path.join(OBJDIR, "stamp")
    stamp = _stamp()
    if (not force and os.path.exists(LIB) and os.path.exists(stamp_file)
            and open(stamp_file).read() == stamp):
        return LIB
    hipcc = _hipcc()
    os.makedirs(OBJDIR, exist_ok=True)

    def cc(src):
        obj = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, "-I", INCLUDE, "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr}")
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(cc, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    with open(stamp_file, "w") as f:
        f.write(stamp)
    if verbose:
        print(f"built {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
