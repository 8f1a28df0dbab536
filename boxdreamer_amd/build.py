"""Build libboxdreamer_hip.so for gfx950 with hipcc (in-tree, so it travels to the GPU box).

    python -m boxdreamer_amd.build [--force]

hipcc cross-compiles without a GPU.  One object per .hip file (compiled in parallel), then one
shared library exposing the C ABI declared in include/boxdreamer_hip.h.
"""
from __future__ import annotations

import hashlib
import json
import os
import re
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
OBJDIR = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libboxdreamer_hip.so")
SOURCES = ["gemm.hip", "gemm_f16c8.hip", "attention.hip", "norm.hip", "layout.hip", "decode.hip", "match.hip", "pnp.hip", "forward.hip", "trace.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc"]
RESOURCES = os.path.join(OBJDIR, "resources.json")      # per kernel: VGPRs, scratch bytes, LDS, occupancy (hipcc's own remarks)
# MFMA kernel families whose register budgets are hand-tuned: a scratch spill there is a silent 10-40 % regression (a spill reload is a
# vector-memory load inside the K / key loop, profiles/r2_gemm_epilogue.md).  The build FAILS if one of them spills, except the
# instances listed here with the reason they are tolerated.
NO_SPILL_FAMILIES = ("gemm_kernel_pc", "gemm_kernel_glds", "attn_kernel")
SPILL_ALLOWED = {      # regex on the mangled name -> tolerated scratch bytes
    # (round 5: the split-bf16 hd-96 3-wave attention instance lost its 72-byte spill with the buffer-load K / V staging; no allowance left for it)
    # pipelined attention with e4m3 output (the fp8 mode's BETR attention): 3 registers of output addressing stored before the key
    # loop and reloaded after it (once per workgroup; re-deriving them after the loop makes the other output kinds spill instead)
    r"attn_kernel_ppIDF16bLi96ELi2EE": 16,
    # the persistent kernels' GENERIC epilogue (EP 0: table add, row remap, GELU into a foreign operand class) -- the patch-embed and
    # heatmap-embed GEMMs (2 of the 101 launches of a step) and the hand-off GEMMs of promoted Linears; its spills live in the epilogue
    # (profiles/r2_gemm_epilogue.md), the specialised epilogues EP 1-3 that every block Linear takes must stay at zero
    r"gemm_kernel_pc_f16c8ILi[23]ELi0ELi0ELb0ELi4ELi2ELi4ELb0ELb0EE": 256,       # (the 256 x 192 form; the small 128 x 96 form has no generic-epilogue instance)
    r"gemm_kernel_pcI.*Li4ELi0ELi0ELb0ELi0ELb0EE": 256,
}


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm's hipcc to build the gfx950 kernels)")


def _parse_resources(stderr: str) -> dict:
    """hipcc -Rpass-analysis=kernel-resource-usage remarks -> {mangled kernel name: {VGPRs, AGPRs, ScratchSize, LDS Size, Occupancy}}."""
    out, cur = {}, None
    for line in stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            out[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur:
            out[cur][m.group(1).strip()] = int(m.group(2))
    return out


def check_spills(resources: dict) -> dict:
    """Kernels of the hand-tuned families with scratch bytes beyond what SPILL_ALLOWED tolerates."""
    bad = {}
    for name, r in resources.items():
        sc = r.get("ScratchSize", 0)
        if sc <= 0 or not any(f in name for f in NO_SPILL_FAMILIES):
            continue
        if any(re.search(k, name) and sc <= v for k, v in SPILL_ALLOWED.items()):
            continue
        bad[name] = sc
    return bad


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

    resources = {}

    def cc(src):
        obj = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, "-Rpass-analysis=kernel-resource-usage", "-I", INCLUDE, "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr}")
        resources.update(_parse_resources(r.stderr))
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(cc, SOURCES))
    spills = check_spills(resources)
    if spills:
        raise RuntimeError("register spills in hand-tuned MFMA kernels (scratch bytes):\n  " + "\n  ".join(f"{k}: {v}" for k, v in spills.items()))
    with open(RESOURCES, "w") as f:
        json.dump(resources, f, indent=0, sort_keys=True)
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
