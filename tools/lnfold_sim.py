"""LayerNorm folded into the neighbouring GEMMs -- what it does to the heatmap logits BEFORE any kernel is written.

    LN(x) W^T + b  =  rstd * (x (g . W)^T  -  mean * s)  +  (beta W^T + b),      s[n] = sum_k (g . W)[n, k]

so the consumer GEMM multiplies the RAW residual row x (rounded to the operand class by the producer GEMM's epilogue) by the gain-folded
weight and applies the row statistics in its epilogue.  What changes numerically: the operand that is rounded is x, not (x - mean) rstd g + beta
(a row mean that is large against the row's spread costs relative precision: the product carries mean * s, which the epilogue subtracts again),
and the weight that is rounded is g . W.  CPU emulation at full depth on top of the default mode's policy (F16C8 Linears with the kernel's fixed
scales, BETR q, k columns one f16 pass), folding every LayerNorm but the first of each stack (the ones a proj / fc2 epilogue precedes):
plain / function-preserving-rescaled / trained-like-outlier weights (oracle/numerics_sim.py).

    python tools/lnfold_sim.py [T] [weight kinds, comma separated]
"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boxdreamer_amd import synth
from oracle import boxdreamer_oracle as orc, numerics_sim as ns

torch.set_num_threads(16)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 6
KINDS = sys.argv[2].split(",") if len(sys.argv) > 2 else ["plain", "rescaled", "outliers:0.25", "outliers:0.5", "outliers:0.75"]
RESID3 = os.environ.get("RESID3", "0") == "1"
f16 = ns.make_linear("f16")
c8 = ns.make_linear("f16c8fix")
LOG = {}


class Shim(ns._FShim):
    """numerics_sim's shim + a layer_norm that remembers its input, so that the Linear that follows can be computed in the folded form."""

    def __init__(self, fold: bool):
        super().__init__(c8)
        self.fold = fold
        self.n_ln = 0

    def layer_norm(self, x, shape, w, b, eps):
        if RESID3 and self.fold and w is not None and x.shape[-1] == 768:
            # the residual stream itself kept in the 3-byte operand form (f16 + e4m3 remainder): every LayerNorm input is the sum a
            # residual Linear just wrote, and the caller goes on adding to the SAME tensor -- rounding it in place emulates a stream that
            # exists only in that form
            xh = x.half().float()
            x.copy_(xh + ((x - xh) * 2048.0).clamp(-448, 448).to(torch.float8_e4m3fn).float() / 2048.0)
        y = F.layer_norm(x, shape, w, b, eps)
        if self.fold and w is not None:
            y._fold = (x, w, b, eps)
        return y

    def linear(self, x, w, b=None):
        kind = {(2304, 768): "qkv", (768, 768): "proj768", (3072, 768): "fc1"}.get(tuple(w.shape), "other")
        n = ns.STATS.get("n_" + kind, 0)
        ns.STATS["n_" + kind] = n + 1
        fold = getattr(x, "_fold", None)
        first_of_stack = kind == "qkv" and n in (0, 12)          # norm1 of block 0: no residual GEMM in front of it -> stays a kernel
        betr = kind == "qkv" and n >= 12

        def product(a, wt):
            if betr:                                   # default mode: BETR's q, k columns one f16 pass, v F16C8
                y8, y16 = c8(a, wt, None), f16(a, wt, None)
                y = y8.clone()
                y[..., :1536] = y16[..., :1536]
                return y
            return c8(a, wt, None)

        if fold is None or kind not in ("qkv", "fc1") or first_of_stack:
            y = product(x.float(), w.float()) + (0 if b is None else b)
        else:
            xr, g, beta, eps = fold
            xr = xr.float()
            wg = (w.double() * g.double()[None, :]).float()                   # pack time: gain folded, THEN rounded to the operand class
            s = wg.double().sum(1)                                            # (the class represents wg to ~2^-15: its own sum is the same to that order)
            bb = (b.double() if b is not None else 0) + w.double() @ beta.double()
            mean = xr.double().mean(-1, keepdim=True)
            var = xr.double().var(-1, unbiased=False, keepdim=True)
            rstd = torch.rsqrt(var + eps)
            acc = product(xr, wg).double()
            y = (rstd * (acc - mean * s) + bb).float()
            r = (mean.abs() * rstd).max().item()
            LOG["max |mean| / std"] = max(LOG.get("max |mean| / std", 0.0), r)
            LOG["max |x|"] = max(LOG.get("max |x|", 0.0), xr.abs().max().item())
        if betr:
            y[..., :1536] = y[..., :1536].half().float()          # f16 results of the q, k columns
        return y


def run(fold, data, bsd, dsd):
    old = orc.F
    ns.STATS.clear()
    orc.F = Shim(fold)
    try:
        with torch.no_grad():
            return orc.boxdreamer_forward(data, bsd, dsd)
    finally:
        orc.F = old


def weights(kind):
    if kind == "plain":
        return synth.betr_state_dict(1234, 12), synth.dino_state_dict(4321, 12)
    if kind == "rescaled":
        d, b = synth.rescale_function_preserving(synth.dino_state_dict(4321, 12), synth.betr_state_dict(1234, 12))
        return b, d
    g = float(kind.split(":")[1])
    return synth.betr_state_dict_outliers(1234, 12, g), synth.dino_state_dict_outliers(4321, 12, g)


for wk in KINDS:
    bsd, dsd = weights(wk)
    for seed in (11, 12):
        data = synth.make_batch(seed=seed, B=1, T=T)
        with torch.no_grad():
            ref = orc.boxdreamer_forward(data, bsd, dsd)
        row = []
        for tag, on in (("default (standalone LayerNorm)", False), ("LayerNorm folded", True)):
            LOG.clear()
            o = run(on, data, bsd, dsd)
            err = (o["logits"] - ref["logits"]).abs().max().item()
            same = (o["topk_idx"].sort(-1)[0] == ref["topk_idx"].sort(-1)[0]).all(-1).float().mean().item()
            row.append(f"{tag}: {err:.3e} (sets {same:.2f}, feats {(o['rgb_feat'] - ref['rgb_feat']).abs().max().item():.2e})")
        print(f"T={T} weights {wk:13s} seed {seed}: " + " | ".join(row) + f" | folded rows: max |mean|/std {LOG.get('max |mean| / std', 0):.2f}, max |x| {LOG.get('max |x|', 0):.1f}",
              flush=True)
