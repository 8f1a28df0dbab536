"""CPU emulation of candidate MFMA operand schemes for the STRICT (<= 1e-3 logits) mode  (TEST INFRASTRUCTURE ONLY).

Runs the fp32 oracle with every nn.Linear replaced by an emulation of a reduced-precision operand scheme
(operands rounded exactly as the kernel would, products/accumulation in fp32/fp64 like the MFMA's fp32 accumulator)
and reports the heatmap-logit error against the plain fp32 oracle.  Used to choose the scheme BEFORE writing the
kernel; nothing here is on the product path.

    python -m oracle.numerics_sim [--views 6] [--depth 12] [--schemes f16,bf16x3,f16c8,f16c6,...]

Schemes (a = activation row block, w = weight row block, both K-contiguous):
  bf16x3   hi*hi + hi*lo + lo*hi with bf16 planes                       (round-1 strict mode, 3 MFMA passes)
  f16      one f16 pass
  f16c8    f16(a)*f16(w)  +  mx8(a - f16 a) * mx8(w)  +  mx8(a) * mx8(w - f16 w)     corrections on MX e4m3 (0.5 pass each)
  f16c6    same with MX e2m3 (fp6) correction operands                                (0.25 pass each)
  f16c4    same with MX e2m1 (fp4)
  f16c8fix f16c8 with the FIXED power-of-two scales the kernel uses (no per-block scales; BD_PREC_F16C8)
"""
from __future__ import annotations

import argparse
import sys
import time
import types

import torch
import torch.nn.functional as F

from boxdreamer_amd import synth
from oracle import boxdreamer_oracle as orc


def _blocks(x: torch.Tensor, blk: int = 32):
    k = x.shape[-1]
    pad = (-k) % blk
    if pad:
        x = F.pad(x, (0, pad))
    return x.reshape(*x.shape[:-1], -1, blk), k


def mx_quant(x: torch.Tensor, fmt: str, blk: int = 32) -> torch.Tensor:
    """Block-scaled (MX) quantise-dequantise along the last dim: one power-of-two scale per `blk` elements chosen so the
    block max lands in the top binade of the element format, elements rounded to nearest-even and saturated."""
    xb, k = _blocks(x.float(), blk)
    amax = xb.abs().amax(-1, keepdim=True).clamp_min(1e-38)
    if fmt == "e4m3":
        top = 8           # 2^8 <= 448 < 2^9
    elif fmt in ("e2m3", "e2m1"):
        top = 2           # 2^2 <= 7.5 (6.0) < 2^3
    else:
        raise ValueError(fmt)
    e = torch.floor(torch.log2(amax)) - top
    s = torch.exp2(e)
    v = xb / s
    if fmt == "e4m3":
        q = v.clamp(-448, 448).to(torch.float8_e4m3fn).float()
    else:
        mbits = 3 if fmt == "e2m3" else 1
        vmax = 7.5 if fmt == "e2m3" else 6.0
        ex = torch.floor(torch.log2(v.abs().clamp_min(1e-30))).clamp(0, 2)
        step = torch.exp2(ex - mbits)
        q = (torch.round(v / step) * step).clamp(-vmax, vmax)      # torch.round = half-to-even
    out = (q * s).reshape(*x.shape[:-1], -1)[..., :k]
    return out


def make_linear(scheme: str):
    def lin_bf16x3(x, w, b):
        xh = x.bfloat16().float(); xl = (x - xh).bfloat16().float()
        wh = w.bfloat16().float(); wl = (w - wh).bfloat16().float()
        return F.linear(xh, wh) + F.linear(xh, wl) + F.linear(xl, wh) + (0 if b is None else b)

    def lin_16(dt):
        def f(x, w, b):
            return F.linear(x.to(dt).float(), w.to(dt).float(), b)
        return f

    def lin_f16c(fmt):
        def f(x, w, b):
            xh = x.half().float(); wh = w.half().float()
            xl, wl = x - xh, w - wh
            y = F.linear(xh, wh)
            y = y + F.linear(mx_quant(xl, fmt), mx_quant(w, fmt)) + F.linear(mx_quant(x, fmt), mx_quant(wl, fmt))
            return y + (0 if b is None else b)
        return f

    def lin_f16c_one(fmt):
        # only the activation is split (weights single f16): isolates how much each side matters
        def f(x, w, b):
            xh = x.half().float(); wh = w.half().float()
            return F.linear(xh, wh) + F.linear(mx_quant(x - xh, fmt), mx_quant(w, fmt)) + (0 if b is None else b)
        return f

    def e4m3(t):
        return t.clamp(-448, 448).to(torch.float8_e4m3fn).float()

    def lin_f16c8_fixed(x, w, b, D=11):
        """The kernel's form (BD_PREC_F16C8): FIXED power-of-two scales instead of per-block ones -- activations' q plane
        un-scaled, weights' q plane scaled per tensor to the top of e4m3's range, lo planes 2^D above their q plane."""
        xh = x.half().float(); wh = w.half().float()
        sw = torch.floor(torch.log2(448.0 / w.abs().max().clamp_min(1e-30)))
        xq, xl = e4m3(x), e4m3((x - xh) * 2.0 ** D) * 2.0 ** -D
        wq, wl = e4m3(w * 2.0 ** sw) * 2.0 ** -sw, e4m3((w - wh) * 2.0 ** (sw + D)) * 2.0 ** -(sw + D)
        return F.linear(xh, wh) + F.linear(xl, wq) + F.linear(xq, wl) + (0 if b is None else b)

    def lin_f16c8_clamp448(x, w, b):
        """Round 2's kernel: the producer clamped the WHOLE activation to +-448 (hi plane included)."""
        return lin_f16c8_fixed(x.clamp(-448.0, 448.0), w, b)

    table = {"f16c8fix": lin_f16c8_fixed, "f16c8clamp": lin_f16c8_clamp448, "bf16x3": lin_bf16x3, "f16": lin_16(torch.float16), "bf16": lin_16(torch.bfloat16),
             "f16c8": lin_f16c("e4m3"), "f16c6": lin_f16c("e2m3"), "f16c4": lin_f16c("e2m1"),
             "f16a8": lin_f16c_one("e4m3"), "fp32": lambda x, w, b: F.linear(x, w, b)}
    return table[scheme]


POLICY = {}     # optional per-call override (see _FShim.linear)
STATS = {}      # range of the A operands seen by the emulated Linears (reset by run())


class _FShim(types.ModuleType):
    """torch.nn.functional with `linear` (and the patch-embed conv, an im2col GEMM on the GPU) swapped out."""

    def __init__(self, lin):
        super().__init__("F_shim")
        self._lin = lin

    def __getattr__(self, name):
        return getattr(F, name)

    def linear(self, x, w, b=None):
        STATS["max_abs_A"] = max(STATS.get("max_abs_A", 0.0), float(x.abs().max()))
        STATS["n_over_448"] = STATS.get("n_over_448", 0) + int((x.abs() > 448).sum())
        if callable(POLICY.get("fn")):      # per-call scheme selection: fn(kind, index, x, w, b) -> result or None (= the default scheme)
            kind = {(2304, 768): "qkv", (768, 768): "proj768", (3072, 768): "fc1", (768, 3072): "fc2"}.get(tuple(w.shape), "other")
            n = STATS.get("n_" + kind, 0)
            STATS["n_" + kind] = n + 1
            y = POLICY["fn"](kind, n, x.float(), w.float(), b)
            if y is not None:
                return y
        return self._lin(x.float(), w.float(), b)

    def conv2d(self, x, w, b, stride):
        # 14x14 / stride 14 patch embedding == unfold + linear (layers/patch_embed.py:65)
        n, c, H, W = x.shape
        p = stride
        cols = F.unfold(x, kernel_size=p, stride=p).transpose(1, 2)          # (n, L, c*p*p)
        y = self._lin(cols, w.reshape(w.shape[0], -1), b)                      # (n, L, out)
        return y.transpose(1, 2).reshape(n, w.shape[0], H // p, W // p)


def run(scheme: str, data, bsd, dsd):
    old = orc.F
    STATS.clear()
    orc.F = _FShim(make_linear(scheme))
    try:
        with torch.no_grad():
            return orc.boxdreamer_forward(data, bsd, dsd)
    finally:
        orc.F = old


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=6)
    ap.add_argument("--depth", type=int, default=12)
    ap.add_argument("--seed", type=int, default=11)
    ap.add_argument("--schemes", default="bf16x3,f16,f16c8,f16c8fix,f16c6,f16c4,f16a8")
    ap.add_argument("--outliers", type=float, default=0.0, help="gain of the trained-like outlier grafts (synth.*_outliers); 0 = plain")
    ap.add_argument("--rescale", action="store_true", help="function-preserving operand-range stress (synth.rescale_function_preserving)")
    a = ap.parse_args()
    torch.set_num_threads(8)
    if a.outliers > 0:
        bsd, dsd = synth.betr_state_dict_outliers(1234, a.depth, a.outliers), synth.dino_state_dict_outliers(4321, a.depth, a.outliers)
    else:
        bsd, dsd = synth.betr_state_dict(1234, a.depth), synth.dino_state_dict(4321, a.depth)
    if a.rescale:
        dsd, bsd = synth.rescale_function_preserving(dsd, bsd)
    data = synth.make_batch(seed=a.seed, B=1, T=a.views)
    with torch.no_grad():
        ref = orc.boxdreamer_forward(data, bsd, dsd)
    print(f"T={a.views} depth={a.depth}: logits rms {ref['logits'].pow(2).mean().sqrt():.3f}  max {ref['logits'].abs().max():.3f}")
    for s in a.schemes.split(","):
        t0 = time.time()
        o = run(s, data, bsd, dsd)
        err = (o["logits"] - ref["logits"]).abs()
        same = (o["topk_idx"].sort(-1)[0] == ref["topk_idx"].sort(-1)[0]).all(-1).float().mean().item()
        print(f"{s:8s} logits max-abs err {err.max():.3e}  rms {err.pow(2).mean().sqrt():.3e}  feats err "
              f"{(o['rgb_feat'] - ref['rgb_feat']).abs().max():.3e}  top20 sets equal {same:.2f}  max|A| {STATS.get('max_abs_A', 0):.0f} "
              f"(> 448: {STATS.get('n_over_448', 0)})  ({time.time() - t0:.1f}s)")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
