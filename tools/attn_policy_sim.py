"""DINOv2's attention in the default (strict) mode: which operand forms can replace split-bf16 (three MFMA passes for S = Q.K^T and three for
P.V) inside the 1e-3 bar?    CPU emulation at full depth (TEST INFRASTRUCTURE: imports the oracle): the oracle's DINOv2
forward with the two attention products replaced by emulations of the kernel's arithmetic -- operands rounded as the kernel rounds them, fp32
accumulation, P = exp2(s - max) UNNORMALISED in (0, 1] with the row sum taken from the unrounded values (csrc/attention.hip:259-269) -- and,
for `--base default`, every Linear in the default mode's policy (tools/unit_sensitivity_sim.py).

    python tools/attn_policy_sim.py [--views 6] [--seed 11] [--base default|fp32] [--weights plain|outliers:0.5]

forms (QK / PV):  x3b = split-bf16, three passes (shipped) | x3h = split-f16, three passes | h = one f16 pass | x2h = f16 P x split-f16 V, two passes
"""
import argparse
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boxdreamer_amd import synth                                    # noqa: E402
from oracle import boxdreamer_oracle as orc, numerics_sim as ns    # noqa: E402
from tools.unit_sensitivity_sim import policy as linear_policy     # noqa: E402


def split(x, dt):
    hi = x.to(dt).float()
    return hi, (x - hi).to(dt).float()


def qk_product(q, k, form):
    """scores q.k^T (q already carries the softmax scale as in the oracle; the kernel folds it into the exponent -- same values up to fp32 rounding)"""
    if form == "fp32":
        return q @ k.transpose(-2, -1)
    if form == "h":
        return q.half().float() @ k.half().float().transpose(-2, -1)
    dt = torch.bfloat16 if form == "x3b" else torch.float16
    qh, ql = split(q, dt)
    kh, kl = split(k, dt)
    return qh @ kh.transpose(-2, -1) + qh @ kl.transpose(-2, -1) + ql @ kh.transpose(-2, -1)


def pv_product(s, v, form):
    m = s.amax(-1, keepdim=True)
    p = torch.exp(s - m)                       # unnormalised probabilities in (0, 1]
    l = p.sum(-1, keepdim=True)                # row sum of the UNROUNDED values (psum += pv before the conversion)
    if form == "fp32":
        return (p @ v) / l
    if form == "h":
        return (p.half().float() @ v.half().float()) / l
    if form == "x2h":
        vh, vl = split(v, torch.float16)
        ph = p.half().float()
        return (ph @ vh + ph @ vl) / l
    dt = torch.bfloat16 if form == "x3b" else torch.float16
    ph, pl = split(p, dt)
    vh, vl = split(v, dt)
    return (ph @ vh + ph @ vl + pl @ vh) / l


def dino_forward(sd, x, qk_form, pv_form, out_round, nheads=12, patch=14):
    """oracle.dino_forward_features with the attention products emulated (same citations); the attention INPUT is rounded as the QKV GEMM
    stores it (two 16-bit planes keep ~16 / ~22 bits: modelled as exact for the split forms, f16 for the single-plane form)"""
    N, _, H, W = x.shape
    dim = sd["cls_token"].shape[-1]
    FF = orc.F
    t = FF.conv2d(x, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=patch)
    t = t.flatten(2).transpose(1, 2)
    t = torch.cat((sd["cls_token"].expand(N, -1, -1), t), dim=1)
    t = t + orc.dino_pos_embed(sd, H // patch, patch)
    nreg = sd["register_tokens"].shape[1]
    t = torch.cat((t[:, :1], sd["register_tokens"].expand(N, -1, -1), t[:, 1:]), dim=1)
    hd = dim // nheads
    for i in range(12):
        p = f"blocks.{i}."
        h = F.layer_norm(t, (dim,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-6)
        L = h.shape[1]
        qkv = FF.linear(h, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"])
        qkv = qkv.reshape(N, L, 3, nheads, hd).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0] * hd ** -0.5, qkv[1], qkv[2]
        s = qk_product(q, k, qk_form)
        h = pv_product(s, v, pv_form).transpose(1, 2).reshape(N, L, dim)
        h = FF.linear(h, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
        t = t + sd[p + "ls1.gamma"] * h
        h = F.layer_norm(t, (dim,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-6)
        h = F.gelu(FF.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
        h = FF.linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
        t = t + sd[p + "ls2.gamma"] * h
    t = F.layer_norm(t, (dim,), sd["norm.weight"], sd["norm.bias"], 1e-6)
    return t[:, nreg + 1:]


def run(data, bsd, dsd, qk_form, pv_form, base):
    old_F, old_dino = orc.F, orc.dino_forward_features
    ns.STATS.clear()                     # (the policy numbers the Linear calls of ONE forward)
    if base == "default":
        ns.POLICY["fn"] = linear_policy(set(), {})
        orc.F = ns._FShim(ns.make_linear("f16c8fix"))
    orc.dino_forward_features = lambda sd, x, nheads=12, patch=14, return_stages=False: dino_forward(sd, x, qk_form, pv_form, None, nheads, patch)
    try:
        with torch.no_grad():
            return orc.boxdreamer_forward(data, bsd, dsd)
    finally:
        orc.F, orc.dino_forward_features = old_F, old_dino
        ns.POLICY.clear()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=6)
    ap.add_argument("--seed", type=int, default=11)
    ap.add_argument("--base", default="default", choices=["default", "fp32"])
    ap.add_argument("--weights", default="plain")
    ap.add_argument("--forms", default="x3b/x3b,x3h/x3h,x3h/x2h,x3h/h,x3b/h,h/x3h,h/h,fp32/fp32")
    a = ap.parse_args()
    torch.set_num_threads(int(os.environ.get("SIM_THREADS", "8")))
    if a.weights.startswith("outliers:"):
        g = float(a.weights.split(":")[1])
        bsd, dsd = synth.betr_state_dict_outliers(1234, 12, g), synth.dino_state_dict_outliers(4321, 12, g)
    else:
        bsd, dsd = synth.betr_state_dict(1234, 12), synth.dino_state_dict(4321, 12)
    data = synth.make_batch(seed=a.seed, B=1, T=a.views)
    with torch.no_grad():
        ref = orc.boxdreamer_forward(data, bsd, dsd)
    print(f"T={a.views} seed {a.seed} weights {a.weights} base {a.base}: logits rms {ref['logits'].pow(2).mean().sqrt():.3f}", flush=True)
    for f in a.forms.split(","):
        qk, pv = f.split("/")
        t0 = time.time()
        o = run(data, bsd, dsd, qk, pv, a.base)
        err = (o["logits"] - ref["logits"]).abs().max().item()
        same = (o["topk_idx"].sort(-1)[0] == ref["topk_idx"].sort(-1)[0]).all(-1).float().mean().item()
        print(f"  QK {qk:5s} PV {pv:5s}: logits max-abs err {err:.3e}  feats {(o['rgb_feat'] - ref['rgb_feat']).abs().max().item():.2e}  "
              f"top-20 sets equal {same:.2f}  ({time.time() - t0:.0f}s)", flush=True)


if __name__ == "__main__":
    main()
