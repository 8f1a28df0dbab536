"""Which SIDE of the f16 + e4m3 scheme carries the error: activation rounding (independent per token) or weight rounding (coherent
over all tokens)?  Variants of the per-Linear emulation (oracle/numerics_sim.py), logits max-abs error vs the fp32 oracle."""
import sys, torch
import torch.nn.functional as F
from boxdreamer_amd import synth
from oracle import boxdreamer_oracle as orc, numerics_sim as ns
torch.set_num_threads(8)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 6
seeds = [int(a) for a in sys.argv[2:]] or [11]
bsd, dsd = synth.betr_state_dict(1234, 12), synth.dino_state_dict(4321, 12)
def e4m3(t): return t.clamp(-448, 448).to(torch.float8_e4m3fn).float()
def parts(x, w, D=11):
    xh, wh = x.half().float(), w.half().float()
    sw = torch.floor(torch.log2(448.0 / w.abs().max().clamp_min(1e-30)))
    xq, xl = e4m3(x), e4m3((x - xh) * 2.0 ** D) * 2.0 ** -D
    wq, wl = e4m3(w * 2.0 ** sw) * 2.0 ** -sw, e4m3((w - wh) * 2.0 ** (sw + D)) * 2.0 ** -(sw + D)
    return xh, xl, xq, wh, wl, wq
def full(x, w, b):
    xh, xl, xq, wh, wl, wq = parts(x, w); return F.linear(xh, wh) + F.linear(xl, wq) + F.linear(xq, wl) + (0 if b is None else b)
def wonly(x, w, b):      # weights corrected, activations single f16: hi_A hi_W + q_A lo_W   (1.5 pass-equivalents, A = plain f16)
    xh, xl, xq, wh, wl, wq = parts(x, w); return F.linear(xh, wh) + F.linear(xq, wl) + (0 if b is None else b)
def aonly(x, w, b):      # activations corrected, weights single f16
    xh, xl, xq, wh, wl, wq = parts(x, w); return F.linear(xh, wh) + F.linear(xl, wq) + (0 if b is None else b)
def f16(x, w, b): return F.linear(x.half().float(), w.half().float(), b)
def run(fn, seed):
    data = synth.make_batch(seed=seed, B=1, T=T)
    with torch.no_grad():
        ref = orc.boxdreamer_forward(data, bsd, dsd)
    ns.POLICY["fn"] = fn
    o = ns.run("f16c8fix", data, bsd, dsd)
    ns.POLICY.clear()
    same = (o["topk_idx"].sort(-1)[0] == ref["topk_idx"].sort(-1)[0]).all(-1).float().mean().item()
    return (o["logits"] - ref["logits"]).abs().max().item(), same
def everywhere(lin):
    return lambda kind, n, x, w, b: lin(x, w, b)
def by_kind(table):          # table: (kind, stack) -> lin ; stack 0 = DINOv2, 1 = BETR ; missing -> full
    def fn(kind, n, x, w, b):
        if kind == "other": return None
        stack = 0 if n < 12 else 1
        if kind == "proj768":   # DINO proj: first 12; BETR: adapter fc1, fc2 (n = 12, 13), then proj
            stack = 0 if n < 12 else 1
        lin = table.get((kind, stack))
        return lin(x, w, b) if lin else None
    return fn
def qk_f16_v(vlin, stack_sel=(1,)):
    def fn(kind, n, x, w, b):
        if kind == "qkv" and ((n >= 12) if stack_sel == (1,) else True):
            y, y16 = vlin(x, w, b), f16(x, w, b)
            y = y.clone(); y[..., :1536] = y16[..., :1536]; return y
        return None
    return fn
tests = {
    "full f16c8 everywhere": None,
    "weights-only correction everywhere (1.5 passes, A = f16)": everywhere(wonly),
    "activations-only correction everywhere": everywhere(aonly),
    "BETR qkv: q,k f16 single + v full": qk_f16_v(full),
    "BETR+DINO qkv: q,k f16 single + v full": qk_f16_v(full, (0, 1)),
}
for kind in ("qkv", "proj768", "fc1", "fc2"):
    for st in (0, 1):
        tests[f"wonly in {kind} of {'DINO' if st == 0 else 'BETR'} only"] = by_kind({(kind, st): wonly})
for name, fn in tests.items():
    r = [run(fn, sd) for sd in seeds]
    print(f"{name:60s} " + "  ".join(f"{e:.2e} ({s:.2f})" for e, s in r), flush=True)
