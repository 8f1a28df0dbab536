"""per-layer policies for BETR's QKV Linear: which of the 12 may run as a single f16 pass, and q/k vs v columns"""
import sys, torch
from boxdreamer_amd import synth
from oracle import boxdreamer_oracle as orc, numerics_sim as ns
torch.set_num_threads(8)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 6
bsd, dsd = synth.betr_state_dict(1234, 12), synth.dino_state_dict(4321, 12)
f16 = ns.make_linear("f16"); c8 = ns.make_linear("f16c8fix")
def run(fn, seed=11):
    data = synth.make_batch(seed=seed, B=1, T=T)
    with torch.no_grad():
        ref = orc.boxdreamer_forward(data, bsd, dsd)
    ns.POLICY["fn"] = fn
    o = ns.run("f16c8fix", data, bsd, dsd)
    ns.POLICY.clear()
    return (o["logits"] - ref["logits"]).abs().max().item()
def betr_qkv(pred):
    def fn(kind, n, x, w, b):
        if kind == "qkv" and n >= 12 and pred(n - 12):
            return f16(x, w, b)
        return None
    return fn
def betr_qkv_cols(which):      # which: "qk" or "v" single-pass, rest f16c8
    def fn(kind, n, x, w, b):
        if kind == "qkv" and n >= 12:
            y8, y16 = c8(x, w, b), f16(x, w, b)
            y = y8.clone()
            if which == "qk": y[..., :1536] = y16[..., :1536]
            else: y[..., 1536:] = y16[..., 1536:]
            return y
        return None
    return fn
print("all f16c8            ", run(None))
print("BETR qkv all f16     ", run(betr_qkv(lambda i: True)))
print("BETR qkv f16: q,k only", run(betr_qkv_cols("qk")))
print("BETR qkv f16: v only ", run(betr_qkv_cols("v")))
for lo, hi in ((0, 6), (6, 12), (0, 4), (4, 8), (8, 12), (0, 8), (2, 12)):
    print(f"BETR qkv f16 in blocks [{lo},{hi})", run(betr_qkv(lambda i: lo <= i < hi)))
