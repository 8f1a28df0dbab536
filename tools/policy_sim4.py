"""DINOv2's QKV Linear in the strict mode: can its q, k columns run as ONE f16 pass with f16 results (as BETR's do in f16c8_qk16), v staying
F16C8?  DINOv2's q, k are NOT normalised before the softmax.  CPU emulation at full depth on top of the default mode's policy (BETR q, k one
f16 pass), plain / function-preserving-rescaled / trained-like-outlier weights (oracle/numerics_sim.py)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boxdreamer_amd import synth
from oracle import boxdreamer_oracle as orc, numerics_sim as ns
torch.set_num_threads(16)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 6
f16 = ns.make_linear("f16"); c8 = ns.make_linear("f16c8fix")
def policy(dino_qk16, round_out=True):
    def fn(kind, n, x, w, b):
        if kind != "qkv":
            return None
        if n >= 12 or dino_qk16:                       # BETR blocks (default mode) / DINOv2 blocks (the candidate)
            y8, y16 = c8(x, w, b), f16(x, w, b)
            y = y8.clone()
            y[..., :1536] = y16[..., :1536].half().float() if round_out else y16[..., :1536]
            return y
        return None
    return fn
def weights(kind):
    if kind == "plain":
        return synth.betr_state_dict(1234, 12), synth.dino_state_dict(4321, 12)
    if kind == "rescaled":
        d, b = synth.rescale_function_preserving(synth.dino_state_dict(4321, 12), synth.betr_state_dict(1234, 12))
        return b, d
    g = float(kind.split(":")[1])
    return synth.betr_state_dict_outliers(1234, 12, g), synth.dino_state_dict_outliers(4321, 12, g)
for wk in ("plain", "rescaled", "outliers:0.5"):
    bsd, dsd = weights(wk)
    for seed in (11, 12):
        data = synth.make_batch(seed=seed, B=1, T=T)
        with torch.no_grad():
            ref = orc.boxdreamer_forward(data, bsd, dsd)
        row = []
        for tag, on in (("default (f16c8_qk16)", False), ("+ DINOv2 q,k one f16 pass", True)):
            ns.POLICY["fn"] = policy(on)
            o = ns.run("f16c8fix", data, bsd, dsd)
            ns.POLICY.clear()
            err = (o["logits"] - ref["logits"]).abs().max().item()
            same = (o["topk_idx"].sort(-1)[0] == ref["topk_idx"].sort(-1)[0]).all(-1).float().mean().item()
            row.append(f"{tag}: {err:.3e} (sets {same:.2f}, feats {(o['rgb_feat'] - ref['rgb_feat']).abs().max().item():.2e})")
        print(f"T={T} weights {wk:13s} seed {seed}: " + " | ".join(row), flush=True)
