"""Per-Linear promotion of the strict default, simulated on the CPU before it is built (oracle/numerics_sim.py).

Default mode = F16C8 Linears, BETR's q, k columns one f16 pass.  Pass 1 records, per Linear call, the per-channel |A| maxima of the
calibration batch and scores  E = max_n sqrt(sum_k (amax_k W_nk)^2)  (the size of the products whose relative rounding error the operand
class leaves behind).  A Linear is promoted to split-bf16 when E exceeds rho x the median E of its type.  Prints the logits error per rho.

    python tools/promote_sim.py [gain] [T]
"""
import sys, os, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boxdreamer_amd import synth
from oracle import boxdreamer_oracle as orc, numerics_sim as ns
torch.set_num_threads(16)
gain = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
T = int(sys.argv[2]) if len(sys.argv) > 2 else 2
f16 = ns.make_linear("f16"); c8 = ns.make_linear("f16c8fix"); x3 = ns.make_linear("bf16x3")
if gain > 0:
    bsd, dsd = synth.betr_state_dict_outliers(1234, 12, gain), synth.dino_state_dict_outliers(4321, 12, gain)
else:
    bsd, dsd = synth.betr_state_dict(1234, 12), synth.dino_state_dict(4321, 12)
data = synth.make_batch(seed=11, B=1, T=T)
with torch.no_grad():
    ref = orc.boxdreamer_forward(data, bsd, dsd)
print(f"gain {gain} T {T}: logits rms {ref['logits'].pow(2).mean().sqrt():.3f} max {ref['logits'].abs().max():.2f}")
scores = {}
def default_lin(kind, n, x, w, b):
    if kind == "qkv" and n >= 12:                      # BETR: q, k one f16 pass, v F16C8
        y8, y16 = c8(x, w, b), f16(x, w, b)
        y = y8.clone(); y[..., :1536] = y16[..., :1536]
        return y
    return c8(x, w, b)
def collect(kind, n, x, w, b):
    a = x.reshape(-1, x.shape[-1]).abs().amax(0)
    scores[(kind, n)] = ((w * a) ** 2).sum(1).sqrt().max().item()
    return default_lin(kind, n, x, w, b)
def run(fn):
    ns.POLICY["fn"] = fn
    o = ns.run("f16c8fix", data, bsd, dsd)
    ns.POLICY.clear()
    err = (o["logits"] - ref["logits"]).abs().max().item()
    same = (o["topk_idx"].sort(-1)[0] == ref["topk_idx"].sort(-1)[0]).all(-1).float().mean().item()
    return err, same, (o["rgb_feat"] - ref["rgb_feat"]).abs().max().item()
e0 = run(collect)
print(f"default: logits err {e0[0]:.3e} sets {e0[1]:.2f} feats {e0[2]:.2e}")
med = {}
for kind in {k for k, _ in scores}:
    med[kind] = statistics.median(v for (k, _), v in scores.items() if k == kind)
ratio = {k: v / med[k[0]] for k, v in scores.items()}
top = sorted(ratio.items(), key=lambda kv: -kv[1])[:16]
print("largest ratios:", ", ".join(f"{k[0]}#{k[1]}={v:.1f}" for k, v in top))
prev = None
for rho in (16, 8, 4, 2, 1.5, 0.0):
    sel = {k for k, v in ratio.items() if v > rho}
    if sel == prev:
        continue
    prev = sel
    e = run(lambda kind, n, x, w, b: x3(x, w, b) if (kind, n) in sel else default_lin(kind, n, x, w, b))
    print(f"rho {rho:5.1f}: promoted {len(sel):3d} / {len(ratio)}  logits err {e[0]:.3e} sets {e[1]:.2f} feats {e[2]:.2e}", flush=True)
