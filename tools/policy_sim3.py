"""DINOv2 attention in the strict mode: what does ONE f16 pass for P.V (P in [0, 1], V rounded to f16) cost on the logits when Q.K^T stays
split-bf16 (emulated exact)?  CPU emulation at full depth on top of the f16c8 Linears (oracle/numerics_sim.py)."""
import sys, torch
from boxdreamer_amd import synth
from oracle import boxdreamer_oracle as orc, numerics_sim as ns
torch.set_num_threads(8)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 6
bsd, dsd = synth.betr_state_dict(1234, 12), synth.dino_state_dict(4321, 12)
data = synth.make_batch(seed=11, B=1, T=T)
with torch.no_grad():
    ref = orc.boxdreamer_forward(data, bsd, dsd)
real_softmax = torch.Tensor.softmax
class PVShim:
    """a @ v with both operands rounded to f16 where `a` is a softmax output of DINOv2's shape (seq 261)"""
    enabled = False
orig_matmul = torch.Tensor.__matmul__
def mm(a, b):
    if PVShim.enabled and a.dim() == 4 and a.shape[-1] == 261 and a.shape[-2] == 261 and b.shape[-1] == 64:
        return orig_matmul(a.half().float(), b.half().float())
    return orig_matmul(a, b)
for tag, on in (("f16c8 Linears, exact attention", False), ("f16c8 Linears + DINOv2 P.V in f16", True)):
    PVShim.enabled = on
    torch.Tensor.__matmul__ = mm
    try:
        o = ns.run("f16c8fix", data, bsd, dsd)
    finally:
        torch.Tensor.__matmul__ = orig_matmul
    err = (o["logits"] - ref["logits"]).abs().max().item()
    same = (o["topk_idx"].sort(-1)[0] == ref["topk_idx"].sort(-1)[0]).all(-1).float().mean().item()
    print(f"{tag:45s} logits max-abs err {err:.3e}  feats {(o['rgb_feat'] - ref['rgb_feat']).abs().max().item():.3e}  sets {same:.2f}")
