"""LayerNorm kernel timing in three cache states (measurement tool; BOXDREAMER_HIP_LIB selects an A/B build):
   back to back on the same rows (infinity-cache resident), right after a kernel that wrote the rows (the step's situation),
   and after 1.2 GB of unrelated traffic (HBM resident).   python tools/ln_probe.py [prec]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from boxdreamer_amd import hip_ops
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
dev = "cuda"
g = torch.ones(768, device=dev); b = torch.zeros(768, device=dev)
junk = torch.empty(300_000_000, device=dev)          # 1.2 GB
def timed(fn, pre=None, reps=30):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for _ in range(5):
        if pre: pre()
        fn()
    torch.cuda.synchronize()
    for e0, e1 in ev:
        if pre: pre()
        e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    ts = sorted(e0.elapsed_time(e1) for e0, e1 in ev)
    return ts[len(ts) // 2] * 1e3
for M in (49152, 24576, 50112):
    x = torch.randn(M, 768, device=dev); y = torch.randn(M, 768, device=dev)
    nb = M * 768 * (7 if prec.startswith("f16c8") else (8 if prec.endswith("x3") else 6))
    fn = lambda: hip_ops.layernorm(x, g, b, 1e-6, prec=prec)
    t0 = timed(fn)
    t1 = timed(fn, pre=lambda: x.add_(y, alpha=1e-3))
    t2 = timed(fn, pre=lambda: junk.zero_())
    print(f"{prec} M={M}: back-to-back {t0:6.1f} us ({nb/t0/1e6:5.2f} TB/s) | after a writer {t1:6.1f} us ({nb/t1/1e6:5.2f}) | after 1.2 GB of other traffic {t2:6.1f} us ({nb/t2/1e6:5.2f})")
