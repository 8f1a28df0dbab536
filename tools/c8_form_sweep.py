"""F16C8 GEMM: time per launch of the path's four block shapes over the row counts of small batches (M = 1536 x B), for choosing between the
persistent kernel's forms (256 x 192 tiles on 8 + 4 waves / the small form).  Run once per library build:
    BOXDREAMER_HIP_LIB=tools/_probe/libbd_<form>.so python tools/c8_form_sweep.py        (builds: tools/r5_ab_c8forms.sh)
HIP events on torch's current stream around 60 back-to-back launches after a warm-up; microseconds per launch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from boxdreamer_amd import hip_ops

dev = torch.device("cuda")
shapes = [("proj", 768, 768, 0), ("fc2", 768, 3072, 0), ("fc1+gelu", 3072, 768, 1), ("qkv", 2304, 768, 0)]
rows = [256, 1536, 2048, 3072, 6144, 8192, 12288, 24576, 49152]
print("shape      " + " ".join(f"{m:>8d}" for m in rows))
for name, N, K, act in shapes:
    wf = torch.randn(N, K, device=dev) * 0.05
    qe = hip_ops.f16c8_qexp(wf)
    w = hip_ops.f16c8_encode(wf, qe, True)
    b = torch.randn(N, device=dev)
    line = []
    for M in rows:
        a = hip_ops.f16c8_encode(torch.randn(M, K, device=dev), 0, False)
        out16 = act == 1 or name == "qkv"
        resid = None if out16 else torch.randn(M, N, device=dev)
        o = hip_ops.gemm(a, w, b, prec="f16c8", act=act, out_f32=not out16, resid=resid, out=resid, w_qexp=qe)
        for _ in range(60):
            hip_ops.gemm(a, w, b, prec="f16c8", act=act, out_f32=not out16, resid=resid, out=o, w_qexp=qe)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 60
        e0.record()
        for _ in range(n):
            hip_ops.gemm(a, w, b, prec="f16c8", act=act, out_f32=not out16, resid=resid, out=o, w_qexp=qe)
        e1.record(); torch.cuda.synchronize()
        line.append(e0.elapsed_time(e1) / n * 1e3)
    print(f"{name:10s} " + " ".join(f"{t:8.1f}" for t in line), flush=True)
