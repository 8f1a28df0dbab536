"""Is the GEMM mainloop limited by operand delivery from L2 / fabric, or inside the CU?  Same launch with every A row (and/or
every W row) aliased to row 0 (leading dimension 0): all operand traffic then hits in L1/L2 (measurement tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from boxdreamer_amd import hip_ops
M = 49152
for N, K in ((2304, 768), (2304, 3072), (768, 3072)):
    a = hip_ops.to_operand(torch.randn(M, K, device="cuda"), "bf16")
    w = hip_ops.to_operand(torch.randn(N, K, device="cuda") * 0.05, "bf16")
    b = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for name, aa, ww in (("normal", a, w), ("A rows aliased", a[:1].expand(M, K), w), ("A and W aliased", a[:1].expand(M, K), w[:1].expand(N, K))):
        for _ in range(3):
            hip_ops.gemm(aa, ww, b, prec="bf16", out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            hip_ops.gemm(aa, ww, b, prec="bf16", out=out)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"N={N} K={K} {name:16s}: {ms*1e3:.0f} us  {2*M*N*K/ms/1e9:.0f} TF/s")
