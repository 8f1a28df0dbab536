"""Fixed-cost vs per-slab cost of bd_gemm: sweep K at fixed (M, N) and fit t = a + b*K/64."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from boxdreamer_amd import hip_ops
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
dev = torch.device("cuda")
M = 49152
for N, f32 in ((2304, False), (3072, False), (768, True), (768, False)):
    pts = []
    for K in (64, 384, 768, 1536, 3072):
        a = hip_ops.to_operand(torch.randn(M, K, device=dev), prec)
        w = hip_ops.to_operand(torch.randn(N, K, device=dev) * 0.05, prec)
        b = torch.randn(N, device=dev)
        resid = torch.randn(M, N, device=dev) if f32 else None
        o = hip_ops.gemm(a, w, b, prec=prec, out_f32=f32, resid=resid, out=resid)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            hip_ops.gemm(a, w, b, prec=prec, out_f32=f32, resid=resid, out=o)
        e1.record(); torch.cuda.synchronize()
        pts.append((K, e0.elapsed_time(e1) / 10))
    (k0, t0), (k1, t1) = pts[1], pts[-1]
    slope = (t1 - t0) / ((k1 - k0) / 64)
    print(f"N={N} f32+resid={f32}: " + "  ".join(f"K={k}:{t*1e3:.0f}us" for k, t in pts) +
          f"  | per-64-slab {slope*1e3:.1f}us, fixed {1e3*(t0 - slope*k0/64):.0f}us, K=768 total {pts[2][1]*1e3:.0f}us")
