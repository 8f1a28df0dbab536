"""Micro-benchmark of bd_gemm on the path's real shapes (config-2: M = 49152 / 50112 rows).
    python tools/gemm_bench.py [prec]          (BOXDREAMER_HIP_LIB=<other build> for A/B runs)
Random (not zero) operands; HIP events on torch's current stream (the launch stream)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from boxdreamer_amd import hip_ops, _lib

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
dev = torch.device("cuda")
shapes = [("qkv", 49152, 2304, 768, 0), ("proj", 49152, 768, 768, 0), ("fc1+gelu", 49152, 3072, 768, 1),
          ("fc2", 49152, 768, 3072, 0), ("dino qkv", 50112, 2304, 768, 0), ("dino fc1", 50112, 3072, 768, 1), ("dino fc2", 50112, 768, 3072, 0),
          ("dino proj", 50112, 768, 768, 0), ("head", 8192, 1568, 768, 0)]
tot_f = tot_t = 0.0
for name, M, N, K, act in shapes:
    a = hip_ops.to_operand(torch.randn(M, K, device=dev), prec)
    wf = torch.randn(N, K, device=dev) * 0.05
    qe = hip_ops.f16c8_qexp(wf) if prec == "f16c8" else 0
    w = hip_ops.f16c8_encode(wf, qe, True) if prec == "f16c8" else hip_ops.to_operand(wf, prec)
    b = torch.randn(N, device=dev)
    out16 = act == 1 or name in ("qkv", "dino qkv")
    resid = None if out16 else torch.randn(M, N, device=dev)
    o = hip_ops.gemm(a, w, b, prec=prec, act=act, out_f32=not out16, resid=resid, out=resid, w_qexp=qe)
    for _ in range(150):          # the clock governor needs tens of milliseconds of load before the timed launches
        hip_ops.gemm(a, w, b, prec=prec, act=act, out_f32=not out16, resid=resid, out=o, w_qexp=qe)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 30
    e0.record()
    for _ in range(n):
        hip_ops.gemm(a, w, b, prec=prec, act=act, out_f32=not out16, resid=resid, out=o, w_qexp=qe)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    tf = 2.0 * M * N * K / ms / 1e9
    tot_f += 2.0 * M * N * K; tot_t += ms
    print(f"{name:10s} M={M} N={N} K={K} act={act}: {ms:.3f} ms  {tf:.0f} TF/s (algorithmic)")
print(f"prec={prec} weighted: {tot_f / tot_t / 1e9:.0f} TF/s")
