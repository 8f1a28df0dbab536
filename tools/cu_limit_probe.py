"""Is the persistent GEMM power-bound?  The same GEMM (whole rounds for the grid it runs on) on 256 / 224 / 192 / 128 CUs
(A/B builds with cu_count() pinned: profiles/r4_cu_limit_probe.diff).  If the TF/s barely drops with fewer CUs, the idle CUs are free
for the other lane's HBM-bound launches.   python tools/cu_limit_probe.py <cus> [prec]   (BOXDREAMER_HIP_LIB = the matching build)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from boxdreamer_amd import hip_ops
cus = int(sys.argv[1]); prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
dev = torch.device("cuda")
for name, N, K, act, rounds in (("qkv", 2304, 768, 0, 9), ("fc1+gelu", 3072, 768, 1, 12), ("fc2", 768, 3072, 0, 3), ("proj", 768, 768, 0, 3)):
    M = rounds * cus // (N // 192) * 256            # whole rounds of `cus` workgroups
    a = hip_ops.to_operand(torch.randn(M, K, device=dev), prec)
    wf = torch.randn(N, K, device=dev) * 0.05
    qe = hip_ops.f16c8_qexp(wf) if prec == "f16c8" else 0
    w = hip_ops.f16c8_encode(wf, qe, True) if prec == "f16c8" else hip_ops.to_operand(wf, prec)
    b = torch.randn(N, device=dev)
    out16 = act == 1 or name == "qkv"
    resid = None if out16 else torch.randn(M, N, device=dev)
    o = hip_ops.gemm(a, w, b, prec=prec, act=act, out_f32=not out16, resid=resid, out=resid, w_qexp=qe)
    for _ in range(150):
        hip_ops.gemm(a, w, b, prec=prec, act=act, out_f32=not out16, resid=resid, out=o, w_qexp=qe)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 30
    e0.record()
    for _ in range(n):
        hip_ops.gemm(a, w, b, prec=prec, act=act, out_f32=not out16, resid=resid, out=o, w_qexp=qe)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"cus {cus} {prec} {name:9s} M={M}: {ms*1e3:.0f} us  {2.0*M*N*K/ms/1e9:.0f} TF/s  ({2.0*M*N*K/ms/1e9/cus:.2f} per CU)")
