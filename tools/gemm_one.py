"""Launch one GEMM shape a few times (for rocprofv3 --pmc runs).  usage: gemm_one.py M N K [prec] [act] [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from boxdreamer_amd import hip_ops
M, N, K = (int(x) for x in sys.argv[1:4])
prec = sys.argv[4] if len(sys.argv) > 4 else "bf16"
act = int(sys.argv[5]) if len(sys.argv) > 5 else 0
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 3
dev = torch.device("cuda")
a = hip_ops.to_operand(torch.randn(M, K, device=dev), prec)
w = hip_ops.to_operand(torch.randn(N, K, device=dev) * 0.05, prec)
b = torch.randn(N, device=dev)
for _ in range(reps):
    o = hip_ops.gemm(a, w, b, prec=prec, act=act)
torch.cuda.synchronize()
