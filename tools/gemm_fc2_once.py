"""One deep-K fp32-residual GEMM (fc2: M x 768 x 3072, in-place residual) launched `reps` times: the subject of rocprofv3 --pmc passes.
usage: gemm_fc2_once.py [M] [prec] [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from boxdreamer_amd import hip_ops
M = int(sys.argv[1]) if len(sys.argv) > 1 else 49152
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
dev = torch.device("cuda")
a = hip_ops.to_operand(torch.randn(M, 3072, device=dev), prec)
w = hip_ops.to_operand(torch.randn(768, 3072, device=dev) * 0.05, prec)
b = torch.randn(768, device=dev)
x = torch.randn(M, 768, device=dev)
for _ in range(reps):
    hip_ops.gemm(a, w, b, prec=prec, out_f32=True, resid=x, out=x)
torch.cuda.synchronize()
