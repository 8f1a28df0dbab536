"""Launch bd_attention a few times (rocprofv3 --pmc / timing).  usage: attn_one.py batch seq heads hd [prec] [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from boxdreamer_amd import hip_ops
b, s, h, d = (int(x) for x in sys.argv[1:5])
prec = sys.argv[5] if len(sys.argv) > 5 else "bf16"
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 3
qkv = hip_ops.to_operand(torch.randn(b * s, 3 * h * d, device="cuda"), prec)
for _ in range(2):
    o = hip_ops.attention(qkv, b, s, h, d, d ** -0.5, prec=prec)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    o = hip_ops.attention(qkv, b, s, h, d, d ** -0.5, prec=prec)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"attention b={b} s={s} h={h} d={d} {prec}: {ms:.3f} ms  {4.0*b*h*s*s*d/ms/1e9:.0f} TF/s")
