"""Round 6: BETR attention (hd 96, 8 heads) one pose at a time: the software-pipelined 256-query kernel (48 workgroups at seq 1536) against the
128-query flash kernel (seq 1600 is not a multiple of 256, so it takes that form: 104 workgroups, 1.085x the work).  20 launches per HIP graph."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from boxdreamer_amd import hip_ops
for prec in ("f16", "bf16"):
    for b, s in ((1, 1536), (1, 1600), (2, 1536), (2, 1600), (16, 1536), (16, 1600)):
        h, d = 8, 96
        qkv = hip_ops.to_operand(torch.randn(b * s, 3 * h * d, device="cuda"), prec)
        hip_ops.attention(qkv, b, s, h, d, d ** -0.5, prec=prec); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20):
                hip_ops.attention(qkv, b, s, h, d, d ** -0.5, prec=prec)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 200 * 1e3
        print(f"{prec} b={b} seq={s}: {us:7.1f} us   {4.0*b*h*s*s*d/us/1e6:6.0f} TF/s   (per unit of seq-1536 work: {us * (1536 / s) ** 2:7.1f} us)", flush=True)
