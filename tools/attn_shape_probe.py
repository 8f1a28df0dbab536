import sys, os
sys.path.insert(0, "/root/repo")
import torch
from boxdreamer_amd import hip_ops
prec = sys.argv[1]
for batch, seq, heads, hd in ((192, 256, 12, 64), (48, 1024, 12, 64), (12, 4096, 12, 64)):
    qkv = hip_ops.to_operand(torch.randn(batch * seq, 3 * heads * hd, device="cuda"), prec)
    for _ in range(3): hip_ops.attention(qkv, batch, seq, heads, hd, hd ** -0.5, prec=prec)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): hip_ops.attention(qkv, batch, seq, heads, hd, hd ** -0.5, prec=prec)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    tiles = batch * heads * (seq // 256) * (seq // 64)
    print(f"batch {batch} seq {seq}: {ms*1e3:.0f} us  {4.0*seq*seq*hd*heads*batch/ms/1e9:.0f} TF/s (x3 passes: {3*4.0*seq*seq*hd*heads*batch/ms/1e9:.0f})  {ms*1e6/ (tiles/256):.0f} ns per (workgroup, key tile)")
