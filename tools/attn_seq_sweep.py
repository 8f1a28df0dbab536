"""Attention time against sequence length around the 64-key tile / 96-query block edges (DINOv2 geometry: 12 heads x 64, batch 192).
    python tools/attn_seq_sweep.py <bf16|fp16|bf16x3>        (BOXDREAMER_HIP_LIB=<other build> for A/B runs)
150 warm-up launches per point: with a handful the clock ramp alone moves a point by +-20 %."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from boxdreamer_amd import hip_ops
prec = sys.argv[1]
heads, hd, batch = 12, 64, 192
for seq in (256, 257, 261, 272, 288, 289, 320, 261, 256):
    qkv = hip_ops.to_operand(torch.randn(batch * seq, 3 * heads * hd, device="cuda"), prec)
    for _ in range(150):
        hip_ops.attention(qkv, batch, seq, heads, hd, hd ** -0.5, prec=prec)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40):
        hip_ops.attention(qkv, batch, seq, heads, hd, hd ** -0.5, prec=prec)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 40
    nqb = (seq + 95) // 96; nt = (seq + 63) // 64
    print(f"{prec} seq {seq}: {ms*1e3:.0f} us  q-blocks {nqb} tiles {nt} units {nqb*nt}  {ms*1e3/(nqb*nt):.2f} us/unit")
