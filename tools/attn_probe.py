"""Attention kernel timing at a few (batch, seq, heads, head_dim) points (measurement tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from boxdreamer_amd import hip_ops
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
for batch, seq, heads, hd in ((192, 261, 12, 64), (192, 256, 12, 64), (192, 288, 12, 64), (192, 320, 12, 64), (32, 1536, 8, 96), (32, 4352, 8, 96)):
    qkv = hip_ops.to_operand(torch.randn(batch * seq, 3 * heads * hd, device="cuda"), prec)
    for _ in range(100):          # (the clock governor: with a handful of warm-up launches a point moves by +-20 %)
        hip_ops.attention(qkv, batch, seq, heads, hd, hd ** -0.5, prec=prec)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40):
        hip_ops.attention(qkv, batch, seq, heads, hd, hd ** -0.5, prec=prec)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 40
    print(f"batch {batch} seq {seq} heads {heads} hd {hd}: {ms*1e3:.0f} us  {4.0*seq*seq*hd*heads*batch/ms/1e9:.0f} TF/s")
    if seq == 261:          # DINOv2: 5 prefix tokens + 256 patches -- the split form the encoder runs (bd_attention_prefix)
        for pq in (True, False):
            for _ in range(100):
                hip_ops.attention_prefix(qkv, batch, seq, heads, hd, hd ** -0.5, 5, prec=prec, prefix_queries=pq)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(40):
                hip_ops.attention_prefix(qkv, batch, seq, heads, hd, hd ** -0.5, 5, prec=prec, prefix_queries=pq)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 40
            print(f"   prefix split ({'patch + prefix launches' if pq else 'patch queries only (last block)'}): {ms*1e3:.0f} us  {4.0*seq*seq*hd*heads*batch/ms/1e9:.0f} TF/s")
