"""DINOv2 attention: the patch-only launch (bd_attention_prefix(prefix_queries = 0)) and the five prefix queries as a second launch of the
SAME tiled kernel (bd_attention_q over the query range [0, 5)) -- one after the other on one stream, and CONCURRENTLY on two streams --
against the one-launch form.   python tools/attn_prefix_concurrent_probe.py [prec]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from boxdreamer_amd import hip_ops
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
batch, seq, heads, hd, npre = 192, 261, 12, 64, 5
qkv = hip_ops.to_operand(torch.randn(batch * seq, 3 * heads * hd, device="cuda"), prec)
zero = torch.zeros(batch, dtype=torch.int32, device="cuda")
out = hip_ops.attention(qkv, batch, seq, heads, hd, hd ** -0.5, prec=prec)
ref = out.clone()
s2 = torch.cuda.Stream()
def one(): hip_ops.attention(qkv, batch, seq, heads, hd, hd ** -0.5, prec=prec)
def patch(): hip_ops.attention_prefix(qkv, batch, seq, heads, hd, hd ** -0.5, npre, prec=prec, prefix_queries=False, out=out)
def prefix(): hip_ops.attention_prefix(qkv, batch, seq, heads, hd, hd ** -0.5, npre, prec=prec, prefix_queries=2, out=out)
def serial(): patch(); prefix()
ev = torch.cuda.Event()
def concurrent():
    main = torch.cuda.current_stream()
    ev.record(main)
    s2.wait_event(ev)
    with torch.cuda.stream(s2):
        prefix()
    patch()
    main.wait_stream(s2)
def timed(fn, reps=40):
    for _ in range(100): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for name, fn in (("one launch, 261 queries", one), ("patch queries only", patch), ("prefix queries only (tiled kernel, range [0,5))", prefix),
                 ("patch, then prefix, one stream", serial), ("patch || prefix on two streams", concurrent), ("one launch again", one)):
    print(f"{prec} {name:52s} {timed(fn):7.1f} us")

out.zero_(); concurrent(); torch.cuda.synchronize()
print("two ranges == one launch (bits):", torch.equal(out.view(torch.uint8), ref.view(torch.uint8)))
