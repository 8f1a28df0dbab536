"""SHA-1 of bd_attention outputs over a fixed set of seeded shapes / operand classes: run with two builds (BOXDREAMER_HIP_LIB=...) and
diff the listings to show a kernel change is bit-identical."""
import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from boxdreamer_amd import hip_ops
for prec in ("bf16", "fp16", "bf16x3"):
    for batch, seq, heads, hd in ((6, 261, 12, 64), (3, 70, 2, 64), (2, 257, 12, 64), (2, 272, 4, 64), (2, 273, 4, 64), (2, 13, 4, 64), (2, 512, 8, 96)):
        g = torch.Generator().manual_seed(seq)
        qkv = hip_ops.to_operand(torch.randn(batch * seq, 3 * heads * hd, generator=g).cuda(), prec)
        out = hip_ops.attention(qkv, batch, seq, heads, hd, hd ** -0.5, prec=prec)
        t = out if not isinstance(out, (tuple, list)) else torch.cat([o.reshape(-1).view(torch.int16) for o in out])
        h = hashlib.sha1(t.contiguous().cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:12]
        print(prec, seq, hd, h)
