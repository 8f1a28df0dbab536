"""Phase timing of attn_kernel on DINOv2's shape (measurement tool; library built with -DBD_ATTN_PROBE into tools/_probe).

    python tools/attn_phase_probe.py build              # here (same probe library)
    python tools/attn_dino_probe.py [prec] [seq]        # on the GPU box: batch 192 x 12 heads x seq (261) x 64
Per key tile: S^T MFMAs, softmax, P.V MFMAs, barrier 1 (single-buffer classes), staging stores (incl. the wait for the tile's loads),
barrier 2; plus prologue (Q + tile 0 staged) and output."""
import ctypes as C, os, sys
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "_probe", "libbd_attn_probe.so")
import numpy as np, torch
sys.path.insert(0, ROOT); os.environ["BOXDREAMER_HIP_LIB"] = LIB
from boxdreamer_amd import _lib, hip_ops
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
seq = int(sys.argv[2]) if len(sys.argv) > 2 else 261
lib = _lib.load()
batch, heads, hd = 192, 12, 64
qkv = hip_ops.to_operand(torch.randn(batch * seq, 3 * heads * hd, device="cuda"), prec)
buf = torch.zeros(512 * 8 * 64, dtype=torch.int32, device="cuda")
for _ in range(3): hip_ops.attention(qkv, batch, seq, heads, hd, hd ** -0.5, prec=prec)
lib.bd_attn_probe_set.argtypes = [C.c_void_p]; assert lib.bd_attn_probe_set(C.c_void_p(buf.data_ptr())) == 0
torch.cuda.synchronize()
for _ in range(20): hip_ops.attention(qkv, batch, seq, heads, hd, hd ** -0.5, prec=prec)
torch.cuda.synchronize()
ts = buf.cpu().numpy().astype(np.uint32).reshape(512, 8, 64).astype(np.int64)
nt = (seq + 63) // 64
for w in (0, 1, 2):
    d = lambda i, j: ((ts[:, w, i] - ts[:, w, j]) & 0xFFFFFFFF).astype(np.float64).mean()
    print(f"wave {w}: prologue {d(61, 60):.0f}  whole loop {d(62, 61):.0f}  output {d(63, 62):.0f}  total {d(63, 60):.0f} cycles")
    for kt in range(nt):
        b = kt * 8
        nxt = d(b + 8, b + 6) if kt + 1 < nt else 0.0
        print(f"   tile {kt}: S {d(b + 1, b):.0f}  softmax {d(b + 2, b + 1):.0f}  PV {d(b + 3, b + 2):.0f}  barrier1 {d(b + 4, b + 3):.0f}  "
              f"stores(+load wait) {d(b + 5, b + 4):.0f}  barrier2 {d(b + 6, b + 5):.0f}  to next top (load issue) {nxt:.0f}")
