"""Segment timing of the ping-pong attention kernel (measurement tool; builds the library with -DBD_ATTN_PROBE into tools/_probe).

    python tools/attn_phase_probe.py build          # here
    python tools/attn_phase_probe.py run [prec]     # on the GPU box: BETR shape (32 x 8 heads x 1536 x 96)
Stamps (s_memtime) of tiles 4..15 per wave: iteration start, after the 24 slots, after the staging stores, after the barrier,
after the end-of-iteration register moves."""
import ctypes as C, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "_probe", "libbd_attn_probe.so")
if sys.argv[1] == "build":
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    src = os.path.join(ROOT, "boxdreamer_amd", "csrc")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-DBD_ATTN_PROBE",
                           "-I", os.path.join(ROOT, "include"), "-shared", "-o", LIB] + sorted(os.path.join(src, f) for f in os.listdir(src) if f.endswith(".hip")))
    print("built", LIB); sys.exit(0)
import numpy as np, torch
sys.path.insert(0, ROOT); os.environ["BOXDREAMER_HIP_LIB"] = LIB
from boxdreamer_amd import _lib, hip_ops
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
lib = _lib.load()
batch, seq, heads, hd = 32, 1536, 8, 96
qkv = hip_ops.to_operand(torch.randn(batch * seq, 3 * heads * hd, device="cuda"), prec)
buf = torch.zeros(512 * 8 * 64, dtype=torch.int32, device="cuda")
for _ in range(3): hip_ops.attention(qkv, batch, seq, heads, hd, hd ** -0.5, prec=prec)
lib.bd_attn_probe_set.argtypes = [C.c_void_p]; assert lib.bd_attn_probe_set(C.c_void_p(buf.data_ptr())) == 0
torch.cuda.synchronize()
for _ in range(20): hip_ops.attention(qkv, batch, seq, heads, hd, hd ** -0.5, prec=prec)
torch.cuda.synchronize()
ts = buf.cpu().numpy().astype(np.uint32).reshape(512, 8, 64).astype(np.int64)[:256]
d = lambda w, i, j: ((ts[:, w, i] - ts[:, w, j]) & 0xFFFFFFFF).astype(np.float64).mean()
for w in (0, 4):
    body = [d(w, 5 * k + 1, 5 * k) for k in range(12)]; st = [d(w, 5 * k + 2, 5 * k + 1) for k in range(12)]
    bar = [d(w, 5 * k + 3, 5 * k + 2) for k in range(12)]; mv = [d(w, 5 * k + 4, 5 * k + 3) for k in range(12)]
    nxt = [d(w, 5 * (k + 1), 5 * k + 4) for k in range(11)]
    print(f"wave {w}: 24 MFMA slots + softmax {np.mean(body):.0f}  staging stores {np.mean(st):.0f}  barrier wait {np.mean(bar):.0f}  register moves {np.mean(mv):.0f}"
          f"   tile period {np.mean(body) + np.mean(st) + np.mean(bar) + np.mean(mv) + np.mean(nxt):.0f} cycles (768 cycles of MFMA work per wave, 2 waves per SIMD)")
