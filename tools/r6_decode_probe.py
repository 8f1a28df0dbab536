"""Round 6: bd_decode_topk per launch (50 launches in a HIP graph) for 8 maps (one pose) and 256 maps (batch 32)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from boxdreamer_amd import hip_ops
for nm in (8, 256):
    heat = torch.tanh(torch.randn(nm // 8, 8, 224, 224, device="cuda"))
    hip_ops.decode_topk(heat); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(50):
            out = hip_ops.decode_topk(heat)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"{os.environ.get('BOXDREAMER_HIP_LIB', 'shipped')}: {nm} maps: {e0.elapsed_time(e1) / 500 * 1e3:.1f} us per launch; checksum {float(out[0].sum()):.3f} {int(out[2].sum())}")
