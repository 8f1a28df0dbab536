"""Per-launch-shape durations of the default mode's step (HIP events around every GEMM / attention launch, one lane, un-graphed):
run twice, BOXDREAMER_HIP_LNFOLD=0 / 1, and compare which Linear pays what for the LayerNorm fold.

    BOXDREAMER_HIP_LNFOLD=1 python tools/r6_lnfold_shapes.py [prec] [B]"""
import argparse, collections, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from boxdreamer_amd import _lib, synth

prec = sys.argv[1] if len(sys.argv) > 1 else "f16c8_qk16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
T = 6
dev = torch.device("cuda", 0)
args = argparse.Namespace(lanes="1", in_flight=1, cache_refs=False, graph=False, single_device_test=False)
one = synth.make_batch(seed=100, B=B, T=T)
img, bb = one["images"].to(torch.bfloat16).to(dev), one["bbox_feat"].to(torch.bfloat16).to(dev)
mask = torch.zeros(B, T, dtype=torch.bool, device=dev); mask[:, T - 1] = True
run = bench.ModeRun(prec, args, dev, 1, 0, None, img, bb, mask)
lib = _lib.load()
for _ in range(3):
    run.eager()
torch.cuda.synchronize()
recs, wall = bench.trace_launches(lib, _lib, run.eager, 5)
agg = collections.OrderedDict()
for kind, m, n, k, ms in recs:
    a = agg.setdefault((kind, m, n, k), [0, 0.0])
    a[0] += 1; a[1] += ms
tot = sum(v[1] for v in agg.values()) / 5
print(f"LNFOLD={os.environ.get('BOXDREAMER_HIP_LNFOLD', '1')} prec {prec} B {B}: traced GEMM + attention time per step {tot:.3f} ms, wall per step {wall / 5:.3f} ms")
for (kind, m, n, k), (c, ms) in agg.items():
    print(f"  {'gemm' if kind == 0 else 'attn'} M={m:6d} N={n:5d} K={k:5d}: {c // 5:3d} launches/step, {ms / c * 1e3:8.1f} us each, {ms / 5:7.3f} ms/step")
