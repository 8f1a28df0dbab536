"""Does interleaving two batches on two streams (two captured graphs, kernels of one filling the tail rounds of the other) raise
throughput?  Measurement tool.   python tools/two_stream_probe.py [prec] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import bench
from boxdreamer_amd import synth
from boxdreamer_amd.graph import GraphedPath
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
B, T = 32, 6
data = synth.make_batch(seed=11, B=B, T=T)
images = data["images"].to(dev, torch.bfloat16); bbox = data["bbox_feat"].to(dev, torch.bfloat16) if "bbox_feat" in data else None
if bbox is None:
    bbox = torch.randn(B, T, 8, 224, 224, device=dev).to(torch.bfloat16)
paths = []
for i in range(2):
    enc, dec = bench.build_models(prec, dev)
    g = GraphedPath(enc, dec, B, T, 224, torch.bfloat16, dev)
    g.set_inputs(images, bbox)
    paths.append(g)
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
torch.cuda.synchronize()
def run(two):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(K):
        i = k & 1 if two else 0
        with torch.cuda.stream(streams[i]):
            paths[i].replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3
for rep in range(3):
    a = run(False); b = run(True)
    print(f"{prec}: one stream {a:.3f} ms/step ({B / a * 1e3:.0f} poses/s)   two streams / two graphs {b:.3f} ms/step ({B / b * 1e3:.0f} poses/s)")
