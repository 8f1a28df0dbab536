"""SHA-1 of bd_im2col_images / bd_patchify_heatmaps outputs over seeded inputs of several sizes, input dtypes and operand classes, plus their
timing at the path's shape: run with two builds (BOXDREAMER_HIP_LIB=...) and diff the listings to show a kernel change is bit-identical."""
import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from boxdreamer_amd import hip_ops

def h(t):
    ts = t if isinstance(t, (tuple, list)) else [t]
    m = hashlib.sha1()
    for x in ts:
        m.update(x.contiguous().view(torch.uint8).cpu().numpy().tobytes())
    return m.hexdigest()[:12]

for prec in ("bf16", "fp16", "bf16x3", "f16c8", "fp8"):
    for size, n in ((224, 3), (112, 2), (98, 2), (56, 5), (28, 1)):
        for dt in (torch.float32, torch.bfloat16, torch.float16):
            g = torch.Generator().manual_seed(size + n)
            img = torch.rand(n, 3, size, size, generator=g).to(dt).cuda()
            heat = (torch.rand(n, 8, size, size, generator=g) * 2 - 1).to(dt).cuda()
            print(prec, size, str(dt).split(".")[1], h(hip_ops.im2col_images(img, prec=prec)), h(hip_ops.patchify_heatmaps(heat, prec=prec)))
if "--time" in sys.argv:
    for prec in ("bf16", "f16c8"):
        img = torch.rand(192, 3, 224, 224).to(torch.bfloat16).cuda(); heat = torch.rand(192, 8, 224, 224).to(torch.bfloat16).cuda()
        for name, fn, x in (("im2col", hip_ops.im2col_images, img), ("patchify", hip_ops.patchify_heatmaps, heat)):
            for _ in range(50): fn(x, prec=prec)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): fn(x, prec=prec)
            e1.record(); torch.cuda.synchronize()
            print(f"# {prec} {name}: {e0.elapsed_time(e1) * 20:.1f} us", file=sys.stderr)
