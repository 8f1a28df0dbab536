"""What this part's memory system delivers to simple streams (torch elementwise kernels), next to LayerNorm / proj: is 4.5-5.2 TB/s the fabric's
rate for a read + write stream, or do those kernels leave bandwidth on the table?   python tools/hbm_stream_probe.py"""
import torch, time
dev = "cuda"
def rate(name, fn, nbytes, reps=40):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name:46s} {ms*1e3:7.1f} us  {nbytes/ms/1e9:6.2f} TB/s")
M, D = 49152, 768
x = torch.randn(M, D, device=dev); y = torch.randn(M, D, device=dev); z = torch.empty_like(x)
xb = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
big = torch.randn(4 * M, D, device=dev)
rate("read only: sum of 604 MB fp32", lambda: big.sum(), big.numel() * 4)
rate("read only: sum of 151 MB fp32", lambda: x.sum(), x.numel() * 4)
rate("copy fp32 151 MB -> 151 MB", lambda: z.copy_(x), x.numel() * 8)
rate("cast fp32 151 MB -> bf16 75 MB (LayerNorm's bytes)", lambda: xb.copy_(x), x.numel() * 6)
rate("x += y in place (2 reads + 1 write, 453 MB)", lambda: x.add_(y), x.numel() * 12)
rate("z = x + bf16 (proj's bytes: 151 + 75 in, 151 out)", lambda: torch.add(x, xb, out=z), x.numel() * 10)
rate("fill 151 MB", lambda: z.zero_(), x.numel() * 4)
