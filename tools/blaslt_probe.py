"""What a tuned library GEMM reaches on the block shapes (measurement only; torch.matmul -> hipBLASLt/rocBLAS).
Gives the headroom of boxdreamer_amd's own kernels on this box; never used on the product path."""
import torch
dev = torch.device("cuda")
M = 49152
for name, N, K in (("qkv", 2304, 768), ("proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)):
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.05
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        torch.matmul(a, w.t(), out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        torch.matmul(a, w.t(), out=out)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20
    print(f"{name:5s} M={M} N={N} K={K}: {t*1e3:.0f} us  {2*M*N*K/t/1e9:.0f} TF/s (bf16 in/out, no epilogue)")
