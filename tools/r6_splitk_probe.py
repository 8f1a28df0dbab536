"""Round 6: time per launch of the residual Linears (proj K = 768, fc2 K = 3072; N = 768) of the F16C8 class at the row counts of one / two poses
at a time, unsplit against split-K factors 2 .. 4 (bd_gemm_args.sk_ws / sk_split).  50 launches captured in one HIP graph (no host launch time in
the figure), replayed 20 times; microseconds per launch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from boxdreamer_amd import hip_ops

dev = torch.device("cuda")
N = 768
kinds = sys.argv[1].split(",") if len(sys.argv) > 1 else ["resid", "emit", "c8"]
print("kind  shape rows   " + " ".join(f"S={s}" .rjust(8) for s in (1, 2, 3, 4)))
for kind in kinds:
    for name, K in (("proj", 768), ("fc2", 3072)):
        wf = torch.randn(N, K, device=dev) * 0.05
        qe = hip_ops.f16c8_qexp(wf)
        w = hip_ops.f16c8_encode(wf, qe, True)
        b = torch.randn(N, device=dev)
        for M in (256, 1536, 3072):
            a = hip_ops.f16c8_encode(torch.randn(M, K, device=dev), 0, False)
            x0 = torch.randn(M, N, device=dev)
            line = []
            for S in (1, 2, 3, 4):
                kw = dict(prec="f16c8", w_qexp=qe)
                if kind in ("resid", "emit"):
                    o = torch.empty_like(x0)
                    kw.update(out_f32=True, resid=x0, out=o)
                    if kind == "emit":
                        kw["ln_emit"] = (torch.zeros((M, 8, 2), dtype=torch.float32, device=dev), torch.zeros((2, M, N), dtype=torch.float16, device=dev))
                else:
                    st = torch.zeros((M, 8, 2), dtype=torch.float32, device=dev)
                    op = hip_ops.f16c8_encode(x0, 0, False)
                    kw.update(ln_emit=(st, op), ln_resid_in_op=True, out_f32=False, out=torch.zeros((2, M, N), dtype=torch.float16, device=dev))
                if S > 1:
                    kw["split_k"] = (hip_ops.splitk_workspace(M, N), S)
                hip_ops.gemm(a, w, b, **kw)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    for _ in range(50):
                        hip_ops.gemm(a, w, b, **kw)
                g.replay(); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    g.replay()
                e1.record(); torch.cuda.synchronize()
                line.append(e0.elapsed_time(e1) / (20 * 50) * 1e3)
            print(f"{kind:5s} {name:5s} {M:5d}  " + " ".join(f"{t:8.1f}" for t in line), flush=True)
