import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boxdreamer_amd import hip_ops, synth
def rnd(n, s, std=1.0): return torch.from_numpy(synth.bell_np(n, s, std, 0.0, 3).astype(np.float32)).cuda()
def t(fn, reps=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (M, N, act, mode) in ((50112, 3072, 1, None), (49152, 768, 0, 2), (50112, 2304, 0, 4)):
    K = 768
    x, w, b = rnd("x", (M, K), 1.5) + 0.4, rnd("w", (N, K), 0.04), rnd("b", (N,), 0.3)
    e = hip_ops.f16c8_qexp(w)
    a16, w16 = hip_ops.f16c8_encode(x, 0, False), hip_ops.f16c8_encode(w, e, True)
    xd = x.double().reshape(M, 8, 96); mean = xd.mean(-1); m2 = ((xd - mean[..., None]) ** 2).sum(-1)
    st = torch.stack([mean, m2], -1).float().contiguous()
    s = w.double().sum(1).float().contiguous()
    out = hip_ops.gemm(a16, w16, b, prec="f16c8", w_qexp=e, act=act, out_mode=mode)
    plain = lambda: hip_ops.gemm(a16, w16, b, prec="f16c8", w_qexp=e, act=act, out_mode=mode, out=out)
    fold = lambda: hip_ops.gemm(a16, w16, b, prec="f16c8", w_qexp=e, act=act, out_mode=mode, out=out, ln_apply=(st, s, 1e-6))
    for _ in range(60): plain(); fold()
    torch.cuda.synchronize()
    P, Fd = [], []
    for _ in range(6):
        P.append(t(plain)); Fd.append(t(fold))
    print(M, N, act, mode, "plain", [round(v, 1) for v in P], "fold", [round(v, 1) for v in Fd], "delta median", round(sorted(Fd)[3] - sorted(P)[3], 1))
