import sys, torch
sys.path.insert(0, "/root/repo")
from boxdreamer_amd import hip_ops
torch.manual_seed(0)
M, N, K = 256, 192, 64
# W exactly representable in f16 and in e4m3 after scaling: small integers / 64
w = (torch.randint(-7, 8, (N, K)).float()) / 64.0
e = hip_ops.f16c8_qexp(w)
w16 = hip_ops.f16c8_encode(w.cuda(), e, True)
wh, wl, wq = (t.cpu().double() for t in hip_ops.f16c8_decode(w16, e, True))
print("E", e, "lo_W max", wl.abs().max().item(), "q_W exact", (wq - w.double()).abs().max().item())
res = []
for k0 in range(0, 64):
    a = torch.ones(M, K)
    a[:, k0] = 1 + 2.0 ** -13
    a16 = hip_ops.f16c8_encode(a.cuda(), 0, False)
    out = hip_ops.gemm(a16, w16, None, prec="f16c8", out_f32=True, w_qexp=e).cpu().double()
    hh = hip_ops.f16c8_decode(a16)[0].cpu().double() @ wh.t()
    d = (out - hh)[0] / 2.0 ** -13          # should equal q_W[:, k0]
    # which column of W does it match?
    errs = ((wq - d[:, None]).abs().sum(0))
    kk = int(errs.argmin())
    res.append((k0, kk, float(errs.min())))
print("lo_A at k0 pairs with q_W at k (err):")
print([(a, b) if c < 1e-6 else (a, None) for a, b, c in res])
