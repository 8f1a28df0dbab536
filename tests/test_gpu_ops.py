"""GPU parity: every exported HIP operator against the CPU oracle / plain torch fp32 on the same
seeded inputs.  Calls go through the C ABI (ctypes); nothing here falls back to PyTorch compute."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from boxdreamer_amd import hip_ops, synth
from oracle import boxdreamer_oracle as orc

pytestmark = pytest.mark.gpu
PRECS = ["bf16", "fp16", "bf16x3"]
PRECS_LIN = PRECS + ["f16x3"]          # operand classes of the Linears / their producers (attention takes f16x3 only as an OUTPUT class)
SPLIT = ("bf16x3", "f16x3")
# operand rounding of the mode (relative); bf16x3 keeps ~16 mantissa bits, f16x3 ~22
EPS = {"bf16": 2.0 ** -8, "fp16": 2.0 ** -11, "bf16x3": 2.0 ** -15, "f16x3": 2.0 ** -20}


def _rand(name, shape, std=1.0, seed=3):
    return torch.from_numpy(synth.bell_np(name, shape, std, 0.0, seed).astype(np.float32))


def _q(x, prec):
    """round through the operand dtype (what the kernel actually multiplies)"""
    return hip_ops.from_operand(hip_ops.to_operand(x, prec), prec)


@pytest.mark.parametrize("prec", PRECS_LIN)
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 200, 128), (77, 1568, 768), (1000, 768, 3072),
                                   (2048, 768, 128), (1500, 1536, 192), (1300, 1568, 64), (2304, 2304, 768)])
def test_gemm_plain(hip, prec, M, N, K):
    a, w, b = _rand("a", (M, K)), _rand("w", (N, K), 0.05), _rand("b", (N,), 0.1)
    a16, w16 = hip_ops.to_operand(a.cuda(), prec), hip_ops.to_operand(w.cuda(), prec)
    out = hip_ops.gemm(a16, w16, b.cuda(), prec=prec, out_f32=True)
    ref = (_q(a, prec).double() @ _q(w, prec).double().t() + b.double()).float()
    err = (out.cpu() - ref).abs().max().item()
    # operands are exactly representable -> only fp32 accumulation order (and the dropped lo*lo term) differ
    tol = 2e-5 * K ** 0.5 if prec != "bf16x3" else 2e-5 * K ** 0.5 + 1e-4      # (f16x3's dropped lo*lo term is ~2^-22: no allowance)
    assert err < tol, (prec, err)
    # transpose / layout detector: asymmetric operands, exact small integers
    ai = torch.arange(M * K, dtype=torch.float32).reshape(M, K).remainder(7) - 3
    wi = torch.arange(N * K, dtype=torch.float32).reshape(N, K).remainder(5) - 2
    out = hip_ops.gemm(hip_ops.to_operand(ai.cuda(), prec), hip_ops.to_operand(wi.cuda(), prec), None, prec=prec,
                       out_f32=True)
    assert torch.equal(out.cpu(), ai @ wi.t())


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 200, 128), (77, 1568, 768), (1000, 768, 3072), (2048, 768, 128),
                                   (7500, 2304, 768), (9000, 3072, 640), (16500, 1024, 320)])
def test_gemm_f16c8(hip, M, N, K):
    """BD_PREC_F16C8: f16 hi pass + ONE e4m3 pass over [lo_A | q_A] . [q_W | lo_W] on the block-scaled MFMA (fixed
    power-of-two scales).  Checked against fp64 arithmetic on EXACTLY the three decoded planes the kernel multiplies
    (hip_ops.f16c8_decode), so only the fp32 accumulation order differs; and against the true product: the scheme must be
    >= 8x closer to it than a plain f16 pass.  Also every output form of the class: fp32 (+ in-place residual), the operand
    class itself with GELU (exact erf), an f16 plane, split-bf16 planes."""
    from boxdreamer_amd import _lib
    a, w, b = _rand("a", (M, K)), _rand("w", (N, K), 0.03), _rand("b", (N,), 0.1)
    e = hip_ops.f16c8_qexp(w)
    a16, w16 = hip_ops.f16c8_encode(a.cuda(), 0, False), hip_ops.f16c8_encode(w.cuda(), e, True)
    ah, al, aq = (t.cpu().double() for t in hip_ops.f16c8_decode(a16))
    wh, wl, wq = (t.cpu().double() for t in hip_ops.f16c8_decode(w16, e, True))
    ref = ah @ wh.t() + al @ wq.t() + aq @ wl.t() + b.double()
    out = hip_ops.gemm(a16, w16, b.cuda(), prec="f16c8", out_f32=True, w_qexp=e)
    err = (out.cpu().double() - ref).abs().max().item()
    assert err < 2e-5 * K ** 0.5, err
    true = a.double() @ w.double().t() + b.double()
    e_scheme = (out.cpu().double() - true).abs().max().item()
    e_f16 = ((ah @ wh.t() + b.double()) - true).abs().max().item()
    assert e_scheme * 8 < e_f16 + 1e-6, (e_scheme, e_f16)
    # in-place fp32 residual
    res = _rand("res", (M, N), 1.0)
    buf = res.clone().cuda()
    hip_ops.gemm(a16, w16, b.cuda(), prec="f16c8", out_f32=True, out=buf, resid=buf, w_qexp=e)
    assert (buf.cpu().double() - (ref + res.double())).abs().max().item() < 2e-5 * K ** 0.5 + 1e-6
    if N % 32 == 0:
        g = F.gelu(ref.float())
        o = hip_ops.gemm(a16, w16, b.cuda(), prec="f16c8", act=_lib.ACT_GELU, w_qexp=e)          # operand-class output
        oh, ol, oq = hip_ops.f16c8_decode(o)
        assert ((oh + ol).cpu() - g).abs().max().item() < 4e-4 * max(1.0, g.abs().max().item()) * 2.0 ** -4 + 2e-5
        assert ((oq.cpu() - g).abs() <= g.abs() * 2.0 ** -4 + 2.0 ** -9).all()
        o = hip_ops.gemm(a16, w16, b.cuda(), prec="f16c8", out_mode=2, w_qexp=e)                 # f16 plane
        assert (o.float().cpu() - ref.float()).abs().max().item() < 2.0 ** -10 * max(1.0, ref.abs().max().item())
        o = hip_ops.gemm(a16, w16, b.cuda(), prec="f16c8", out_mode=4, w_qexp=e)                 # split-bf16 planes
        assert ((o[0].float() + o[1].float()).cpu() - ref.float()).abs().max().item() < 2.0 ** -15 * max(1.0, ref.abs().max().item()) + 2e-5 * K ** 0.5
        # kind 5: split-f16 planes (an F16C8 Linear feeding a PROMOTED one; ADVICE r4: the wrapper used to lay this kind out wrongly and
        # no op-level test covered it) -- plain and with GELU (the fc1 -> promoted fc2 hand-off), hi / lo consistent (lo = f16(x - hi))
        for act, want in ((_lib.ACT_NONE, ref.float()), (_lib.ACT_GELU, g)):
            o = hip_ops.gemm(a16, w16, b.cuda(), prec="f16c8", out_mode=5, act=act, w_qexp=e)
            assert o.dtype == torch.float16 and tuple(o.shape) == (2, M, N)
            assert ((o[0].float() + o[1].float()).cpu() - want).abs().max().item() < 2.0 ** -20 * max(1.0, want.abs().max().item()) + 2e-5 * K ** 0.5
            assert (o[1].float().abs() <= o[0].float().abs() * 2.0 ** -10 + 2.0 ** -24).all()


@pytest.mark.parametrize("prec", PRECS_LIN + ["fp8"])
@pytest.mark.parametrize("M,N,K", [(7500, 2304, 768), (9000, 3072, 640), (16500, 1024, 320)])
def test_gemm_block_sized(hip, prec, M, N, K):
    """Transformer-block sized Linears (hundreds of 256x256 / 128x128 tiles, several tile rounds per CU, ragged last
    M-tile): exact on small integers through the 16-bit output path, GELU + bias, and the in-place fp32 residual form."""
    if prec == "fp8" and K % 128:
        pytest.skip("fp8 slabs are 128 deep")
    dev = torch.device("cuda")
    ai = ((torch.arange(M * K, dtype=torch.float32, device=dev).reshape(M, K) * 7).remainder(5) - 2)
    wi = ((torch.arange(N * K, dtype=torch.float32, device=dev).reshape(N, K) * 3).remainder(3) - 1)
    bi = (torch.arange(N, dtype=torch.float32, device=dev).remainder(9) - 4)
    exact = ai @ wi.t() + bi                                    # small integers: exact in fp32
    a16, w16 = hip_ops.to_operand(ai, prec), hip_ops.to_operand(wi, prec)
    if prec == "fp8":
        o = hip_ops.gemm(a16, w16, bi, prec=prec, out_mode=3)   # bf16 plane (qkv of the fp8 mode)
        assert o.dtype == torch.bfloat16 and torch.equal(o, exact.to(torch.bfloat16))
    else:
        o = hip_ops.gemm(a16, w16, bi, prec=prec)               # operand-dtype output, RNE of the exact value
        got = hip_ops.from_operand(o, prec)
        want = hip_ops.from_operand(hip_ops.to_operand(exact, prec), prec)
        assert torch.equal(got, want)
    # random data + bias + GELU (fc1), against fp32 math on the dequantised operands
    a, w, b = _rand("pa", (M, K)).to(dev), _rand("pw", (N, K), 0.05).to(dev), _rand("pb", (N,), 0.1).to(dev)
    a16, w16 = hip_ops.to_operand(a, prec), hip_ops.to_operand(w, prec)
    aq, wq = hip_ops.from_operand(a16, prec), hip_ops.from_operand(w16, prec)
    ref = F.gelu(aq @ wq.t() + b)
    o = hip_ops.from_operand(hip_ops.gemm(a16, w16, b, prec=prec, act=1), prec) if prec != "fp8" else \
        hip_ops.gemm(a16, w16, b, prec=prec, act=1).float()
    eps = 2.0 ** -4 if prec == "fp8" else EPS[prec]
    assert (o - ref).abs().max().item() <= eps * max(1.0, ref.abs().max().item()) + 3e-4 * K ** 0.5
    # fp32 output accumulated in place onto the residual stream (proj / fc2)
    res = _rand("pr", (M, N)).to(dev)
    buf = res.clone()
    hip_ops.gemm(a16, w16, b, prec=prec, out_f32=True, out=buf, resid=buf)
    want = aq @ wq.t() + b + res
    tol = (2e-4 if prec != "bf16x3" else 4e-4) * K ** 0.5
    assert (buf - want).abs().max().item() <= tol


@pytest.mark.parametrize("prec", PRECS_LIN)
def test_gemm_epilogues(hip, prec):
    M, N, K, P, TPI, OFF = 512, 256, 128, 256, 261, 5
    a, w, b = _rand("a2", (M, K)), _rand("w2", (N, K), 0.05), _rand("b2", (N,), 0.1)
    tab, res = _rand("tab", (P, N)), _rand("res", ((M // P) * TPI, N))
    a16, w16 = hip_ops.to_operand(a.cuda(), prec), hip_ops.to_operand(w.cuda(), prec)
    base = (_q(a, prec).double() @ _q(w, prec).double().t() + b.double()).float()
    # gelu + 16-bit output
    o = hip_ops.gemm(a16, w16, b.cuda(), prec=prec, act=1)
    ref = F.gelu(base)
    # the 16-bit store rounds the result itself: half an operand ulp of the largest value
    assert (hip_ops.from_operand(o, prec).cpu() - ref).abs().max().item() < EPS[prec] * max(1.0, ref.abs().max().item()) + 2e-4
    # row remap + table + residual, fp32 in-place style output
    out_rows = (M // P) * TPI
    buf = res.clone().cuda()
    hip_ops.gemm(a16, w16, b.cuda(), prec=prec, out_f32=True, out=buf, resid=buf, addtab=tab.cuda(),
                 rpg=(P, TPI, OFF), out_rows=out_rows)
    exp = res.clone()
    rows = torch.arange(M)
    orow = (rows // P) * TPI + rows % P + OFF
    exp[orow] = base + tab[rows % P] + res[orow]
    assert (buf.cpu() - exp).abs().max().item() < 3e-4
    untouched = torch.ones(out_rows, dtype=torch.bool); untouched[orow] = False
    assert torch.equal(buf.cpu()[untouched], res[untouched])


@pytest.mark.parametrize("prec", PRECS_LIN)
@pytest.mark.parametrize("affine,eps", [(True, 1e-5), (True, 1e-6), (False, 1e-6)])
def test_layernorm(hip, prec, affine, eps):
    x = _rand("lnx", (1001, 768), 2.0) + 0.3
    g = _rand("lng", (768,), 0.1) + 1 if affine else None
    b = _rand("lnb", (768,), 0.1) if affine else None
    o16, o32 = hip_ops.layernorm(x.cuda(), g.cuda() if affine else None, b.cuda() if affine else None, eps,
                                 prec=prec, want32=True)
    ref = F.layer_norm(x, (768,), g, b, eps)
    assert (o32.cpu() - ref).abs().max().item() < 2e-5
    assert (hip_ops.from_operand(o16, prec).cpu() - ref).abs().max().item() < 8 * EPS[prec] + 1e-5
    # gathered rows (DINO final norm drops 5 prefix tokens per 261)
    o16, o32 = hip_ops.layernorm(x.cuda(), None, None, eps, prec=prec, want32=True, rows=3 * 256, rpg=(256, 261, 5))
    idx = (torch.arange(768) // 256) * 261 + torch.arange(768) % 256 + 5
    assert (o32.cpu() - F.layer_norm(x[idx], (768,), None, None, eps)).abs().max().item() < 2e-5


@pytest.mark.parametrize("prec", PRECS_LIN)
@pytest.mark.parametrize("hd,heads", [(96, 8), (64, 12)])
def test_qk_rmsnorm(hip, prec, hd, heads):
    rows = 333
    qkv = _rand("rmsq", (rows, 3, heads, hd), 1.5) + 0.1
    wq, wk = _rand("wq", (hd,), 0.1) + 1, _rand("wk", (hd,), 0.1) + 1
    t = hip_ops.to_operand(qkv.reshape(rows, -1).cuda(), prec)
    hip_ops.qk_rmsnorm_(t, wq.cuda(), wk.cuda(), 1e-6, heads, hd, prec=prec)
    got = hip_ops.from_operand(t, prec).cpu().reshape(rows, 3, heads, hd)
    src = _q(qkv.reshape(rows, -1), prec).reshape(rows, 3, heads, hd)
    tol = 8 * EPS[prec] + 1e-5
    assert (got[:, 0] - orc._rmsnorm(src[:, 0], wq)).abs().max().item() < tol
    assert (got[:, 1] - orc._rmsnorm(src[:, 1], wk)).abs().max().item() < tol
    assert torch.equal(got[:, 2], src[:, 2])           # v untouched


@pytest.mark.parametrize("prec", ["bf16", "fp16", "bf16x3", "f16x3", "fp8"])
@pytest.mark.parametrize("N,K,kind", [(2304, 768, "plain"), (2304, 768, "rms"), (3072, 768, "gelu"), (768, 768, "resid"),
                                      (768, 3072, "resid"), (768, 768, "f32")])
def test_gemm_row_result_independent_of_tile_shape(hip, prec, N, K, kind):
    """A row's result must not depend on the kernel / tile shape that computed it: the first rows of a 49152-row launch
    (persistent 256x192 kernel, specialised epilogues) equal, bit for bit, the same rows launched alone at 1536 / 300 / 4096
    rows (128x128, 64x64 and hybrid tiles, generic epilogue).  This is what makes a sample's output independent of its batch
    (tests/test_gpu_path.py::test_full_size_properties); it pins the accumulator-start / bias / residual convention, the
    GELU arithmetic and the separate fp32 -> 16-bit rounding across every kernel."""
    M = 49152
    a, w, b = _rand("a", (M, K)).cuda(), (_rand("w", (N, K), 0.05)).cuda(), _rand("b", (N,), 0.5).cuda()
    a16, w16 = hip_ops.to_operand(a, prec), hip_ops.to_operand(w, prec)
    res = _rand("r", (M, N)).cuda() if kind == "resid" else None
    ws = (torch.rand(N, generator=torch.Generator().manual_seed(5)) + 0.5).cuda() if prec == "fp8" else None

    def go(rows):
        aa = a16[:, :rows].contiguous() if a16.dim() == 3 else a16[:rows].contiguous()
        kw = {"wscale": ws}
        if kind == "resid":
            kw.update(resid=res[:rows].clone(), out_f32=True)
        elif kind == "gelu":
            kw.update(act=1)
        elif kind == "f32":
            kw.update(out_f32=True)
        elif kind == "rms":
            kw.update(rms=((_rand("wq", (96,), 0.1) + 1).cuda(), (_rand("wk", (96,), 0.1) + 1).cuda(), 1e-6))
        return hip_ops.gemm(aa, w16, b, prec=prec, **kw)

    big = go(M)
    for rows in (1536, 300, 4096):
        small = go(rows)
        head = big[:, :rows] if big.dim() == 3 else big[:rows]
        assert torch.equal(head.contiguous().view(torch.uint8), small.contiguous().view(torch.uint8)), (prec, kind, rows)


@pytest.mark.parametrize("N,K,kind", [(768, 768, "resid"), (768, 3072, "resid"), (3072, 768, "gelu"), (2304, 768, "plain"), (2304, 768, "rms_f16"),
                                      (2304, 768, "rms_operand"), (2304, 768, "split_bf16"), (768, 768, "f32")])
def test_gemm_f16c8_small_form_rows_equal_the_large_form(hip, N, K, kind):
    """The F16C8 class's persistent kernel has two forms (round 5): 256 x 192 tiles on 8 + 4 waves, and -- for launches whose large tiles would
    fill their rounds badly (one pose at a time: 1536 rows) -- 128 x 192 tiles on 4 + 4 waves, one consumer wave per SIMD.  Same wave tile, K
    order and epilogue arithmetic: the first rows of a 49152-row launch (large form) must equal, bit for bit, the same rows launched alone at
    1536 / 300 / 4096 rows (small form; ragged last tile at 300), for every specialised epilogue of the path."""
    M = 49152
    a, w, b = _rand("a", (M, K)).cuda(), _rand("w", (N, K), 0.05).cuda(), _rand("b", (N,), 0.5).cuda()
    e = hip_ops.f16c8_qexp(w)
    w16 = hip_ops.f16c8_encode(w, e, True)
    res = _rand("r", (M, N)).cuda() if kind == "resid" else None
    rms = ((_rand("wq", (96,), 0.1) + 1).cuda(), (_rand("wk", (96,), 0.1) + 1).cuda(), 1e-6)

    def go(rows):
        kw = {"w_qexp": e}
        if kind == "resid":
            kw.update(resid=res[:rows].clone(), out_f32=True)
        elif kind == "f32":
            kw.update(out_f32=True)
        elif kind == "gelu":
            kw.update(act=1)
        elif kind == "rms_f16":
            kw.update(rms=rms, out_mode=2)
        elif kind == "rms_operand":
            kw.update(rms=rms)
        elif kind == "split_bf16":
            kw.update(out_mode=4)
        return hip_ops.gemm(hip_ops.f16c8_encode(a[:rows], 0, False), w16, b, prec="f16c8", **kw)

    big = go(M)
    for rows in (1536, 300, 4096):
        small = go(rows)
        if kind in ("rms_operand", "gelu", "plain"):                      # F16C8 operand out: f16 plane + the byte plane's first rows * N bytes
            assert torch.equal(big[0][:rows], small[0]), (kind, rows, "f16 plane")
            nb = rows * N
            assert torch.equal(big[1].view(torch.uint8).reshape(-1)[:nb], small[1].view(torch.uint8).reshape(-1)[:nb]), (kind, rows, "lo8 plane")
        else:
            head = big[:, :rows] if big.dim() == 3 else big[:rows]
            assert torch.equal(head.contiguous().view(torch.uint8), small.contiguous().view(torch.uint8)), (kind, rows)


@pytest.mark.parametrize("prec", ["bf16", "f16x3", "fp8", "f16c8"])
@pytest.mark.parametrize("K", [768, 3072])
def test_gemm_rows_of_a_sparse_last_round_take_smaller_tiles(hip, prec, K):
    """DINOv2's M = 50112 rows are 195.75 row tiles of 256: at N = 768 (proj, fc2) 3 full rounds of 256 x 192 tiles + 16 tiles.  bd_gemm
    then runs the last 960 rows as a second launch of one-tile kernels (gemm_common.h: pc192_main_rows; the F16C8 class keeps its one
    launch, profiles/r4_gemm_tail_rows.md).  Every row must equal, bit for bit, the same row launched without the split: rows
    [0, 49152) alone (whole rounds, one launch) and rows [49152, 50112) alone (below the split's threshold)."""
    M, N, main = 50112, 768, 49152
    a, w, b = _rand("a", (M, K)).cuda(), _rand("w", (N, K), 0.05).cuda(), _rand("b", (N,), 0.5).cuda()
    res = _rand("r", (M, N)).cuda()
    kw = {}
    if prec == "f16c8":
        kw["w_qexp"] = hip_ops.f16c8_qexp(w)
        w16 = hip_ops.f16c8_encode(w, kw["w_qexp"], True)
    else:
        w16 = hip_ops.to_operand(w, prec)
    if prec == "fp8":
        kw["wscale"] = (torch.rand(N, generator=torch.Generator().manual_seed(5)) + 0.5).cuda()

    def go(r0, r1):
        return hip_ops.gemm(hip_ops.to_operand(a[r0:r1], prec), w16, b, prec=prec, resid=res[r0:r1].clone(), out_f32=True, **kw)

    full = go(0, M)
    assert torch.equal(full[:main], go(0, main)), (prec, K, "rows of the full rounds")
    assert torch.equal(full[main:], go(main, M)), (prec, K, "rows of the sparse round")
    ref = hip_ops.from_operand(hip_ops.to_operand(a[main:], prec), prec).double() @ (
        hip_ops.f16c8_decode(w16, kw["w_qexp"], True)[0].double() if prec == "f16c8" else hip_ops.from_operand(w16, prec).double()).t()
    if prec == "fp8":
        ref = ref * kw["wscale"].double()
    ref = ref + b.double() + res[main:].double()
    tol = {"bf16": 1e-3, "f16x3": 1e-4, "fp8": 1e-3, "f16c8": 2e-2}[prec]       # (F16C8: ref uses W's f16 plane only)
    assert (full[main:].double() - ref).abs().max().item() < tol * K ** 0.5, (prec, K)


@pytest.mark.parametrize("prec,out_mode", [("bf16", None), ("fp16", None), ("bf16x3", None), ("bf16x3", 2), ("f16c8", 2), ("f16c8", 4)])
@pytest.mark.parametrize("M", [300, 4096, 49152 // 8])
def test_gemm_fused_qk_rmsnorm(hip, prec, out_mode, M):
    """QKV Linear with the q/k RMSNorm (LlamaRMSNorm, blocks.py:44-56, applied after the head split, :257) fused into the
    epilogue: q and k heads come out as  w * x * rsqrt(mean_96(x^2) + eps)  of the fp32 Linear result, v untouched.  Checked
    against fp64 on the operands as stored; every output form the whole-path modes use; M below one tile and ragged."""
    from boxdreamer_amd import _lib
    heads, hd, K = 8, 96, 768
    N = 3 * heads * hd
    a, w, b = _rand("a", (M, K)), _rand("w", (N, K), 0.05), _rand("b", (N,), 0.1)
    wq, wk = (_rand("wq", (hd,), 0.1) + 1).cuda(), (_rand("wk", (hd,), 0.1) + 1).cuda()
    if prec == "f16c8":
        e = hip_ops.f16c8_qexp(w)
        a16, w16 = hip_ops.f16c8_encode(a.cuda(), 0, False), hip_ops.f16c8_encode(w.cuda(), e, True)
        ah, al, aq = (t.cpu().double() for t in hip_ops.f16c8_decode(a16))
        wh, wl, wqq = (t.cpu().double() for t in hip_ops.f16c8_decode(w16, e, True))
        lin = ah @ wh.t() + al @ wqq.t() + aq @ wl.t() + b.double()
    else:
        e = 0
        a16, w16 = hip_ops.to_operand(a.cuda(), prec), hip_ops.to_operand(w.cuda(), prec)
        lin = _q(a, prec).double() @ _q(w, prec).double().t() + b.double()
    g = _lib.GemmArgs()
    out = hip_ops.gemm(a16, w16, b.cuda(), prec=prec, out_mode=out_mode, w_qexp=e, rms=(wq, wk, 1e-6))
    x = lin.reshape(M, 3, heads, hd)
    ref = x.clone()
    ref[:, 0] = wq.cpu().double() * (x[:, 0] * torch.rsqrt(x[:, 0].pow(2).mean(-1, keepdim=True) + 1e-6))
    ref[:, 1] = wk.cpu().double() * (x[:, 1] * torch.rsqrt(x[:, 1].pow(2).mean(-1, keepdim=True) + 1e-6))
    ref = ref.reshape(M, N).float()
    if out_mode == 2:
        got, eps = out.float().cpu(), 2.0 ** -11
    elif out_mode == 4:
        got, eps = (out[0].float() + out[1].float()).cpu(), 2.0 ** -15
    else:
        got, eps = hip_ops.from_operand(out, prec).cpu(), EPS[prec]
    err = (got - ref).abs().max().item()
    assert err < 2 * eps * max(1.0, ref.abs().max().item()) + 1e-4, (prec, out_mode, err)


@pytest.mark.parametrize("M", [300, 6144])
def test_gemm_fused_qk_rmsnorm_two_parts(hip, M):
    """rms_parts = 2: the q, k launch of a column-split QKV Linear (BD_PREC_F16C8_QK16): N = 2 x heads x 96 output columns, BOTH halves
    normalised (q with wq, k with wk), written with the full [q | k | v] row stride next to a v block that must stay untouched."""
    heads, hd, K = 8, 96, 768
    D = heads * hd
    a, w, b = _rand("a", (M, K)), _rand("w", (2 * D, K), 0.05), _rand("b", (2 * D,), 0.1)
    wq, wk = (_rand("wq", (hd,), 0.1) + 1).cuda(), (_rand("wk", (hd,), 0.1) + 1).cuda()
    a16, w16 = hip_ops.to_operand(a.cuda(), "fp16"), hip_ops.to_operand(w.cuda(), "fp16")
    full = torch.full((M, 3 * D), 7.0, dtype=torch.float16, device="cuda")
    out = full[:, : 2 * D]                                   # a view with row stride 3 D
    hip_ops.gemm(a16, w16, b.cuda(), prec="fp16", out=out, rms=(wq, wk, 1e-6, 2))
    lin = _q(a, "fp16").double() @ _q(w, "fp16").double().t() + b.double()
    x = lin.reshape(M, 2, heads, hd)
    ref = x.clone()
    ref[:, 0] = wq.cpu().double() * (x[:, 0] * torch.rsqrt(x[:, 0].pow(2).mean(-1, keepdim=True) + 1e-6))
    ref[:, 1] = wk.cpu().double() * (x[:, 1] * torch.rsqrt(x[:, 1].pow(2).mean(-1, keepdim=True) + 1e-6))
    got = full[:, : 2 * D].float().cpu()
    err = (got - ref.reshape(M, 2 * D).float()).abs().max().item()
    assert err < 2 * 2.0 ** -11 * max(1.0, ref.abs().max().item()) + 1e-4, err
    assert torch.equal(full[:, 2 * D:], torch.full((M, D), 7.0, dtype=torch.float16, device="cuda"))


@pytest.mark.parametrize("prec", ["bf16", "bf16x3", "f16c8"])
def test_fused_qk_rmsnorm_refuses_odd_head_counts(hip, prec):
    """with an odd head count (N = 3 x 3 x 96 = 864, not a multiple of the 192-column workgroup tile) the last column
    tile's second 96-column wave tile lies past N and the fused branch has no column guard.  The geometry check must say "not
    fused" (the whole-path entry points then run the separate bd_qk_rmsnorm kernel) and bd_gemm must reject rms_wq outright --
    never write past column N.  A guard row/column around the output proves nothing was written outside."""
    import ctypes as C
    from boxdreamer_amd import _lib
    heads, hd, K, M = 3, 96, 768, 700
    N = 3 * heads * hd
    a, w, b = _rand("a", (M, K)), _rand("w", (N, K), 0.05), _rand("b", (N,), 0.1)
    wq, wk = (_rand("wq", (hd,), 0.1) + 1).cuda(), (_rand("wk", (hd,), 0.1) + 1).cuda()
    e = hip_ops.f16c8_qexp(w) if prec == "f16c8" else 0
    a16 = hip_ops.to_operand(a.cuda(), prec)
    w16 = hip_ops.f16c8_encode(w.cuda(), e, True) if prec == "f16c8" else hip_ops.to_operand(w.cuda(), prec)
    g = _lib.GemmArgs()
    g.M, g.N, g.K = M, N, K
    g.rms_wq, g.rms_wk, g.rms_eps = wq.data_ptr(), wk.data_ptr(), 1e-6
    g.A, g.W, g.lda, g.ldw, g.ldo = a16.data_ptr(), w16.data_ptr(), K, K, N
    assert _lib.load().bd_gemm_fuses_qk_rmsnorm(C.byref(g), _lib.prec_id(prec)) == 0
    with pytest.raises(_lib.HipLibraryError, match="BD_ERR_SHAPE"):
        hip_ops.gemm(a16, w16, b.cuda(), prec=prec, w_qexp=e, out_mode=2, rms=(wq, wk, 1e-6))
    # the unfused pair still gives the right answer for this geometry: Linear (f16 plane out) + in-place q/k RMSNorm
    out = hip_ops.gemm(a16, w16, b.cuda(), prec=prec, w_qexp=e, out_mode=2 if prec != "bf16" else None)
    if prec == "bf16":
        hip_ops.qk_rmsnorm_(out, wq, wk, 1e-6, heads, hd, prec="bf16")
        got, eps = out.float().cpu(), EPS["bf16"]
    else:
        hip_ops.qk_rmsnorm_(out, wq, wk, 1e-6, heads, hd, prec="fp16")
        got, eps = out.float().cpu(), 2.0 ** -11
    if prec == "f16c8":
        ah, al, aq = (t.cpu().double() for t in hip_ops.f16c8_decode(a16))
        wh, wl, wqq = (t.cpu().double() for t in hip_ops.f16c8_decode(w16, e, True))
        lin = ah @ wh.t() + al @ wqq.t() + aq @ wl.t() + b.double()
    else:
        lin = _q(a, prec).double() @ _q(w, prec).double().t() + b.double()
    x = lin.reshape(M, 3, heads, hd)
    ref = x.clone()
    ref[:, 0] = wq.cpu().double() * (x[:, 0] * torch.rsqrt(x[:, 0].pow(2).mean(-1, keepdim=True) + 1e-6))
    ref[:, 1] = wk.cpu().double() * (x[:, 1] * torch.rsqrt(x[:, 1].pow(2).mean(-1, keepdim=True) + 1e-6))
    err = (got - ref.reshape(M, N).float()).abs().max().item()
    assert err < 4 * eps * max(1.0, ref.abs().max().item()) + 1e-4, (prec, err)


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("batch,seq,heads,hd", [(2, 261, 12, 64), (2, 512, 8, 96), (1, 1536, 8, 96), (3, 70, 2, 64)])
def test_attention(hip, prec, batch, seq, heads, hd):
    qkv = _rand("attq", (batch, seq, 3, heads, hd), 1.0)
    qkv[:, :, 0] *= 1.7                                            # non-trivial softmax
    if seq >= 200:                                                 # force an online-softmax rescale late in the row
        qkv[:, 150, 1] = qkv[:, 7, 0] * 3.0
    t = hip_ops.to_operand(qkv.reshape(batch * seq, -1).cuda(), prec)
    out = hip_ops.attention(t, batch, seq, heads, hd, hd ** -0.5, prec=prec)
    got = hip_ops.from_operand(out, prec).cpu().reshape(batch, seq, heads, hd)
    src = _q(qkv.reshape(batch * seq, -1), prec).reshape(batch, seq, 3, heads, hd).double()
    q, k, v = (src[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    ref = ((q @ k.transpose(-1, -2)) * hd ** -0.5).softmax(-1) @ v
    ref = ref.permute(0, 2, 1, 3).float()
    err = (got - ref).abs().max().item()
    assert err < 6 * EPS[prec] * max(1.0, ref.abs().max().item()) + 2e-5, (prec, err)


@pytest.mark.parametrize("variant", ["f16_out_bf16x3", "bf16_out_fp8"])
@pytest.mark.parametrize("batch,seq,heads,hd", [(2, 261, 12, 64), (1, 1536, 8, 96), (1, 4352, 8, 96), (3, 70, 2, 64)])
def test_attention_mixed_output_variants(hip, variant, batch, seq, heads, hd):
    """The two attention variants the default modes really run, against fp64 torch on the operands as stored:
      BD_PREC_F16_OUT_BF16X3  (strict mode, BETR): f16 qkv in, one f16 MFMA pass, split-bf16 (hi, lo) planes out;
      BD_PREC_BF16_OUT_FP8    (fp8 mode): bf16 qkv in, e4m3 out.
    seq 4352 = BASELINE configs[3] (T = 17)."""
    from boxdreamer_amd import _lib
    qkv = _rand("attq", (batch, seq, 3, heads, hd), 1.0)
    qkv[:, :, 0] *= 1.7
    if seq >= 200:
        qkv[:, 150, 1] = qkv[:, 7, 0] * 3.0
    in_prec = "fp16" if variant == "f16_out_bf16x3" else "bf16"
    t = hip_ops.to_operand(qkv.reshape(batch * seq, -1).cuda(), in_prec)
    src = _q(qkv.reshape(batch * seq, -1), in_prec).reshape(batch, seq, 3, heads, hd).double()
    q, k, v = (src[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    ref = (((q @ k.transpose(-1, -2)) * hd ** -0.5).softmax(-1) @ v).permute(0, 2, 1, 3).float()
    lib = _lib.load()
    if variant == "f16_out_bf16x3":
        out = torch.empty((2, batch * seq, heads * hd), dtype=torch.bfloat16, device="cuda")
        _lib.check(lib.bd_attention(_lib.ptr(t), 0, _lib.ptr(out), out[0].numel(), batch, seq, heads, hd, hd ** -0.5,
                                    _lib.PREC_F16_OUT_BF16X3, _lib.stream()), "bd_attention")
        got = (out[0].float() + out[1].float()).cpu().reshape(batch, seq, heads, hd)
        tol = 6 * 2.0 ** -11 * max(1.0, ref.abs().max().item()) + 2e-5      # f16 P and V rounding; the (hi, lo) store is ~2^-16
    else:
        out = torch.empty((batch * seq, heads * hd), dtype=torch.float8_e4m3fn, device="cuda")
        _lib.check(lib.bd_attention(_lib.ptr(t), 0, _lib.ptr(out), 0, batch, seq, heads, hd, hd ** -0.5,
                                    _lib.PREC_BF16_OUT_FP8, _lib.stream()), "bd_attention")
        got = out.float().cpu().reshape(batch, seq, heads, hd)
        # e4m3 output rounding: 2^-4 relative (normal range >= 2^-6), 2^-10 absolute below it; plus the bf16 pass
        err = (got - ref).abs()
        assert (err <= ref.abs() * 2.0 ** -4 + 2.0 ** -10 + 6 * 2.0 ** -8).all(), err.max().item()
        return
    err = (got - ref).abs().max().item()
    assert err < tol, (variant, err)


@pytest.mark.parametrize("variant", ["f16_out_f16c8", "bf16x3_out_f16c8"])
@pytest.mark.parametrize("batch,seq,heads,hd", [(2, 261, 12, 64), (1, 1536, 8, 96), (2, 512, 8, 96), (3, 70, 2, 64)])
def test_attention_f16c8_output_variants(hip, variant, batch, seq, heads, hd):
    """The attention variants of the strict mode the bench reports (f16c8_qkv16), against fp64 torch on the operands as stored:
      BD_PREC_F16_OUT_F16C8     (BETR): f16 qkv (one plane) in, one f16 MFMA pass;
      BD_PREC_BF16X3_OUT_F16C8  (DINOv2): split-bf16 qkv planes in, three MFMA passes per product;
    both write the F16C8 operand class (f16 plane + k-permuted e4m3 correction plane) that the proj GEMM reads.  Sequences that are
    multiples of 256 at head_dim 96 take the pipelined kernel, whose rows leave through its LDS staging block."""
    from boxdreamer_amd import _lib
    qkv = _rand("attq", (batch, seq, 3, heads, hd), 1.0)
    qkv[:, :, 0] *= 1.7
    if seq >= 200:
        qkv[:, 150, 1] = qkv[:, 7, 0] * 3.0
    in_prec = "fp16" if variant == "f16_out_f16c8" else "bf16x3"
    t = hip_ops.to_operand(qkv.reshape(batch * seq, -1).cuda(), in_prec)
    src = _q(qkv.reshape(batch * seq, -1), in_prec).reshape(batch, seq, 3, heads, hd).double()
    q, k, v = (src[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    ref = (((q @ k.transpose(-1, -2)) * hd ** -0.5).softmax(-1) @ v).permute(0, 2, 1, 3).float()
    out = torch.zeros((2, batch * seq, heads * hd), dtype=torch.float16, device="cuda")
    pid = _lib.PREC_F16_OUT_F16C8 if variant == "f16_out_f16c8" else _lib.PREC_BF16X3_OUT_F16C8
    qkv_plane = 0 if variant == "f16_out_f16c8" else t[0].numel()
    _lib.check(_lib.load().bd_attention(_lib.ptr(t), qkv_plane, _lib.ptr(out), out[0].numel(), batch, seq, heads, hd, hd ** -0.5,
                                        pid, _lib.stream()), "bd_attention")
    hi, lo, _ = hip_ops.f16c8_decode(out)
    got = (hi + lo).cpu().reshape(batch, seq, heads, hd)
    # f16 variant: P and V carry 11 bits; split-bf16: ~16; the stored class keeps hi + lo to ~2^-15 relative
    eps = 2.0 ** -11 if variant == "f16_out_f16c8" else 2.0 ** -15
    err = (got - ref).abs().max().item()
    assert err < 6 * eps * max(1.0, ref.abs().max().item()) + 2e-5, (variant, err)
    # the f16 plane alone must be the correctly rounded f16 image of the stored value (the GEMM's first pass reads only it)
    assert (hi.cpu().reshape(batch, seq, heads, hd) - got).abs().max().item() <= 2.0 ** -11 * max(1.0, got.abs().max().item())


_PREFIX_VARIANTS = {   # name -> (input operand class, attention code, how to read the result)
    "bf16": ("bf16", "PREC_BF16", "native"), "fp16": ("fp16", "PREC_F16", "native"), "bf16x3": ("bf16x3", "PREC_BF16X3", "planes"),
    "f16_out_bf16x3": ("fp16", "PREC_F16_OUT_BF16X3", "planes_bf16"), "bf16_out_fp8": ("bf16", "PREC_BF16_OUT_FP8", "fp8"),
    "f16_out_f16c8": ("fp16", "PREC_F16_OUT_F16C8", "f16c8"), "bf16x3_out_f16c8": ("bf16x3", "PREC_BF16X3_OUT_F16C8", "f16c8"),
    "f16_out_f16x3": ("fp16", "PREC_F16_OUT_F16X3", "planes_f16"), "bf16x3_out_f16x3": ("bf16x3", "PREC_BF16X3_OUT_F16X3", "planes_f16")}


@pytest.mark.parametrize("variant", sorted(_PREFIX_VARIANTS))
@pytest.mark.parametrize("batch,seq,heads,npre", [(3, 261, 12, 5), (2, 70, 2, 1), (1, 320, 3, 8), (1, 517, 2, 5)])
def test_attention_prefix_split(hip, variant, batch, seq, heads, npre):
    """bd_attention_prefix (round 4): with prefix_queries = 0 only the patch queries run (exact tiles; the last DINOv2 block), their rows
    bit-identical to bd_attention's and the prefix rows of `out` untouched; with prefix_queries = 1 it is bd_attention.  Every attention
    code the whole-path entry points use, against fp64 torch on the operands as stored."""
    from boxdreamer_amd import _lib
    hd = 64
    in_prec, code, kind = _PREFIX_VARIANTS[variant]
    pid = getattr(_lib, code)
    qkv = _rand("attp", (batch, seq, 3, heads, hd), 1.0)
    qkv[:, :, 0] *= 1.7
    qkv[:, min(150, seq - 1), 1] = qkv[:, 3, 0] * 3.0
    qkv[:, 0, 0] *= 2.5                                             # a peaked prefix query
    t = hip_ops.to_operand(qkv.reshape(batch * seq, -1).cuda(), in_prec)
    src = _q(qkv.reshape(batch * seq, -1), in_prec).reshape(batch, seq, 3, heads, hd).double()
    q, k, v = (src[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    ref = (((q @ k.transpose(-1, -2)) * hd ** -0.5).softmax(-1) @ v).permute(0, 2, 1, 3).float()
    lib = _lib.load()
    qkv_plane = t[0].numel() if in_prec == "bf16x3" else 0

    def alloc():
        if kind == "native":
            return torch.full((batch * seq, heads * hd), 7.0, dtype=_lib.op_dtype(in_prec), device="cuda")
        if kind == "fp8":
            return torch.full((batch * seq, heads * hd), 7.0, device="cuda").to(torch.float8_e4m3fn)
        dt = torch.bfloat16 if kind in ("planes", "planes_bf16") else torch.float16
        return torch.full((2, batch * seq, heads * hd), 7.0, dtype=dt, device="cuda")

    def read(o):
        if kind in ("native", "fp8"):
            return o.float().cpu().reshape(batch, seq, heads, hd)
        if kind == "f16c8":
            hi, lo, _ = hip_ops.f16c8_decode(o)
            return (hi + lo).cpu().reshape(batch, seq, heads, hd)
        return (o[0].float() + o[1].float()).cpu().reshape(batch, seq, heads, hd)

    plane = lambda o: 0 if kind in ("native", "fp8") else o[0].numel()
    one = alloc()
    _lib.check(lib.bd_attention(_lib.ptr(t), qkv_plane, _lib.ptr(one), plane(one), batch, seq, heads, hd, hd ** -0.5, pid, _lib.stream()),
               "bd_attention")
    two = alloc()
    _lib.check(lib.bd_attention_prefix(_lib.ptr(t), qkv_plane, _lib.ptr(two), plane(two), batch, seq, heads, hd, hd ** -0.5, npre, 1, pid,
                                       _lib.stream()), "bd_attention_prefix")
    skip = alloc()
    keep = skip.clone()
    _lib.check(lib.bd_attention_prefix(_lib.ptr(t), qkv_plane, _lib.ptr(skip), plane(skip), batch, seq, heads, hd, hd ** -0.5, npre, 0, pid,
                                       _lib.stream()), "bd_attention_prefix")
    torch.cuda.synchronize()
    g1, g2, g3 = read(one), read(two), read(skip)
    in_eps = {"bf16": 2.0 ** -8, "fp16": 2.0 ** -11, "bf16x3": 2.0 ** -15}[in_prec]
    scale = max(1.0, ref.abs().max().item())
    if kind == "fp8":
        tol = ref.abs() * 2.0 ** -4 + 2.0 ** -10 + 6 * 2.0 ** -8
        assert ((g2 - ref).abs() <= tol).all()
    else:
        out_eps = {"native": in_eps, "planes": 2.0 ** -15, "planes_bf16": 2.0 ** -15, "planes_f16": 2.0 ** -20, "f16c8": 2.0 ** -15}[kind]
        err = (g2 - ref).abs().max().item()
        assert err < 6 * max(in_eps, out_eps) * scale + 2e-5, (variant, err)
    assert torch.equal(g2[:, :npre], g1[:, :npre])                  # prefix_queries = 1 IS bd_attention
    # patch rows: same kernel, same tiles' arithmetic as the one-launch form (query blocks are cut differently: not bit-identical by
    # construction only where a block boundary changes which lanes share an online-softmax rescale -- it does not: per-row arithmetic)
    assert torch.equal(g2[:, npre:], g1[:, npre:])
    assert torch.equal(g3[:, npre:], g2[:, npre:])
    # skipped prefix queries: every byte of those rows is as the caller left it (F16C8's lo8 plane is packed row-wise into the plane's
    # first half: compare the hi plane rows and the packed lo8 rows)
    if kind in ("native", "fp8"):
        rows = lambda o: o.view(torch.uint8).reshape(batch, seq, -1)[:, :npre]
        assert torch.equal(rows(skip), rows(keep))
    elif kind == "f16c8":
        assert torch.equal(skip[0].reshape(batch, seq, -1)[:, :npre], keep[0].reshape(batch, seq, -1)[:, :npre])
        lo8 = lambda o: o[1].view(torch.uint8).reshape(-1)[: batch * seq * heads * hd].reshape(batch, seq, -1)[:, :npre]
        assert torch.equal(lo8(skip), lo8(keep))
    else:
        assert torch.equal(skip.reshape(2, batch, seq, -1)[:, :, :npre], keep.reshape(2, batch, seq, -1)[:, :, :npre])


@pytest.mark.parametrize("prec", PRECS_LIN)
def test_im2col_and_patchify(hip, prec):
    data = synth.make_batch(seed=21, B=1, T=2)
    img = data["images"][0]                                          # (2,3,224,224)
    for dt in (torch.float32, torch.bfloat16):
        a = hip_ops.im2col_images(img.to(dt).cuda(), prec=prec)
        mean = torch.tensor(orc._IMAGENET_MEAN).view(1, 3, 1, 1)
        std = torch.tensor(orc._IMAGENET_STD).view(1, 3, 1, 1)
        xn = (img.to(dt).float() - mean) / std
        cols = F.unfold(xn, 14, stride=14).transpose(1, 2).reshape(2 * 256, 588)
        got = hip_ops.from_operand(a, prec).cpu()
        assert torch.equal(got[:, 588:], torch.zeros(512, 52))
        assert (got[:, :588] - cols).abs().max().item() < 8 * EPS[prec] * 3 + 1e-6
    heat = data["bbox_feat"][0]                                      # (2,8,224,224)
    a = hip_ops.patchify_heatmaps(heat.cuda(), prec=prec)
    got = hip_ops.from_operand(a, prec).cpu()
    ref = orc.patchify(heat, 14, 8).reshape(512, 1568)
    assert torch.equal(got[:, 1568:], torch.zeros(512, 32))
    assert (got[:, :1568] - ref).abs().max().item() < 2 * EPS[prec] + 1e-7


def test_unpatchify_sigmoid(hip):
    proj = _rand("proj", (2 * 256, 1568), 2.0)
    logits, heat = hip_ops.unpatchify_sigmoid(proj.cuda(), 2)
    ref = orc.unpatchify(proj.reshape(2, 256, 1568), 14, 8)
    assert torch.equal(logits.cpu(), ref)
    assert (heat.cpu() - (2 * torch.sigmoid(ref) - 1)).abs().max().item() < 5e-7


def test_decode_topk(hip, golden_dir):
    # adversarial plateaus (the fixture's construction, oracle/make_golden.py) + the reference's answers
    u = torch.from_numpy(synth.uniform_np("unit.decode", (2, 8, 224 * 224), -1.0, 0.5, 5).astype(np.float32))
    hm = u.reshape(2, 8, 224, 224).clone()
    for b in range(2):
        for c in range(8):
            ys, n = 10 + 13 * c + b, (20 if c < 4 else 7)
            hm[b, c, ys, 30:30 + n] = 1.0
            if c >= 4:
                hm[b, c, ys + 1, 100:113] = torch.linspace(0.9, 0.6, 13)
    kp, kn, idx = hip_ops.decode_topk(hm.cuda())
    g = np.load(f"{golden_dir}/unit_vectors.npz")
    assert np.array_equal(kp.cpu().numpy(), g["decode_kp"])
    assert np.abs(kn.cpu().numpy() - g["decode_norm_kp"]).max() < 1e-6
    on, okp, oidx = orc.recover_bb8_corners(hm)
    assert torch.equal(idx.cpu().long(), oidx)                       # same order, same tie rule
    # random smooth-ish maps + exact ties across the rank-20 boundary (lower index must win)
    hm2 = torch.from_numpy(synth.uniform_np("dec2", (4, 8, 224, 224), -1, 1, 9).astype(np.float32))
    hm2[0, 0].fill_(0.25)                                            # all equal -> indices 0..19
    hm2[1, 3, 100, 50:90] = 0.999                                    # 40-way tie over the boundary
    kp, kn, idx = hip_ops.decode_topk(hm2.cuda())
    _, okp, oidx = orc.recover_bb8_corners(hm2)
    assert torch.equal(idx.cpu().long(), oidx)
    assert torch.equal(kp.cpu(), okp)
    assert idx[0, 0].cpu().tolist() == list(range(20))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_render_corner_heatmaps(hip, golden_dir, dtype):
    """bd_render_corner_heatmaps vs the oracle restatement of make_bbox_features and the reference's own output."""
    from boxdreamer_amd.bbox_features import make_bbox_features
    corners = torch.from_numpy(synth.uniform_np("unit.corners", (3, 8, 2), 20.0, 204.0, 17).astype(np.float32))
    corners[1, 2] = torch.tensor([100.0, 57.0])
    corners[2, 5] = torch.tensor([-3.25, 230.5])
    got = make_bbox_features(corners.cuda(), type="heatmap", shape=(224, 224), dtype=dtype).float().cpu()
    ref = orc.make_bbox_features(corners, (224, 224))
    tol = 4e-6 if dtype == torch.float32 else 2.0 ** -8
    assert (got - ref).abs().max().item() <= tol
    if dtype == torch.float32:
        g = np.load(f"{golden_dir}/unit_vectors.npz")
        assert np.abs(got.reshape(-1)[::11].numpy() - g["bbox_features_strided"]).max() <= 4e-6
        assert got.max().item() == 1.0
    # several samples in one launch: the normalisation max is per group of views, as in per-sample dataset calls
    two = torch.cat([corners, corners.flip(0) * 0.9 + 5.0])
    got2 = make_bbox_features(two.cuda(), shape=(224, 224), group=3).cpu()
    ref2 = torch.cat([orc.make_bbox_features(two[:3], (224, 224)), orc.make_bbox_features(two[3:], (224, 224))])
    assert (got2 - ref2).abs().max().item() <= 4e-6


@pytest.mark.parametrize("prec", PRECS)
def test_attention_query_range(hip, prec):
    """bd_attention_q (last decoder block): queries from one 256-row view per sample, keys/values from all rows."""
    batch, T, P, heads, hd = 3, 4, 256, 8, 96
    seq = T * P
    qkv = _rand("attqr", (batch, seq, 3, heads, hd), 1.0)
    qv = torch.tensor([3, 0, 2], dtype=torch.int32)
    t = hip_ops.to_operand(qkv.reshape(batch * seq, -1).cuda(), prec)
    out = hip_ops.attention_q(t, batch, seq, heads, hd, hd ** -0.5, qv.cuda(), P, prec=prec)
    got = hip_ops.from_operand(out, prec).cpu().reshape(batch, P, heads, hd)
    full = hip_ops.from_operand(hip_ops.attention(t, batch, seq, heads, hd, hd ** -0.5, prec=prec), prec).cpu()
    full = full.reshape(batch, seq, heads, hd)
    for b in range(batch):
        assert torch.equal(got[b], full[b, qv[b] * P:(qv[b] + 1) * P])       # bit-identical to the full-width kernel


def test_attention_refuses_a_sample_beyond_the_descriptor_range(hip):
    """The K / V tiles are fetched through 32-bit buffer descriptors spanning ONE sample's rows: a sample of 2 GiB or more must be refused
    (BD_ERR_SHAPE) before anything is launched -- the arguments are validated first, so a small dummy buffer is enough here."""
    from boxdreamer_amd import _lib
    lib = _lib.load()
    t = torch.zeros(4096, dtype=torch.bfloat16, device="cuda")
    heads, hd = 8, 96
    seq = (1 << 31) // (3 * heads * hd * 2) + 1
    with pytest.raises(_lib.HipLibraryError, match="BD_ERR_SHAPE"):
        _lib.check(lib.bd_attention(_lib.ptr(t), 0, _lib.ptr(t), 0, 1, seq, heads, hd, hd ** -0.5, _lib.PREC_BF16, _lib.stream()), "bd_attention")
    with pytest.raises(_lib.HipLibraryError, match="BD_ERR_SHAPE"):
        _lib.check(lib.bd_attention_prefix(_lib.ptr(t), 0, _lib.ptr(t), 0, 1, seq, heads, hd, hd ** -0.5, 5, 1, _lib.PREC_BF16, _lib.stream()),
                   "bd_attention_prefix")


# ----------------------------------------------------------------------------- fp8 (e4m3) mode, BASELINE configs[4]

@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (1500, 1536, 256), (300, 200, 384), (2048, 768, 3072)])
def test_gemm_fp8(hip, M, N, K):
    """v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales: exact on small integers (also a layout / transpose
    detector), and equal to the fp32 product of the DEQUANTISED operands on random data (per-channel weight scales)."""
    ai = (torch.arange(M * K, dtype=torch.float32).reshape(M, K) * 7).remainder(9) - 4
    wi = (torch.arange(N * K, dtype=torch.float32).reshape(N, K) * 5).remainder(7) - 3
    out = hip_ops.gemm(hip_ops.to_operand(ai.cuda(), "fp8"), hip_ops.to_operand(wi.cuda(), "fp8"), None, prec="fp8",
                       out_f32=True)
    assert torch.equal(out.cpu(), ai @ wi.t())
    from boxdreamer_amd import pack
    a, w, b = _rand("a8", (M, K), 1.5), _rand("w8", (N, K), 0.03), _rand("b8", (N,), 0.1)
    w8, sc = pack.pack_linear_weight(w, "fp8", return_scale=True)
    a8 = hip_ops.to_operand(a.cuda(), "fp8")
    out = hip_ops.gemm(a8, w8.cuda(), b.cuda(), prec="fp8", out_f32=True, wscale=sc.cuda())
    ref = (a8.float().cpu().double() @ w8.float().double().t()) * sc.double() + b.double()
    assert (out.cpu() - ref.float()).abs().max().item() < 2e-4 * K ** 0.5
    # and the quantisation itself is what e4m3 promises (relative step 2^-3 -> <= 2^-4 rounding error)
    full = a.double() @ w.double().t() + b.double()
    rel = ((out.cpu().double() - full).abs().mean() / full.abs().mean()).item()
    assert rel < 0.08, rel
    # bf16 single-plane output (qkv for the bf16 attention) and e4m3 output with GELU (fc1 -> fc2)
    o_bf = hip_ops.gemm(a8, w8.cuda(), b.cuda(), prec="fp8", wscale=sc.cuda(), out_mode=3)
    assert o_bf.dtype == torch.bfloat16 and (o_bf.float().cpu() - ref.float()).abs().max().item() < 2.0 ** -8 * ref.abs().max().item() + 1e-3
    if N % 8 == 0:
        o8 = hip_ops.gemm(a8, w8.cuda(), b.cuda(), prec="fp8", wscale=sc.cuda(), act=1)
        g = F.gelu(ref.float())
        assert o8.dtype == torch.float8_e4m3fn
        assert (o8.float().cpu() - g).abs().max().item() <= 2.0 ** -4 * g.abs().max().item() + 2.0 ** -9
        # kind 3 WITH GELU: an e4m3 fc1 feeding a PROMOTED (bf16) fc2 -- the hand-off of the mixed e4m3 policy
        og = hip_ops.gemm(a8, w8.cuda(), b.cuda(), prec="fp8", wscale=sc.cuda(), act=1, out_mode=3)
        assert og.dtype == torch.bfloat16 and (og.float().cpu() - g).abs().max().item() <= 2.0 ** -8 * g.abs().max().item() + 1e-3


def test_layernorm_and_layout_fp8(hip):
    x = _rand("lnx8", (777, 768), 2.0) + 0.3
    g, b = _rand("lng8", (768,), 0.1) + 1, _rand("lnb8", (768,), 0.1)
    o8, o32 = hip_ops.layernorm(x.cuda(), g.cuda(), b.cuda(), 1e-5, prec="fp8", want32=True)
    ref = F.layer_norm(x, (768,), g, b, 1e-5)
    assert o8.dtype == torch.float8_e4m3fn
    assert (o32.cpu() - ref).abs().max().item() < 2e-5
    err = (o8.float().cpu() - ref).abs()
    assert (err <= ref.abs() * 2.0 ** -4 + 2.0 ** -10).all()           # RNE to 3 mantissa bits (subnormal floor 2^-9)
    data = synth.make_batch(seed=21, B=1, T=2)
    a = hip_ops.im2col_images(data["images"][0].cuda(), kpad=640, prec="fp8")
    mean = torch.tensor(orc._IMAGENET_MEAN).view(1, 3, 1, 1); std = torch.tensor(orc._IMAGENET_STD).view(1, 3, 1, 1)
    cols = F.unfold((data["images"][0] - mean) / std, 14, stride=14).transpose(1, 2).reshape(512, 588)
    got = a.float().cpu()
    assert torch.equal(got[:, 588:], torch.zeros(512, 52)) and ((got[:, :588] - cols).abs() <= cols.abs() * 2.0 ** -4 + 2.0 ** -10).all()
    hmap = hip_ops.patchify_heatmaps(data["bbox_feat"][0].cuda(), kpad=1664, prec="fp8").float().cpu()
    refp = orc.patchify(data["bbox_feat"][0], 14, 8).reshape(512, 1568)
    assert torch.equal(hmap[:, 1568:], torch.zeros(512, 96)) and ((hmap[:, :1568] - refp).abs() <= refp.abs() * 2.0 ** -4 + 2.0 ** -10).all()


# ----------------------------------------------------------------------------- GPU PnP

@pytest.mark.parametrize("npts", [8, 24])
def test_gpu_pnp_matches_host_form(hip, npts):
    """bd_solve_pnp (one pose per thread, fp64) follows boxdreamer_amd/pnp.py step by step: same poses as the batched numpy
    form on noisy corners, the true pose on exact corners, zeros for a sample with a NaN corner."""
    from boxdreamer_amd import pnp
    from boxdreamer_amd.box_utils import solve_poses_device
    rng = np.random.default_rng(3)
    N = 40
    box = np.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], float) * [0.1, 0.07, 0.05]
    p3 = np.tile(np.tile(box, (npts // 8, 1)), (N, 1, 1))
    K = np.tile(np.array([[600.0, 0, 112], [0, 600, 112], [0, 0, 1]]), (N, 1, 1))
    Rt = pnp._rodrigues_b(rng.normal(size=(N, 3)) * 0.9)
    tt = np.stack([rng.normal(size=N) * 0.05, rng.normal(size=N) * 0.05, 0.6 + rng.random(N) * 0.4], 1)
    pc = p3 @ np.swapaxes(Rt, 1, 2) + tt[:, None]
    exact = pc[..., :2] / pc[..., 2:] * 600 + 112
    noisy = exact + rng.normal(size=exact.shape) * 0.7
    noisy[7, 3, 0] = np.nan
    f32 = lambda a: torch.from_numpy(a.astype(np.float32)).cuda()
    got = solve_poses_device(f32(noisy), f32(p3), f32(K)).cpu().numpy()
    ok, Rb, tb = pnp.solve_pnp_batched(p3.astype(np.float32), noisy.astype(np.float32), K.astype(np.float32))
    assert not ok[7] and (got[7] == 0).all()
    for i in range(N):
        if i == 7:
            continue
        assert np.abs(got[i, :3, :3] - Rb[i]).max() < 1e-4 and np.abs(got[i, :3, 3] - tb[i]).max() < 1e-4, i
        assert got[i, 3, 3] == 1.0
    got = solve_poses_device(f32(exact), f32(p3), f32(K)).cpu().numpy()
    assert np.abs(got[:, :3, :3] - Rt).max() < 1e-4 and np.abs(got[:, :3, 3] - tt).max() < 1e-4


@pytest.mark.parametrize("prec", ["bf16", "fp16", "bf16x3"])
@pytest.mark.parametrize("batch,seq,heads,hd", [(2, 261, 12, 64), (3, 70, 2, 64), (2, 13, 2, 64), (2, 70, 8, 96), (1, 300, 8, 96)])
def test_attention_ragged_tail_never_reads_past_the_sample(hip, prec, batch, seq, heads, hd):
    """the K / V tiles of a ragged last key tile are fetched through a buffer descriptor whose range ends with the sample; the
    rows past `seq` must come back as ZERO (hardware bounds check), never as whatever lies behind the sample -- P = 0 times NaN / Inf is NaN.
    The qkv planes are carved out of an arena whose guard rows behind every plane are NaN (and, between samples, the next sample's rows are
    real data): outputs must be finite and equal, bit for bit, to the run on an arena whose guard rows are zero."""
    from boxdreamer_amd import _lib
    lib = _lib.load()
    np_ = 2 if prec == "bf16x3" else 1
    rows, cols, guard = batch * seq, 3 * heads * hd, 96
    qkv = _rand("attq_guard", (rows, cols), 1.0)
    t = hip_ops.to_operand(qkv.cuda(), prec)
    planes = [t[0], t[1]] if np_ == 2 else [t]

    def run(fill):
        arena = torch.full((np_, rows + guard, cols), fill, dtype=planes[0].dtype, device="cuda")
        for i, pl in enumerate(planes):
            arena[i, :rows] = pl
        out = torch.full((np_, rows + guard, heads * hd), 0.0, dtype=planes[0].dtype, device="cuda")
        _lib.check(lib.bd_attention(arena.data_ptr(), (rows + guard) * cols if np_ == 2 else 0, out.data_ptr(),
                                    (rows + guard) * heads * hd if np_ == 2 else 0, batch, seq, heads, hd, hd ** -0.5, _lib.prec_id(prec),
                                    _lib.stream()), "bd_attention")
        torch.cuda.synchronize()
        return out[:, :rows].clone()
    clean, poisoned = run(0.0), run(float("nan"))
    assert torch.isfinite(poisoned.float()).all(), "a key / value row past the sample leaked into the result"
    assert torch.equal(clean.view(torch.int16), poisoned.view(torch.int16))
