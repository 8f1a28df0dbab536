"""GPU: the LayerNorm fold (ABI 8, include/boxdreamer_hip.h bd_gemm_args.ln_*) at the operator level.

    LN(x) W^T + b  =  rstd (x (g . W)^T - mean s) + (beta W^T + b)            (blocks.py:35-41, 876-886; DINOv2 layers/block.py:89-114)

Producer side: an fp32-residual Linear also emits the rows' F16C8 operand copy and per 96-column wave tile the pair (mean, M2).
Consumer side: a Linear whose A operand is such a raw copy applies the row statistics in its epilogue.  Checked against fp64 torch
arithmetic on the same operands, against the un-folded operators (bd_layernorm + bd_gemm), and for form independence (a row's bits do
not depend on the tile form that computed it)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from boxdreamer_amd import hip_ops, synth

pytestmark = pytest.mark.gpu


def _rand(name, shape, std=1.0, seed=3):
    return torch.from_numpy(synth.bell_np(name, shape, std, 0.0, seed).astype(np.float32))


def _lo8(t, rows, cols):
    return t[1].contiguous().view(torch.uint8).reshape(-1)[: rows * cols]


def _emit(M, K, with_fold=True, resid=True, seed=3):
    N = 768
    a, w, b = _rand("a", (M, K), seed=seed).cuda(), _rand("w", (N, K), 0.04, seed).cuda(), _rand("b", (N,), 0.3, seed).cuda()
    res = (_rand("r", (M, N), 2.0, seed) + 0.7).cuda()          # a row mean that is not zero
    e = hip_ops.f16c8_qexp(w)
    kw = dict(prec="f16c8", w_qexp=e, out_f32=True)
    if resid:
        kw["resid"] = res.clone()
    st = op = None
    if with_fold:
        st = torch.full((M, N // 96, 2), float("nan"), dtype=torch.float32, device="cuda")
        op = torch.zeros((2, M, N), dtype=torch.float16, device="cuda")
        kw["ln_emit"] = (st, op)
    out = hip_ops.gemm(hip_ops.f16c8_encode(a, 0, False), hip_ops.f16c8_encode(w, e, True), b, **kw)
    return out, st, op


@pytest.mark.parametrize("M,K", [(49152, 768), (1536, 768), (1000, 3072), (1536, 3072), (300, 768), (50112, 3072)])
def test_producer_emits_operand_copy_and_row_statistics(hip, M, K):
    """proj (K = 768) / fc2 (K = 3072) of both stacks, large / small / 128 x 96 forms, ragged last tiles."""
    plain, _, _ = _emit(M, K, with_fold=False)
    out, st, op = _emit(M, K)
    assert torch.equal(out, plain), "the fp32 rows must not change"
    ref = hip_ops.f16c8_encode(out, 0, False)                     # the reference packer on the fp32 rows the launch wrote
    assert torch.equal(op[0], ref[0]), "f16 plane of the operand copy"
    assert torch.equal(_lo8(op, M, 768), _lo8(ref, M, 768)), "lo8 plane of the operand copy"
    x = out.double().reshape(M, 8, 96)
    mean = x.mean(-1)
    m2 = ((x - mean[..., None]) ** 2).sum(-1)
    assert torch.isfinite(st).all()
    assert (st[..., 0].double() - mean).abs().max().item() <= 2e-6 * float(x.abs().max())
    assert ((st[..., 1].double() - m2).abs() / m2.clamp_min(1e-3)).max().item() <= 2e-5


@pytest.mark.parametrize("K", [768, 3072])
def test_producer_rows_do_not_depend_on_the_launch_form(hip, K):
    big = _emit(49152, K)
    for rows in (1536, 300, 4096):
        a = _rand("a", (49152, K)).cuda()[:rows]
        w, b = _rand("w", (768, K), 0.04).cuda(), _rand("b", (768,), 0.3).cuda()
        res = (_rand("r", (49152, 768), 2.0) + 0.7).cuda()[:rows]
        e = hip_ops.f16c8_qexp(w)
        st = torch.zeros((rows, 8, 2), dtype=torch.float32, device="cuda")
        op = torch.zeros((2, rows, 768), dtype=torch.float16, device="cuda")
        out = hip_ops.gemm(hip_ops.f16c8_encode(a, 0, False), hip_ops.f16c8_encode(w, e, True), b, prec="f16c8", w_qexp=e, out_f32=True,
                           resid=res.clone(), ln_emit=(st, op))
        assert torch.equal(out, big[0][:rows]) and torch.equal(st, big[1][:rows]), (K, rows)
        assert torch.equal(op[0], big[2][0][:rows]) and torch.equal(_lo8(op, rows, 768), _lo8(big[2], rows, 768)), (K, rows)


@pytest.mark.parametrize("M,K", [(49152, 768), (1536, 768), (1000, 3072), (300, 768), (50112, 3072)])
def test_producer_with_the_residual_in_the_operand_copy(hip, M, K):
    """bd_gemm_args.ln_resid_in_op (the 3-byte residual stream): the residual rows are the F16C8 copy the previous residual Linear left, the
    sum goes back in place (+ row statistics); fp32 rows only on request.  Against fp64 on the decoded planes + the decoded residual."""
    N = 768
    a, w, b = _rand("a", (M, K), seed=7).cuda(), _rand("w", (N, K), 0.04, 7).cuda(), _rand("b", (N,), 0.3, 7).cuda()
    x0 = (_rand("r", (M, N), 2.0, 7) + 0.7).cuda()
    e = hip_ops.f16c8_qexp(w)
    a16, w16 = hip_ops.f16c8_encode(a, 0, False), hip_ops.f16c8_encode(w, e, True)
    outs = []
    for f32 in (False, True):
        op = hip_ops.f16c8_encode(x0, 0, False)                          # the stream as the previous residual Linear left it
        st = torch.full((M, 8, 2), float("nan"), dtype=torch.float32, device="cuda")
        dummy = torch.zeros((2, M, N), dtype=torch.float16, device="cuda") if not f32 else None
        out = hip_ops.gemm(a16, w16, b, prec="f16c8", w_qexp=e, out_f32=f32, out=dummy, ln_emit=(st, op), ln_resid_in_op=True)
        outs.append((out, st, op))
    (_, st0, op0), (o32, st1, op1) = outs
    assert torch.equal(op0[0], op1[0]) and torch.equal(_lo8(op0, M, N), _lo8(op1, M, N)) and torch.equal(st0, st1)
    ref = hip_ops.f16c8_encode(o32, 0, False)                            # the copy IS the reference packing of the fp32 rows
    assert torch.equal(op1[0], ref[0]) and torch.equal(_lo8(op1, M, N), _lo8(ref, M, N))
    rows = torch.arange(0, M, max(1, M // 129), device="cuda")
    ah, al, aq = (t.double()[rows] for t in hip_ops.f16c8_decode(a16))
    wh, wl, wq = (t.double() for t in hip_ops.f16c8_decode(w16, e, True))
    want = ah @ wh.t() + al @ wq.t() + aq @ wl.t() + b.double() + hip_ops.from_operand(hip_ops.f16c8_encode(x0, 0, False), "f16c8")[rows].double()
    assert (o32[rows].double() - want).abs().max().item() <= 2e-5 * K ** 0.5 + 1e-5
    xd = o32.double().reshape(M, 8, 96)
    mean = xd.mean(-1)
    m2 = ((xd - mean[..., None]) ** 2).sum(-1)
    assert (st1[..., 0].double() - mean).abs().max().item() <= 2e-6 * float(xd.abs().max())
    assert ((st1[..., 1].double() - m2).abs() / m2.clamp_min(1e-3)).max().item() <= 2e-5
    # form independence: the head rows of the large launch launched alone (small forms)
    if M == 49152:
        for rws in (1536, 300):
            op = hip_ops.f16c8_encode(x0[:rws], 0, False)
            st = torch.zeros((rws, 8, 2), dtype=torch.float32, device="cuda")
            hip_ops.gemm(hip_ops.f16c8_encode(a[:rws], 0, False), w16, b, prec="f16c8", w_qexp=e, out_f32=False,
                         out=torch.zeros((2, rws, N), dtype=torch.float16, device="cuda"), ln_emit=(st, op), ln_resid_in_op=True)
            assert torch.equal(op[0], op0[0][:rws]) and torch.equal(_lo8(op, rws, N), _lo8(op0, rws, N)) and torch.equal(st, st0[:rws]), rws


def _stats_of(x):
    """(mean, M2) per 96-column group of fp32 rows, as the producer writes them (fp64 math, rounded)."""
    M = x.shape[0]
    xd = x.double().reshape(M, 8, 96)
    mean = xd.mean(-1)
    m2 = ((xd - mean[..., None]) ** 2).sum(-1)
    return torch.stack([mean, m2], -1).float().contiguous()


def _consumer_case(M, N, seed=5):
    K = 768
    x = (_rand("x", (M, K), 1.5, seed) + 0.4).cuda()
    g, beta = (_rand("g", (K,), 0.2, seed) + 1.0).cuda(), _rand("beta", (K,), 0.2, seed).cuda()
    w, b = _rand("w", (N, K), 0.04, seed).cuda(), _rand("b", (N,), 0.3, seed).cuda()
    wf = (w.double() * g.double()[None, :]).float()
    bf = (b.double() + w.double() @ beta.double()).float()
    return x, g, beta, w, b, wf, bf


def _ref_rows(x, eps):
    xd = x.double()
    mean = xd.mean(-1, keepdim=True)
    rstd = torch.rsqrt(xd.var(-1, unbiased=False, keepdim=True) + eps)
    return mean, rstd


@pytest.mark.parametrize("kind", ["gelu", "f16", "split_bf16", "rms_f16"])
@pytest.mark.parametrize("M", [1536, 7000, 49152])
def test_consumer_f16c8_applies_row_statistics(hip, kind, M):
    """fc1 + GELU (operand out), BETR's v columns (f16 plane), DINOv2's QKV (split-bf16 planes), a whole BETR QKV with the fused q/k RMSNorm:
    against fp64 arithmetic on the three planes the kernel multiplies (f16c8_decode) with the statistics it is given."""
    N = {"gelu": 3072, "f16": 768, "split_bf16": 2304, "rms_f16": 2304}[kind]
    eps = 1e-5
    x, g, beta, w, b, wf, bf = _consumer_case(M, N)
    e = hip_ops.f16c8_qexp(wf)
    a16, w16 = hip_ops.f16c8_encode(x, 0, False), hip_ops.f16c8_encode(wf, e, True)
    wh, wl, wq = (t.double() for t in hip_ops.f16c8_decode(w16, e, True))
    s = (wh + wl).sum(1).float().contiguous()
    st = _stats_of(x)
    kw = dict(prec="f16c8", w_qexp=e, ln_apply=(st, s, eps))
    rms = ((_rand("wq", (96,), 0.1) + 1).cuda(), (_rand("wk", (96,), 0.1) + 1).cuda(), 1e-6)
    if kind == "gelu":
        kw.update(act=1)
    elif kind == "f16":
        kw.update(out_mode=2)
    elif kind == "split_bf16":
        kw.update(out_mode=4)
    else:
        kw.update(out_mode=2, rms=rms)
    out = hip_ops.gemm(a16, w16, bf, **kw)
    # reference on a row sample (fp64 on the decoded planes)
    rows = torch.arange(0, M, max(1, M // 257), device="cuda")
    ah, al, aq = (t.double()[rows] for t in hip_ops.f16c8_decode(a16))
    acc = ah @ wh.t() + al @ wq.t() + aq @ wl.t()
    mean, rstd = _ref_rows(x[rows], eps)
    y = rstd * (acc - mean * s.double()[None, :]) + bf.double()[None, :]
    if kind == "gelu":
        y = F.gelu(y)
        got = hip_ops.from_operand(out, "f16c8")[rows].double()
    elif kind == "split_bf16":
        got = (out[0].float() + out[1].float())[rows].double()
    else:
        if kind == "rms_f16":
            yq = y[:, :1536].reshape(-1, 2, 8, 96)
            wn = torch.stack([rms[0], rms[1]]).double()[None, :, None, :]
            y = torch.cat([(wn * yq * torch.rsqrt((yq ** 2).mean(-1, keepdim=True) + 1e-6)).reshape(-1, 1536), y[:, 1536:]], 1)
        got = out[rows].double()
    err = (got - y).abs().max().item()
    scale = float(y.abs().max())
    tol = {"gelu": 2.0 ** -14, "split_bf16": 2.0 ** -14, "f16": 2.0 ** -10, "rms_f16": 2.0 ** -10}[kind] * scale + 2e-4
    assert err <= tol, (kind, M, err, tol)
    # ... and against what the fold replaces: LayerNorm (fp32) -> the un-folded Linear in fp64
    lnx = F.layer_norm(x[rows].double(), (768,), g.double(), beta.double(), eps)
    z = lnx @ w.double().t() + b.double()
    if kind == "gelu":
        z = F.gelu(z)
    if kind != "rms_f16":
        assert (got - z).abs().max().item() <= 3e-3 * float(z.abs().max()), (kind, M)


def test_consumer_rows_do_not_depend_on_the_launch_form(hip):
    x, g, beta, w, b, wf, bf = _consumer_case(49152, 3072)
    e = hip_ops.f16c8_qexp(wf)
    w16 = hip_ops.f16c8_encode(wf, e, True)
    s = wf.double().sum(1).float().contiguous()
    st = _stats_of(x)

    def go(rows):
        return hip_ops.gemm(hip_ops.f16c8_encode(x[:rows], 0, False), w16, bf, prec="f16c8", w_qexp=e, act=1, ln_apply=(st[:rows].contiguous(), s, 1e-6))
    big = go(49152)
    for rows in (1536, 300):
        small = go(rows)
        assert torch.equal(big[0][:rows], small[0]) and torch.equal(_lo8(big, rows, 3072), _lo8(small, rows, 3072)), rows


@pytest.mark.parametrize("M", [1536, 9216, 49152])
def test_consumer_f16_qk_launch(hip, M):
    """BETR's q, k columns in the default mode: ONE f16 pass on the f16 plane of the raw operand copy, fused q/k RMSNorm, f16 result."""
    N, eps = 1536, 1e-5
    x, g, beta, w, b, wf, bf = _consumer_case(M, N, seed=9)
    a16 = hip_ops.f16c8_encode(x, 0, False)
    w16 = wf.half().contiguous()
    s = w16.double().sum(1).float().contiguous()
    st = _stats_of(x)
    rms = ((_rand("wq", (96,), 0.1) + 1).cuda(), (_rand("wk", (96,), 0.1) + 1).cuda(), 1e-6, 2)
    out = hip_ops.gemm(a16[0], w16, bf, prec="fp16", rms=rms, ln_apply=(st, s, eps))
    rows = torch.arange(0, M, max(1, M // 257), device="cuda")
    acc = a16[0][rows].double() @ w16.double().t()
    mean, rstd = _ref_rows(x[rows], eps)
    y = (rstd * (acc - mean * s.double()[None, :]) + bf.double()[None, :]).reshape(-1, 2, 8, 96)
    wn = torch.stack([rms[0], rms[1]]).double()[None, :, None, :]
    y = (wn * y * torch.rsqrt((y ** 2).mean(-1, keepdim=True) + 1e-6)).reshape(-1, N)
    err = (out[rows].double() - y).abs().max().item()
    assert err <= 2.0 ** -10 * float(y.abs().max()) + 1e-4, (M, err)
    # form independence: the launch is pinned to the persistent 256 x 192 kernel for every row count
    head = hip_ops.gemm(a16[0][:300].contiguous(), w16, bf, prec="fp16", rms=rms, ln_apply=(st[:300].contiguous(), s, eps))
    assert torch.equal(head, out[:300])


def test_chain_matches_the_unfolded_operators(hip):
    """proj -> (LayerNorm) -> fc1 + GELU: the folded pair against bd_gemm + bd_layernorm + bd_gemm on the same inputs (two different
    roundings of the same function: the operand that is rounded is x instead of LN(x))."""
    M, K = 4096, 768
    out, st, op = _emit(M, K, seed=4)
    _, g, beta, w, b, wf, bf = _consumer_case(M, 3072, seed=6)
    e0, e1 = hip_ops.f16c8_qexp(w), hip_ops.f16c8_qexp(wf)
    xn, _ = hip_ops.layernorm(out, g, beta, 1e-6, prec="f16c8")
    plain = hip_ops.gemm(xn, hip_ops.f16c8_encode(w, e0, True), b, prec="f16c8", w_qexp=e0, act=1)
    w16 = hip_ops.f16c8_encode(wf, e1, True)
    wh, wl, _ = hip_ops.f16c8_decode(w16, e1, True)
    s = (wh.double() + wl.double()).sum(1).float().contiguous()
    fold = hip_ops.gemm(op, w16, bf, prec="f16c8", w_qexp=e1, act=1, ln_apply=(st, s, 1e-6))
    a, c = hip_ops.from_operand(plain, "f16c8"), hip_ops.from_operand(fold, "f16c8")
    ref = F.gelu(F.layer_norm(out.double(), (768,), g.double(), beta.double(), 1e-6) @ w.double().t() + b.double())
    ea, ec = (a.double() - ref).abs().max().item(), (c.double() - ref).abs().max().item()
    print(f"[ln fold chain] un-folded err {ea:.3e}, folded err {ec:.3e} (|ref| max {float(ref.abs().max()):.2f})")
    assert ec <= max(2.0 * ea, 1e-3)


def test_predicate_refuses_what_no_kernel_form_serves(hip):
    import ctypes as C
    from boxdreamer_amd import _lib
    lib = _lib.load()
    g = _lib.GemmArgs()
    assert lib.bd_gemm_takes_ln_fold(C.byref(g), _lib.prec_id("f16c8")) == 1            # nothing asked for
    x = torch.zeros(1024, dtype=torch.float32, device="cuda")
    g.ln_stats_in, g.ln_colsum = x.data_ptr(), x.data_ptr()
    g.M, g.N, g.K = 512, 768, 3072                                                      # K != 768
    assert lib.bd_gemm_takes_ln_fold(C.byref(g), _lib.prec_id("f16c8")) == 0
    assert lib.bd_gemm_takes_ln_fold(C.byref(g), _lib.prec_id("bf16")) == 0
    with pytest.raises(ValueError):
        w = torch.zeros((2, 768, 768), dtype=torch.float16, device="cuda")
        a = torch.zeros((2, 512, 768), dtype=torch.float16, device="cuda")
        hip_ops.gemm(a, w, x[:768], prec="f16x3", ln_apply=(x, x[:768], 1e-6))


# ---- split-K of the residual Linears for launches of a few tiles (ABI 9, bd_gemm_args.sk_ws / sk_split): one pose at a time

def _splitk_case(M, K, kind, split, seed=11):
    """kind: "resid" (fp32 residual, EP 3), "emit" (+ LayerNorm-fold producer, EP 4), "c8" / "c8f32" (residual in the operand copy, EP 5)."""
    N = 768
    a, w, b = _rand("a", (M, K), seed=seed).cuda(), _rand("w", (N, K), 0.04, seed).cuda(), _rand("b", (N,), 0.3, seed).cuda()
    x0 = (_rand("r", (M, N), 2.0, seed) + 0.7).cuda()
    e = hip_ops.f16c8_qexp(w)
    a16, w16 = hip_ops.f16c8_encode(a, 0, False), hip_ops.f16c8_encode(w, e, True)
    kw = dict(prec="f16c8", w_qexp=e)
    st = op = None
    if kind in ("resid", "emit"):
        kw.update(out_f32=True, resid=x0.clone())
    if kind != "resid":
        st = torch.full((M, 8, 2), float("nan"), dtype=torch.float32, device="cuda")
        op = hip_ops.f16c8_encode(x0, 0, False) if kind.startswith("c8") else torch.zeros((2, M, N), dtype=torch.float16, device="cuda")
        kw["ln_emit"] = (st, op)
    if kind.startswith("c8"):
        kw.update(ln_resid_in_op=True, out_f32=kind == "c8f32")
        if kind == "c8":
            kw["out"] = torch.zeros((2, M, N), dtype=torch.float16, device="cuda")
    ws = None
    if split is not None:
        ws = hip_ops.splitk_workspace(M, N)
        kw["split_k"] = (ws, split)
    out = hip_ops.gemm(a16, w16, b, **kw)
    torch.cuda.synchronize()
    return out, st, op, ws


@pytest.mark.parametrize("M,K", [(1536, 3072), (1536, 768), (256, 3072), (256, 768), (300, 768), (1566, 3072), (3072, 768)])
@pytest.mark.parametrize("kind", ["resid", "emit", "c8", "c8f32"])
def test_split_k_matches_the_unsplit_launch(hip, M, K, kind):
    """S workgroups per 128 x 96 tile over disjoint K ranges, summed in a fixed order: equal to the unsplit launch up to fp32 association
    (not bit-identical by contract), deterministic, and the flag region is left zero."""
    base, st0, op0, _ = _splitk_case(M, K, kind, None)
    for split in (2, 3, 4):
        out, st, op, ws = _splitk_case(M, K, kind, split)
        assert int(ws[:16384].view(torch.int32).abs().max()) == 0, "flags must be left zero"
        if kind != "c8":
            scale = float(base.abs().max())
            assert (out - base).abs().max().item() <= 4e-6 * scale, (split, (out - base).abs().max().item())
            assert not torch.equal(out, torch.zeros_like(out))
        if kind != "resid":
            x0 = hip_ops.from_operand(op0, "f16c8").double()
            x1 = hip_ops.from_operand(op, "f16c8").double()
            assert (x1 - x0).abs().max().item() <= 2.0 ** -14 * float(x0.abs().max()), split       # one operand-class rounding step at most
            assert (st[..., 0] - st0[..., 0]).abs().max().item() <= 1e-4 and ((st[..., 1] - st0[..., 1]).abs() / st0[..., 1].clamp_min(1e-3)).max().item() <= 1e-3
        again, st2, op2, _ = _splitk_case(M, K, kind, split)
        if kind != "c8":
            assert torch.equal(out, again), "split-K must be deterministic"
        if kind != "resid":
            assert torch.equal(op[0], op2[0]) and torch.equal(_lo8(op, M, 768), _lo8(op2, M, 768)) and torch.equal(st, st2)


def test_split_k_library_choice_and_reused_scratch(hip):
    """sk_split = 0: the library chooses the factor; one scratch region serves launches of different row counts back to back (the whole-path use:
    the stream's proj / fc2, then the last decoder block's compact rows)."""
    ws = hip_ops.splitk_workspace(1536, 768)
    for M, K in ((1536, 3072), (256, 3072), (1536, 768), (256, 768), (1536, 3072)):
        N = 768
        a, w, b = _rand("a", (M, K), seed=13).cuda(), _rand("w", (N, K), 0.04, 13).cuda(), _rand("b", (N,), 0.3, 13).cuda()
        x0 = (_rand("r", (M, N), 2.0, 13) + 0.7).cuda()
        e = hip_ops.f16c8_qexp(w)
        a16, w16 = hip_ops.f16c8_encode(a, 0, False), hip_ops.f16c8_encode(w, e, True)
        base = hip_ops.gemm(a16, w16, b, prec="f16c8", w_qexp=e, out_f32=True, resid=x0.clone())
        out = hip_ops.gemm(a16, w16, b, prec="f16c8", w_qexp=e, out_f32=True, resid=x0.clone(), split_k=(ws, 0))
        assert (out - base).abs().max().item() <= 4e-6 * float(base.abs().max()), (M, K)
        assert int(ws[:16384].view(torch.int32).abs().max()) == 0
    assert hip_ops.splitk_workspace(8192, 768) is None and hip_ops.splitk_workspace(1536, 1000) is None
