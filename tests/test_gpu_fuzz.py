"""GPU parity, randomised shapes: the C-ABI operators on shapes NOT taken from the path (ragged tile edges, tiny and odd sizes),
drawn from a fixed seed so a failure reproduces.  GEMMs use small-integer operands: every product and partial sum is exact in fp32,
so the result must equal the integer answer bit for bit whatever tile shape, ring depth or accumulation order a launch picks.
Nothing here falls back to PyTorch compute."""
import numpy as np
import pytest
import torch

from boxdreamer_amd import hip_ops

pytestmark = pytest.mark.gpu

KSTEP = {"bf16": 64, "fp16": 64, "bf16x3": 64, "f16c8": 64, "fp8": 128}
EPS = {"bf16": 2.0 ** -8, "fp16": 2.0 ** -11, "bf16x3": 2.0 ** -15}


def _gemm_cases(n=100, seed=20260928):
    rng = np.random.default_rng(seed)
    precs = ["bf16", "fp16", "bf16x3", "f16c8", "fp8"]
    cases = []
    for i in range(n):
        prec = precs[i % len(precs)]
        # M: anywhere, with a bias towards one-past / one-short of the 64 / 128 / 256-row tiles
        M = int(rng.integers(1, 2600))
        if rng.random() < 0.4:
            M = int(rng.choice([64, 128, 256, 512, 1024, 2048]) + rng.integers(-1, 2))
        # N: multiples of 8 (16-byte rows of every output kind), incl. multiples of 192 (persistent kernel) and their neighbours
        N = int(rng.integers(1, 300)) * 8
        if rng.random() < 0.4:
            N = int(rng.choice([192, 384, 768, 1536, 2304]) + 8 * rng.integers(-1, 2))
        if rng.random() < 0.2:                                            # several tile rounds per CU of the persistent kernel
            M = int(rng.integers(2600, 9000))
        K = int(rng.integers(1, 9)) * KSTEP[prec]
        if rng.random() < 0.15:
            K = int(rng.choice([768, 1536, 3072]))
        cases.append((prec, M, N, K, bool(rng.integers(0, 2)), int(rng.integers(0, 3)), int(rng.integers(0, 2 ** 31))))
    return cases


@pytest.mark.parametrize("prec,M,N,K,bias,form,seed", _gemm_cases())
def test_gemm_random_shapes_exact_on_integers(hip, prec, M, N, K, bias, form, seed):
    """form 0: fp32 output; 1: fp32 output accumulated in place onto a residual; 2: the operand-class (16-bit) output."""
    g = torch.Generator().manual_seed(seed)
    dev = torch.device("cuda")
    ai = torch.randint(-3, 4, (M, K), generator=g).float()
    wi = torch.randint(-2, 3, (N, K), generator=g).float()
    bi = torch.randint(-4, 5, (N,), generator=g).float() if bias else None
    exact = ai @ wi.t() + (bi if bias else 0.0)                       # |sum| <= 6 K + 4 < 2^24: exact in fp32 on the host too
    if prec == "f16c8":
        a16, w16 = hip_ops.f16c8_encode(ai.to(dev), 0, False), hip_ops.f16c8_encode(wi.to(dev), 0, True)
        kw = dict(prec="f16c8", w_qexp=0)
    else:
        a16, w16 = hip_ops.to_operand(ai.to(dev), prec), hip_ops.to_operand(wi.to(dev), prec)
        kw = dict(prec=prec)
    bd = bi.to(dev) if bias else None
    if form == 0:
        out = hip_ops.gemm(a16, w16, bd, out_f32=True, **kw)
        assert torch.equal(out.cpu(), exact), (prec, M, N, K)
    elif form == 1:
        res = torch.randint(-50, 51, (M, N), generator=g).float()
        buf = res.clone().to(dev)
        hip_ops.gemm(a16, w16, bd, out_f32=True, out=buf, resid=buf, **kw)
        assert torch.equal(buf.cpu(), exact + res), (prec, M, N, K)
    else:
        if prec == "fp8":
            out = hip_ops.gemm(a16, w16, bd, out_mode=3, **kw)        # the fp8 mode's 16-bit form: a bf16 plane
            assert torch.equal(out.cpu(), exact.to(torch.bfloat16)), (prec, M, N, K)
        elif prec == "f16c8":
            out = hip_ops.gemm(a16, w16, bd, out_mode=2, **kw)        # an f16 plane (saturating at +-65504: far away here)
            assert torch.equal(out.float().cpu(), exact.to(torch.float16).float()), (prec, M, N, K)
        else:
            out = hip_ops.from_operand(hip_ops.gemm(a16, w16, bd, **kw), prec).cpu()
            want = hip_ops.from_operand(hip_ops.to_operand(exact.to(dev), prec), prec).cpu()
            assert torch.equal(out, want), (prec, M, N, K)


def _attention_cases(n=18, seed=7):
    rng = np.random.default_rng(seed)
    cases = []
    for i in range(n):
        hd = (64, 96)[i % 2]
        seq = int(rng.integers(1, 700))
        if i % 3 == 0:
            seq = int(rng.choice([64, 128, 256, 512]) + rng.integers(-1, 2))     # one past / short of the 64-key tiles
        cases.append((("bf16", "fp16", "bf16x3")[i % 3], int(rng.integers(1, 4)), seq, int(rng.integers(1, 5)), hd,
                      int(rng.integers(0, 2 ** 31))))
    return cases


@pytest.mark.parametrize("prec,batch,seq,heads,hd,seed", _attention_cases())
def test_attention_random_shapes(hip, prec, batch, seq, heads, hd, seed):
    g = torch.Generator().manual_seed(seed)
    qkv = torch.randn(batch, seq, 3, heads, hd, generator=g)
    qkv[:, :, 0] *= 1.5
    t = hip_ops.to_operand(qkv.reshape(batch * seq, -1).cuda(), prec)
    out = hip_ops.attention(t, batch, seq, heads, hd, hd ** -0.5, prec=prec)
    got = hip_ops.from_operand(out, prec).cpu().reshape(batch, seq, heads, hd)
    src = hip_ops.from_operand(t, prec).cpu().reshape(batch, seq, 3, heads, hd).double()
    q, k, v = (src[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    ref = (((q @ k.transpose(-1, -2)) * hd ** -0.5).softmax(-1) @ v).permute(0, 2, 1, 3).float()
    assert torch.isfinite(got).all()
    err = (got - ref).abs().max().item()
    assert err < 6 * EPS[prec] * max(1.0, ref.abs().max().item()) + 2e-5, (prec, batch, seq, heads, hd, err)


@pytest.mark.parametrize("prec", ["bf16", "fp16", "bf16x3", "f16c8"])
def test_layernorm_random_row_counts(hip, prec):
    """Row counts around the kernel's rows-per-workgroup; widths the path uses (768) and a narrower one."""
    g = torch.Generator().manual_seed(5)
    for rows, C in [(1, 768), (7, 768), (63, 768), (65, 768), (1000, 768), (33, 384)]:
        x = torch.randn(rows, C, generator=g) * 3.0 + 0.5
        gamma, beta = 1.0 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
        out, _ = hip_ops.layernorm(x.cuda(), gamma.cuda(), beta.cuda(), 1e-5, prec=prec)
        if prec == "f16c8":
            hi, lo, _ = hip_ops.f16c8_decode(out)
            got, eps = (hi + lo).cpu(), 2.0 ** -14
        else:
            got, eps = hip_ops.from_operand(out, prec).cpu(), EPS[prec]
        ref = torch.nn.functional.layer_norm(x.double(), (C,), gamma.double(), beta.double(), 1e-5).float()
        assert (got - ref).abs().max().item() < 2 * eps * max(1.0, ref.abs().max().item()) + 1e-5, (prec, rows, C)


@pytest.mark.parametrize("n_maps,H,W,levels,seed", [(3, 224, 224, 5, 1), (16, 224, 224, 3, 2), (8, 112, 112, 7, 3), (5, 56, 98, 2, 4),
                                                     (1, 224, 224, 1, 5), (64, 28, 28, 4, 6), (2, 224, 224, 1000, 7)])
def test_decode_topk_random_maps_with_ties(hip, n_maps, H, W, levels, seed):
    """Top-20 decode on maps quantised to a few levels (ties everywhere, across the rank-20 boundary too): index sets, order and
    the mean keypoints must equal the oracle's (lower index wins a tie), bit for bit."""
    from oracle import boxdreamer_oracle as orc
    g = torch.Generator().manual_seed(seed)
    q = torch.randint(0, levels, (1, n_maps, H, W), generator=g).float()
    hm = (q / max(levels - 1, 1)) * 2.0 - 1.0 if levels > 1 else torch.full((1, n_maps, H, W), 0.25)
    kp, kn, idx = hip_ops.decode_topk(hm.cuda())
    on, okp, oidx = orc.recover_bb8_corners(hm)
    assert torch.equal(idx.cpu().long().reshape(oidx.shape), oidx)
    assert torch.equal(kp.cpu().reshape(okp.shape), okp)
    assert (kn.cpu().reshape(on.shape) - on).abs().max().item() < 1e-6


@pytest.mark.parametrize("prec", ["bf16", "fp16", "bf16x3", "f16c8"])
@pytest.mark.parametrize("size,n", [(224, 2), (112, 3), (98, 2), (56, 5), (28, 1)])
def test_im2col_and_patchify_sizes_and_dtypes(hip, prec, size, n):
    """The strip / pair forms of the two input-layout kernels (patch 14) on grids that are not a multiple of the 8-patch strip, a single
    image, and all three input dtypes, against torch.unfold / the oracle's patchify."""
    import torch.nn.functional as F
    from oracle import boxdreamer_oracle as orc
    g = torch.Generator().manual_seed(size * 7 + n)
    grid = size // 14
    eps = {"bf16": 2.0 ** -8, "fp16": 2.0 ** -11, "bf16x3": 2.0 ** -15, "f16c8": 2.0 ** -14}[prec]

    def decode(a):
        if prec == "f16c8":
            hi, lo, _ = hip_ops.f16c8_decode(a)
            return (hi + lo).cpu()
        return hip_ops.from_operand(a, prec).cpu()

    for dt in (torch.float32, torch.bfloat16, torch.float16):
        img = torch.rand(n, 3, size, size, generator=g).to(dt)
        heat = (torch.rand(n, 8, size, size, generator=g) * 2 - 1).to(dt)
        got = decode(hip_ops.im2col_images(img.cuda(), prec=prec))
        mean = torch.tensor(orc._IMAGENET_MEAN).view(1, 3, 1, 1)
        std = torch.tensor(orc._IMAGENET_STD).view(1, 3, 1, 1)
        cols = F.unfold((img.float() - mean) / std, 14, stride=14).transpose(1, 2).reshape(n * grid * grid, 588)
        assert torch.equal(got[:, 588:], torch.zeros(n * grid * grid, 52))
        assert (got[:, :588] - cols).abs().max().item() < 8 * eps * 3 + 1e-6, (prec, size, dt)
        got = decode(hip_ops.patchify_heatmaps(heat.cuda(), prec=prec))
        ref = orc.patchify(heat.float(), 14, 8).reshape(n * grid * grid, 1568)
        assert torch.equal(got[:, 1568:], torch.zeros(n * grid * grid, 32))
        assert (got[:, :1568] - ref).abs().max().item() < 2 * eps + 1e-7, (prec, size, dt)
