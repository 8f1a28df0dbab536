"""Thin tensor-level wrappers over the C ABI (one function per exported operator).

Used by the GPU parity tests and by the module mirrors.  All tensors must live on the HIP device;
every call is enqueued on torch's current stream.  In the "bf16x3" mode a 16-bit activation is a
stacked pair [2, rows, cols] (hi plane, lo plane).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import check, planes, prec_id, ptr, stream


def _alloc16(rows: int, cols: int, prec, device) -> torch.Tensor:
    shape = (2, rows, cols) if planes(prec) == 2 else (rows, cols)
    return torch.empty(shape, dtype=_lib.op_dtype(prec), device=device)


def _plane(t: torch.Tensor, prec) -> int:
    return t[0].numel() if planes(prec) == 2 else 0


E4M3_MAX = 448.0


def f16c8_qexp(w: torch.Tensor) -> int:
    """Exponent E of a weight tensor's e4m3 planes: the largest power of two that keeps max|w| * 2^E inside e4m3."""
    m = float(w.detach().abs().max())
    return 0 if m <= 0 else int(torch.floor(torch.log2(torch.tensor(E4M3_MAX / m))).item())


def _f16c8_perm(K: int, device) -> torch.Tensor:
    """Byte position -> k inside every 32-block of the lo8 plane: byte 16 h + 8 a + j holds k = 16 a + 8 h + j."""
    pos = torch.arange(32, device=device)
    h, a, j = pos // 16, (pos // 8) % 2, pos % 8
    k_of_pos = 16 * a + 8 * h + j
    return (torch.arange(0, K, 32, device=device)[:, None] + k_of_pos[None, :]).reshape(-1)


def f16c8_encode(x: torch.Tensor, qexp: int = 0, weight: bool = False) -> torch.Tensor:
    """fp32 [rows, K] (K % 32 == 0) -> [2, rows, K] float16 STORAGE of the F16C8 operand class (include/boxdreamer_hip.h):
    plane 0 = f16(x) (clamped to f16's finite range only); plane 1 = raw bytes (both e4m3 images saturate at +-448):
      activations: the first rows*K bytes are the lo8 plane e4m3((x - hi) 2^D), k-permuted per 32-block (rest unused);
      weights:     per 32-block 64 bytes = for each lane half h: [q8 x 16 | lo8 x 16] of its sixteen k (same k order),
                   q8 = e4m3(hi 2^E), lo8 = e4m3((w - hi) 2^(E + D)).
    Torch ops: weight packing at load time and test helpers (not on the hot path)."""
    x = x.float().clamp(-65504.0, 65504.0)
    rows, K = x.shape
    assert K % 32 == 0, "F16C8 operands are laid out in 32-element blocks"
    hi = x.half()
    lo = (x - hi.float()) * 2.0 ** (qexp + _lib.F16C8_D)
    perm = _f16c8_perm(K, x.device)
    l8 = lo.clamp(-E4M3_MAX, E4M3_MAX).to(torch.float8_e4m3fn).view(torch.uint8)[:, perm]
    if weight:
        q8 = (hi.float() * 2.0 ** qexp).clamp(-E4M3_MAX, E4M3_MAX).to(torch.float8_e4m3fn).view(torch.uint8)[:, perm]
        mix = torch.stack([q8.reshape(rows, K // 16, 16), l8.reshape(rows, K // 16, 16)], dim=2)     # [rows, K/16, 2, 16]
        plane1 = mix.reshape(-1).contiguous()
    else:
        plane1 = torch.zeros((rows * K * 2,), dtype=torch.uint8, device=x.device)
        plane1[: rows * K] = l8.reshape(-1)
    return torch.stack([hi, plane1.view(torch.float16).reshape(rows, K)]).contiguous()


def f16c8_decode(t: torch.Tensor, qexp: int = 0, weight: bool = False):
    """(hi, lo, q) as fp32 [rows, K] -- exactly the three values the GEMM multiplies."""
    hi = t[0].float()
    rows, K = hi.shape
    raw = t[1].contiguous().view(torch.uint8).reshape(-1)
    inv = torch.empty(K, dtype=torch.long, device=t.device)
    inv[_f16c8_perm(K, t.device)] = torch.arange(K, device=t.device)
    dec = lambda b: b[:, inv].contiguous().view(torch.float8_e4m3fn).float()
    if weight:
        mix = raw.reshape(rows, K // 16, 2, 16)
        q = dec(mix[:, :, 0].reshape(rows, K)) * 2.0 ** -qexp
        lo = dec(mix[:, :, 1].reshape(rows, K)) * 2.0 ** -(qexp + _lib.F16C8_D)
    else:
        lo = dec(raw[: rows * K].reshape(rows, K)) * 2.0 ** -(qexp + _lib.F16C8_D)
        q = (hi * 2.0 ** qexp).clamp(-E4M3_MAX, E4M3_MAX).to(torch.float8_e4m3fn).float() * 2.0 ** -qexp     # derived by the GEMM in registers (saturating)
    return hi, lo, q


def to_operand(x: torch.Tensor, prec) -> torch.Tensor:
    """fp32 [rows, cols] -> operand tensor (test helper; torch ops, not on the product path)."""
    dt = _lib.op_dtype(prec)
    if prec_id(prec) == _lib.PREC_F16C8:
        return f16c8_encode(x, 0, False)
    if prec_id(prec) == _lib.PREC_FP8:
        return x.float().clamp(-448.0, 448.0).to(dt).contiguous()
    if dt == torch.float16:
        x = x.float().clamp(-65504.0, 65504.0)             # the kernels' f16 conversions saturate (bd_common.h: RANGE)
    hi = x.to(dt)
    if planes(prec) == 1:
        return hi.contiguous()
    return torch.stack([hi, (x.float() - hi.float()).to(dt)]).contiguous()


def from_operand(t: torch.Tensor, prec) -> torch.Tensor:
    if prec_id(prec) == _lib.PREC_F16C8:
        hi, lo, _ = f16c8_decode(t)
        return hi + lo
    return t.float() if planes(prec) == 1 else t[0].float() + t[1].float()


_OUT_DTYPES = {2: torch.float16, 3: torch.bfloat16}      # single planes; 4 / 5: split-bf16 / split-f16 planes [2, rows, N]
_OUT_SPLIT = {4: torch.bfloat16, 5: torch.float16}


def gemm(a16, w16, bias=None, *, prec="bf16", act=_lib.ACT_NONE, resid=None, addtab=None, out=None,
         out_f32=False, n=None, out_rows=None, rpg=(0, 0, 0), wscale=None, out_mode=None, w_qexp=0, rms=None,
         ln_emit=None, ln_apply=None, ln_resid_in_op=False, split_k=None):
    """out[map(r)] = act(wscale * (A W^T) + bias) + addtab[r % rows(addtab)] + resid[map(r)].
    out_mode: None -> operand dtype (or fp32 with out_f32), 2 -> f16 single plane, 3 -> bf16 single plane, 4 -> split-bf16 (hi, lo)
    planes, 5 -> split-f16 (hi, lo) planes (bd_gemm_args.out_f32, include/boxdreamer_hip.h).
    LayerNorm fold (ABI 8): ln_emit = (stats [M, N / 96, 2] fp32, operand copy [2, M, N] F16C8 storage) -- the producer side;
    ln_apply = (stats [M, 8, 2], column sums [N], eps) -- the consumer side.
    split_k = (scratch uint8 tensor of splitk_workspace_bytes(M, N) whose first 16 KiB are zero, factor 0 = library's choice | 2 .. 4):
    bd_gemm_args.sk_ws / sk_split (ABI 9)."""
    lib = _lib.load()
    np_ = planes(prec)
    A2 = a16[0] if np_ == 2 else a16
    W2 = w16[0] if np_ == 2 else w16
    M, K = A2.shape
    N = n if n is not None else W2.shape[0]
    rows_out = out_rows if out_rows is not None else M
    mode = int(out_f32) if out_mode is None else out_mode
    if out is None:
        if mode == 1:
            out = torch.empty((rows_out, N), dtype=torch.float32, device=A2.device)
        elif mode in _OUT_DTYPES:
            out = torch.empty((rows_out, N), dtype=_OUT_DTYPES[mode], device=A2.device)
        elif mode in _OUT_SPLIT:
            out = torch.empty((2, rows_out, N), dtype=_OUT_SPLIT[mode], device=A2.device)
        else:
            out = _alloc16(rows_out, N, prec, A2.device)
    g = _lib.GemmArgs()
    g.A, g.lda, g.a_plane = ptr(a16), A2.stride(0), _plane(a16, prec)
    g.W, g.ldw, g.w_plane = ptr(w16), W2.stride(0), _plane(w16, prec)
    g.bias = ptr(bias)
    g.wscale = ptr(wscale)
    g.resid, g.ldr = ptr(resid), (resid.stride(0) if resid is not None else 0)
    g.addtab, g.tab_rows = ptr(addtab), (addtab.shape[0] if addtab is not None else 0)
    o2 = out[0] if (mode in _OUT_SPLIT or (not mode and np_ == 2)) else out
    g.out, g.ldo, g.out_f32 = ptr(out), o2.stride(0), mode
    g.out_plane = out[0].numel() if mode in _OUT_SPLIT else (0 if mode else _plane(out, prec))
    g.w_qexp = int(w_qexp)
    if rms is not None:                      # (wq, wk, eps[, parts]): fused q/k RMSNorm of a QKV Linear ([q | k | v] columns, or [q | k])
        g.rms_wq, g.rms_wk, g.rms_eps = ptr(rms[0]), ptr(rms[1]), float(rms[2])
        g.rms_parts = int(rms[3]) if len(rms) > 3 else 0
    g.M, g.N, g.K, g.act = M, N, K, act
    g.rpg_in, g.rpg_out, g.row_off = rpg
    if ln_emit is not None:
        st, op = ln_emit
        g.ln_stats_out, g.ln_op_out, g.ln_op_plane, g.ln_op_ld = ptr(st), ptr(op), op[0].numel(), op[0].stride(0)
        g.ln_resid_in_op = int(bool(ln_resid_in_op))      # the residual rows are read from (and the sum written back to) `op`; fp32 rows only with out_f32
    if ln_apply is not None:
        g.ln_stats_in, g.ln_colsum, g.ln_eps = ptr(ln_apply[0]), ptr(ln_apply[1]), float(ln_apply[2])
    if split_k is not None:
        g.sk_ws, g.sk_split = ptr(split_k[0]), int(split_k[1])
    if (ln_emit is not None or ln_apply is not None) and not lib.bd_gemm_takes_ln_fold(C.byref(g), prec_id(prec)):
        raise ValueError("bd_gemm_takes_ln_fold: this launch has no kernel form with the LayerNorm-fold epilogues")
    check(lib.bd_gemm(C.byref(g), prec_id(prec), stream()), "bd_gemm")
    return out


def splitk_workspace(M, N, device="cuda"):
    """Zeroed scratch region for gemm(..., split_k=(ws, factor)); None when (M, N) has no split-K form."""
    nbytes = _lib.load().bd_gemm_splitk_workspace_bytes(int(M), int(N))
    return torch.zeros(nbytes, dtype=torch.uint8, device=device) if nbytes else None


def layernorm(x, gamma, beta, eps, *, prec="bf16", want16=True, want32=False, rows=None, rpg=(0, 0, 0)):
    lib = _lib.load()
    M = rows if rows is not None else x.shape[0]
    cols = x.shape[1]
    o16 = _alloc16(M, cols, prec, x.device) if want16 else None
    o32 = torch.empty((M, cols), dtype=torch.float32, device=x.device) if want32 else None
    check(lib.bd_layernorm(ptr(x), x.stride(0), ptr(gamma), ptr(beta), eps, ptr(o16),
                           _plane(o16, prec) if o16 is not None else 0, ptr(o32), cols, M, cols, *rpg,
                           prec_id(prec), stream()), "bd_layernorm")
    return o16, o32


def qk_rmsnorm_(qkv16, wq, wk, eps, heads, head_dim, *, prec="bf16"):
    lib = _lib.load()
    q2 = qkv16[0] if planes(prec) == 2 else qkv16
    rows = q2.numel() // (3 * heads * head_dim)
    check(lib.bd_qk_rmsnorm(ptr(qkv16), _plane(qkv16, prec), ptr(wq), ptr(wk), eps, rows, heads, head_dim,
                            prec_id(prec), stream()), "bd_qk_rmsnorm")
    return qkv16


def attention(qkv16, batch, seq, heads, head_dim, scale, *, prec="bf16"):
    lib = _lib.load()
    dev = qkv16.device
    out = _alloc16(batch * seq, heads * head_dim, prec, dev)
    check(lib.bd_attention(ptr(qkv16), _plane(qkv16, prec), ptr(out), _plane(out, prec), batch, seq, heads,
                           head_dim, scale, prec_id(prec), stream()), "bd_attention")
    return out


def attention_prefix(qkv16, batch, seq, heads, head_dim, scale, n_prefix, *, prec="bf16", prefix_queries=True, out=None):
    """bd_attention_prefix: patch queries in the tiled kernel, the n_prefix leading queries in a side launch (or skipped)."""
    lib = _lib.load()
    if out is None:
        out = _alloc16(batch * seq, heads * head_dim, prec, qkv16.device)
    check(lib.bd_attention_prefix(ptr(qkv16), _plane(qkv16, prec), ptr(out), _plane(out, prec), batch, seq, heads, head_dim, scale,
                                  n_prefix, int(bool(prefix_queries)), prec_id(prec), stream()), "bd_attention_prefix")
    return out


def attention_q(qkv16, batch, seq, heads, head_dim, scale, q_view, q_len, *, prec="bf16"):
    """Attention with queries restricted to rows [q_view[b]*q_len, +q_len) of each sequence; compact output."""
    lib = _lib.load()
    out = _alloc16(batch * q_len, heads * head_dim, prec, qkv16.device)
    check(lib.bd_attention_q(ptr(qkv16), _plane(qkv16, prec), ptr(out), _plane(out, prec), batch, seq, heads, head_dim,
                             scale, ptr(q_view), q_len, prec_id(prec), stream()), "bd_attention_q")
    return out


def im2col_images(images, patch=14, kpad=640, *, prec="bf16"):
    lib = _lib.load()
    images = images.contiguous()
    n, size = images.shape[0], images.shape[-1]
    grid = size // patch
    out = _alloc16(n * grid * grid, kpad, prec, images.device)
    check(lib.bd_im2col_images(ptr(images), _lib.dtype_id(images), ptr(out), _plane(out, prec), n, size, patch,
                               kpad, prec_id(prec), stream()), "bd_im2col_images")
    return out


def patchify_heatmaps(heat, patch=14, kpad=1600, *, prec="bf16"):
    lib = _lib.load()
    heat = heat.contiguous()
    n, c, size = heat.shape[0], heat.shape[1], heat.shape[-1]
    grid = size // patch
    out = _alloc16(n * grid * grid, kpad, prec, heat.device)
    check(lib.bd_patchify_heatmaps(ptr(heat), _lib.dtype_id(heat), ptr(out), _plane(out, prec), n, c, size, patch,
                                   kpad, prec_id(prec), stream()), "bd_patchify_heatmaps")
    return out


def unpatchify_sigmoid(proj, B, size=224, patch=14):
    lib = _lib.load()
    logits = torch.empty((B, 8, size, size), dtype=torch.float32, device=proj.device)
    heat = torch.empty_like(logits)
    check(lib.bd_unpatchify_sigmoid(ptr(proj), ptr(logits), ptr(heat), B, 8, size, patch, stream()),
          "bd_unpatchify_sigmoid")
    return logits, heat


def decode_topk(heat, k=20, want_idx=True):
    """heat: fp32 [..., H, W] in [-1, 1] -> (kp_px [..., 2], kp_norm [..., 2], idx [..., k] int32)."""
    lib = _lib.load()
    heat = heat.contiguous()
    if heat.dtype != torch.float32:
        heat = heat.float()
    H, W = heat.shape[-2:]
    lead = heat.shape[:-2]
    n = heat.numel() // (H * W)
    kp = torch.empty((n, 2), dtype=torch.float32, device=heat.device)
    kn = torch.empty_like(kp)
    idx = torch.empty((n, k), dtype=torch.int32, device=heat.device) if want_idx else None
    check(lib.bd_decode_topk(ptr(heat), n, H, W, k, ptr(kp), ptr(kn), ptr(idx), stream()), "bd_decode_topk")
    return kp.reshape(*lead, 2), kn.reshape(*lead, 2), (idx.reshape(*lead, k) if want_idx else None)
