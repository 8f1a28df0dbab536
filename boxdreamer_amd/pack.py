"""Weight re-packing at load time (pure host logic; unit-tested on CPU).

fp32 checkpoint tensors (reference key names) -> device arrays the HIP kernels consume:
  * nn.Linear weights stay [N, K] (both GEMM operands are K-contiguous), K zero-padded to a
    multiple of 64, cast to the operand dtype; BF16X3 stores two planes (hi, lo = bf16(w - hi));
  * DINOv2 LayerScale gamma is folded into proj / fc2 (weight rows and bias) in fp32 before the cast
    (src/models/sources/DINOv2/layers/block.py:89-114, layer_scale.py:26-27);
  * input-independent tables are precomputed once: the DINOv2 positional embedding resampled
    37x37 -> grid x grid (vision_transformer.py:179-211) and BETR's 2-D sincos table
    (pos_encodiong.py:125-213, consumed at betr.py:357-364).
"""
from __future__ import annotations

import ctypes as C
import math
import os

import torch
import torch.nn.functional as F

from . import _lib


E4M3_MAX = 448.0


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def split_planes(w: torch.Tensor, prec) -> torch.Tensor:
    """fp32 -> operand dtype; BF16X3 -> stacked (hi, lo) planes, shape [2, ...]."""
    dt = _lib.op_dtype(prec)
    if _lib.prec_id(prec) == _lib.PREC_FP8:
        return w.clamp(-E4M3_MAX, E4M3_MAX).to(dt).contiguous()
    if dt == torch.float16:
        w = w.clamp(-65504.0, 65504.0)
    hi = w.to(dt)
    if _lib.planes(prec) == 1:
        return hi.contiguous()
    lo = (w - hi.float()).to(dt)
    return torch.stack([hi, lo]).contiguous()


def embed_k_multiple(prec) -> int:
    """K padding of the two embedding GEMMs (patch embed: K = 588, heat-map embed: K = 1568 -- the only K of the path that the model does not
    fix).  F16C8 family: a multiple of 192 (= 6 slabs of 32), so that these launches take the THREE-stage operand ring of gemm_kernel_pc_f16c8
    like every other Linear (K / 32 % 3 == 0) instead of the two-stage form (~1.3 us per slab against ~0.7 one pose at a time, ~2800 against
    ~2000 cycles at B = 32): 768 / 1728 instead of 640 / 1600 -- the padded columns are zeros on both operands, the sums do not change."""
    return 192 if _lib.operand_prec(prec) == _lib.PREC_F16C8 else _lib.k_multiple(prec)


def pack_linear_weight(weight: torch.Tensor, prec, kpad: int | None = None,
                       row_scale: torch.Tensor | None = None, return_scale: bool = False):
    """[N, K] fp32 -> [planes?, N, Kpad] operand dtype (row_scale folds LayerScale).

    FP8: each output channel n is stored as e4m3(w[n, :] / s_n) with s_n = max|w[n, :]| / 448; the GEMM epilogue
    multiplies the accumulator by s_n.  With return_scale the fp32 [N] scale vector is returned too (None otherwise)."""
    w = weight.detach().float().reshape(weight.shape[0], -1)
    if row_scale is not None:
        w = w * row_scale.detach().float().reshape(-1, 1)
    k = w.shape[1]
    mult = _lib.k_multiple(prec)
    kp = kpad if kpad is not None else round_up(k, mult)
    if kp % mult or kp < k:
        raise ValueError(f"bad kpad {kp} for K={k}")
    if kp != k:
        w = F.pad(w, (0, kp - k))
    scale = None
    if _lib.prec_id(prec) == _lib.PREC_F16C8:
        from . import hip_ops
        qexp = hip_ops.f16c8_qexp(w)
        packed = hip_ops.f16c8_encode(w, qexp, weight=True)
        return (packed, qexp) if return_scale else packed
    if _lib.prec_id(prec) == _lib.PREC_FP8:
        scale = (w.abs().amax(dim=1).clamp_min(1e-30) / E4M3_MAX).contiguous()
        packed = (w / scale[:, None]).clamp(-E4M3_MAX, E4M3_MAX).to(torch.float8_e4m3fn).contiguous()
    else:
        packed = split_planes(w, prec)
    return (packed, scale) if return_scale else packed


def pack_bias(bias: torch.Tensor, row_scale: torch.Tensor | None = None) -> torch.Tensor:
    b = bias.detach().float()
    if row_scale is not None:
        b = b * row_scale.detach().float()
    return b.contiguous()


def sincos_table(dim: int, grid: int) -> torch.Tensor:
    """BETR positional table (grid*grid, dim): token i*grid+j -> [sin(j w)|cos(j w)|sin(i w)|cos(i w)],
    w_d = 10000^(-d/(dim/4)), float64 math then fp32 (pos_encodiong.py:139-150, 200-213)."""
    q = dim // 4
    omega = 1.0 / 10000 ** (torch.arange(q, dtype=torch.float64) / q)
    idx = torch.arange(grid, dtype=torch.float64)
    ii = idx.repeat_interleave(grid).reshape(-1, 1)
    jj = idx.repeat(grid).reshape(-1, 1)
    ow, oh = jj * omega, ii * omega
    return torch.cat([ow.sin(), ow.cos(), oh.sin(), oh.cos()], dim=1).float().contiguous()


def dino_pos_tables(pos_embed: torch.Tensor, cls_token: torch.Tensor, register_tokens: torch.Tensor | None,
                    grid: int):
    """(prefix_tokens [1+nreg, C], pos_patch [grid*grid, C]) in fp32.

    prepare_tokens_with_masks (vision_transformer.py:213-232): cls gets pos[0] added, patch tokens get
    the bicubic+antialias resample of pos[1:]; registers are inserted afterwards without pos."""
    pe = pos_embed.detach().float()
    n = pe.shape[1] - 1
    dim = pe.shape[-1]
    if n == grid * grid:
        patch_pe = pe[0, 1:]
    else:
        m = int(math.sqrt(n))
        if m * m != n:
            raise ValueError("pos_embed is not square")
        patch_pe = F.interpolate(pe[:, 1:].reshape(1, m, m, dim).permute(0, 3, 1, 2), mode="bicubic",
                                 antialias=True, size=(grid, grid))
        patch_pe = patch_pe.permute(0, 2, 3, 1).reshape(-1, dim)
    rows = [cls_token.detach().float().reshape(1, dim) + pe[0, :1]]
    if register_tokens is not None:
        rows.append(register_tokens.detach().float().reshape(-1, dim))
    return torch.cat(rows, 0).contiguous(), patch_pe.contiguous()


class Packed:
    """Owns the device tensors behind a ctypes weight struct (keeps them alive).

    Linears are packed per operand CLASS on demand and cached (`linear_of`), so that the per-Linear promotion of the F16C8 family
    (include/boxdreamer_hip.h: BD_PROMOTE_*; boxdreamer_amd/calibrate.py) only swaps pointers in the struct (`set_promote`)."""

    def __init__(self, base_prec=None, device=None):
        self.tensors = []
        self.struct = None
        self.blocks = None
        self.base = None if base_prec is None else _lib.prec_id(base_prec)
        self.device = device
        self._src = {}        # linear name -> (weight, bias, row_scale, kpad)
        self._cache = {}      # (linear name, operand class) -> _lib.Linear
        self.block_names = [] # per block: prefix of its four Linears' names
        self.promote = None   # (per-block masks, misc mask, feats_prec) currently written into the struct
        self.named = {}       # introspection (tests): "pos_table" / (linear name, class) -> the device tensor behind the pointer

    def keep(self, t: torch.Tensor, device) -> C.c_void_p:
        t = t.to(device).contiguous()
        self.tensors.append(t)
        return C.c_void_p(t.data_ptr())

    def linear(self, w, b: torch.Tensor, device) -> _lib.Linear:
        """w: packed weight, or (packed weight, per-channel scale | None) from pack_linear_weight(return_scale=True)."""
        scale, qexp = None, 0
        if isinstance(w, tuple):
            w, scale = w
            if isinstance(scale, int):            # F16C8: the tensor's e4m3 exponent, not a per-channel scale vector
                scale, qexp = None, scale
        ws = self.keep(scale, device) if scale is not None else C.c_void_p(0)
        return _lib.Linear(self.keep(w, device), self.keep(b, device), ws, qexp)

    def register(self, name: str, weight, bias, row_scale=None, kpad=None) -> None:
        self._src[name] = (weight, bias, row_scale, kpad)

    def linear_of(self, name: str, cls=None) -> _lib.Linear:
        """The Linear `name` packed in operand class `cls` (default: the base class); packed once per class."""
        cls = self.base if cls is None else _lib.prec_id(cls)
        key = (name, cls)
        if key not in self._cache:
            w, b, rs, kpad = self._src[name]
            self._cache[key] = self.linear(pack_linear_weight(w, cls, kpad=kpad, row_scale=rs, return_scale=True), pack_bias(b, rs),
                                           self.device)
            self.named[key] = next(t for t in self.tensors[::-1] if t.dim() >= 2)        # the packed weight just kept
        return self._cache[key]

    def set_promote(self, masks=None, misc: int = 0, feats_prec: int = 0) -> None:
        """Write the per-Linear operand classes into the struct: bit set -> that Linear's weight in the promoted class's layout (F16C8 family: split-f16 planes; e4m3: bf16)."""
        depth = len(self.block_names)
        masks = [0] * depth if masks is None else [int(m) for m in masks]
        if len(masks) != depth:
            raise ValueError(f"{len(masks)} promotion masks for {depth} blocks")
        masks = [m | _lib.PROMOTE_FC2 if m & _lib.PROMOTE_FC1 else m for m in masks]
        if misc & _lib.PROMOTE_ADAPTER_FC1 and "adapter_fc1" in self._src:
            misc |= _lib.PROMOTE_ADAPTER_FC2
        state = (tuple(masks), int(misc), int(feats_prec))
        if state == self.promote:
            return
        if (any(masks) or misc or feats_prec) and self.base not in (_lib.PREC_F16C8, _lib.PREC_FP8):
            raise ValueError("per-Linear promotion exists for the F16C8 family (-> split-f16) and for fp8 (-> bf16) only")
        x3 = _lib.promoted_class(self.base)
        for i, (bw, pre) in enumerate(zip(self.blocks, self.block_names)):
            m = masks[i]
            bw.qkv = self.linear_of(pre + "qkv", x3 if m & _lib.PROMOTE_QKV else None)
            bw.proj = self.linear_of(pre + "proj", x3 if m & _lib.PROMOTE_PROJ else None)
            bw.fc1 = self.linear_of(pre + "fc1", x3 if m & _lib.PROMOTE_FC1 else None)
            bw.fc2 = self.linear_of(pre + "fc2", x3 if m & _lib.PROMOTE_FC2 else None)
            bw.promote = m
        w = self.struct
        if "patch_embed" in self._src:
            w.patch_embed = self.linear_of("patch_embed", x3 if misc & _lib.PROMOTE_PATCH_EMBED else None)
            w.feats_prec = int(feats_prec)
        else:
            w.adapter_fc1 = self.linear_of("adapter_fc1", x3 if misc & _lib.PROMOTE_ADAPTER_FC1 else None)
            w.adapter_fc2 = self.linear_of("adapter_fc2", x3 if misc & _lib.PROMOTE_ADAPTER_FC2 else None)
            w.bbox_emb = self.linear_of("bbox_emb", x3 if misc & _lib.PROMOTE_BBOX_EMB else None)
            w.bbox_proj = self.linear_of("bbox_proj", x3 if misc & _lib.PROMOTE_BBOX_PROJ else None)
        w.promote_misc = int(misc)
        self.promote = state

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.tensors)


def _pack_block(pk: Packed, sd: dict, p: str, prec, device, ls: bool, qk_norm: bool) -> _lib.BlockWeights:
    g1 = sd[p + "ls1.gamma"] if ls else None
    g2 = sd[p + "ls2.gamma"] if ls else None
    bw = _lib.BlockWeights()
    bw.ln1_w = pk.keep(sd[p + "norm1.weight"].float(), device)
    bw.ln1_b = pk.keep(sd[p + "norm1.bias"].float(), device)
    bw.ln2_w = pk.keep(sd[p + "norm2.weight"].float(), device)
    bw.ln2_b = pk.keep(sd[p + "norm2.bias"].float(), device)
    pk.register(p + "qkv", sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"])
    pk.register(p + "proj", sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"], g1)
    pk.register(p + "fc1", sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])
    pk.register(p + "fc2", sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"], g2)
    pk.block_names.append(p)
    if qk_norm:
        bw.q_norm_w = pk.keep(sd[p + "attn.q_norm.weight"].float(), device)
        bw.k_norm_w = pk.keep(sd[p + "attn.k_norm.weight"].float(), device)
        if _lib.prec_id(prec) == _lib.PREC_F16C8:   # f16 single-plane copy whose q, k rows BD_PREC_F16C8_QK16 reads (3.5 MB per block)
            bw.qkv16 = pk.linear(pack_linear_weight(sd[p + "attn.qkv.weight"], "fp16"), pack_bias(sd[p + "attn.qkv.bias"]), device)
    if _lib.prec_id(prec) == _lib.PREC_F16C8 and ln_fold_enabled():
        _pack_ln_fold(pk, bw, sd, p, device, qk_norm)
    return bw


def ln_fold_enabled() -> bool:
    """The LayerNorm fold of the F16C8 family (ABI 8) is on unless $BOXDREAMER_HIP_LNFOLD = 0 (A/B measurements; read at pack time -- the
    library itself reads no environment)."""
    return os.environ.get("BOXDREAMER_HIP_LNFOLD", "1") != "0"


def fold_layernorm(weight: torch.Tensor, bias: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor):
    """LN(x) W^T + b = rstd (x W'^T - mean s) + b'  with  W' = W . gamma (per input column),  b' = b + W beta  (fp64, then fp32)."""
    w = weight.detach().double().reshape(weight.shape[0], -1)
    return (w * gamma.detach().double()[None, :]).float(), (bias.detach().double() + w @ beta.detach().double()).float()


def _pack_ln_fold(pk: "Packed", bw, sd: dict, p: str, device, qk_norm: bool) -> None:
    """The gain-folded copies of the two Linears that follow a LayerNorm, in the F16C8 layout, + the column sums of the ROUNDED weights
    each launch multiplies (include/boxdreamer_hip.h: bd_block_weights.qkv_f ...)."""
    from . import hip_ops
    bw.ln_resid3 = int(os.environ.get("BOXDREAMER_HIP_RESID3", "1") != "0")       # (A/B switch, read at pack time like BOXDREAMER_HIP_LNFOLD)
    for name, norm, lin, fs in (("qkv_f", "norm1", "attn.qkv", "qkv_s"), ("fc1_f", "norm2", "mlp.fc1", "fc1_s")):
        wf, bf = fold_layernorm(sd[p + lin + ".weight"], sd[p + lin + ".bias"], sd[p + norm + ".weight"], sd[p + norm + ".bias"])
        packed, qexp = pack_linear_weight(wf, _lib.PREC_F16C8, return_scale=True)
        hi, lo, _ = hip_ops.f16c8_decode(packed, qexp, True)
        setattr(bw, name, pk.linear((packed, qexp), bf, device))
        pk.named[(p + name, _lib.PREC_F16C8)] = pk.tensors[-2]
        setattr(bw, fs, pk.keep((hi.double() + lo.double()).sum(1).float(), device))
        if name == "qkv_f" and qk_norm:                # the f16 copy of the folded QKV weight (BD_PREC_F16C8_QK16 reads its q, k rows)
            w16 = pack_linear_weight(wf, "fp16")
            bw.qkv16_f = pk.linear(w16, bf, device)
            bw.qkv16_s = pk.keep(w16.double().sum(1).float(), device)


def pack_dino(sd: dict, prec, device, heads: int, patch: int = 14, img_size: int = 224) -> Packed:
    """DINOv2 ViT state_dict (hub key names) -> bd_dino_weights."""
    pk = Packed(prec, device)
    dim = sd["cls_token"].shape[-1]
    depth = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
    grid = img_size // patch
    reg = sd.get("register_tokens")
    prefix, pos_patch = dino_pos_tables(sd["pos_embed"], sd["cls_token"], reg, grid)
    kpad = round_up(3 * patch * patch, embed_k_multiple(prec))
    blocks = (_lib.BlockWeights * depth)()
    for i in range(depth):
        blocks[i] = _pack_block(pk, sd, f"blocks.{i}.", prec, device, ls=f"blocks.{i}.ls1.gamma" in sd, qk_norm=False)
    w = _lib.DinoWeights()
    w.depth, w.dim, w.heads, w.n_prefix = depth, dim, heads, prefix.shape[0]
    w.grid, w.patch, w.kpad = grid, patch, kpad
    w.ln_eps = 1e-6                                              # vision_transformer.py:95
    pk.register("patch_embed", sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], None, kpad)
    w.pos_patch = pk.keep(pos_patch, device)
    w.prefix_tokens = pk.keep(prefix, device)
    w.norm_w = pk.keep(sd["norm.weight"].float(), device)
    w.norm_b = pk.keep(sd["norm.bias"].float(), device)
    w.blocks = C.cast(blocks, C.POINTER(_lib.BlockWeights))
    pk.struct, pk.blocks = w, blocks
    pk.set_promote()
    return pk


def pack_betr(sd: dict, prec, device, heads: int, patch: int = 14, img_size: int = 224, box_dim: int = 8) -> Packed:
    """BETR state_dict (reference key names, no 'decoder.' prefix) -> bd_betr_weights."""
    pk = Packed(prec, device)
    dim = sd["bbox_learnable_query"].shape[-1]
    depth = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("attn."))
    grid = img_size // patch
    kpad = round_up(patch * patch * box_dim, embed_k_multiple(prec))
    blocks = (_lib.BlockWeights * depth)()
    for i in range(depth):
        blocks[i] = _pack_block(pk, sd, f"attn.{i}.", prec, device, ls=False, qk_norm=True)
    w = _lib.BetrWeights()
    w.depth, w.dim, w.heads, w.grid, w.patch, w.box_dim, w.kpad = depth, dim, heads, grid, patch, box_dim, kpad
    w.ln_eps = 1e-5              # get_layernorm ignores its eps argument: blocks.py:805
    w.adapter_ln_eps = 1e-6      # betr.py:161
    w.rms_eps = 1e-6             # blocks.py:45
    pk.register("adapter_fc1", sd["input_transform.fc1.weight"], sd["input_transform.fc1.bias"])
    pk.register("adapter_fc2", sd["input_transform.fc2.weight"], sd["input_transform.fc2.bias"])
    pk.register("bbox_emb", sd["bbox_emb.weight"], sd["bbox_emb.bias"], None, kpad)
    pk.register("bbox_proj", sd["bbox_proj.weight"], sd["bbox_proj.bias"])
    w.pos_table = pk.keep(sincos_table(dim, grid), device)
    pk.named["pos_table"] = pk.tensors[-1]
    w.query_token = pk.keep(sd["bbox_learnable_query"].float().reshape(-1), device)
    w.blocks = C.cast(blocks, C.POINTER(_lib.BlockWeights))
    pk.struct, pk.blocks = w, blocks
    pk.set_promote()
    return pk
