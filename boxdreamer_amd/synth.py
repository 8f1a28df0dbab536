"""Deterministic synthetic weights and inputs for the corner-heatmap path.

Everything here is a pure function of (seed, tensor name, element index): a 64-bit
integer mix (splitmix64 finaliser) turned into floats with exact arithmetic only
(no log/cos), so this container and the GPU box regenerate bit-identical tensors
without shipping 350 MB of weights.  `torch.manual_seed` streams are NOT used.

Shapes / key names follow the reference checkpoints:
  * BETR decoder   -- /root/reference/src/models/modules/backbone/betr.py:131-176
  * DINOv2 ViT-B/14 reg4 -- /root/reference/src/models/sources/DINOv2/vision_transformer.py:106-168
Weight scales follow SURVEY.md §8(d) "Config 2" (trunc-normal-like 0.02 linears, small
biases, LN weights near 1) so that softmax / LayerNorm are not degenerate.
"""
from __future__ import annotations

import functools

import numpy as np
import torch

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a64(name: str) -> np.uint64:
    h = 0xCBF29CE484222325
    for b in name.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return np.uint64(h)


def _mix(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on a uint64 array (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    return x


def _bits(name: str, n: int, seed: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        base = _mix(np.array([np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15) + _fnv1a64(name)],
                             dtype=np.uint64))[0]
        idx = np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95) + base
    return _mix(idx)


def uniform_np(name: str, shape, lo: float = 0.0, hi: float = 1.0, seed: int = 0) -> np.ndarray:
    """U[lo, hi) with 24-bit resolution, float64 result (exact)."""
    n = int(np.prod(shape)) if len(shape) else 1
    u = (_bits(name, n, seed) >> np.uint64(40)).astype(np.float64) * (1.0 / 16777216.0)
    return (lo + (hi - lo) * u).reshape(shape)


def bell_np(name: str, shape, std: float = 1.0, mean: float = 0.0, seed: int = 0) -> np.ndarray:
    """Bell-shaped, unit-variance, bounded to +-3.46 sigma (sum of four 16-bit uniforms).

    Stands in for the reference's trunc_normal_(std=0.02)
    (/root/reference/src/models/sources/DINOv2/vision_transformer.py:332-337) with exact,
    platform-independent arithmetic."""
    n = int(np.prod(shape)) if len(shape) else 1
    b = _bits(name, n, seed)
    m16 = np.uint64(0xFFFF)
    s = ((b & m16) + ((b >> np.uint64(16)) & m16) + ((b >> np.uint64(32)) & m16)
         + ((b >> np.uint64(48)) & m16)).astype(np.float64)
    z = (s - 2.0 * 65535.0) * (1.7320508075688772 / 65536.0)
    return (mean + std * z).reshape(shape)


def _t(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a.astype(np.float32)))


# --------------------------------------------------------------------------- weights

def betr_state_dict(seed: int = 1234, depth: int = 12, d_model: int = 768, nhead: int = 8,
                    patch: int = 14, box_dim: int = 8) -> dict:
    """Random BETR decoder weights keyed exactly like the reference state_dict.  (A fresh dict per call over memoised tensors --
    generating 88 M seeded values takes seconds; callers replace entries, never write into a tensor in place.)"""
    return dict(_betr_state_dict(int(seed), int(depth), int(d_model), int(nhead), int(patch), int(box_dim)))


@functools.lru_cache(maxsize=4)
def _betr_state_dict(seed, depth, d_model, nhead, patch, box_dim) -> dict:
    hd = d_model // nhead
    pf = patch * patch * box_dim
    sd = {}

    def lin(prefix, out_f, in_f, wstd=0.02):
        sd[prefix + ".weight"] = _t(bell_np(prefix + ".weight", (out_f, in_f), wstd, 0.0, seed))
        sd[prefix + ".bias"] = _t(bell_np(prefix + ".bias", (out_f,), 0.01, 0.0, seed))

    def norm(prefix, n, bias=True):
        sd[prefix + ".weight"] = _t(bell_np(prefix + ".weight", (n,), 0.02, 1.0, seed))
        if bias:
            sd[prefix + ".bias"] = _t(bell_np(prefix + ".bias", (n,), 0.01, 0.0, seed))

    sd["bbox_learnable_query"] = _t(bell_np("bbox_learnable_query", (1, d_model), 0.02, 0.0, seed))
    for i in range(depth):
        p = f"attn.{i}"
        norm(p + ".norm1", d_model)
        lin(p + ".attn.qkv", 3 * d_model, d_model)
        norm(p + ".attn.q_norm", hd, bias=False)
        norm(p + ".attn.k_norm", hd, bias=False)
        lin(p + ".attn.proj", d_model, d_model)
        norm(p + ".norm2", d_model)
        lin(p + ".mlp.fc1", 4 * d_model, d_model)
        lin(p + ".mlp.fc2", d_model, 4 * d_model)
    lin("bbox_proj", pf, d_model)
    lin("input_transform.fc1", d_model, d_model)
    lin("input_transform.fc2", d_model, d_model)
    lin("bbox_emb", d_model, pf)
    return sd


def dino_state_dict(seed: int = 4321, depth: int = 12, dim: int = 768, nheads: int = 12,
                    patch: int = 14, img_size: int = 518, nreg: int = 4) -> dict:
    """Random DINOv2 ViT weights keyed like hub `dinov2_vitb14_reg` (vendored copy's names).  (Memoised like betr_state_dict.)"""
    return dict(_dino_state_dict(int(seed), int(depth), int(dim), int(nheads), int(patch), int(img_size), int(nreg)))


@functools.lru_cache(maxsize=4)
def _dino_state_dict(seed, depth, dim, nheads, patch, img_size, nreg) -> dict:
    sd = {}
    npos = (img_size // patch) ** 2 + 1

    def lin(prefix, out_f, in_f, wstd=0.02):
        sd[prefix + ".weight"] = _t(bell_np("dino." + prefix + ".weight", (out_f, in_f), wstd, 0.0, seed))
        sd[prefix + ".bias"] = _t(bell_np("dino." + prefix + ".bias", (out_f,), 0.01, 0.0, seed))

    def norm(prefix, n):
        sd[prefix + ".weight"] = _t(bell_np("dino." + prefix + ".weight", (n,), 0.02, 1.0, seed))
        sd[prefix + ".bias"] = _t(bell_np("dino." + prefix + ".bias", (n,), 0.01, 0.0, seed))

    sd["cls_token"] = _t(bell_np("dino.cls_token", (1, 1, dim), 0.02, 0.0, seed))
    sd["pos_embed"] = _t(bell_np("dino.pos_embed", (1, npos, dim), 0.02, 0.0, seed))
    sd["register_tokens"] = _t(bell_np("dino.register_tokens", (1, nreg, dim), 0.02, 0.0, seed))
    sd["mask_token"] = torch.zeros(1, dim)
    sd["patch_embed.proj.weight"] = _t(bell_np("dino.patch_embed.proj.weight", (dim, 3, patch, patch),
                                               0.02, 0.0, seed))
    sd["patch_embed.proj.bias"] = _t(bell_np("dino.patch_embed.proj.bias", (dim,), 0.01, 0.0, seed))
    for i in range(depth):
        p = f"blocks.{i}"
        norm(p + ".norm1", dim)
        # wider q/k so that 64-dim unnormalised attention logits have O(1) spread
        lin(p + ".attn.qkv", 3 * dim, dim, wstd=0.05)
        lin(p + ".attn.proj", dim, dim)
        sd[p + ".ls1.gamma"] = _t(bell_np("dino." + p + ".ls1.gamma", (dim,), 0.05, 1.0, seed))
        norm(p + ".norm2", dim)
        lin(p + ".mlp.fc1", 4 * dim, dim)
        lin(p + ".mlp.fc2", dim, 4 * dim)
        sd[p + ".ls2.gamma"] = _t(bell_np("dino." + p + ".ls2.gamma", (dim,), 0.05, 1.0, seed))
    norm("norm", dim)
    return sd


# --------------------------------------------------------------------------- "trained-like" statistics
# Random N(0, 0.02^2) weights keep every activation O(1-10).  Trained DINOv2 / BETR checkpoints do not: ViTs develop a few
# "massive activation" channels in the residual stream (hundreds, fed by single MLP hidden units with very large
# pre-activations), LayerNorm gains with outliers, and attention logits with a large dynamic range.  These variants graft
# such statistics onto the seeded weights (same key names, deterministic) so that the 16-bit / e4m3 operand classes of the
# strict mode can be checked against the fp32 oracle where their range limits (f16: 65504, e4m3: 448) actually bite.
# Reference modules whose operands are affected: src/models/sources/DINOv2/layers/block.py:89-114 (norm1 / norm2 -> attn / mlp),
# layers/mlp.py:34-40 (fc1 -> GELU -> fc2), src/models/modules/backbone/utils/blocks.py:243-302 (q / k RMSNorm gains).

OUTLIER_CHANNELS = (137, 481)          # residual-stream channels that carry the massive activations
OUTLIER_HIDDEN = (5, 77, 1999, 3000)   # MLP hidden units with very large pre-activations


def _scale_rows(sd, key, rows, f):
    w = sd[key].clone()
    w[list(rows)] *= f
    sd[key] = w


def dino_state_dict_outliers(seed: int = 4321, depth: int = 12, gain: float = 1.0, **kw) -> dict:
    """dino_state_dict + trained-like outliers (gain = 1: LayerNorm outputs reach ~800 and GELU outputs several hundred, i.e.
    beyond e4m3's 448; gain < 1 scales the grafts down)."""
    sd = dino_state_dict(seed, depth, **kw)
    g = float(gain)
    blk = lambda i: f"blocks.{min(i, depth - 1)}"
    # (1) two massive-activation channels: one early MLP writes hundreds into them
    _scale_rows(sd, blk(1) + ".mlp.fc2.weight", OUTLIER_CHANNELS, 60.0 * g)
    b = sd[blk(1) + ".mlp.fc2.bias"].clone(); b[list(OUTLIER_CHANNELS)] += 3.0 * g; sd[blk(1) + ".mlp.fc2.bias"] = b
    # (2) LayerNorm gains: most later norms damp the massive channels (as trained nets do), one does the opposite, and a few
    #     ordinary channels carry large gains
    for i in range(2, depth):
        for nm in (".norm1.weight", ".norm2.weight"):
            _scale_rows(sd, blk(i) + nm, OUTLIER_CHANNELS, 0.05)
    if depth > 4:
        _scale_rows(sd, blk(4) + ".norm2.weight", OUTLIER_CHANNELS[:1], 20.0 * 30.0 * g)       # undo the damping, then x30
        _scale_rows(sd, blk(7) + ".norm1.weight", (23, 300, 655), 30.0 * g)
    # (3) MLP hidden units with huge pre-activations (-> GELU outputs in the hundreds feeding fc2's A operand)
    _scale_rows(sd, blk(min(6, depth - 1)) + ".mlp.fc1.weight", OUTLIER_HIDDEN, 40.0 * g)
    _scale_rows(sd, blk(min(9, depth - 1)) + ".mlp.fc1.weight", OUTLIER_HIDDEN[:2], 100.0 * g)
    # (4) attention logits with a large dynamic range: a few q / k feature rows x8
    _scale_rows(sd, blk(min(3, depth - 1)) + ".attn.qkv.weight", (0, 1, 70, 768 + 0, 768 + 1, 768 + 70), 8.0 * g)
    return sd


def betr_state_dict_outliers(seed: int = 1234, depth: int = 12, gain: float = 1.0, **kw) -> dict:
    """betr_state_dict + trained-like outliers: LayerNorm gains, MLP hidden units, q / k RMSNorm gains, one massive channel."""
    sd = betr_state_dict(seed, depth, **kw)
    g = float(gain)
    blk = lambda i: f"attn.{min(i, depth - 1)}"
    _scale_rows(sd, blk(1) + ".mlp.fc2.weight", OUTLIER_CHANNELS[1:], 60.0 * g)
    for i in range(2, depth):
        for nm in (".norm1.weight", ".norm2.weight"):
            _scale_rows(sd, blk(i) + nm, OUTLIER_CHANNELS[1:], 0.05)
    if depth > 3:
        _scale_rows(sd, blk(3) + ".norm2.weight", OUTLIER_CHANNELS[1:], 20.0 * 30.0 * g)
        _scale_rows(sd, blk(min(5, depth - 1)) + ".norm1.weight", (11, 402), 30.0 * g)
    _scale_rows(sd, blk(min(2, depth - 1)) + ".mlp.fc1.weight", OUTLIER_HIDDEN, 40.0 * g)
    _scale_rows(sd, blk(min(8, depth - 1)) + ".mlp.fc1.weight", OUTLIER_HIDDEN[2:], 100.0 * g)
    _scale_rows(sd, blk(min(4, depth - 1)) + ".attn.q_norm.weight", (3, 40), 4.0 * g)
    _scale_rows(sd, blk(min(4, depth - 1)) + ".attn.k_norm.weight", (3, 40), 4.0 * g)
    _scale_rows(sd, "input_transform.fc1.weight", (9, 500), 40.0 * g)
    return sd


def rescale_function_preserving(dino_sd: dict, betr_sd: dict, s_norm: float = 256.0, s_v: float = 64.0, s_qk: float = 32.0):
    """Operand-RANGE stress that leaves the network's function unchanged: power-of-two gains moved between a producer and its only
    consumer, so every fp32 product is the same number scaled by a power of two and the fp32 forward is (bit for bit, barring
    under / overflow) the forward of the original weights -- while the 16-bit / e4m3 operands see LayerNorm outputs up to ~1000,
    weight columns down to ~1e-4 next to ordinary ones in the same tensor, attention values x64 and DINOv2 q / k features x32 / 32:
      * LayerNorm gain and shift of a few channels x s_norm, the consuming Linear's columns / s_norm   (norm1 -> qkv, norm2 -> fc1,
        DINOv2's final norm -> BETR's adapter fc1);
      * value features x s_v (qkv rows + bias), proj columns / s_v   (attention output is linear in v);
      * DINOv2 only (BETR RMS-normalises q, k): q feature x s_qk, the same k feature / s_qk  (logits unchanged).
    Returns new (dino_sd, betr_sd)."""
    d, b = {k: v.clone() for k, v in dino_sd.items()}, {k: v.clone() for k, v in betr_sd.items()}

    def chans(tag, n, count, i):
        return sorted({int(c) for c in (uniform_np(f"rescale.{tag}.{i}", (count,), 0.0, float(n), 97) // 1)})

    def norm_to_linear(sd, norm, lin, cs, f):
        sd[norm + ".weight"][cs] *= f
        if norm + ".bias" in sd:
            sd[norm + ".bias"][cs] *= f
        sd[lin + ".weight"][:, cs] /= f

    def stack(sd, prefix, depth, dim, qk: bool):
        for i in range(depth):
            p = f"{prefix}{i}"
            norm_to_linear(sd, p + ".norm1", p + ".attn.qkv", chans("n1" + prefix, dim, 6, i), s_norm)
            norm_to_linear(sd, p + ".norm2", p + ".mlp.fc1", chans("n2" + prefix, dim, 6, i), s_norm)
            vj = chans("v" + prefix, dim, 5, i)
            rows = [2 * dim + j for j in vj]
            sd[p + ".attn.qkv.weight"][rows] *= s_v
            sd[p + ".attn.qkv.bias"][rows] *= s_v
            sd[p + ".attn.proj.weight"][:, vj] /= s_v
            if qk:
                qj = chans("q" + prefix, dim, 5, i)
                sd[p + ".attn.qkv.weight"][qj] *= s_qk
                sd[p + ".attn.qkv.bias"][qj] *= s_qk
                kr = [dim + j for j in qj]
                sd[p + ".attn.qkv.weight"][kr] /= s_qk
                sd[p + ".attn.qkv.bias"][kr] /= s_qk

    dim = d["cls_token"].shape[-1]
    ddepth = 1 + max(int(k.split(".")[1]) for k in d if k.startswith("blocks."))
    bdepth = 1 + max(int(k.split(".")[1]) for k in b if k.startswith("attn."))
    stack(d, "blocks.", ddepth, dim, qk=True)
    stack(b, "attn.", bdepth, dim, qk=False)
    cn = chans("final", dim, 6, 0)
    d["norm.weight"][cn] *= s_norm
    d["norm.bias"][cn] *= s_norm
    b["input_transform.fc1.weight"][:, cn] /= s_norm
    return d, b


# --------------------------------------------------------------------------- inputs

def corner_heatmaps_np(corners: np.ndarray, size: int) -> np.ndarray:
    """Corner heatmaps in [-1, 1] as the dataset renders them.

    Restates the 'heatmap' branch of make_bbox_features
    (/root/reference/src/datasets/utils/base/bbox_utils.py:263-303): per corner
    exp(-dist / (dist_to_centroid/10)^2), divided by its max, mapped to [-1, 1].
    corners: (..., 8, 2) pixel (x, y).  Returns (..., 8, size, size) float64."""
    lead = corners.shape[:-2]
    c = corners.reshape(-1, 8, 2).astype(np.float64)
    ix = np.arange(size, dtype=np.float64)[None, None, None, :]
    iy = np.arange(size, dtype=np.float64)[None, None, :, None]
    center = c.mean(axis=1, keepdims=True)
    dx = c[:, :, 0][:, :, None, None] - ix
    dy = c[:, :, 1][:, :, None, None] - iy
    d = np.sqrt(dx * dx + dy * dy)
    dis = np.sqrt(((center - c) ** 2).sum(-1))
    scale = (dis / 10.0) ** 2
    h = np.exp(-d / scale[:, :, None, None])
    h = h / h.max(axis=(2, 3), keepdims=True)
    return (h * 2.0 - 1.0).reshape(*lead, 8, size, size)


def make_batch(seed: int = 7, B: int = 1, T: int = 2, size: int = 224,
               dtype: torch.dtype = torch.float32, quantize: torch.dtype | None = torch.bfloat16) -> dict:
    """A synthetic BoxDreamer batch dict with the keys `BoxDreamer.forward` reads
    (/root/reference/src/models/BoxDreamerModel.py:193-215, 268-270, 364).

    images: U[0,1) with a zeroed 16-px border band (the dataset masks background to 0);
    bbox_feat: rendered from 8 seeded corners in [m, size-m)^2 with m = 30 px at 224 (scaled with `size`: a fixed
    30-px margin is an EMPTY range at 56 px and rendered NaN maps there); query_idx = T-1.
    Values are rounded through `quantize` (the dataset casts every tensor to the run
    precision, /root/reference/src/datasets/base.py:715-752) and returned in `dtype`."""
    img = uniform_np("images", (B, T, 3, size, size), 0.0, 1.0, seed)
    band = 16
    img[..., :band, :] = 0.0
    img[..., -band:, :] = 0.0
    img[..., :, :band] = 0.0
    img[..., :, -band:] = 0.0
    margin = 30.0 * size / 224.0                      # == 30.0 exactly at the default size (golden checksums unchanged)
    corners = uniform_np("corners", (B, T, 8, 2), margin, float(size) - margin, seed)
    heat = corner_heatmaps_np(corners, size)
    if not np.isfinite(heat).all():
        raise ValueError(f"synth.make_batch(seed={seed}, size={size}): a corner coincides with the box centroid (0/0 in the renderer)")

    def q(a):
        t = torch.from_numpy(np.ascontiguousarray(a.astype(np.float32)))
        if quantize is not None:
            t = t.to(quantize).to(torch.float32)
        return t.to(dtype)

    eye = np.broadcast_to(np.eye(4), (B, T, 4, 4)).copy()
    eye[..., 2, 3] = 5.0
    f = 1.2 * size
    K = np.broadcast_to(np.array([[f, 0, size / 2.0], [0, f, size / 2.0], [0, 0, 1.0]]), (B, T, 3, 3)).copy()
    bb3 = uniform_np("bbox_3d", (B, 1, 8, 3), -0.5, 0.5, seed)
    bb3 = np.broadcast_to(bb3, (B, T, 8, 3)).copy()
    data = {
        "images": q(img),
        "bbox_feat": q(heat),
        "query_idx": torch.full((B,), T - 1, dtype=torch.long),
        "poses": q(eye),
        "non_ndc_intrinsics": q(K),
        "intrinsics": q(K),
        "crop_parameters": q(np.zeros((B, T, 4))),
        "image_masks": q((img.sum(2, keepdims=True) > 0).astype(np.float64)),
        "bbox_3d": q(bb3),
        "bbox_proj_crop": q(corners),
    }
    return data
