"""CPU ORACLE for BoxDreamer's corner-heatmap inference path  (TEST INFRASTRUCTURE ONLY).

A plain-PyTorch fp32 restatement of the reference algorithm, keyed by the reference's
state_dict names.  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline`
leg may import this module -- the product path (`boxdreamer_amd/`) never does and fails
loudly when the HIP library is missing.

Parity status: PINNED.  `oracle/make_golden.py` imports the real reference modules
(`BETR`, vendored `DinoVisionTransformer`, `recover_bb8_corners`) in the build container,
runs them on seeded inputs/weights from `boxdreamer_amd.synth`, asserts this restatement
agrees to <= 2e-5 max-abs, and commits the reference's outputs under `tests/golden/`.
The reference holds no tests or golden vectors of its own for this path.
Un-pinned: OpenCV `solvePnP` (cv2 absent; host-side, out of the GPU path).

Every function cites the reference lines (relative to /root/reference) it follows.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

_IMAGENET_MEAN = (0.485, 0.456, 0.406)   # src/models/modules/encoder/dinov2.py:4-5
_IMAGENET_STD = (0.229, 0.224, 0.225)


# ----------------------------------------------------------------------------- DINOv2

def dino_pos_embed(sd: dict, grid: int, patch: int = 14) -> torch.Tensor:
    """interpolate_pos_encoding, src/models/sources/DINOv2/vision_transformer.py:179-211
    (hub variant: interpolate_offset=0.0 -> `size=` path, antialias=True, fp32).
    Returns (1, 1+grid*grid, C): class pos + resampled patch pos."""
    pe = sd["pos_embed"].float()
    n = pe.shape[1] - 1
    if n == grid * grid:
        return pe
    m = int(math.sqrt(n))
    assert m * m == n
    dim = pe.shape[-1]
    patch_pe = F.interpolate(pe[:, 1:].reshape(1, m, m, dim).permute(0, 3, 1, 2),
                             mode="bicubic", antialias=True, size=(grid, grid))
    patch_pe = patch_pe.permute(0, 2, 3, 1).reshape(1, -1, dim)
    return torch.cat((pe[:, :1], patch_pe), dim=1)


def dino_forward_features(sd: dict, x: torch.Tensor, nheads: int = 12, patch: int = 14,
                          return_stages: bool = False):
    """DinoVisionTransformer.forward_features -> x_norm_patchtokens.

    x: (N, 3, H, W) already ImageNet-normalised.  Follows
    vision_transformer.py:213-232 (prepare_tokens_with_masks), layers/patch_embed.py:68-81,
    layers/block.py:89-114 (x + ls1(attn(norm1 x)); x + ls2(mlp(norm2 x))),
    layers/attention.py:56-69 (naive softmax attention, scale hd^-0.5 on q),
    layers/mlp.py:34-40 (fc1 -> exact GELU -> fc2), layers/layer_scale.py:26-27,
    vision_transformer.py:254-270 (final LayerNorm eps 1e-6, drop cls + registers)."""
    stages = {}
    N, _, H, W = x.shape
    dim = sd["cls_token"].shape[-1]
    t = F.conv2d(x, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=patch)
    t = t.flatten(2).transpose(1, 2)                                   # (N, P, C)
    t = torch.cat((sd["cls_token"].expand(N, -1, -1), t), dim=1)
    t = t + dino_pos_embed(sd, H // patch, patch)
    nreg = sd["register_tokens"].shape[1]
    t = torch.cat((t[:, :1], sd["register_tokens"].expand(N, -1, -1), t[:, 1:]), dim=1)
    stages["tokens"] = t
    depth = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
    hd = dim // nheads
    for i in range(depth):
        p = f"blocks.{i}."
        h = F.layer_norm(t, (dim,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-6)
        L = h.shape[1]
        qkv = F.linear(h, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"])
        qkv = qkv.reshape(N, L, 3, nheads, hd).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0] * hd ** -0.5, qkv[1], qkv[2]
        a = (q @ k.transpose(-2, -1)).softmax(dim=-1)
        h = (a @ v).transpose(1, 2).reshape(N, L, dim)
        h = F.linear(h, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
        t = t + sd[p + "ls1.gamma"] * h
        h = F.layer_norm(t, (dim,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-6)
        h = F.gelu(F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
        h = F.linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
        t = t + sd[p + "ls2.gamma"] * h
        stages[f"block{i}"] = t
    t = F.layer_norm(t, (dim,), sd["norm.weight"], sd["norm.bias"], 1e-6)
    out = t[:, nreg + 1:]
    if return_stages:
        return out, stages
    return out


def encoder_predict(sd: dict, images: torch.Tensor, nheads: int = 12, patch: int = 14) -> torch.Tensor:
    """DinoV2Wrapper.predict, src/models/modules/encoder/dinov2.py:45-60:
    flatten (B,T), (x - mean)/std, forward_features()['x_norm_patchtokens'], view (B,T,P,C)."""
    B, T = images.shape[:2]
    x = images.float().flatten(0, 1)
    mean = torch.tensor(_IMAGENET_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(_IMAGENET_STD).view(1, 3, 1, 1)
    x = (x - mean) / std
    f = dino_forward_features(sd, x, nheads, patch)
    return f.view(B, T, *f.shape[1:])


# ----------------------------------------------------------------------------- BETR

def sincos_pos_embed(dim: int, grid: int) -> torch.Tensor:
    """get_2d_sincos_pos_embed, src/models/modules/backbone/utils/pos_encodiong.py:125-213,
    as consumed at betr.py:357-364.  Token t = i*grid + j (row i, col j) gets
    [sin(j w) | cos(j w) | sin(i w) | cos(i w)], w_d = 10000^(-d/(dim/4)), float64 -> fp32.
    Returns (grid*grid, dim)."""
    q = dim // 4
    omega = torch.arange(q, dtype=torch.float64) / q
    omega = 1.0 / 10000 ** omega
    ii, jj = torch.meshgrid(torch.arange(grid, dtype=torch.float64),
                            torch.arange(grid, dtype=torch.float64), indexing="ij")
    ow = jj.reshape(-1, 1) * omega[None]
    oh = ii.reshape(-1, 1) * omega[None]
    return torch.cat([ow.sin(), ow.cos(), oh.sin(), oh.cos()], dim=1).float()


def patchify(x: torch.Tensor, p: int, c: int) -> torch.Tensor:
    """BETR.patchify, betr.py:211-228: (N,c,H,W) -> (N, L, p*p*c), feature (pi*p+qi)*c + ci."""
    n, _, H, _ = x.shape
    h = H // p
    x = x.reshape(n, c, h, p, h, p)
    return torch.einsum("nchpwq->nhwpqc", x).reshape(n, h * h, p * p * c)


def unpatchify(x: torch.Tensor, p: int, c: int) -> torch.Tensor:
    """BETR.unpatchify, betr.py:230-247."""
    n, L, _ = x.shape
    h = int(L ** 0.5)
    x = x.reshape(n, h, h, p, p, c)
    return torch.einsum("nhwpqc->nchpwq", x).reshape(n, c, h * p, h * p)


def _rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """LlamaRMSNorm, src/models/modules/backbone/utils/blocks.py:44-56."""
    return w * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps))


def betr_block(sd: dict, p: str, x: torch.Tensor, nhead: int) -> torch.Tensor:
    """SelfAttentionBlock.forward, blocks.py:876-886, with Attention.forward, blocks.py:243-302
    (SDPA branch): LayerNorm eps 1e-5 (get_layernorm ignores its eps, blocks.py:805),
    q/k RMSNorm over head_dim after the head split, scale hd^-0.5, timm Mlp with exact GELU."""
    B, N, C = x.shape
    hd = C // nhead
    h = F.layer_norm(x, (C,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-5)
    qkv = F.linear(h, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"])
    qkv = qkv.view(B, N, 3, nhead, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.unbind(0)
    q = _rmsnorm(q, sd[p + "attn.q_norm.weight"])
    k = _rmsnorm(k, sd[p + "attn.k_norm.weight"])
    a = ((q * hd ** -0.5) @ k.transpose(-2, -1)).softmax(dim=-1)
    h = (a @ v).transpose(1, 2).reshape(B, N, C)
    h = F.linear(h, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
    x = x + h
    h = F.layer_norm(x, (C,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-5)
    h = F.gelu(F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
    h = F.linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    return x + h


def betr_forward(sd: dict, pose_feat: torch.Tensor, masks: torch.Tensor, rgb_feat: torch.Tensor,
                 nhead: int = 8, patch: int = 14, return_stages: bool = False):
    """BETR.forward, betr.py:249-308 (use_pretrained, bb8/heatmap).

    pose_feat (B,T,8,H,W) in [-1,1]; masks (B,T) bool one-hot on the query view;
    rgb_feat (B,T,P,C).  Returns (logits (B,8,H,W), heat = 2*sigmoid(logits)-1)."""
    stages = {}
    B, T, c, H, _ = pose_feat.shape
    C = rgb_feat.shape[-1]
    P = rgb_feat.shape[2]
    # betr.py:313-317 adapter: vggsfm Mlp (modules.py:156-162) then LayerNorm(no affine, eps 1e-6)
    r = rgb_feat.float().reshape(B * T, P, C)
    r = F.linear(F.gelu(F.linear(r, sd["input_transform.fc1.weight"], sd["input_transform.fc1.bias"])),
                 sd["input_transform.fc2.weight"], sd["input_transform.fc2.bias"])
    r = F.layer_norm(r, (C,), None, None, 1e-6).reshape(B, T, P, C)
    stages["rgb"] = r
    # betr.py:324-329 heatmap patch embedding
    pf = patchify(pose_feat.float().reshape(B * T, c, H, H), patch, c).reshape(B, T, P, -1)
    pf = F.linear(pf, sd["bbox_emb.weight"], sd["bbox_emb.bias"])
    # betr.py:286-290 query substitution (pose stream only)
    pf = pf.clone()
    pf[masks] = sd["bbox_learnable_query"].expand(B, P, C).to(pf.dtype)     # the reference casts too (betr.py:287-289)
    # betr.py:351-401 fuse + positional table
    g = int(P ** 0.5)
    x = pf + r + sincos_pos_embed(C, g).reshape(1, 1, P, C)
    stages["fuse"] = x
    x = x.reshape(B, T * P, C)
    depth = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("attn."))
    for i in range(depth):
        x = betr_block(sd, f"attn.{i}.", x, nhead)
        stages[f"block{i}"] = x
    x = x.reshape(B, T, P, C)
    qf = x[masks]                                                      # (B, P, C)  betr.py:303
    logits = unpatchify(F.linear(qf, sd["bbox_proj.weight"], sd["bbox_proj.bias"]), patch, c)
    heat = 2 * torch.sigmoid(logits) - 1                               # betr.py:432-435
    if return_stages:
        return logits, heat, stages
    return logits, heat


# ----------------------------------------------------------------------------- decode

def topk_lowest_index(v: torch.Tensor, k: int):
    """Top-k along the last dim with the tie rule this build pins: larger value first,
    then LOWER index first.  (`torch.topk` leaves tie order unspecified; the reference
    never pins it -- src/models/utils/box_utils.py:87.)  Stable descending sort does it."""
    vals, idx = torch.sort(v, dim=-1, descending=True, stable=True)
    return vals[..., :k], idx[..., :k]


def recover_bb8_corners(heat: torch.Tensor, k: int = 20):
    """Heatmap branch of recover_bb8_corners, src/models/utils/box_utils.py:75-110.

    heat: (B, 8, H, W) in [-1, 1] (the decoder output; the reference permutes it to
    (B,1,H,W,8) and back, prediction_utils.py:65 / box_utils.py:82).
    Returns (normalised (B,8,2), pixel (B,8,2), idx (B,8,k) int64)."""
    B, c, H, W = heat.shape
    h = ((heat.float() + 1) / 2).reshape(B, c, H * W)
    _, idx = topk_lowest_index(h, k)
    xs = (idx % W).float().mean(-1)
    ys = (idx // W).float().mean(-1)
    kp = torch.stack([xs, ys], dim=-1)
    norm = kp / torch.tensor([W, H], dtype=torch.float32) * 2 - 1
    return norm, kp, idx


# ----------------------------------------------------------------------------- input rendering ("next" row f2)

def make_bbox_features(bbox: torch.Tensor, size) -> torch.Tensor:
    """'heatmap' branch of make_bbox_features, src/datasets/utils/base/bbox_utils.py:263-303 (the dataset-side
    producer of `bbox_feat`, called at src/datasets/base.py:689-693): bbox (B,8,2) pixel (x,y) -> (B,8,H,W) fp32.
    Per corner: exp(-dist / (dist_to_centroid / 10)^2) / max, mapped to [-1, 1]; all torch fp32 like the reference."""
    H, W = size
    B = bbox.shape[0]
    bbox = bbox.float().view(B, 8, 2)
    out = torch.zeros((B, H, W, 8), dtype=torch.float32)
    ix = torch.arange(W, dtype=torch.float32)
    iy = torch.arange(H, dtype=torch.float32)
    center = bbox.mean(dim=1)
    for i in range(8):
        dx = bbox[:, i, 0].view(B, 1, 1).expand(B, H, W) - ix.view(1, 1, W).expand(B, H, W)
        dy = bbox[:, i, 1].view(B, 1, 1).expand(B, H, W) - iy.view(1, H, 1).expand(B, H, W)
        d = torch.sqrt(dx ** 2 + dy ** 2)
        dis = torch.sqrt((center[:, 0] - bbox[:, i, 0]) ** 2 + (center[:, 1] - bbox[:, i, 1]) ** 2)
        scale = (dis / 10) ** 2
        v = torch.exp(-d / scale.unsqueeze(-1).unsqueeze(-1))
        v = v / v.max()          # NOTE: the reference normalises by the max over the WHOLE batch (bbox_map[..., i].max())
        out[..., i] = v * 2 - 1
    return out.permute(0, 3, 1, 2)


# ----------------------------------------------------------------------------- facade

def boxdreamer_forward(data: dict, betr_sd: dict, dino_sd: dict, nhead: int = 8,
                       dino_heads: int = 12, patch: int = 14) -> dict:
    """GPU-able part of BoxDreamer.forward, src/models/BoxDreamerModel.py:112-191:
    camera_mask one-hot at query_idx (:193-215), encoder (:274-285), decoder (:326-333),
    pred_bbox write-back (:335-348), corner decode (prediction_utils.py:63-93 up to PnP)."""
    B, T = data["images"].shape[:2]
    mask = torch.zeros(B, T, dtype=torch.bool)
    mask[torch.arange(B), data["query_idx"]] = True
    feats = encoder_predict(dino_sd, data["images"], dino_heads, patch)
    logits, heat = betr_forward(betr_sd, data["bbox_feat"], mask, feats, nhead, patch)
    norm, kp, idx = recover_bb8_corners(heat)
    pred_bbox = data["bbox_feat"].float().clone()
    pred_bbox[mask] = heat.to(pred_bbox.dtype)
    return {"camera_mask": mask, "rgb_feat": feats, "logits": logits, "heat": heat,
            "pred_bbox": pred_bbox, "corners_px": kp, "corners_norm": norm, "topk_idx": idx}
