"""CPU restatement of the dense-reference mode helpers.  TEST INFRASTRUCTURE ONLY.

Plain PyTorch fp32, each function citing the reference lines it follows (paths relative to /root/reference).
Pinned against the real functions by oracle/make_golden_dense.py -> tests/golden/dense_vectors.npz.
Only tests/ may import this file; the product path (boxdreamer_amd/dense.py) is HIP.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def foreground_mask(images: torch.Tensor, L: int, threshold: float = 0.05) -> torch.Tensor:
    """src/models/utils/matching.py:91-106: luminance > threshold, nearest-resized to the sqrt(L) feature grid.
    images (V, 3, H, W) -> (V, L) float 0/1."""
    lum = 0.299 * images[:, 0] + 0.587 * images[:, 1] + 0.114 * images[:, 2]
    fg = (lum > threshold).float()
    g = int(math.sqrt(L))
    fg = F.interpolate(fg.unsqueeze(1), size=(g, g), mode="nearest")
    return fg.reshape(fg.shape[0], -1)


def dino_matching_scores(ref_features, query_features, ref_images, query_images):
    """matching.py:64-165 up to `mean_similarity`, in the reference's own order of operations (bmm of the masked,
    L2-normalised patch features; pairs without two foreground patches are filled with -1e4 and -- as in the reference,
    whose later `== -1e9` test never fires -- averaged in; valid_count is therefore always L*L).
    ref_features (B, N, L, D), query_features (B, L, D), ref_images (B, N, 3, H, W), query_images (B, 3, H, W)."""
    B, N, L, D = ref_features.shape
    rf = ref_features.reshape(B * N, L, D).float()
    qf = query_features.float().unsqueeze(1).expand(-1, N, -1, -1).reshape(B * N, L, D)
    qm = foreground_mask(query_images.float(), L).unsqueeze(1).expand(-1, N, -1).reshape(B * N, L)
    rm = foreground_mask(ref_images.reshape(B * N, *ref_images.shape[2:]).float(), L)
    qn = F.normalize(qf * qm.unsqueeze(-1), dim=-1)
    rn = F.normalize(rf * rm.unsqueeze(-1), dim=-1)
    sim = torch.bmm(qn, rn.transpose(-2, -1))
    valid = torch.bmm(qm.unsqueeze(-1), rm.unsqueeze(-1).transpose(-2, -1))
    sim = sim.masked_fill(valid == 0, -1e4)
    mean = sim.sum(dim=[1, 2]) / float(L * L)
    return torch.nan_to_num(mean.reshape(B, N), 0.0, 0.0, 0.0)


def dino_matching_scores_closed_form(ref_features, query_features, ref_images, query_images):
    """The same quantity without the L x L product, in float64 (what the HIP kernels compute in fp32):
        mean = ( s_q . s_r  -  1e4 * (L^2 - c_q c_r) ) / L^2,   s = sum of masked unit patch vectors, c = foreground count.
    The reference's fp32 sum of up to 65536 terms of magnitude 1e4 carries ~1e-3 of rounding noise; this form does not."""
    B, N, L, D = ref_features.shape
    def sums(feat, img):
        m = foreground_mask(img.float(), L).double()
        f = feat.double()
        n = f / f.norm(dim=-1, keepdim=True).clamp_min(1e-12)
        return (n * m.unsqueeze(-1)).sum(1), m.sum(1)
    sq, cq = sums(query_features, query_images)
    sr, cr = sums(ref_features.reshape(B * N, L, D), ref_images.reshape(B * N, *ref_images.shape[2:]))
    sr, cr = sr.reshape(B, N, D), cr.reshape(B, N)
    dot = (sr * sq.unsqueeze(1)).sum(-1)
    return (dot - 1e4 * (L * L - cq.unsqueeze(1) * cr)) / float(L * L)


def topk_mask(scores: torch.Tensor, k: int) -> torch.Tensor:
    """matching.py:167-173."""
    idx = torch.topk(scores, k=k, dim=-1)[1]
    m = torch.zeros_like(scores, dtype=torch.bool)
    m.scatter_(1, idx, True)
    return m


def filter_views(x: torch.Tensor, camera_mask: torch.Tensor, neighbor_mask: torch.Tensor) -> torch.Tensor:
    """src/models/utils/data_processing.py:9-75 for one per-view tensor x (B, T, ...): the selected references in their
    original order, then the query view last."""
    B, T = camera_mask.shape
    ref = x[~camera_mask].reshape(B, T - 1, *x.shape[2:])[neighbor_mask].reshape(B, -1, *x.shape[2:])
    return torch.cat([ref, x[camera_mask].unsqueeze(1)], dim=1)


def sub_batchify_views(x: torch.Tensor, camera_mask: torch.Tensor, sub: int) -> torch.Tensor:
    """src/models/utils/data_utils.py:5-94 for one per-view tensor: (B, T, ...) -> (B, ceil((T-1)/sub), sub+1, ...);
    each round holds `sub` consecutive references (zero-padded in the last round) and the query in the last slot."""
    B, T = camera_mask.shape
    q = x[camera_mask]
    ref = x[~camera_mask].reshape(B, T - 1, *x.shape[2:])
    rounds = (T - 1 + sub - 1) // sub
    out = torch.zeros(B, rounds, sub + 1, *x.shape[2:], dtype=torch.float32)
    for i in range(rounds):
        end = min((i + 1) * sub, T - 1)
        out[:, i, :end - i * sub] = ref[:, i * sub:end]
        out[:, i, sub] = q
    return out


def neighbors_by_pose_similarity(ref_poses: torch.Tensor, pred_pose: torch.Tensor, topk: int) -> torch.Tensor:
    """data_utils.py:97-135: geodesic rotation distance + translation distance, k smallest.
    ref_poses (B, N, 4, 4), pred_pose (B, 1, 4, 4) -> indices (B, topk)."""
    B, N = ref_poses.shape[:2]
    g = ref_poses.reshape(B * N, 4, 4)
    p = pred_pose.reshape(B, 1, 4, 4).repeat(1, N, 1, 1).reshape(B * N, 4, 4)
    Rd = torch.matmul(p[:, :3, :3], g[:, :3, :3].transpose(1, 2))
    tr = torch.diagonal(Rd, dim1=-2, dim2=-1).sum(-1)
    rot = torch.acos(torch.clamp((tr - 1) / 2, -1, 1))
    dist = (rot + torch.norm(p[:, :3, 3] - g[:, :3, 3], dim=-1)).reshape(B, N)
    return torch.topk(dist, k=topk, dim=1, largest=False)[1]
