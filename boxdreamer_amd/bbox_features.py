"""GPU rendering of the reference views' corner heatmaps ("next" row f2 of SURVEY.md §8).

Mirror of `make_bbox_features` (/root/reference/src/datasets/utils/base/bbox_utils.py:215-303, heatmap branch), which the
dataset runs on the CPU in DataLoader workers and ships as (T, 8, 224, 224) tensors (4.8 MB of H2D per pose).  With this
entry point the caller ships only the (T, 8, 2) projected corners.
"""
from __future__ import annotations

import torch

from . import _lib


def make_bbox_features(bbox: torch.Tensor, type: str = "heatmap", shape=(64, 64), dtype: torch.dtype = torch.float32,
                       group: int | None = None) -> torch.Tensor:
    """bbox: (B, 8, 2) pixel (x, y) on the HIP device -> (B, 8, H, W) in [-1, 1].

    Like the reference, the per-corner normalisation max spans ALL B views of the call; pass `group` to render several
    samples in one launch (B = n_groups * group, max taken inside each run of `group` views)."""
    if type not in ("heatmap",):
        raise NotImplementedError("only the 'heatmap' representation is on the MI355X path")
    lib = _lib.load()
    _lib.require_gpu()
    B = bbox.shape[0]
    group = B if group is None else group
    if B % group:
        raise ValueError("B must be a multiple of group")
    H, W = shape
    c = bbox.reshape(B, 8, 2).to(torch.float32).contiguous()
    out = torch.empty((B, 8, H, W), dtype=dtype, device=c.device)
    _lib.check(lib.bd_render_corner_heatmaps(_lib.ptr(c), B // group, group, H, W, _lib.ptr(out), _lib.dtype_id(out),
                                             _lib.stream()), "bd_render_corner_heatmaps")
    return out
