"""Reference-feature caching across queries ("next" row f1 of SURVEY.md §8).

In the test / demo loops the SAME N posed reference crops accompany every query frame of an object, but the reference
re-encodes them for every query (/root/reference/src/models/BoxDreamerModel.py:274-285; SURVEY.md §7 "hard parts").
DINOv2 features are per-image and input-independent of the other views, so they can be computed once per object:
per pose the encoder then runs on 1 crop instead of T (T=6: 282 -> 47 GFLOP of the 638 GFLOP/pose).

Usage (caller side):
    cache = RefFeatureCache(model.rgb_encoder)
    ref_feats = cache.encode(ref_images)                      # (B, T-1, 3, H, W) -> tagged features, once per object
    data["cached_rgb_feat"], data["cached_rgb_mask"] = cache.place(ref_feats, query_idx, T)
    model(data)                                               # encodes only the views whose mask is False
Results are bit-identical to the uncached forward (tests/test_gpu_facade.py).

The cached features carry the encoder state they were computed under (operand class + per-Linear promotion state, features.stamp_of).
The facade's load-time calibration (model.py / calibrate.py) runs inside the FIRST forward and may promote encoder Linears: features cached
BEFORE that are stale.  `merge_cached_features` notices (stamp mismatch), warns once, and encodes every view of the batch afresh instead --
never a silent mix of two promotion states, never an exception in the middle of a sweep.  Call `model.calibrate(data)` (or one forward)
before `cache.encode` to keep the saving.
"""
from __future__ import annotations

import warnings

import torch

from . import _lib, features

_WARNED_STALE = False


def _new_operand(pid: int, n_views: int, P: int, C: int, dtype, dev) -> torch.Tensor:
    pid = _lib.operand_prec(pid)        # a whole-path id (f16c8_qk16, ...) carries the byte layout of its operand class
    rows = n_views * P
    return torch.zeros((2, rows, C) if _lib.planes(pid) == 2 else (rows, C), dtype=dtype, device=dev)


def _plane_views(t16: torch.Tensor, pid: int, n_views: int, P: int, C: int):
    """Per-view row views [n_views, P, C] of every plane of an operand tensor, in the plane's OWN element type.
    Split-bf16: two elementwise 16-bit planes.  F16C8 (include/boxdreamer_hip.h): plane 0 is f16, plane 1 is one e4m3 BYTE per
    element -- rows of C bytes packed into the first rows*C bytes of the plane's storage (the rest is unused) -- so it must be
    moved as uint8 rows, never as 16-bit rows."""
    pid = _lib.operand_prec(pid)
    rows = n_views * P
    if _lib.planes(pid) == 1:
        return [t16.reshape(n_views, P, C)]
    if pid == _lib.PREC_F16C8:
        lo8 = t16[1].view(torch.uint8).reshape(-1)[: rows * C].reshape(n_views, P, C)
        return [t16[0].reshape(n_views, P, C), lo8]
    return [t16[0].reshape(n_views, P, C), t16[1].reshape(n_views, P, C)]


def _init_last_stale():
    merge_cached_features.last_stale = False


def take_views(feats: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """feats (B, T, P, C) fp32 from the HIP encoder; idx (B, T') long, -1 = an all-zero view -> (B, T', P, C) with
    out[b, j] = feats[b, idx[b, j]], its operand-dtype copy re-packed the same way and attached (features.attach): what the dense-reference
    helpers (dense.py: view selection, sub-batches of references) need so that BETR does not re-cast on every forward.  Features that carry
    no operand copy come back plain (BETR then re-casts, as before)."""
    B, T, P, C = feats.shape
    dev = feats.device
    idx = idx.to(dev).long()
    Tn = idx.shape[1]
    valid = (idx >= 0).reshape(-1)
    flat = (torch.arange(B, device=dev)[:, None] * T + idx.clamp_min(0)).reshape(-1)
    out32 = feats.reshape(B * T, P, C)[flat]
    out32[~valid] = 0
    out32 = out32.reshape(B, Tn, P, C)
    tag = features.tag_of(feats)
    if tag is None:
        return out32
    f16, pid = tag
    out16 = _new_operand(pid, B * Tn, P, C, f16.dtype, dev)
    for dst, src in zip(_plane_views(out16, pid, B * Tn, P, C), _plane_views(f16, pid, B * T, P, C)):
        g = src[flat]
        g[~valid] = 0
        dst.copy_(g)
    return features.attach(out32, out16, pid, features.stamp_of(feats))


class RefFeatureCache:
    def __init__(self, encoder):
        self.encoder = encoder                                 # a DinoV2Wrapper

    def encode(self, images: torch.Tensor) -> torch.Tensor:
        """(B, R, 3, H, W) -> (B, R, P, C) fp32 features tagged with their operand-dtype copy."""
        return self.encoder.predict(images)

    @staticmethod
    def _feats16_view(feats: torch.Tensor):
        tag = features.tag_of(feats)
        if tag is None:
            raise ValueError("features were not produced by the HIP encoder (no operand-dtype copy attached)")
        return tag

    def place(self, ref_feats: torch.Tensor, query_idx: torch.Tensor, T: int):
        """Scatter R = T-1 cached reference features into a (B, T, P, C) layout leaving the query slot empty.
        Returns (features, valid_mask (B, T) bool)."""
        B, R, P, C = ref_feats.shape
        assert R == T - 1
        f16, pid = self._feats16_view(ref_feats)
        dev = ref_feats.device
        valid = torch.ones((B, T), dtype=torch.bool, device=dev)
        valid[torch.arange(B, device=dev), query_idx.to(dev).long()] = False
        full32 = torch.zeros((B, T, P, C), dtype=torch.float32, device=dev)
        full32[valid] = ref_feats.reshape(B * R, P, C)
        full16 = _new_operand(pid, B * T, P, C, f16.dtype, dev)
        for dst, src in zip(_plane_views(full16, pid, B * T, P, C), _plane_views(f16, pid, B * R, P, C)):
            dst.reshape(B, T, P, C)[valid] = src
        return features.attach(full32, full16, pid, features.stamp_of(ref_feats)), valid


def merge_cached_features(encoder, images: torch.Tensor, cached: torch.Tensor, valid: torch.Tensor) -> torch.Tensor:
    """Encode only the views with valid == False and write them into (a copy of) the cached layout.  `merge_cached_features.last_stale`
    says whether THIS call fell back to encoding every view (the facade records it per forward in data["hip_precision"]["cache_stale"]:
    the warning fires once per process, the fallback every time)."""
    B, T = images.shape[:2]
    f16, pid = RefFeatureCache._feats16_view(cached)
    stamp, now = features.stamp_of(cached), encoder.model.state_stamp(encoder.prec)
    # (a missing stamp -- an older producer, or a tag lost through .to() / .clone() -- counts as stale whenever the encoder's state carries
    # a promotion: such features cannot be told apart from ones computed under another state)
    promoted_now = any(getattr(encoder.model, "promote", ())) or bool(getattr(encoder.model, "promote_misc", 0))
    merge_cached_features.last_stale = bool((stamp is not None and stamp != now) or (stamp is None and promoted_now))
    if merge_cached_features.last_stale:
        global _WARNED_STALE
        if not _WARNED_STALE:
            _WARNED_STALE = True
            warnings.warn("BoxDreamer HIP path: the cached reference features were encoded under another precision / promotion state of the "
                          "encoder (the load-time calibration ran, or calibrate.set_state was applied, after RefFeatureCache.encode); "
                          "encoding every view afresh.  Re-encode the references after the first forward to keep the cache's saving.",
                          stacklevel=2)
        return encoder.predict(images)
    miss = ~valid
    new = encoder.predict(images[miss])                       # (n_miss, P, C), tagged
    n16, npid = RefFeatureCache._feats16_view(new)
    if npid != pid:
        raise ValueError("cached features were produced in a different precision mode")
    P, C = cached.shape[2:]
    out32 = cached.clone()
    out32[miss] = new
    out16 = f16.clone()
    n_miss = int(new.shape[0])
    for dst, src in zip(_plane_views(out16, pid, B * T, P, C), _plane_views(n16, pid, n_miss, P, C)):
        dst.reshape(B, T, P, C)[miss] = src
    return features.attach(out32, out16, pid, now)


_init_last_stale()
