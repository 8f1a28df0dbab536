"""Explicit hand-off of the encoder's operand-dtype feature copy to the decoder.

`bd_encoder_forward` writes the patch features twice: fp32 (what the reference's `predict` returns,
/root/reference/src/models/modules/encoder/dinov2.py:45-60) and in the GEMM operand format of the active precision mode
(what `bd_decoder_forward` consumes).  The operand copy travels with the fp32 tensor OBJECT: `attach` ties it to that
object's lifetime, `operand_of` returns it only when the object still describes the same data, and `carry` moves it onto
another tensor object that aliases the same storage (a `.view()` / `.reshape()` / `.contiguous()` no-op).  Anything else
-- a copy, a slice, features computed elsewhere -- has NO operand copy: `operand_of` returns None and the decoder re-casts
explicitly (counted in `BETR.recast_count`, warned about once), never silently on stale data.
"""
from __future__ import annotations

import torch

_ATTR = "_bd_operand"


def _version(t: torch.Tensor):
    """In-place-modification stamp of `t`.  Tensors created under `torch.inference_mode()` (Lightning's test / validate /
    predict loops run there by default) do not track a version counter -- reading `_version` raises -- and cannot be
    modified in place outside inference mode; for them the identity check is storage address + element count only."""
    return None if t.is_inference() else t._version


def attach(feats32: torch.Tensor, feats16: torch.Tensor, pid: int, stamp=None) -> torch.Tensor:
    """`stamp`: what produced the features beyond the operand class -- the encoder's per-Linear promotion state (calibrate.py): two
    feature sets of the same class computed under different promotion states are NOT interchangeable (cache.py checks it)."""
    setattr(feats32, _ATTR, (feats16, int(pid), feats32.data_ptr(), feats32.numel(), _version(feats32), stamp))
    return feats32


def operand_of(feats32: torch.Tensor, pid: int):
    """The operand copy for precision id `pid`, or None if `feats32` carries none / was modified in place since."""
    tag = getattr(feats32, _ATTR, None)
    if tag is None:
        return None
    f16, tpid, ptr, numel, version = tag[:5]
    if tpid != int(pid) or ptr != feats32.data_ptr() or numel != feats32.numel() or version != _version(feats32):
        return None
    return f16


def tag_of(feats32: torch.Tensor):
    """(operand copy, precision id) or None -- for the reference-feature cache, which re-packs both copies."""
    tag = getattr(feats32, _ATTR, None)
    if tag is None or tag[2] != feats32.data_ptr() or tag[4] != _version(feats32):
        return None
    return tag[0], tag[1]


def stamp_of(feats32: torch.Tensor):
    """The producer stamp `attach` recorded (None: none recorded / no tag)."""
    tag = getattr(feats32, _ATTR, None)
    return tag[5] if tag is not None and len(tag) > 5 else None


def carry(src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    """Move src's operand copy onto dst when dst aliases exactly the same elements (same storage start, same count)."""
    tag = getattr(src, _ATTR, None)
    if tag is not None and dst is not src and dst.data_ptr() == tag[2] and dst.numel() == tag[3] and dst.is_contiguous():
        setattr(dst, _ATTR, (tag[0], tag[1], tag[2], tag[3], _version(dst), tag[5] if len(tag) > 5 else None))
    return dst
