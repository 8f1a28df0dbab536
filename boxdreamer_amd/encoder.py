"""Encoder plugin: the reference's `PretrainedModelWrapper` / `DinoV2Wrapper` surface
(/root/reference/src/models/modules/encoder/base.py:3-13, encoder/dinov2.py:6-60) backed by the
HIP DINOv2 forward (`bd_encoder_forward`).

Differences that are deliberate:
  * weights are never fetched with `torch.hub` (no network; and the hub model's xformers path is the
    CUDA dependency being replaced).  `ckpt_path` is a local state_dict file (.pth/.pt/.safetensors)
    with the hub key names, or cfg['state_dict'] / cfg['synthetic_seed'] supply tensors directly;
  * `predict` returns fp32 features with the operand-dtype copy the decoder consumes attached (features.py).
"""
from __future__ import annotations

import os
import weakref

import torch

from . import _lib, features, pack, synth

_ARCH = {  # model_type -> (dim, depth, heads)
    "dinov2_vits14_reg": (384, 12, 6),
    "dinov2_vitb14_reg": (768, 12, 12),
    "dinov2_vitl14_reg": (1024, 24, 16),
}


class PretrainedModelWrapper:
    """Mirror of encoder/base.py:3-13 (a plain object, not an nn.Module: encoder weights are
    outside the checkpoint)."""

    def __init__(self, model_name_or_path):
        self.model_name_or_path = model_name_or_path
        self.model = None

    def load_model(self, device="cuda"):
        raise NotImplementedError("Subclasses should implement this method")

    def predict(self, input_tensor):
        raise NotImplementedError


class DinoV2Encoder:
    """DINOv2 ViT-*/14 + 4 registers, inference only, on the HIP library."""

    def __init__(self, state_dict: dict, heads: int, patch: int = 14, prec=_lib.DEFAULT_PREC):
        # own copies: synth's seeded state dicts are cached per process and `.float()` of an fp32 tensor shares its storage
        self.sd = {k: v.detach().float().clone() for k, v in state_dict.items()}
        self.heads, self.patch, self.prec = heads, patch, prec
        self.device = torch.device("cpu")
        self._packed = {}       # (device, operand class, img_size) -> pack.Packed
        self._ws = None
        self._frozen_by = weakref.WeakSet()  # live GraphedPaths that captured raw pointers into _packed / _ws (graph.py)
        depth = 1 + max(int(k.split(".")[1]) for k in self.sd if k.startswith("blocks."))
        # per-Linear promotion (F16C8 family -> split-f16, e4m3 -> bf16) (include/boxdreamer_hip.h: BD_PROMOTE_*), set by calibrate.py
        self.lanes = "auto"          # sub-batch lanes of one predict() call ("auto" | 1..4; bit-identical results, _lib.resolve_lanes)
        self.latency = False         # opt-in latency forms (bd_dino_weights.latency_mode; see BETR.hip_latency)
        self.promote = [0] * depth
        self.promote_misc = 0
        self.feats_prec = 0          # 0: feats16 in the class of `prec`; the promoted class when the decoder's adapter fc1 is promoted
        if prec == "fp8_mixed":      # (the decoder's mixed policy keeps its adapter in bf16: hand the features over in bf16)
            self.promote, self.promote_misc = _lib.fp8_mixed_policy(depth, normed=False)
            self.feats_prec = _lib.PREC_BF16

    def _check_not_frozen(self, what: str):
        if len(self._frozen_by):
            raise RuntimeError(f"{what} would free memory a live GraphedPath still replays on; delete the graph first")

    def to(self, device):
        device = torch.device(device)
        if device != self.device:
            self._check_not_frozen("moving the encoder")
            self._packed.clear()
        self.device = device
        return self

    def _workspace(self, need: int, dev) -> torch.Tensor:
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._check_not_frozen("growing the encoder workspace (more images than the captured batch)")
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
        return self._ws

    def eval(self):
        return self

    def parameters(self):
        return iter(())          # frozen: nothing to hand to an optimiser

    def _weights(self, size: int, prec) -> pack.Packed:
        key = (str(self.device), _lib.operand_prec(prec), size)
        if key not in self._packed:
            self._packed[key] = pack.pack_dino(self.sd, _lib.operand_prec(prec), self.device, self.heads, self.patch, size)
        pk = self._packed[key]
        if _lib.operand_prec(prec) in (_lib.PREC_F16C8, _lib.PREC_FP8):
            want = (tuple(m | _lib.PROMOTE_FC2 if m & _lib.PROMOTE_FC1 else m for m in self.promote), self.promote_misc, self.feats_prec)
            if pk.promote != want:
                self._check_not_frozen("changing the per-Linear promotion")
                pk.set_promote(self.promote, self.promote_misc, self.feats_prec)
        return pk

    def state_stamp(self, prec=None):
        """What, beyond the operand class, decides the bits of the features: the per-Linear promotion state (calibrate.py moves Linears
        between operand classes at load time).  Travels with every feature tensor `predict` returns (features.attach)."""
        cls = _lib.operand_prec(self.prec if prec is None else prec)
        if cls not in (_lib.PREC_F16C8, _lib.PREC_FP8):
            return (cls,)
        return (cls, tuple(m | _lib.PROMOTE_FC2 if m & _lib.PROMOTE_FC1 else m for m in self.promote), int(self.promote_misc), int(self.feats_prec))

    def feats_class(self, prec=None) -> int:
        """Operand class of the 16-bit feature copy `patch_tokens` hands to the decoder."""
        cls = _lib.operand_prec(self.prec if prec is None else prec)
        return self.feats_prec if (cls in (_lib.PREC_F16C8, _lib.PREC_FP8) and self.feats_prec) else cls

    @torch.no_grad()
    def patch_tokens(self, images: torch.Tensor, prec=None):
        """images (N, 3, S, S) in [0, 1] (bf16/fp16/fp32, NCHW) -> (feats32 (N, P, C), feats16)."""
        _lib.require_gpu()
        lib = _lib.load()
        prec = self.prec if prec is None else prec
        if images.device != self.device:
            self.to(images.device)
        images = images.contiguous()
        n, c, size, size2 = images.shape
        if c != 3 or size != size2 or size % self.patch:
            raise ValueError(f"expected (N,3,S,S) with S % {self.patch} == 0, got {tuple(images.shape)}")
        pk = self._weights(size, prec)
        w = pk.struct
        if w.latency_mode != int(self.latency):
            if len(self._frozen_by):
                raise RuntimeError("switching hip_latency would change the workspace layout a live GraphedPath still replays on; delete the graph first")
            w.latency_mode = int(self.latency)
        P, D = w.grid * w.grid, w.dim
        lanes = _lib.resolve_lanes(self.lanes, n, n, prec)
        # sized for the laned AND the plain form: switching `lanes` under a live captured graph must never grow the workspace
        ws = self._workspace(max(lib.bd_encoder_workspace_bytes(w, n, _lib.prec_id(prec)),
                                 lib.bd_encoder_workspace_bytes_lanes(w, n, _lib.prec_id(prec), lanes)), images.device)
        feats32 = torch.empty((n, P, D), dtype=torch.float32, device=images.device)
        fcls = self.feats_class(prec)
        np_ = _lib.planes(fcls)
        feats16 = torch.empty((np_, n * P, D) if np_ == 2 else (n * P, D), dtype=_lib.op_dtype(fcls),
                              device=images.device)
        _lib.check(lib.bd_encoder_forward_lanes(w, _lib.ptr(images), _lib.dtype_id(images), n, size, _lib.ptr(feats32),
                                                _lib.ptr(feats16), n * P * D if np_ == 2 else 0, _lib.ptr(ws),
                                                ws.numel(), _lib.prec_id(prec), lanes, _lib.stream()), "bd_encoder_forward_lanes")
        return feats32, feats16


def _load_state_dict_file(path: str) -> dict:
    if path.endswith((".safetensors", ".safetensor")):
        from safetensors.torch import load_file
        return load_file(path)
    sd = torch.load(path, map_location="cpu")
    return sd.get("state_dict", sd) if isinstance(sd, dict) else sd


class DinoV2Wrapper(PretrainedModelWrapper):
    """Same constructor / API as encoder/dinov2.py:6-60."""

    def __init__(self, ckpt_path=None, cfg=None):
        super().__init__(model_name_or_path="dinov2")
        cfg = cfg or {}
        self.ckpt_path = ckpt_path
        self.model_type = cfg.get("model_type", "dinov2_vits14_reg")
        assert self.model_type in ["dinov2_vits14_reg", "dinov2_vitb14_reg", "dinov2_vitl14_reg",
                                   "dinov2_vitg14_reg"]
        if self.model_type not in _ARCH:
            raise NotImplementedError("dinov2_vitg14_reg (SwiGLU FFN) is outside the MI355X hot path")
        self.freeze = cfg.get("freeze", True)
        self.cfg = cfg
        self.device = None
        self.prec = cfg.get("hip_precision", os.environ.get("BOXDREAMER_HIP_PREC", _lib.DEFAULT_PREC))
        self.load_model()

    def get_device(self):
        return self.device

    def to_device(self, device):
        self.model = self.model.to(device)
        self.device = device

    def load_model(self, device="cuda"):
        dim, depth, heads = _ARCH[self.model_type]
        path = self.ckpt_path or os.environ.get("BOXDREAMER_DINO_WEIGHTS")
        if "state_dict" in self.cfg:
            sd = self.cfg["state_dict"]
        elif path is not None and os.path.isfile(path):
            print(f"Loading model from {path}")
            sd = _load_state_dict_file(path)
        elif "synthetic_seed" in self.cfg:
            sd = synth.dino_state_dict(seed=int(self.cfg["synthetic_seed"]), depth=self.cfg.get("depth", depth),
                                       dim=dim, nheads=heads)
        else:
            raise FileNotFoundError(
                "DINOv2 weights: give encoder.dino.ckpt_path (or $BOXDREAMER_DINO_WEIGHTS) pointing at a local "
                f"{self.model_type} state_dict file; torch.hub download is not available on this path")
        self.model = DinoV2Encoder(sd, heads=heads, prec=self.prec)
        self.model.lanes = self.cfg.get("hip_lanes", "auto")
        self.model.latency = bool(self.cfg.get("hip_latency", False))
        if torch.cuda.is_available():
            self.model.to(device)
            self.device = torch.device(device) if not isinstance(device, torch.device) else device
        else:
            self.device = torch.device("cpu")

    def predict(self, input_tensor):
        """(B,T,3,H,W) or (N,3,H,W) in [0,1] -> x_norm_patchtokens (B,T,P,C) / (N,P,C), fp32."""
        flag = False
        if input_tensor.dim() == 5:
            B, T = input_tensor.shape[:2]
            input_tensor = input_tensor.flatten(0, 1)
            flag = True
        with torch.no_grad():
            feats32, feats16 = self.model.patch_tokens(input_tensor, self.prec)
            ret = feats32.view(B, T, *feats32.shape[1:]) if flag else feats32
            return features.attach(ret, feats16, self.model.feats_class(self.prec), self.model.state_stamp(self.prec))   # explicit hand-off to BETR (features.py)
