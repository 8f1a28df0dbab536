"""Import the REAL reference modules (build container only -- /root/reference never travels).

TEST INFRASTRUCTURE.  Used only by `oracle/make_golden.py` to pin the oracle and to
generate the committed fixtures under `tests/golden/`.  Nothing in `-m gpu` tests,
`smoke()` or `bench.py` imports this file.

The reference's hot-path modules import a few packages this image lacks; none of them
contributes arithmetic on the path, so they are satisfied with `sys.modules` stand-ins
(SURVEY.md §8c / Appendix A):
  * timm 1.0.15 `Mlp` / `DropPath` (blocks.py:28-29) -- `Mlp` restated from timm's published
    definition: fc1 -> act -> drop1 -> norm(Identity) -> fc2 -> drop2, bias=True.
  * pytorch3d `HarmonicEmbedding` (pos_encodiong.py:107), import-only.
  * `src.models.sources.vggsfm[.models]` registered as bare packages so betr.py:7-8 loads
    `modules.py` without running `vggsfm/models/__init__.py` (hydra / pycolmap).
  * empty `cv2` so box_utils imports (the heatmap decode branch never calls cv2).
"""
from __future__ import annotations

import os
import sys
import types
import warnings

import torch.nn as nn

REF = os.environ.get("BOXDREAMER_REFERENCE", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REF, "src", "models"))


def _mk(name, path=None):
    m = types.ModuleType(name)
    if path is not None:
        m.__path__ = [path]
    sys.modules[name] = m
    return m


class _TimmMlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU,
                 norm_layer=None, bias=True, drop=0.0, use_conv=False):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features, bias=bias)
        self.act = act_layer()
        self.drop1 = nn.Dropout(drop)
        self.norm = nn.Identity()
        self.fc2 = nn.Linear(hidden_features, out_features, bias=bias)
        self.drop2 = nn.Dropout(drop)

    def forward(self, x):
        return self.drop2(self.fc2(self.norm(self.drop1(self.act(self.fc1(x))))))


class _DropPath(nn.Module):
    def __init__(self, p=0.0):
        super().__init__()

    def forward(self, x):
        return x


_loaded = None


def load():
    """Returns (BETR, vit_base, recover_bb8_corners, get_2d_sincos_pos_embed) from the reference."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError(f"reference tree not found at {REF}")
    if REF not in sys.path:
        sys.path.insert(0, REF)
    _mk("timm", "x"); _mk("timm.models", "x")
    _mk("timm.models.vision_transformer").Mlp = _TimmMlp
    _mk("timm.layers").DropPath = _DropPath
    _mk("pytorch3d", "x"); _mk("pytorch3d.renderer", "x")
    _mk("pytorch3d.renderer.implicit").HarmonicEmbedding = object
    _mk("src.models.sources.vggsfm", REF + "/src/models/sources/vggsfm")
    _mk("src.models.sources.vggsfm.models", REF + "/src/models/sources/vggsfm/models")
    if "cv2" not in sys.modules:
        _mk("cv2")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from src.models.modules.backbone.betr import BETR
        from src.models.modules.backbone.utils.pos_encodiong import get_2d_sincos_pos_embed
        from src.models.sources.DINOv2.vision_transformer import vit_base
        from src.models.utils.box_utils import recover_bb8_corners
    _loaded = (BETR, vit_base, recover_bb8_corners, get_2d_sincos_pos_embed)
    return _loaded


def build_betr(depth=12, d_model=768, nhead=8, patch=14, img=224):
    BETR = load()[0]
    return BETR(d_model=d_model, nhead=nhead, num_decoder_layers=depth, decoder_only=True,
                patch_size=patch, img_size=img, diff_emb=False, nvs_supervision=False,
                ray_supervision=True, use_mask=False,
                use_pretrained=True, patchify_rays=True,          # config_utils.py:84-87
                pose_representation="bb8", bbox_representation="heatmap").eval()


def build_dino(depth=12):
    """== hub dinov2_vitb14_reg constructor args."""
    vit_base = load()[1]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = vit_base(patch_size=14, img_size=518, init_values=1.0, ffn_layer="mlp", block_chunks=0,
                     num_register_tokens=4, interpolate_antialias=True, interpolate_offset=0.0)
    if depth != 12:                      # reduced-depth fixtures: keep the first `depth` blocks
        m.blocks = m.blocks[:depth]
    return m.eval()


def load_make_bbox_features():
    """The dataset-side `make_bbox_features` (src/datasets/utils/base/bbox_utils.py:215-303).  Its module imports cv2 /
    PIL / a sibling `preprocess` module at import time only; the heatmap branch is pure torch."""
    load()
    for pkg, path in (("src.datasets", REF + "/src/datasets"), ("src.datasets.utils", REF + "/src/datasets/utils"),
                      ("src.datasets.utils.base", REF + "/src/datasets/utils/base")):
        if pkg not in sys.modules:
            _mk(pkg, path)
    if "src.datasets.utils.preprocess" not in sys.modules:
        _mk("src.datasets.utils.preprocess").generate_cornernet_heatmap = None
    import importlib
    return importlib.import_module("src.datasets.utils.base.bbox_utils").make_bbox_features
