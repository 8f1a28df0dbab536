"""Model-config validation, mirroring /root/reference/src/models/utils/config_utils.py:10-96
(same asserts, same keys injected into the decoder config)."""
from __future__ import annotations

import copy


def validate_model_config(config):
    config = copy.copy(config)
    assert config["pose_representation"] in ["plucker", "vector", "bb8"]
    assert config["bbox_representation"] in ["heatmap", "voting", "cornernet"]
    if config["bbox_representation"] in ["cornernet"]:          # treated as heatmap (config_utils.py:27-28)
        config["bbox_representation"] = "heatmap"
    assert config["coordinate"] in ["first_camera", "object"]
    if config["use_rgb"] and config["encoder"]["name"] == "dino":
        assert config["decoder"]["patch_size"] == 14, "Dinov2 only supports patch size 14"
    if config["use_rgb"] and config["encoder"]["name"] == "spa":
        assert config["decoder"]["patch_size"] == 16, "SPA only supports patch size 16"
    assert (config["patchify_rays"] and config["use_rgb"]) or (
        not config["patchify_rays"] and not config["use_rgb"]
    ), "patchify_rays should be True when use_rgb is True"
    return config


def setup_camera_params(config):
    """config_utils.py:44-96: rotation/camera dims and the keys BETR's ctor reads."""
    rotation_length = 0
    camera_dim = 0
    if config["rotation_type"] is not None:
        assert config["rotation_type"] in ["quat", "6d", "euler", "so3", "ray"]
        rotation_length = {"6d": 6, "quat": 4}.get(config["rotation_type"], 3)
        if config["regression_intri"]:
            camera_dim = rotation_length + 3 + 1 + (2 if config["use_pp"] else 0)
        else:
            camera_dim = rotation_length + 3
    else:
        assert config["pose_representation"] == "bb8"
    dec = dict(config["decoder"])
    dec["rotation_type"] = config["rotation_type"]
    dec["camera_dim"] = camera_dim
    dec["rotation_length"] = rotation_length
    dec["use_pretrained"] = config["use_rgb"]
    dec["patchify_rays"] = config["patchify_rays"]
    dec["pose_representation"] = config["pose_representation"]
    dec["bbox_representation"] = config["bbox_representation"]
    if config["use_rgb"] and dec.get("diff_emb"):
        dec["diff_emb"] = False                                  # config_utils.py:90-94
    config = dict(config)
    config["decoder"] = dec
    return config, camera_dim, rotation_length
