"""Resolve the reference's own YAML to the dict its model constructor receives  (TEST INFRASTRUCTURE; build container only).

    python -m oracle.make_model_config        # writes tests/golden/model_modules_config.json

What the harness does (Hydra): `configs/test.yaml` is the root (its scalars `image_size`, `patch_size`, `length`, `coordinate`,
`pose_representation`, `bbox_representation`, `precision`: /root/reference/configs/test.yaml:8-24), the group `model: transformer`
mounts /root/reference/configs/model/transformer.yaml under `model`, whose `modules` block (transformer.yaml:10-71) refers back to
the root scalars through `${...}` interpolations; `run.py:63` instantiates it with `_recursive_=False` and
`PL_BoxDreamer.__init__` hands `self.hparams` to `BoxDreamer(config)` (src/lightning/BoxDreamer_lightning_model.py:34), which
reads `config["modules"]` (src/models/BoxDreamerModel.py:41-69).  hydra / omegaconf are not in this image, so the two files are
read with PyYAML and the root-scalar interpolations resolved here -- the only interpolation kind the `modules` block uses (asserted).
The result is DATA: the resolved `modules` mapping plus the root scalars it was resolved against.  `tests/test_gpu_facade.py` and
`tests/test_host_logic.py` build the facade from this file instead of a hand-typed dict.
"""
from __future__ import annotations

import json
import os
import re

import yaml

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "model_modules_config.json")
_INTERP = re.compile(r"^\$\{([A-Za-z_][A-Za-z0-9_]*)\}$")


def resolve(node, root: dict):
    if isinstance(node, dict):
        return {k: resolve(v, root) for k, v in node.items()}
    if isinstance(node, list):
        return [resolve(v, root) for v in node]
    if isinstance(node, str) and "${" in node:
        m = _INTERP.match(node)
        assert m, f"unsupported interpolation in the modules block: {node!r}"
        assert m.group(1) in root, f"{node!r} names no root scalar of configs/test.yaml"
        return root[m.group(1)]
    return node


def main():
    root = yaml.safe_load(open(os.path.join(REF, "configs", "test.yaml")))
    assert {"model": "transformer"} in root["defaults"], "configs/test.yaml no longer mounts model: transformer"
    model = yaml.safe_load(open(os.path.join(REF, "configs", "model", "transformer.yaml")))
    scalars = {k: v for k, v in root.items() if not isinstance(v, (dict, list))}
    modules = resolve(model["modules"], scalars)
    out = {"source": {"root": "configs/test.yaml", "model_group": "configs/model/transformer.yaml:10-71",
                      "_target_": model["_target_"], "instantiate": "run.py:63 (_recursive_=False) -> BoxDreamer_lightning_model.py:34 -> BoxDreamerModel.py:41-69"},
           "root_scalars": {k: scalars[k] for k in ("precision", "image_size", "patch_size", "length", "coordinate", "pose_representation",
                                                     "bbox_representation", "use_pretrained", "mask_bg")},
           "modules": modules}
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(f"wrote {OUT}: modules keys {sorted(modules)}")


if __name__ == "__main__":
    main()
