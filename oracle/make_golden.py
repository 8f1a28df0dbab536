"""Generate tests/golden/* from the REAL reference and pin the oracle against it.

TEST INFRASTRUCTURE -- runs only in the build container (needs /root/reference):

    python -m oracle.make_golden

For every case: build the reference modules (oracle/ref_import.py), load the seeded
weights from boxdreamer_amd.synth with strict=True (pins every state_dict key and shape),
run the reference forward, run the oracle restatement on the same tensors, assert
max-abs agreement <= TOL, and store the REFERENCE's outputs as fixtures.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from boxdreamer_amd import synth                     # noqa: E402
from oracle import boxdreamer_oracle as orc          # noqa: E402
from oracle import ref_import                        # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
TOL = 2e-5

MEAN = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
STD = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)


def summary(t: torch.Tensor, n: int = 256) -> dict:
    f = t.detach().float().reshape(-1)
    step = max(1, f.numel() // n)
    return {"mean": float(f.double().mean()), "absmax": float(f.abs().max()),
            "samples": f[::step][:n].numpy().astype(np.float32)}


def case_weights(kind: str, dino_depth: int, betr_depth: int):
    """Seeded weights of a fixture case: "plain"; "rescaled" = synth.rescale_function_preserving (operand-range stress, same function);
    "outliers_g<gain>" = synth.*_state_dict_outliers (trained-like statistics, function changes)."""
    if kind.startswith("outliers_g"):
        g = float(kind[len("outliers_g"):])
        return synth.betr_state_dict_outliers(1234, betr_depth, g), synth.dino_state_dict_outliers(4321, dino_depth, g)
    bsd = synth.betr_state_dict(seed=1234, depth=betr_depth)
    dsd = synth.dino_state_dict(seed=4321, depth=dino_depth)
    if kind == "rescaled":
        dsd, bsd = synth.rescale_function_preserving(dsd, bsd)
    else:
        assert kind == "plain", kind
    return bsd, dsd


def run_case(name, B, T, dino_depth, betr_depth, seed, full_logits, weights="plain", tol=TOL):
    torch.set_grad_enabled(False)
    bsd, dsd = case_weights(weights, dino_depth, betr_depth)
    betr = ref_import.build_betr(depth=betr_depth)
    dino = ref_import.build_dino(depth=dino_depth)
    betr.load_state_dict(bsd, strict=True)
    dino.load_state_dict(dsd, strict=True)
    data = synth.make_batch(seed=seed, B=B, T=T)
    mask = torch.zeros(B, T, dtype=torch.bool)
    mask[torch.arange(B), data["query_idx"]] = True

    # ---- reference forward (encoder/dinov2.py:45-60 then betr.py:249-308)
    x = (data["images"].flatten(0, 1) - MEAN) / STD
    ff = dino.forward_features(x)
    feats = ff["x_norm_patchtokens"].view(B, T, 256, 768)
    hooked = {}
    h = betr.bbox_proj.register_forward_hook(lambda m, i, o: hooked.__setitem__("proj", o.detach()))
    heat = betr(data["bbox_feat"], data["images"], mask, feats, None)
    h.remove()
    logits = betr.unpatchify(hooked["proj"], c=8)
    recover = ref_import.load()[2]
    norm_kp, kp = recover(heat.permute(0, 2, 3, 1).unsqueeze(1), "heatmap")
    norm_kp, kp = norm_kp[:, 0], kp[:, 0]
    ref_idx = torch.topk(((heat + 1) / 2).reshape(B, 8, -1), k=20, dim=2)[1].sort(-1)[0]

    # ---- oracle restatement on the same tensors
    o = orc.boxdreamer_forward(data, bsd, dsd)
    errs = {
        "rgb_feat": float((o["rgb_feat"] - feats).abs().max()),
        "logits": float((o["logits"] - logits).abs().max()),
        "heat": float((o["heat"] - heat).abs().max()),
        "corners_px": float((o["corners_px"] - kp).abs().max()),
        "corners_norm": float((o["corners_norm"] - norm_kp).abs().max()),
    }
    same_sets = bool((o["topk_idx"].sort(-1)[0] == ref_idx).all())
    print(f"[{name}] oracle-vs-reference max-abs: {errs}  top20 sets equal: {same_sets}")
    fscale = max(1.0, float(feats.abs().max()) / 8.0)       # (rescaled / outlier weights: features in the hundreds)
    for k, v in errs.items():
        lim = tol if k not in ("corners_px",) else max(1e-4, tol)
        if k == "rgb_feat":
            lim = tol * fscale
        assert v <= lim or (k.startswith("corners") and not same_sets), (name, k, v)
    assert errs["logits"] <= tol and errs["rgb_feat"] <= tol * fscale

    out = {
        "meta": np.array(json.dumps({"B": B, "T": T, "dino_depth": dino_depth, "betr_depth": betr_depth,
                                     "input_seed": seed, "betr_seed": 1234, "dino_seed": 4321, "weights": weights,
                                     "oracle_vs_reference": errs, "top20_sets_equal": same_sets})),
        "corners_px": kp.numpy(), "corners_norm": norm_kp.numpy(),
        "topk_idx_sorted": ref_idx.numpy().astype(np.int32),
        "logits_strided": logits.reshape(B, -1)[:, ::7].numpy(),
        "rgb_feat_strided": feats.reshape(B, -1)[:, ::97].numpy(),
    }
    for k, v in (("rgb_feat", feats), ("logits", logits), ("heat", heat)):
        s = summary(v)
        out[k + "_mean"] = np.float64(s["mean"]); out[k + "_absmax"] = np.float64(s["absmax"])
    if full_logits:
        out["logits"] = logits.numpy()
    np.savez_compressed(os.path.join(GOLD, f"case_{name}.npz"), **out)
    return betr, dino


def unit_vectors(betr, dino):
    """Per-kernel known answers taken from the reference's own modules/functions."""
    out = {}
    # sincos table (pos_encodiong.py:125-213 as used by betr.py:357-364)
    g2d = ref_import.load()[3]
    pe = g2d(768, grid_size=(16, 16), device="cpu").permute(0, 2, 3, 1).reshape(256, 768)
    assert float((orc.sincos_pos_embed(768, 16) - pe).abs().max()) <= 1e-6
    out["sincos_256x768_strided"] = pe.reshape(-1)[::61].numpy()
    out["sincos_absmax_sum"] = np.float64(pe.double().abs().sum())
    # DINO pos-embed resample 37x37 -> 16x16 (vision_transformer.py:179-211)
    dsd = synth.dino_state_dict(seed=4321, depth=1)
    x = torch.zeros(1, 257, 768)
    ref_pe = dino.interpolate_pos_encoding(x, 224, 224)
    assert float((orc.dino_pos_embed(dsd, 16) - ref_pe).abs().max()) <= 1e-6
    out["dino_pos_257x768_strided"] = ref_pe.reshape(-1)[::53].numpy()
    # block-level vectors: one BETR SelfAttentionBlock and its pieces on a seeded input
    xin = torch.from_numpy(synth.bell_np("unit.x", (2, 64, 768), 1.0, 0.0, 99).astype(np.float32))
    blk = betr.attn[0]
    st = lambda t: t.reshape(-1)[::5].numpy()          # strided: enough to pin the oracle
    out["betr_norm1"] = st(blk.norm1(xin))
    q = torch.from_numpy(synth.bell_np("unit.q", (2, 8, 64, 96), 1.5, 0.1, 99).astype(np.float32))
    out["betr_qnorm"] = st(blk.attn.q_norm(q))
    out["betr_attn"] = st(blk.attn(blk.norm1(xin)))
    out["betr_block0"] = st(blk(xin))
    out["gelu"] = st(torch.nn.functional.gelu(xin[0, :4]))
    dblk = dino.blocks[0]
    xd = torch.from_numpy(synth.bell_np("unit.xd", (2, 261, 768), 1.0, 0.0, 99).astype(np.float32))
    out["dino_attn"] = st(dblk.attn(dblk.norm1(xd)))
    out["dino_block0"] = st(dblk(xd))
    # patchify index map (betr.py:225-227): feature f of token t reads pixel (c, y, x)
    probe = torch.arange(8 * 224 * 224, dtype=torch.float32).reshape(1, 8, 224, 224)
    pm = betr.patchify(probe, c=8)[0].to(torch.int32)
    assert torch.equal(orc.patchify(probe, 14, 8)[0].to(torch.int32), pm)
    out["patchify_index_map_tok37"] = pm[37].numpy()
    out["patchify_index_map_tok255"] = pm[255].numpy()
    # decode on adversarial (but tie-unambiguous) heatmaps
    recover = ref_import.load()[2]
    hm = torch.full((2, 8, 224, 224), -1.0)
    u = torch.from_numpy(synth.uniform_np("unit.decode", (2, 8, 224 * 224), -1.0, 0.5, 5).astype(np.float32))
    hm = u.reshape(2, 8, 224, 224).clone()
    for b in range(2):
        for c in range(8):
            # a plateau of exactly 20 saturated pixels (ties inside the set only), and for c>=4 a
            # plateau of 7 plus 13 distinct runners-up
            ys = 10 + 13 * c + b
            n = 20 if c < 4 else 7
            hm[b, c, ys, 30:30 + n] = 1.0
            if c >= 4:
                hm[b, c, ys + 1, 100:113] = torch.linspace(0.9, 0.6, 13)
    nk, kp = recover(hm.permute(0, 2, 3, 1).unsqueeze(1), "heatmap")
    onk, okp, _ = orc.recover_bb8_corners(hm)
    assert float((okp - kp[:, 0]).abs().max()) <= 1e-4
    out["decode_heat_seed"] = np.int64(5)
    out["decode_kp"] = kp[:, 0].numpy()
    out["decode_norm_kp"] = nk[:, 0].numpy()
    # make_bbox_features (dataset-side producer of bbox_feat; "next" row f2) -- the real reference function
    mbf = ref_import.load_make_bbox_features()
    corners = torch.from_numpy(synth.uniform_np("unit.corners", (3, 8, 2), 20.0, 204.0, 17).astype(np.float32))
    corners[1, 2] = torch.tensor([100.0, 57.0])                   # a corner exactly on a pixel centre
    corners[2, 5] = torch.tensor([-3.25, 230.5])                  # and one outside the crop
    ref_h = mbf(corners, type="heatmap", shape=(224, 224))
    assert float((orc.make_bbox_features(corners, (224, 224)) - ref_h).abs().max()) <= 1e-6
    out["bbox_features_strided"] = ref_h.reshape(-1)[::11].numpy()
    out["bbox_features_max_per_map"] = ref_h.reshape(3, 8, -1).max(-1)[0].numpy()
    np.savez_compressed(os.path.join(GOLD, "unit_vectors.npz"), **out)


def robustness_cases():
    """Round 3: the REAL reference on weights whose operand ranges stress the 16-bit / e4m3 classes."""
    # same function as `full_T2` (power-of-two gains moved between a producer and its only consumer), operands up to ~1000
    run_case("rescaled_T2", 1, 2, 12, 12, 7, False, weights="rescaled")
    # trained-like outliers at half gain: logits rms 1.9 / max 9.5; the reference's and the oracle's fp32 forwards differ by their own
    # re-association noise there, so the restatement is pinned at 5e-4 instead of 2e-5
    run_case("outliers_g0.5_T2", 1, 2, 12, 12, 11, False, weights="outliers_g0.5", tol=5e-4)
    # round 4: the same grafts at a quarter and at three quarters of the gain
    run_case("outliers_g0.25_T2", 1, 2, 12, 12, 11, False, weights="outliers_g0.25", tol=5e-4)
    run_case("outliers_g0.75_T2", 1, 2, 12, 12, 11, False, weights="outliers_g0.75", tol=1e-3)


def main():
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    assert ref_import.available(), "reference tree required"
    if "--only-robustness" in sys.argv:
        robustness_cases()
        return
    # key/shape manifest of the reference modules (pins the drop-in state_dict contract)
    betr = ref_import.build_betr(12)
    dino = ref_import.build_dino(12)
    manifest = {"betr": {k: list(v.shape) for k, v in betr.state_dict().items()},
                "dino": {k: list(v.shape) for k, v in dino.state_dict().items()}}
    with open(os.path.join(GOLD, "state_dict_manifest.json"), "w") as f:
        json.dump(manifest, f, indent=0, sort_keys=True)
    # cases: name, B, T, dino_depth, betr_depth, input seed, store full logits
    run_case("tiny_d2_T2", 1, 2, 2, 2, 7, True)
    run_case("tiny_d2_T3_B2", 2, 3, 2, 2, 8, False)
    b, d = run_case("full_T2", 1, 2, 12, 12, 7, True)
    unit_vectors(b, d)
    run_case("full_T6", 1, 6, 12, 12, 11, False)
    robustness_cases()
    print("golden fixtures written to", GOLD)


if __name__ == "__main__":
    main()
