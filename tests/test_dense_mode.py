"""Dense-reference mode: reference selection by DINO patch similarity + batch re-packing.

CPU part: the restatement (oracle/dense_oracle.py) against the fixtures written from the REAL reference functions
(oracle/make_golden_dense.py), and the host mirror's pure-torch helpers against the restatement.
GPU part: the HIP scoring / top-k kernels against the float64 closed form and the fixtures."""
import os

import numpy as np
import pytest
import torch

from boxdreamer_amd import dense
from oracle import dense_oracle as do
from oracle.make_golden_dense import dense_inputs

CASES = {"a": (5, 2, 9, 4), "b": (6, 1, 17, 5)}


def _split(feats, rgb, mask):
    B, T = mask.shape
    return (feats[~mask].reshape(B, T - 1, *feats.shape[2:]), feats[mask], rgb[~mask].reshape(B, T - 1, *rgb.shape[2:]),
            rgb[mask])


@pytest.mark.parametrize("name", ["a", "b"])
def test_oracle_matches_reference_fixture(golden_dir, name):
    g = np.load(os.path.join(golden_dir, "dense_vectors.npz"))
    seed, B, T, k = CASES[name]
    assert list(g[f"{name}_meta"]) == [seed, B, T, k]
    feats, rgb, mask = dense_inputs(seed, B, T)
    rf, qf, ri, qi = _split(feats, rgb, mask)
    scores = do.dino_matching_scores(rf, qf, ri, qi)
    # same fp32 evaluation order as the reference; scores are O(1e4) so compare at that scale
    assert np.abs(scores.numpy() - g[f"{name}_scores_ref_order_fp32"]).max() <= 2e-2
    assert np.array_equal(do.topk_mask(scores, k).numpy(), g[f"{name}_topk_mask"])      # the REFERENCE's selection
    exact = do.dino_matching_scores_closed_form(rf, qf, ri, qi)
    assert np.abs(exact.numpy() - g[f"{name}_scores_exact"]).max() <= 1e-6
    assert np.abs(exact.numpy() - g[f"{name}_scores_ref_order_fp32"]).max() <= 5e-3     # = the reference's rounding noise


@pytest.mark.parametrize("name", ["a", "b"])
def test_host_repacking_matches_restatement(golden_dir, name):
    g = np.load(os.path.join(golden_dir, "dense_vectors.npz"))
    seed, B, T, k = CASES[name]
    feats, rgb, mask = dense_inputs(seed, B, T)
    nm = torch.from_numpy(g[f"{name}_topk_mask"])
    bbox = torch.randn(B, T, 8, 16, 16)
    poses = torch.randn(B, T, 4, 4)
    data = {"poses": poses.clone(), "bbox_3d": torch.randn(B, T, 8, 3)}
    keep_b3 = data["bbox_3d"].clone()
    d2, pf, fr, cm, rf, im = dense.filter_by_neighbor_mask(data, nm, bbox, rgb, mask, feats, None)
    assert torch.equal(pf, do.filter_views(bbox, mask, nm)) and torch.equal(rf, do.filter_views(feats, mask, nm))
    assert torch.equal(d2["poses"], do.filter_views(poses, mask, nm)) and torch.equal(d2["bbox_3d"], do.filter_views(keep_b3, mask, nm))
    assert cm[:, -1].all() and int(cm.sum()) == B and torch.equal(d2["query_idx"], torch.full((B,), k))
    chk = g[f"{name}_filtered_feat_checksum"]
    assert abs(float(rf.double().sum()) - chk[0]) <= 1e-6 * chk[1]                      # reference's filtered features
    sb = dense.sub_batchify(bbox, rgb, mask, feats, None, 3)
    assert list(sb[0].shape) == list(g[f"{name}_subbatch_shape"])
    assert torch.equal(sb[0], do.sub_batchify_views(bbox, mask, 3)) and torch.equal(sb[3], do.sub_batchify_views(feats, mask, 3))
    assert sb[2][:, :, 3].all() and not sb[2][:, :, :3].any()
    gp = torch.from_numpy(g[f"{name}_poses"])
    pred = gp[:, 2:3].clone(); pred[:, :, :3, 3] += 0.01
    idx = dense.fetch_neighbors_by_pose_similarity(gp, pred, topk=3)
    assert np.array_equal(idx.numpy(), g[f"{name}_pose_neighbors"])                     # the REFERENCE's neighbours


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["a", "b"])
@pytest.mark.parametrize("img_dtype", [torch.float32, torch.bfloat16])
def test_hip_selection(hip, golden_dir, name, img_dtype):
    g = np.load(os.path.join(golden_dir, "dense_vectors.npz"))
    seed, B, T, k = CASES[name]
    feats, rgb, mask = dense_inputs(seed, B, T)
    rgb_in = rgb.to(img_dtype)
    q = mask.to(torch.int32).argmax(1)
    scores, nm = dense.match_views(feats.cuda(), rgb_in.cuda(), q.cuda(), k)
    rf, qf, ri, qi = _split(feats, rgb_in.float(), mask)
    exact = do.dino_matching_scores_closed_form(rf, qf, ri, qi)
    # fp32 ulp at |score| ~ 8e3 is 1e-3: a few ulps
    assert (scores.cpu().double() - exact).abs().max().item() <= 4e-3
    if img_dtype == torch.float32:
        assert np.abs(scores.cpu().numpy() - g[f"{name}_scores_exact"]).max() <= 4e-3
    # a valid top-k of the exact scores: k selected, none beaten by an unselected score by more than the tolerance
    sel = nm.cpu()
    assert (sel.sum(1) == k).all()
    for b in range(B):
        lo = exact[b][sel[b]].min().item()
        hi = exact[b][~sel[b]].max().item() if (~sel[b]).any() else -1e30
        assert lo >= hi - 8e-3
    # the signature-compatible wrapper (refs + query passed separately)
    nm2 = dense.dino_matching(rf.cuda(), qf.cuda(), ri.cuda(), qi.cuda(), topk=k)
    ex2 = exact
    for b in range(B):
        s2 = nm2.cpu()[b]
        assert int(s2.sum()) == k and ex2[b][s2].min().item() >= (ex2[b][~s2].max().item() if (~s2).any() else -1e30) - 8e-3


@pytest.mark.gpu
def test_hip_topk_mask_ties(hip):
    from boxdreamer_amd import _lib
    lib = _lib.load()
    s = torch.tensor([[1.0, 3.0, 3.0, -2.0, 3.0, 0.5], [0.0, 0.0, 0.0, 0.0, 0.0, 0.0]], device="cuda")
    m = torch.empty((2, 6), dtype=torch.uint8, device="cuda")
    _lib.check(lib.bd_topk_mask(_lib.ptr(s), 2, 6, 2, _lib.ptr(m), _lib.stream()), "bd_topk_mask")
    assert m.cpu().tolist() == [[0, 1, 1, 0, 0, 0], [1, 1, 0, 0, 0, 0]]                # ties: lower index first
