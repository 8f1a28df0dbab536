"""Dense-reference mode: pick the best references out of a dense database view set, optionally in
several decoder rounds.  Host mirror of the reference's helpers, same names / arguments / dict side effects:

  dino_matching                     /root/reference/src/models/utils/matching.py:64-174
  filter_by_neighbor_mask           /root/reference/src/models/utils/data_processing.py:9-176
  process_dense_input               /root/reference/src/models/utils/data_processing.py:179-225
  sub_batchify                      /root/reference/src/models/utils/data_utils.py:5-94
  fetch_neighbors_by_pose_similarity /root/reference/src/models/utils/data_utils.py:97-135
  process_multi_round               /root/reference/src/models/utils/dense_processing.py:8-158

The arithmetic that scales with the database size -- the L x L patch-similarity of every (query, reference) pair -- runs in
HIP (csrc/match.hip: one pass over the patch features per view, closed form of the reference's masked mean) together with
the top-k selection; the decoder rounds reuse BETR unchanged.  View selection / re-packing of the batch tensors is torch
indexing on the device (plumbing).  Pose recovery inside the multi-round mode goes through the host RANSAC PnP (pnp.solve_pnp_ransac:
the reference's cv2.solvePnPRansac call where cv2 imports, a restatement of the scheme otherwise), whose parity against OpenCV is
un-pinned in this image.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib, pnp
from .box_utils import recover_bb8_corners_chw


def _get(cfg, key, default=None):
    try:
        return cfg[key]
    except (KeyError, AttributeError, TypeError):
        return getattr(cfg, key, default)


# ------------------------------------------------------------------------------------------- reference selection

def match_views(rgb_feature: torch.Tensor, frames: torch.Tensor, query_idx: torch.Tensor, topk: int,
                threshold: float = 0.05):
    """rgb_feature (B, T, L, D) fp32 encoder output, frames (B, T, 3, H, W) RGB in [0, 1], query_idx (B,) ->
    (scores (B, T-1) fp32, neighbor_mask (B, T-1) bool), references in view order (query view skipped)."""
    _lib.require_gpu()
    lib = _lib.load()
    B, T, L, D = rgb_feature.shape
    H, W = frames.shape[-2:]
    feats = rgb_feature.contiguous()
    if feats.dtype != torch.float32:
        feats = feats.float()
    frames = frames.contiguous()
    dev = feats.device
    q = query_idx.to(device=dev, dtype=torch.int32).contiguous()
    sums = torch.empty((B * T, D), dtype=torch.float32, device=dev)
    counts = torch.empty((B * T,), dtype=torch.float32, device=dev)
    scores = torch.empty((B, T - 1), dtype=torch.float32, device=dev)
    _lib.check(lib.bd_dino_match_scores(_lib.ptr(feats), _lib.ptr(frames), _lib.dtype_id(frames), _lib.ptr(q), B, T, L, D,
                                        H, W, float(threshold), _lib.ptr(sums), _lib.ptr(counts), _lib.ptr(scores),
                                        _lib.stream()), "bd_dino_match_scores")
    mask = torch.empty((B, T - 1), dtype=torch.uint8, device=dev)
    _lib.check(lib.bd_topk_mask(_lib.ptr(scores), B, T - 1, int(topk), _lib.ptr(mask), _lib.stream()), "bd_topk_mask")
    return scores, mask.bool()


def dino_matching(ref_features, query_features, ref_images, query_images, similarity_type="dot_product", topk=10,
                  similarity_params=None):
    """Same signature as the reference: ref_features (B, N, L, D), query_features (B, L, D), ref_images (B, N, 3, H, W),
    query_images (B, 3, H, W) -> bool top-k mask (B, N)."""
    if similarity_type != "dot_product":
        raise NotImplementedError("only the reference's default 'dot_product' similarity is implemented")
    B, N = ref_features.shape[:2]
    feats = torch.cat([ref_features, query_features.unsqueeze(1)], dim=1)
    frames = torch.cat([ref_images, query_images.unsqueeze(1)], dim=1)
    q = torch.full((B,), N, dtype=torch.int32, device=feats.device)
    return match_views(feats, frames, q, topk)[1]


def _filter(x, camera_mask, neighbor_mask):
    B, T = camera_mask.shape
    ref = x[~camera_mask].reshape(B, T - 1, *x.shape[2:])[neighbor_mask].reshape(B, -1, *x.shape[2:])
    return torch.cat([ref, x[camera_mask].unsqueeze(1)], dim=1)


def _filter_index(camera_mask, neighbor_mask):
    """The view indices `_filter` keeps, (B, k + 1): the selected references in their original order, the query last."""
    B, T = camera_mask.shape
    ar = torch.arange(T, device=camera_mask.device).expand(B, T)
    ref = ar[~camera_mask].reshape(B, T - 1)[neighbor_mask].reshape(B, -1)
    return torch.cat([ref, ar[camera_mask].reshape(B, 1)], dim=1)


def _filter_features(rgb_feature, camera_mask, neighbor_mask):
    """`_filter` for the encoder features: the same views, WITH their operand-dtype copy (cache.take_views) -- a plain index would drop
    it and BETR would re-cast the features on every forward (the dense path always ran that slow path)."""
    from .cache import take_views
    if rgb_feature.dim() != 4:
        return _filter(rgb_feature, camera_mask, neighbor_mask)
    return take_views(rgb_feature, _filter_index(camera_mask, neighbor_mask))


def _sub_index(camera_mask, sub):
    """View indices of `_sub`, (B, rounds, sub + 1): `sub` consecutive references per round (-1 where the last round runs out), query last."""
    B, T = camera_mask.shape
    ar = torch.arange(T, device=camera_mask.device).expand(B, T)
    ref, q = ar[~camera_mask].reshape(B, T - 1), ar[camera_mask].reshape(B)
    rounds = (T - 1 + sub - 1) // sub
    idx = torch.full((B, rounds, sub + 1), -1, dtype=torch.long, device=camera_mask.device)
    for i in range(rounds):
        end = min((i + 1) * sub, T - 1)
        idx[:, i, :end - i * sub] = ref[:, i * sub:end]
        idx[:, i, sub] = q
    return idx


def filter_by_neighbor_mask(data, neighbor_mask, pose_feat, frames, camera_mask, rgb_feature, image_masks):
    """Keep the selected references (original order) followed by the query view; updates the batch dict like the
    reference's update_filtered_data (bbox_feat, images, query_idx, camera_mask, poses and the per-view tensors)."""
    B = frames.shape[0]
    new_pose_feat = _filter(pose_feat, camera_mask, neighbor_mask)
    new_frames = _filter(frames, camera_mask, neighbor_mask)
    new_rgb = _filter_features(rgb_feature, camera_mask, neighbor_mask) if rgb_feature is not None else None
    new_masks = _filter(image_masks, camera_mask, neighbor_mask) if image_masks is not None else None
    T = new_frames.shape[1]
    new_camera_mask = torch.zeros(B, T, dtype=torch.bool, device=camera_mask.device)
    new_camera_mask[:, -1] = True
    data["bbox_feat"] = new_pose_feat.clone()
    data["images"] = new_frames.clone()
    data["query_idx"] = torch.tensor([T - 1], device=frames.device).repeat(B)
    data["camera_mask"] = new_camera_mask.clone()
    for key in ("poses", "original_poses", "intrinsics", "non_ndc_intrinsics", "original_intrinsics", "scale", "bbox_3d",
                "bbox_proj_crop"):
        if key in data:
            data[key] = _filter(data[key], camera_mask, neighbor_mask)
    if "original_images" in data:                              # list [T][B] -> keep the selected views (data_processing.py:150-176)
        nm = neighbor_mask.cpu().numpy()
        org = data["original_images"]
        n_keep = int(nm[0].sum())
        new = [[] for _ in range(n_keep + 1)]
        for b in range(B):
            r = 0
            for t in range(len(org) - 1):
                if nm[b, t]:
                    new[r].append(org[t][b]); r += 1
            new[-1].append(org[-1][b])
        data["original_images"] = new
    return data, new_pose_feat, new_frames, new_camera_mask, new_rgb, new_masks


def process_dense_input(data, pose_feat, frames, camera_mask, rgb_feature, image_masks, dense_cfg):
    if _get(dense_cfg, "filter") == "dino" and _get(dense_cfg, "filter_enable"):
        q = camera_mask.to(torch.int32).argmax(dim=1)
        _, neighbor_mask = match_views(rgb_feature, frames, q, int(_get(dense_cfg, "filter_topk")))
        return filter_by_neighbor_mask(data, neighbor_mask, pose_feat, frames, camera_mask, rgb_feature, image_masks)
    return data, pose_feat, frames, camera_mask, rgb_feature, image_masks


# ------------------------------------------------------------------------------------------- multi-round decode

def _sub(x, camera_mask, sub):
    B, T = camera_mask.shape
    q = x[camera_mask]
    ref = x[~camera_mask].reshape(B, T - 1, *x.shape[2:])
    rounds = (T - 1 + sub - 1) // sub
    out = torch.zeros(B, rounds, sub + 1, *x.shape[2:], dtype=x.dtype, device=x.device)
    for i in range(rounds):
        end = min((i + 1) * sub, T - 1)
        out[:, i, :end - i * sub] = ref[:, i * sub:end]
        out[:, i, sub] = q
    return out


def sub_batchify(pose_feat, frames, camera_mask, rgb_feature, image_masks, sub_batch_size):
    """(B, T, ...) -> (B, rounds, sub+1, ...): `sub` consecutive references per round (zero-padded at the end), query last."""
    B, T = camera_mask.shape
    rounds = (T - 1 + sub_batch_size - 1) // sub_batch_size
    new_camera_mask = torch.zeros(B, rounds, sub_batch_size + 1, dtype=torch.bool, device=camera_mask.device)
    new_camera_mask[:, :, sub_batch_size] = True
    return (_sub(pose_feat, camera_mask, sub_batch_size), _sub(frames, camera_mask, sub_batch_size), new_camera_mask,
            _sub(rgb_feature, camera_mask, sub_batch_size),
            _sub(image_masks, camera_mask, sub_batch_size) if image_masks is not None else None)


def fetch_neighbors_by_pose_similarity(gt_poses, pred_pose, topk=5):
    """gt_poses (B, N, 4, 4), pred_pose (B, 1, 4, 4) -> (B, topk) indices of the closest reference poses
    (geodesic rotation distance + translation distance)."""
    B, N = gt_poses.shape[:2]
    g = gt_poses.reshape(B * N, 4, 4).float()
    p = pred_pose.reshape(B, 1, 4, 4).repeat(1, N, 1, 1).reshape(B * N, 4, 4).float()
    Rd = torch.matmul(p[:, :3, :3], g[:, :3, :3].transpose(1, 2))
    tr = torch.diagonal(Rd, dim1=-2, dim2=-1).sum(-1)
    rot = torch.acos(torch.clamp((tr - 1) / 2, -1, 1))
    dist = (rot + torch.norm(p[:, :3, 3] - g[:, :3, 3], dim=-1)).reshape(B, N)
    return torch.topk(dist, k=topk, dim=1, largest=False)[1]


def recover_pose_from_dense_bb8(query_rets, bbox_3d, K):
    """query_rets (B, R, 8, H, W) heatmaps of R decoder rounds; bbox_3d (B, 8, 3); K (B, 3, 3).  All R*8 decoded corners of
    a sample go into one RANSAC PnP (box_utils.py:202-300: solvePnPRansac at 2 px / 0.99 / 1000 trials, ITERATIVE on failure --
    pnp.solve_pnp_ransac: cv2's own call where importable, else this repo's restatement of the scheme): a round whose corners are off
    is rejected instead of pulling the coarse pose (and with it the fine level's neighbour choice).  Returns poses (B, 1, 4, 4),
    normalised corners (B, R, 8, 2)."""
    B, R = query_rets.shape[:2]
    norm_kp, kp_px, _ = recover_bb8_corners_chw(query_rets.reshape(B * R, *query_rets.shape[2:]))
    kp = kp_px.reshape(B, R * 8, 2).cpu().numpy()
    b3 = bbox_3d.float().cpu().numpy()
    Kh = K.float().cpu().numpy()
    poses = np.zeros((B, 1, 4, 4), np.float32)
    for b in range(B):
        try:
            ok, Rm, t, _ = pnp.solve_pnp_ransac(np.tile(b3[b], (R, 1)), kp[b], Kh[b], reproj_err=2.0, confidence=0.99, max_trials=1000, seed=b)
        except Exception as e:  # noqa: BLE001
            print(f"PnP failed due to exception: {e}")
            continue
        if ok:
            poses[b, 0, :3, :3] = Rm
            poses[b, 0, :3, 3] = t
            poses[b, 0, 3, 3] = 1.0
    return torch.from_numpy(poses).to(query_rets.device), norm_kp.reshape(B, R, 8, 2)


def process_multi_round(data, pose_feat, frames, camera_mask, rgb_feature, image_masks, decoder, dense_cfg,
                        bbox_representation="heatmap"):
    """Decoder over sub-batches of the references (query appended to each), one PnP over all rounds' corners, optional
    fine round on the references nearest to that pose.  Returns the decoder output of the fine round (tensor) or, without
    it, the batch dict with the coarse prediction (as the reference does)."""
    B = frames.shape[0]
    poses = data["poses"].clone()
    K = data["non_ndc_intrinsics"].clone()
    bbox_3d = data["bbox_3d"].clone()
    sub = int(_get(dense_cfg, "sub_batch_size"))
    from .cache import take_views
    npf, nfr, ncm, nrf, _ = sub_batchify(pose_feat, frames, camera_mask, rgb_feature, None, sub)
    R = npf.shape[1]
    # the rounds' features WITH their operand-dtype copies (same values as `nrf`; cache.take_views): BETR does not re-cast them
    sidx = _sub_index(camera_mask, sub)                                   # (B, R, sub + 1) view indices, -1 = zero padding
    tagged = rgb_feature.dim() == 4
    if _get(dense_cfg, "dense_mem_friendly"):
        outs = [decoder(npf[:, i].contiguous(), nfr[:, i].contiguous(), ncm[:, i],
                        take_views(rgb_feature, sidx[:, i]) if tagged else nrf[:, i].contiguous(), None).clone() for i in range(R)]
        query_rets = torch.stack(outs, dim=1)
    else:
        feats = (take_views(rgb_feature, sidx.reshape(B, R * (sub + 1))) if tagged else nrf)
        if tagged:      # (B, R (sub + 1), P, C) -> (B R, sub + 1, P, C): the same storage, the operand copy follows the alias
            from . import features as _features
            feats = _features.carry(feats, feats.reshape(B * R, sub + 1, *feats.shape[2:]))
        else:
            feats = feats.reshape(B * R, *nrf.shape[2:])
        flat = decoder(npf.reshape(B * R, *npf.shape[2:]), nfr.reshape(B * R, *nfr.shape[2:]), ncm.reshape(B * R, -1), feats, None)
        query_rets = flat.reshape(B, R, *flat.shape[1:])
    query_poses, pred_proj = recover_pose_from_dense_bb8(query_rets, bbox_3d[camera_mask], K[camera_mask])
    if _get(dense_cfg, "fine_level"):
        idx = fetch_neighbors_by_pose_similarity(poses[~camera_mask].reshape(B, poses.shape[1] - 1, 4, 4), query_poses,
                                                 topk=int(_get(dense_cfg, "fine_topk")))
        neighbor_mask = torch.zeros(B, poses.shape[1] - 1, dtype=torch.bool, device=poses.device)
        neighbor_mask.scatter_(1, idx.to(poses.device), True)
        data, pose_feat, frames, camera_mask, rgb_feature, image_masks = filter_by_neighbor_mask(
            data, neighbor_mask, pose_feat, frames, camera_mask, rgb_feature, image_masks)
        from . import features as _features
        return decoder(pose_feat.contiguous(), frames.contiguous(), camera_mask, _features.carry(rgb_feature, rgb_feature.contiguous()), None)
    pred_poses = poses.clone()
    data["pred_bbox"] = pose_feat.clone()
    data["pred_bbox"][camera_mask] = query_rets[:, 0].to(pose_feat.dtype)
    pred_poses[camera_mask] = query_poses.squeeze(1).to(pred_poses.dtype)
    data["regression_boxes"] = data["bbox_proj_crop"].clone()
    data["regression_boxes"][camera_mask] = pred_proj[:, 0].to(data["regression_boxes"].dtype)
    data["pred_poses"] = pred_poses
    data["pred_intrinsics"] = data["intrinsics"].clone()
    return data
