"""Corner decode + pose recovery with the reference's function names
(/root/reference/src/models/utils/box_utils.py:14-199, heatmap branch only).

`recover_bb8_corners` runs the HIP top-20 decode (bd_decode_topk); `recover_pose_from_bb8` moves the 8 corners,
the 3-D box and K to the host ONCE per batch and solves PnP there (the reference does a D2H + two OpenCV
calls per sample inside a Python loop, box_utils.py:139-199)."""
from __future__ import annotations


import numpy as np
import torch

from . import _lib, hip_ops, pnp



def recover_bb8_corners(bbox_feat: torch.Tensor, bbox_representation: str = "heatmap"):
    """bbox_feat: [B, T, H, W, 8] in [-1, 1] (the reference's layout, prediction_utils.py:65) ->
    (normalised [B,T,8,2] in [-1,1], pixel [B,T,8,2])."""
    if bbox_representation != "heatmap":
        raise NotImplementedError("the MI355X path implements the heatmap representation only")
    B, T, H, W, C = bbox_feat.shape
    heat = bbox_feat.permute(0, 1, 4, 2, 3).contiguous().float()      # [B,T,8,H,W]
    kp, kn, _ = hip_ops.decode_topk(heat, k=20, want_idx=False)
    return kn, kp


def recover_bb8_corners_chw(heat: torch.Tensor, want_idx: bool = False):
    """Same decode on the decoder's native [B, 8, H, W] output (no permute round trip)."""
    kp, kn, idx = hip_ops.decode_topk(heat, k=20, want_idx=want_idx)
    return kn, kp, idx


def _host_workers() -> int:
    """Threads of the native host solver: the CPUs this process may use (affinity and cgroup quota), at most 16."""
    global _WORKERS
    if _WORKERS is None:
        import os
        n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        try:
            q, p = open("/sys/fs/cgroup/cpu.max").read().split()
            if q != "max":
                n = min(n, max(1, int(int(q) / int(p))))
        except Exception:  # noqa: BLE001
            pass
        _WORKERS = max(1, min(16, n))
    return _WORKERS


_WORKERS = None


def solve_poses_host(kp_px: np.ndarray, bbox_3d: np.ndarray, K: np.ndarray, workers: int | None = None) -> np.ndarray:
    """kp_px [N,8,2], bbox_3d [N,8,3], K [N,3,3] (host) -> poses [N,4,4] ([R|t], zeros on failure).
    One batched solve for all N poses ("next" row f3: no per-sample Python loop; pnp.solve_pnp_batched)."""
    n = kp_px.shape[0]
    out = np.zeros((n, 4, 4), np.float32)
    if n == 0:
        return out
    if not pnp._HAVE_CV2:
        # native host solver of the HIP library (csrc/pnp.hip: bd_solve_pnp_host, `workers` threads): the algorithm of
        # pnp.solve_pnp_batched step by step (which stays as its numpy cross-check, tests/test_host_logic.py)
        # The PnP post-solve is HOST work (north_star); on a host-only box without the built library (the GPU path itself has raised
        # long before) or when the native solver rejects its arguments, the batched numpy form below solves the same problem.
        try:
            lib = _lib.load()
            kp32, p32, K32 = (np.ascontiguousarray(a, np.float32) for a in (kp_px, bbox_3d, K))
            rc = lib.bd_solve_pnp_host(kp32.ctypes.data, p32.ctypes.data, K32.ctypes.data, n, kp32.shape[1], 30, out.ctypes.data,
                                       _host_workers() if workers is None else workers)
            if rc == 0:
                return out
            out[:] = 0.0
        except (_lib.HipLibraryError, OSError):
            pass
    try:
        ok, R, t = pnp.solve_pnp_batched(bbox_3d, kp_px, K)
    except Exception as e:  # noqa: BLE001  (reference: print and leave zeros, box_utils.py:192-195)
        print(f"PnP failed due to exception: {e}")
        return out
    out[ok, :3, :3] = R[ok]
    out[ok, :3, 3] = t[ok]
    out[ok, 3, 3] = 1.0
    return out


def solve_poses_device(kp_px: torch.Tensor, bbox_3d: torch.Tensor, K: torch.Tensor, iters: int = 30) -> torch.Tensor:
    """kp_px [N,n,2], bbox_3d [N,n,3], K [N,3,3] on the GPU -> poses [N,4,4] on the GPU ([R|t], zeros on failure):
    `bd_solve_pnp`, one pose per thread in fp64 -- the corners never leave the device ("next" row f3)."""
    _lib.require_gpu()
    lib = _lib.load()
    kp = kp_px.float().contiguous()
    p3 = bbox_3d.to(kp.device).float().contiguous()
    Kd = K.to(kp.device).float().contiguous()
    n, npts = kp.shape[0], kp.shape[1]
    out = torch.empty((n, 4, 4), dtype=torch.float32, device=kp.device)
    _lib.check(lib.bd_solve_pnp(_lib.ptr(kp), _lib.ptr(p3), _lib.ptr(Kd), n, npts, iters, _lib.ptr(out), _lib.stream()),
               "bd_solve_pnp")
    return out


def recover_pose_from_bb8(bbox_feat, bbox_3d, K, bbox_representation="heatmap"):
    """box_utils.py:113-199: (poses [B,T,4,4], normalised corners [B,T,8,2])."""
    B, T = bbox_feat.shape[:2]
    normalized_keypoints_2d, keypoints_2d = recover_bb8_corners(bbox_feat, bbox_representation)
    kp = keypoints_2d.reshape(B * T, 8, 2).cpu().numpy()              # ONE device->host sync per batch
    b3 = bbox_3d.float().reshape(B * T, 8, 3).cpu().numpy()
    Kh = K.float()
    Kh = (Kh.reshape(B * T, 3, 3) if Kh.dim() == 4 else Kh.reshape(1, 3, 3).expand(B * T, 3, 3)).cpu().numpy()
    poses = torch.from_numpy(solve_poses_host(kp, b3, Kh)).reshape(B, T, 4, 4).to(bbox_feat.device)
    return poses, normalized_keypoints_2d
