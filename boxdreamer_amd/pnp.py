"""Host-side PnP for the 8 decoded box corners (stays on the CPU by design, north_star / SURVEY.md §8 K12).

The reference calls cv2.solvePnPRansac (result discarded), then cv2.solvePnP(SOLVEPNP_ITERATIVE) and
cv2.Rodrigues per sample (/root/reference/src/models/utils/box_utils.py:139-199).  When OpenCV is
importable that exact solvePnP call is used.  This image has no cv2, so the fallback below restates the
published ITERATIVE algorithm for non-planar points (DLT initialisation + Levenberg-Marquardt on the
reprojection error over (rvec, tvec)) in numpy.  PARITY UNPINNED: no OpenCV here to check against; only the
corners fed to it are pinned.
"""
from __future__ import annotations

import numpy as np

try:  # pragma: no cover - not available in the build image
    import cv2  # type: ignore
    _HAVE_CV2 = hasattr(cv2, "solvePnP")
except Exception:  # noqa: BLE001
    cv2 = None
    _HAVE_CV2 = False


def rodrigues(rvec: np.ndarray) -> np.ndarray:
    th = float(np.linalg.norm(rvec))
    if th < 1e-12:
        return np.eye(3)
    k = rvec.reshape(3) / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * (Kx @ Kx)


def _rvec_from_R(R: np.ndarray) -> np.ndarray:
    c = np.clip((np.trace(R) - 1) / 2, -1.0, 1.0)
    th = np.arccos(c)
    if th < 1e-8:
        return np.zeros(3)
    if np.pi - th < 1e-4:                      # near pi: take the dominant column of (R + I)/2
        A = (R + np.eye(3)) / 2
        i = int(np.argmax(np.diag(A)))
        v = A[:, i] / np.sqrt(max(A[i, i], 1e-12))
        return v * th
    w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) / (2 * np.sin(th))
    return w * th


def _dlt_init(p3: np.ndarray, p2n: np.ndarray):
    """Linear pose from >= 6 non-planar points in normalised image coordinates."""
    n = p3.shape[0]
    A = np.zeros((2 * n, 12))
    X = np.concatenate([p3, np.ones((n, 1))], 1)
    A[0::2, 0:4] = X
    A[0::2, 8:12] = -p2n[:, :1] * X
    A[1::2, 4:8] = X
    A[1::2, 8:12] = -p2n[:, 1:2] * X
    _, _, vt = np.linalg.svd(A)
    P = vt[-1].reshape(3, 4)
    U, s, Vt = np.linalg.svd(P[:, :3])
    R = U @ Vt
    scale = s.mean()
    if np.linalg.det(R) < 0:
        R, scale = -R, -scale
    t = P[:, 3] / scale
    if (R @ p3.mean(0) + t)[2] < 0:            # points must be in front of the camera
        R, t = -R, -t
        if np.linalg.det(R) < 0:
            U[:, -1] *= -1
            R = U @ Vt
    return R, t


def _project(rvec, t, p3):
    pc = p3 @ rodrigues(rvec).T + t
    return pc[:, :2] / pc[:, 2:3]


def solve_pnp_iterative(p3: np.ndarray, p2: np.ndarray, K: np.ndarray, iters: int = 30):
    """(success, R (3,3), t (3,)) minimising reprojection error; p3 (n,3), p2 (n,2) pixels."""
    p3 = np.asarray(p3, np.float64)
    p2 = np.asarray(p2, np.float64)
    K = np.asarray(K, np.float64)
    if _HAVE_CV2:  # pragma: no cover
        ok, rvec, tvec = cv2.solvePnP(p3.astype(np.float32), p2.astype(np.float32), K.astype(np.float32), None,
                                      flags=cv2.SOLVEPNP_ITERATIVE)
        if not ok:
            return False, np.eye(3), np.zeros(3)
        return True, cv2.Rodrigues(rvec)[0], tvec.reshape(3)
    p2n = (p2 - K[:2, 2]) / np.array([K[0, 0], K[1, 1]])
    try:
        R, t = _dlt_init(p3, p2n)
    except np.linalg.LinAlgError:
        return False, np.eye(3), np.zeros(3)
    x = np.concatenate([_rvec_from_R(R), t])
    lam = 1e-3

    def resid(v):
        return (_project(v[:3], v[3:], p3) - p2n).reshape(-1)

    r = resid(x)
    if not np.all(np.isfinite(r)):
        return False, np.eye(3), np.zeros(3)
    for _ in range(iters):
        J = np.zeros((r.size, 6))
        for j in range(6):
            d = np.zeros(6)
            d[j] = 1e-6
            J[:, j] = (resid(x + d) - r) / 1e-6
        H, g = J.T @ J, J.T @ r
        try:
            step = np.linalg.solve(H + lam * np.diag(np.diag(H) + 1e-12), -g)
        except np.linalg.LinAlgError:
            break
        r_new = resid(x + step)
        if np.all(np.isfinite(r_new)) and r_new @ r_new < r @ r:
            x, r, lam = x + step, r_new, max(lam * 0.3, 1e-9)
            if np.linalg.norm(step) < 1e-10:
                break
        else:
            lam *= 10
    return True, rodrigues(x[:3]), x[3:]
