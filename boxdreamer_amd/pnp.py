"""Host-side PnP for the 8 decoded box corners (stays on the CPU by design, north_star / SURVEY.md §8 K12).

The reference calls cv2.solvePnPRansac (result discarded), then cv2.solvePnP(SOLVEPNP_ITERATIVE) and
cv2.Rodrigues per sample (/root/reference/src/models/utils/box_utils.py:139-199).  When OpenCV is
importable that exact solvePnP call is used.  This image has no cv2, so the fallback below restates the
published ITERATIVE algorithm for non-planar points (DLT initialisation + Levenberg-Marquardt on the
reprojection error IN PIXELS over (rvec, tvec)) in numpy.  PARITY AGAINST OPENCV ITSELF IS UNPINNED (no cv2 here); what is pinned is
the objective: tests/test_host_logic.py::test_pnp_is_the_minimiser_of_the_pixel_reprojection_error checks every form of this solver
(this file's two and the native one in csrc/pnp.hip) against an independent minimiser of that objective (scipy's MINPACK
Levenberg-Marquardt) on noisy corners with square and non-square pixels, to 5e-7 / 2e-6 in R and t.  OpenCV's own termination
(at most 20 iterations, FLT_EPSILON) and its float32 interface are not reproduced: both stop at the same minimum to that accuracy.
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
    # OpenCV's ITERATIVE solver minimises the reprojection error in PIXELS: the normalised x / y residuals weighted by fx / fy.
    # (fx, fy) / sqrt(fx fy) gives the same minimiser and is exactly (1, 1) for square pixels.
    wxy = np.abs(np.array([K[0, 0], K[1, 1]])) / np.sqrt(abs(K[0, 0] * K[1, 1]))

    def resid(v):
        return ((_project(v[:3], v[3:], p3) - p2n) * wxy).reshape(-1)

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


# ------------------------------------------------------------------------------------------------
# Batched form ("next" row f3 of SURVEY.md §8: removes the per-sample Python loop + double OpenCV call
# of box_utils.py:139-199).  The same algorithm as solve_pnp_iterative, vectorised over N poses with
# batched numpy linear algebra: one SVD call for all DLT systems, Levenberg-Marquardt with per-pose
# damping and accept / reject masks.  PARITY UNPINNED against OpenCV like the scalar form; the two forms
# agree with each other to ~1e-9 (tests/test_host_logic.py).

def _rodrigues_b(rvec: np.ndarray) -> np.ndarray:
    th = np.linalg.norm(rvec, axis=1)
    small = th < 1e-12
    k = rvec / np.where(small, 1.0, th)[:, None]
    Kx = np.zeros((rvec.shape[0], 3, 3))
    Kx[:, 0, 1], Kx[:, 0, 2] = -k[:, 2], k[:, 1]
    Kx[:, 1, 0], Kx[:, 1, 2] = k[:, 2], -k[:, 0]
    Kx[:, 2, 0], Kx[:, 2, 1] = -k[:, 1], k[:, 0]
    R = np.eye(3)[None] + np.sin(th)[:, None, None] * Kx + (1 - np.cos(th))[:, None, None] * (Kx @ Kx)
    R[small] = np.eye(3)
    return R


def _project_b(x: np.ndarray, p3: np.ndarray) -> np.ndarray:
    pc = p3 @ np.swapaxes(_rodrigues_b(x[:, :3]), 1, 2) + x[:, None, 3:]
    return pc[..., :2] / pc[..., 2:3]


def solve_pnp_batched(p3: np.ndarray, p2: np.ndarray, K: np.ndarray, iters: int = 30):
    """p3 (N, n, 3), p2 (N, n, 2) pixels, K (N, 3, 3) -> (ok (N,) bool, R (N, 3, 3), t (N, 3))."""
    p3 = np.asarray(p3, np.float64)
    p2 = np.asarray(p2, np.float64)
    K = np.asarray(K, np.float64)
    N, n = p3.shape[:2]
    if _HAVE_CV2:  # pragma: no cover - the reference's own solver when it is importable
        ok = np.zeros(N, bool); R = np.tile(np.eye(3), (N, 1, 1)); t = np.zeros((N, 3))
        for i in range(N):
            ok[i], R[i], t[i] = solve_pnp_iterative(p3[i], p2[i], K[i], iters)
        return ok, R, t
    f = np.stack([K[:, 0, 0], K[:, 1, 1]], 1)
    p2n = (p2 - K[:, None, :2, 2]) / f[:, None, :]
    # ---- DLT initialisation, all poses in one SVD
    X = np.concatenate([p3, np.ones((N, n, 1))], 2)
    A = np.zeros((N, 2 * n, 12))
    A[:, 0::2, 0:4] = X
    A[:, 0::2, 8:12] = -p2n[:, :, :1] * X
    A[:, 1::2, 4:8] = X
    A[:, 1::2, 8:12] = -p2n[:, :, 1:2] * X
    ok = np.isfinite(A).all(axis=(1, 2))
    A[~ok] = 0.0
    _, _, vt = np.linalg.svd(A)
    P = vt[:, -1].reshape(N, 3, 4)
    U, s, Vt = np.linalg.svd(P[:, :, :3])
    R = U @ Vt
    scale = s.mean(1)
    neg = np.linalg.det(R) < 0
    R[neg], scale[neg] = -R[neg], -scale[neg]
    t = P[:, :, 3] / np.where(scale == 0, 1.0, scale)[:, None]
    behind = (np.einsum("nij,nj->ni", R, p3.mean(1)) + t)[:, 2] < 0
    R[behind], t[behind] = -R[behind], -t[behind]
    fix = behind & (np.linalg.det(R) < 0)
    if fix.any():
        U2 = U[fix].copy(); U2[:, :, -1] *= -1
        R[fix] = U2 @ Vt[fix]
    # ---- rotation vector of R (vectorised _rvec_from_R)
    c = np.clip((np.trace(R, axis1=1, axis2=2) - 1) / 2, -1.0, 1.0)
    th = np.arccos(c)
    w = np.stack([R[:, 2, 1] - R[:, 1, 2], R[:, 0, 2] - R[:, 2, 0], R[:, 1, 0] - R[:, 0, 1]], 1)
    rvec = w / (2 * np.where(np.sin(th) == 0, 1.0, np.sin(th)))[:, None] * th[:, None]
    rvec[th < 1e-8] = 0.0
    for i in np.nonzero(np.pi - th < 1e-4)[0]:          # near pi (rare): scalar branch
        rvec[i] = _rvec_from_R(R[i])
    x = np.concatenate([rvec, t], 1)
    lam = np.full(N, 1e-3)
    wxy = (np.abs(f) / np.sqrt(np.abs(f[:, :1] * f[:, 1:2])))[:, None, :]        # pixel-error weights (see solve_pnp_iterative)
    p2n_w = p2n * wxy

    def _proj_w(v):
        return _project_b(v, p3) * wxy
    r = (_proj_w(x) - p2n_w).reshape(N, -1)
    ok &= np.isfinite(r).all(1)
    r[~ok] = 0.0
    x[~ok] = 0.0; x[~ok, 5] = 1.0
    active = ok.copy()
    eye6 = np.eye(6)
    for _ in range(iters):
        if not active.any():
            break
        J = np.empty((N, r.shape[1], 6))
        for j in range(6):
            J[:, :, j] = ((_proj_w(x + 1e-6 * eye6[j]) - p2n_w).reshape(N, -1) - r) / 1e-6
        J[~np.isfinite(J)] = 0.0
        H = np.swapaxes(J, 1, 2) @ J
        g = np.einsum("nij,ni->nj", J, r)
        D = np.einsum("nii->ni", H) + 1e-12
        Hd = H + lam[:, None, None] * (D[:, :, None] * eye6[None])
        sing = np.abs(np.linalg.det(Hd)) < 1e-300
        Hd[sing] = eye6
        step = np.linalg.solve(Hd, -g[:, :, None])[:, :, 0]
        step[sing | ~active] = 0.0
        active &= ~sing                                   # scalar form: LinAlgError -> stop iterating this pose
        r_new = (_proj_w(x + step) - p2n_w).reshape(N, -1)
        better = active & np.isfinite(r_new).all(1) & ((r_new * r_new).sum(1) < (r * r).sum(1))
        x[better] += step[better]
        r[better] = r_new[better]
        lam = np.where(better, np.maximum(lam * 0.3, 1e-9), np.where(active, lam * 10, lam))
        active &= ~(better & (np.linalg.norm(step, axis=1) < 1e-10))
    Rout = _rodrigues_b(x[:, :3])
    tout = x[:, 3:].copy()
    Rout[~ok] = np.eye(3); tout[~ok] = 0.0
    return ok, Rout, tout


# ------------------------------------------------------------------------------------------------
# RANSAC form (round 6): the dense multi-round pose.  The reference feeds the R x 8 corners of all decoder rounds to
# cv2.solvePnPRansac(reprojectionError=2.0, confidence=0.99, iterationsCount=1000, flags=SOLVEPNP_ITERATIVE) and falls back to
# cv2.solvePnP only when that fails (reference: src/models/utils/box_utils.py:266-285): one bad round's corners are REJECTED, not
# averaged in.  With cv2 importable that exact call is used.  Without it: the same scheme -- minimal-sample hypotheses (6 distinct 3-D
# points: the DLT's minimum; the rounds repeat the same 8 box corners, so a sample never takes one corner twice), inliers within
# `reproj_err` pixels, an adaptive trial count from the best inlier ratio, a final ITERATIVE solve on the inliers of the best hypothesis.
# Hypotheses are generated and scored in vectorised batches (solve_pnp_batched).  Sampling is deterministic (a Philox counter stream
# keyed by `seed`): the same corners give the same pose on every box.  OpenCV's own sampler / minimal solver (EPnP on 5 points) are NOT
# reproduced: parity against its binary is un-pinned, like the rest of this file.

def _reproj_px(R, t, p3, p2, K):
    pc = p3 @ np.swapaxes(R, -1, -2) + t[..., None, :]
    z = pc[..., 2:3]
    uv = pc[..., :2] / np.where(np.abs(z) < 1e-12, 1e-12, z) * np.stack([K[0, 0], K[1, 1]]) + K[:2, 2]
    err = np.linalg.norm(uv - p2, axis=-1)
    return np.where(z[..., 0] > 0, err, np.inf)            # a point behind the camera is never an inlier


def solve_pnp_ransac(p3: np.ndarray, p2: np.ndarray, K: np.ndarray, reproj_err: float = 2.0, confidence: float = 0.99,
                     max_trials: int = 1000, seed: int = 0, batch: int = 32):
    """(success, R (3,3), t (3,), inlier mask (n,) bool).  p3 (n,3) may repeat 3-D points (several 2-D observations of one corner)."""
    p3 = np.asarray(p3, np.float64)
    p2 = np.asarray(p2, np.float64)
    K = np.asarray(K, np.float64)
    n = p3.shape[0]
    if _HAVE_CV2:  # pragma: no cover - the reference's own call when it is importable
        ok, rvec, tvec, inl = cv2.solvePnPRansac(p3.astype(np.float32), p2.astype(np.float32), K.astype(np.float32), None,
                                                 reprojectionError=float(reproj_err), confidence=float(confidence),
                                                 flags=cv2.SOLVEPNP_ITERATIVE, iterationsCount=int(max_trials))
        if ok:
            mask = np.zeros(n, bool)
            if inl is not None:
                mask[np.asarray(inl).reshape(-1)] = True
            return True, cv2.Rodrigues(rvec)[0], tvec.reshape(3), mask
        ok, R, t = solve_pnp_iterative(p3, p2, K)
        return ok, R, t, np.ones(n, bool)
    # distinct 3-D points (the rounds repeat the box corners): a sample draws S distinct corners, then one observation of each
    _, corner_of = np.unique(np.round(p3, 9), axis=0, return_inverse=True)
    corner_of = corner_of.reshape(-1)
    n_corners = int(corner_of.max()) + 1
    S = 6
    if n_corners < S or not np.isfinite(p2).all():
        ok, R, t = solve_pnp_iterative(p3, p2, K)
        return ok, R, t, np.ones(n, bool)
    obs = [np.nonzero(corner_of == c)[0] for c in range(n_corners)]
    rng = np.random.Generator(np.random.Philox(key=int(seed) & 0xFFFFFFFFFFFFFFFF))
    best_mask, best_cnt, best_err = None, -1, np.inf
    trials, need = 0, max_trials
    while trials < min(need, max_trials):
        h = min(batch, max_trials - trials)
        corners = np.argsort(rng.random((h, n_corners)), axis=1)[:, :S]               # S distinct corners per hypothesis
        pick = np.stack([[obs[c][int(rng.integers(len(obs[c])))] for c in row] for row in corners])      # one observation of each
        ok, R, t = solve_pnp_batched(p3[pick], p2[pick], np.broadcast_to(K, (h, 3, 3)), iters=5)
        err = _reproj_px(R, t, p3[None], p2[None], K)                                     # (h, n)
        inl = err < reproj_err
        cnt = np.where(ok, inl.sum(1), -1)
        tot = np.where(inl, err, 0.0).sum(1)
        for i in np.argsort(-cnt, kind="stable")[:1]:                                     # the batch's best; ties: lowest index
            if cnt[i] > best_cnt or (cnt[i] == best_cnt and tot[i] < best_err):
                best_cnt, best_err, best_mask = int(cnt[i]), float(tot[i]), inl[i].copy()
        trials += h
        w = max(best_cnt, 0) / n
        if w >= 1.0:
            break
        if w > 0:                                                                          # trials for `confidence` at this inlier ratio
            need = int(np.ceil(np.log(1.0 - confidence) / np.log(max(1.0 - w ** S, 1e-300))))
    if best_mask is None or best_cnt < S or len(np.unique(corner_of[best_mask])) < S:
        ok, R, t = solve_pnp_iterative(p3, p2, K)                                         # the reference's fallback (box_utils.py:280-285)
        return ok, R, t, np.ones(n, bool)
    ok, R, t = solve_pnp_iterative(p3[best_mask], p2[best_mask], K)
    if not ok:
        ok, R, t = solve_pnp_iterative(p3, p2, K)
        return ok, R, t, np.ones(n, bool)
    return True, R, t, best_mask
