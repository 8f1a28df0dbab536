// Pose from the decoded box corners on the GPU ("next" row f3 of SURVEY.md section 8): the reference solves one
// cv2.solvePnP(SOLVEPNP_ITERATIVE) per sample in a Python loop on the host (src/models/utils/box_utils.py:139-199).
// This kernel restates the same published algorithm as boxdreamer_amd/pnp.py -- DLT initialisation for non-planar
// points, then Levenberg-Marquardt on the reprojection error over (rvec, tvec) with a forward-difference Jacobian --
// one thread per pose in fp64, so the 8 corners never leave the device and the per-batch host solve (9 ms per 32
// poses) disappears from the serving loop.  The work is tiny (~2e5 flops per pose) and latency-bound: no attempt is
// made to spread one pose over a wavefront.  PARITY against OpenCV is un-pinned (no cv2 in the image); the kernel is
// tested against the numpy form, which it follows step by step.
#include "bd_common.h"
#include <cmath>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <functional>
#include <unistd.h>
#include <vector>

namespace {

constexpr int MAXPTS = 64;

// (isfinite is a __device__-only overload in HIP's headers: a portable test for the code shared by the GPU and the host threads)
__host__ __device__ inline bool finite_d(double v) { return v - v == 0.0; }

// cyclic Jacobi eigen-decomposition of a symmetric n x n matrix (row-major, destroyed); V columns = eigenvectors
template <int N> __host__ __device__ void jacobi_eig(double* A, double* V, double* w) {
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) V[i * N + j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < N; ++i) {
            diag += A[i * N + i] * A[i * N + i];
            for (int j = i + 1; j < N; ++j) off += A[i * N + j] * A[i * N + j];
        }
        if (off <= 1e-60 || off <= 1e-32 * diag) break;
        for (int p = 0; p < N - 1; ++p)
            for (int q = p + 1; q < N; ++q) {
                const double apq = A[p * N + q];
                if (apq == 0.0) continue;
                const double theta = (A[q * N + q] - A[p * N + p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < N; ++k) {
                    const double akp = A[k * N + p], akq = A[k * N + q];
                    A[k * N + p] = c * akp - s * akq;
                    A[k * N + q] = s * akp + c * akq;
                }
                for (int k = 0; k < N; ++k) {
                    const double apk = A[p * N + k], aqk = A[q * N + k];
                    A[p * N + k] = c * apk - s * aqk;
                    A[q * N + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < N; ++k) {
                    const double vkp = V[k * N + p], vkq = V[k * N + q];
                    V[k * N + p] = c * vkp - s * vkq;
                    V[k * N + q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < N; ++i) w[i] = A[i * N + i];
}

__host__ __device__ double det3(const double* R) {
    return R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) + R[2] * (R[3] * R[7] - R[4] * R[6]);
}

__host__ __device__ void rodrigues(const double* rv, double* R) {
    const double th = sqrt(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);
    if (th < 1e-12) {
        for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
        return;
    }
    const double k0 = rv[0] / th, k1 = rv[1] / th, k2 = rv[2] / th, s = sin(th), c1 = 1.0 - cos(th);
    const double Kx[9] = {0, -k2, k1, k2, 0, -k0, -k1, k0, 0};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double kk = 0.0;
            for (int m = 0; m < 3; ++m) kk += Kx[i * 3 + m] * Kx[m * 3 + j];
            R[i * 3 + j] = (i == j ? 1.0 : 0.0) + s * Kx[i * 3 + j] + c1 * kk;
        }
}

__host__ __device__ void rvec_from_R(const double* R, double* rv) {
    double c = (R[0] + R[4] + R[8] - 1.0) / 2.0;
    c = c < -1.0 ? -1.0 : (c > 1.0 ? 1.0 : c);
    const double th = acos(c);
    if (th < 1e-8) { rv[0] = rv[1] = rv[2] = 0.0; return; }
    if (3.14159265358979323846 - th < 1e-4) {            // near pi: dominant column of (R + I) / 2
        double A[9];
        for (int i = 0; i < 9; ++i) A[i] = (R[i] + (i % 4 == 0 ? 1.0 : 0.0)) / 2.0;
        int k = 0;
        if (A[4] > A[k * 4]) k = 1;
        if (A[8] > A[k * 4]) k = 2;
        const double d = sqrt(A[k * 4] > 1e-12 ? A[k * 4] : 1e-12);
        for (int i = 0; i < 3; ++i) rv[i] = A[i * 3 + k] / d * th;
        return;
    }
    const double f = th / (2.0 * sin(th));
    rv[0] = (R[7] - R[5]) * f; rv[1] = (R[2] - R[6]) * f; rv[2] = (R[3] - R[1]) * f;
}

// residuals of pose x = (rvec, t) for n points; returns false on a non-finite value.  OpenCV's ITERATIVE solver minimises the
// reprojection error in PIXELS, i.e. it weights the normalised x / y residuals by fx / fy: (wx, wy) = (fx, fy) / sqrt(fx fy) gives the
// same minimiser and is exactly (1, 1) for square pixels.
__host__ __device__ bool residuals(const double* x, const double* p3, const double* p2n, int n, double* r, double wx, double wy) {
    double R[9];
    rodrigues(x, R);
    bool ok = true;
    for (int i = 0; i < n; ++i) {
        const double X = p3[i * 3], Y = p3[i * 3 + 1], Z = p3[i * 3 + 2];
        const double cx = R[0] * X + R[1] * Y + R[2] * Z + x[3], cy = R[3] * X + R[4] * Y + R[5] * Z + x[4],
                     cz = R[6] * X + R[7] * Y + R[8] * Z + x[5];
        r[2 * i] = (cx / cz - p2n[2 * i]) * wx;
        r[2 * i + 1] = (cy / cz - p2n[2 * i + 1]) * wy;
        ok = ok && finite_d(r[2 * i]) && finite_d(r[2 * i + 1]);
    }
    return ok;
}

// solve the 6 x 6 system H d = g (partial pivoting); false if singular
__host__ __device__ bool solve6(double* H, double* g, double* d) {
    for (int c = 0; c < 6; ++c) {
        int piv = c;
        for (int r = c + 1; r < 6; ++r) if (fabs(H[r * 6 + c]) > fabs(H[piv * 6 + c])) piv = r;
        if (fabs(H[piv * 6 + c]) < 1e-300) return false;
        if (piv != c) {
            for (int k = 0; k < 6; ++k) { const double t = H[c * 6 + k]; H[c * 6 + k] = H[piv * 6 + k]; H[piv * 6 + k] = t; }
            const double t = g[c]; g[c] = g[piv]; g[piv] = t;
        }
        for (int r = c + 1; r < 6; ++r) {
            const double f = H[r * 6 + c] / H[c * 6 + c];
            for (int k = c; k < 6; ++k) H[r * 6 + k] -= f * H[c * 6 + k];
            g[r] -= f * g[c];
        }
    }
    for (int r = 5; r >= 0; --r) {
        double s = g[r];
        for (int k = r + 1; k < 6; ++k) s -= H[r * 6 + k] * d[k];
        d[r] = s / H[r * 6 + r];
    }
    return true;
}

// One pose: kp [n, 2] pixels, pts3 [n, 3], Kp [3, 3] -> out [4, 4] = [R|t; 0 0 0 1], all zeros where the solve fails.  The same code
// runs per GPU thread (pnp_kernel) and per host worker thread (bd_solve_pnp_host).
__host__ __device__ void solve_one_pose(const float* kp, const float* pts3, const float* Kp, int n, int iters, float* out) {
    for (int i = 0; i < 16; ++i) out[i] = 0.f;
    double p3[MAXPTS * 3], p2n[MAXPTS * 2];
    const double fx = Kp[0], fy = Kp[4], cx = Kp[2], cy = Kp[5];
    double mean[3] = {0, 0, 0};
    bool ok = true;
    for (int i = 0; i < n; ++i) {
        for (int c = 0; c < 3; ++c) { p3[i * 3 + c] = pts3[i * 3 + c]; mean[c] += p3[i * 3 + c] / n; }
        p2n[2 * i] = ((double)kp[i * 2] - cx) / fx;
        p2n[2 * i + 1] = ((double)kp[i * 2 + 1] - cy) / fy;
        ok = ok && finite_d(p2n[2 * i]) && finite_d(p2n[2 * i + 1]);
    }
    if (!ok) return;
    // ---- DLT: null vector of A (2n x 12) = eigenvector of A^T A with the smallest eigenvalue
    double AtA[144], V[144], w[12];
    for (int i = 0; i < 144; ++i) AtA[i] = 0.0;
    for (int i = 0; i < n; ++i) {
        const double X[4] = {p3[i * 3], p3[i * 3 + 1], p3[i * 3 + 2], 1.0};
        double r1[12], r2[12];
        for (int c = 0; c < 4; ++c) {
            r1[c] = X[c]; r1[4 + c] = 0.0; r1[8 + c] = -p2n[2 * i] * X[c];
            r2[c] = 0.0; r2[4 + c] = X[c]; r2[8 + c] = -p2n[2 * i + 1] * X[c];
        }
        for (int a = 0; a < 12; ++a)
            for (int b = 0; b < 12; ++b) AtA[a * 12 + b] += r1[a] * r1[b] + r2[a] * r2[b];
    }
    jacobi_eig<12>(AtA, V, w);
    int kmin = 0;
    for (int i = 1; i < 12; ++i) if (w[i] < w[kmin]) kmin = i;
    double P[12];
    for (int i = 0; i < 12; ++i) P[i] = V[i * 12 + kmin];
    // ---- nearest rotation to M = P[:, :3]: M = U S V^T, R = U V^T (through the eigen-decomposition of M^T M)
    double M[9], MtM[9], V3[9], s2[3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) M[i * 3 + j] = P[i * 4 + j];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0.0;
            for (int k = 0; k < 3; ++k) s += M[k * 3 + i] * M[k * 3 + j];
            MtM[i * 3 + j] = s;
        }
    jacobi_eig<3>(MtM, V3, s2);
    int ord[3] = {0, 1, 2};                                   // descending singular values (numpy's order)
    for (int a = 0; a < 2; ++a) for (int b = a + 1; b < 3; ++b) if (s2[ord[b]] > s2[ord[a]]) { const int t = ord[a]; ord[a] = ord[b]; ord[b] = t; }
    double sig[3], U[9], Vs[9];
    for (int c = 0; c < 3; ++c) {
        sig[c] = sqrt(s2[ord[c]] > 0 ? s2[ord[c]] : 0.0);
        for (int i = 0; i < 3; ++i) Vs[i * 3 + c] = V3[i * 3 + ord[c]];
    }
    if (!(sig[2] > 0.0)) return;
    for (int c = 0; c < 3; ++c)
        for (int i = 0; i < 3; ++i) {
            double s = 0.0;
            for (int k = 0; k < 3; ++k) s += M[i * 3 + k] * Vs[k * 3 + c];
            U[i * 3 + c] = s / sig[c];
        }
    double R[9], t[3];
    auto UVt = [&]() {
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double s = 0.0;
                for (int c = 0; c < 3; ++c) s += U[i * 3 + c] * Vs[j * 3 + c];
                R[i * 3 + j] = s;
            }
    };
    UVt();
    double scale = (sig[0] + sig[1] + sig[2]) / 3.0;
    if (det3(R) < 0) { for (int i = 0; i < 9; ++i) R[i] = -R[i]; scale = -scale; }
    for (int i = 0; i < 3; ++i) t[i] = P[i * 4 + 3] / scale;
    const double zc = R[6] * mean[0] + R[7] * mean[1] + R[8] * mean[2] + t[2];
    if (zc < 0) {                                             // points must be in front of the camera
        for (int i = 0; i < 9; ++i) R[i] = -R[i];
        for (int i = 0; i < 3; ++i) t[i] = -t[i];
        if (det3(R) < 0) { for (int i = 0; i < 3; ++i) U[i * 3 + 2] = -U[i * 3 + 2]; UVt(); }
    }
    // ---- Levenberg-Marquardt on (rvec, t)
    double x[6], r[MAXPTS * 2], rn[MAXPTS * 2], J[MAXPTS * 2 * 6];
    rvec_from_R(R, x);
    x[3] = t[0]; x[4] = t[1]; x[5] = t[2];
    const double fgm = sqrt(fabs(fx * fy)), wx = fgm > 0 ? fabs(fx) / fgm : 1.0, wy = fgm > 0 ? fabs(fy) / fgm : 1.0;
    if (!residuals(x, p3, p2n, n, r, wx, wy)) return;
    double lam = 1e-3;
    const int m = 2 * n;
    for (int it = 0; it < iters; ++it) {
        for (int j = 0; j < 6; ++j) {
            double xd[6];
            for (int k = 0; k < 6; ++k) xd[k] = x[k];
            xd[j] += 1e-6;
            residuals(xd, p3, p2n, n, rn, wx, wy);
            for (int i = 0; i < m; ++i) {
                const double d = (rn[i] - r[i]) / 1e-6;
                J[i * 6 + j] = finite_d(d) ? d : 0.0;
            }
        }
        double H[36], g[6], step[6];
        for (int a = 0; a < 6; ++a) {
            double s = 0.0;
            for (int i = 0; i < m; ++i) s += J[i * 6 + a] * r[i];
            g[a] = -s;
            for (int b = 0; b < 6; ++b) {
                double h = 0.0;
                for (int i = 0; i < m; ++i) h += J[i * 6 + a] * J[i * 6 + b];
                H[a * 6 + b] = h;
            }
        }
        for (int a = 0; a < 6; ++a) H[a * 6 + a] += lam * (H[a * 6 + a] + 1e-12);
        if (!solve6(H, g, step)) break;
        double xn[6], e0 = 0.0, e1 = 0.0, sn = 0.0;
        for (int k = 0; k < 6; ++k) { xn[k] = x[k] + step[k]; sn += step[k] * step[k]; }
        const bool fin = residuals(xn, p3, p2n, n, rn, wx, wy);
        for (int i = 0; i < m; ++i) { e0 += r[i] * r[i]; e1 += rn[i] * rn[i]; }
        if (fin && e1 < e0) {
            for (int k = 0; k < 6; ++k) x[k] = xn[k];
            for (int i = 0; i < m; ++i) r[i] = rn[i];
            lam = lam * 0.3 > 1e-9 ? lam * 0.3 : 1e-9;
            if (sqrt(sn) < 1e-10) break;
        } else {
            lam *= 10.0;
        }
    }
    rodrigues(x, R);
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) out[i * 4 + j] = (float)R[i * 3 + j];
        out[i * 4 + 3] = (float)x[3 + i];
    }
    out[15] = 1.f;
}

__global__ __launch_bounds__(64) void pnp_kernel(const float* __restrict__ kp, const float* __restrict__ pts3,
                                                 const float* __restrict__ Kmat, int N, int n, int iters,
                                                 float* __restrict__ poses) {
    const int id = blockIdx.x * 64 + threadIdx.x;
    if (id >= N) return;
    solve_one_pose(kp + (size_t)id * n * 2, pts3 + (size_t)id * n * 3, Kmat + (size_t)id * 9, n, iters, poses + (size_t)id * 16);
}

}  // namespace

// Host worker threads of bd_solve_pnp_host, kept between calls (round 6: starting seven std::threads per call was ~0.25 ms of the 0.36 ms
// a batch of 32 poses took -- device idle time in a facade forward, profiles/r6_facade_gaps.txt).  Library-owned HOST state (no device state):
// up to 15 detached threads per process, parked on a condition variable; one call at a time uses them (a mutex serialises callers); a forked
// child finds the parent's pid in the pool and starts its own.  Never torn down.
namespace {
struct HostPool {
    std::mutex m;
    std::condition_variable wake, done;
    const std::function<void(int)>* job = nullptr;
    int threads = 0, nt = 0, gen = 0, pending = 0;
    pid_t pid = 0;
};
std::mutex g_pool_mu;
HostPool* g_pool = nullptr;

void host_pool_worker(HostPool* P, int t) {
    int seen = 0;
    for (;;) {
        std::unique_lock<std::mutex> lk(P->m);
        P->wake.wait(lk, [&] { return P->gen != seen; });
        seen = P->gen;
        const std::function<void(int)>* job = P->job;
        const bool mine = t < P->nt;
        lk.unlock();
        if (!mine) continue;
        (*job)(t);
        lk.lock();
        if (--P->pending == 0) P->done.notify_one();
    }
}

void host_pool_run(int nt, const std::function<void(int)>& job) {      // job(0) on the caller, job(1 .. nt-1) on the pool
    std::lock_guard<std::mutex> callers(g_pool_mu);
    if (!g_pool || g_pool->pid != getpid() || g_pool->threads < nt - 1) {
        HostPool* P = new HostPool();                                    // (an older / inherited pool is left behind, its threads parked or gone)
        P->pid = getpid();
        P->threads = nt - 1 > 15 ? nt - 1 : 15;
        for (int t = 1; t <= P->threads; ++t) std::thread(host_pool_worker, P, t).detach();
        g_pool = P;
    }
    HostPool* P = g_pool;
    {
        std::lock_guard<std::mutex> lk(P->m);
        P->job = &job;
        P->nt = nt;
        P->pending = nt - 1;
        ++P->gen;
    }
    P->wake.notify_all();
    job(0);
    std::unique_lock<std::mutex> lk(P->m);
    P->done.wait(lk, [&] { return P->pending == 0; });
    P->job = nullptr;
}
}  // namespace

// The host form ("PnP post-solve stays on the host CPU", north_star): the same per-pose solver on a pool of host threads, poses
// dealt out in contiguous chunks.  All pointers are HOST pointers.  Replaces the per-sample Python loop around cv2.solvePnP of
// src/models/utils/box_utils.py:139-199 when OpenCV is not importable (and the single-threaded numpy restatement of round 2:
// 9 ms per 32 poses).  n_threads <= 0: one thread per 4 poses, at most 16.
extern "C" int bd_solve_pnp_host(const float* kp_px, const float* pts3, const float* K, int n_poses, int n_points, int iters,
                                 float* poses, int n_threads) {
    if (!kp_px || !pts3 || !K || !poses) return BD_ERR_NULL;
    if (n_poses <= 0 || n_points < 6 || n_points > MAXPTS || iters < 0) return BD_ERR_SHAPE;
    int nt = n_threads > 0 ? n_threads : (n_poses + 3) / 4;
    nt = nt > 16 ? 16 : (nt > n_poses ? n_poses : nt);
    auto work = [=](int t) {
        const int lo = (int)((int64_t)n_poses * t / nt), hi = (int)((int64_t)n_poses * (t + 1) / nt);
        for (int i = lo; i < hi; ++i)
            solve_one_pose(kp_px + (size_t)i * n_points * 2, pts3 + (size_t)i * n_points * 3, K + (size_t)i * 9, n_points, iters,
                           poses + (size_t)i * 16);
    };
    if (nt <= 1) { work(0); return BD_OK; }
    host_pool_run(nt, work);
    return BD_OK;
}

extern "C" int bd_solve_pnp(const float* kp_px, const float* pts3, const float* K, int n_poses, int n_points, int iters,
                            float* poses, void* stream) {
    if (!kp_px || !pts3 || !K || !poses) return BD_ERR_NULL;
    if (n_poses <= 0 || n_points < 6 || n_points > MAXPTS || iters < 0) return BD_ERR_SHAPE;
    hipLaunchKernelGGL(pnp_kernel, dim3((n_poses + 63) / 64), dim3(64), 0, (hipStream_t)stream, kp_px, pts3, K, n_poses,
                       n_points, iters, poses);
    BD_CHECK_LAUNCH();
    return BD_OK;
}
