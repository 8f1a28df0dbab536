"""Where does a GEMM tile's time go?  Measurement tool (not part of the product library).

    python tools/gemm_phase_probe.py build            # here (no GPU): tools/_probe/libbd_probe.so = the library built with -DBD_GEMM_PROBE
    python tools/gemm_phase_probe.py run M N K [prec] # on the GPU box

Every wave stamps the shader clock (s_memtime) at three points of every K-slab -- before the slab barrier, after it, after
issuing the next slab's LDS-DMA -- plus kernel start, mainloop end, epilogue start and end.  Printed: mean cycles per phase
over the workgroups of the first round, for the older (wave 0) and younger (wave 4) wave of SIMD 0."""
import ctypes as C
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "_probe", "libbd_probe.so")


def build():
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    src = os.path.join(ROOT, "boxdreamer_amd", "csrc")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-DBD_GEMM_PROBE",
           "-I", os.path.join(ROOT, "include"), "-shared", "-o", LIB] + sorted(
        os.path.join(src, f) for f in os.listdir(src) if f.endswith(".hip"))
    subprocess.check_call(cmd)
    print("built", LIB)


def run(M, N, K, prec="bf16"):
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    os.environ["BOXDREAMER_HIP_LIB"] = LIB
    from boxdreamer_amd import _lib, hip_ops
    lib = _lib.load()
    a = hip_ops.to_operand(torch.randn(M, K, device="cuda"), prec)
    wf = torch.randn(N, K, device="cuda") * 0.05
    qe = hip_ops.f16c8_qexp(wf) if prec == "f16c8" else 0
    w = hip_ops.f16c8_encode(wf, qe, True) if prec == "f16c8" else hip_ops.to_operand(wf, prec)
    b = torch.randn(N, device="cuda")
    nwave = 8
    buf = torch.zeros(1024 * nwave * 64, dtype=torch.int32, device="cuda")
    rms = None
    if os.environ.get("BD_PROBE_RMS") == "1":          # fused q/k RMSNorm: also forces the persistent kernel at ANY M (lone-CU runs)
        rms = (torch.ones(96, device="cuda"), torch.ones(96, device="cuda"), 1e-6)
    for _ in range(3):
        hip_ops.gemm(a, w, b, prec=prec, w_qexp=qe, rms=rms)
    for setter in ("bd_gemm_probe_set", "bd_gemm_f16c8_probe_set"):      # one probe buffer pointer per GEMM translation unit
        fn = getattr(lib, setter)
        fn.argtypes = [C.c_void_p]
        assert fn(C.c_void_p(buf.data_ptr())) == 0
    torch.cuda.synchronize()
    for _ in range(int(os.environ.get("BD_PROBE_LAUNCHES", "60"))):      # back to back: the stamps of the LAST launch survive,
        hip_ops.gemm(a, w, b, prec=prec, w_qexp=qe, rms=rms)                          # taken at sustained (DVFS-settled) clocks
    torch.cuda.synchronize()
    if os.environ.get("BD_PROBE_PC", "1") == "1":
        return report_pc(buf.cpu().numpy().astype(np.uint32).reshape(512, 16, 64), M, N, K, prec)
    ts = buf.cpu().numpy().astype(np.uint32).reshape(1024, nwave, 64)
    nk = min(K // (32 if prec in ("bf16x3", "f16c8") else (128 if prec == "fp8" else 64)), 20)
    first = ts[:256]                                   # first round: one workgroup per CU
    for wv in (0, 4):
        t = first[:, wv, :].astype(np.int64)
        d = lambda i, j: ((t[:, i] - t[:, j]) & 0xFFFFFFFF).astype(np.float64)
        bar = np.stack([d(3 * k + 1, 3 * k) for k in range(nk)], 1)
        dma = np.stack([d(3 * k + 2, 3 * k + 1) for k in range(nk)], 1)
        mma = np.stack([d(3 * (k + 1), 3 * k + 2) for k in range(nk - 1)] + [d(61, 3 * (nk - 1) + 2)], 1)
        print(f"wave {wv}: per-slab cycles (mean over 256 workgroups)")
        print("  barrier wait :", np.round(bar.mean(0)).astype(int).tolist())
        print("  DMA issue    :", np.round(dma.mean(0)).astype(int).tolist())
        print("  frags + MFMA :", np.round(mma.mean(0)).astype(int).tolist())
        print(f"  prologue {d(0, 60).mean():.0f}  mainloop {d(61, 0).mean():.0f}  sync {d(62, 61).mean():.0f}  epilogue {d(63, 62).mean():.0f}"
              f"  total {d(63, 60).mean():.0f} cycles;  slab mean: barrier {bar[:, 1:].mean():.0f} dma {dma[:, :-1].mean():.0f} mfma {mma.mean():.0f}")


def report_pc(ts, M, N, K, prec):
    """gemm_kernel_pc (8 consumer + 4 producer waves, persistent): first tile of each of the 256 workgroups."""
    import numpy as np
    nk = min(K // (32 if prec in ("bf16x3", "f16c8") else (128 if prec == "fp8" else 64)), 20)
    first = ts[:min(256, ((M + 255) // 256) * (N // 192))].astype(np.int64)
    d = lambda w, i, j: ((first[:, w, i] - first[:, w, j]) & 0xFFFFFFFF).astype(np.float64)
    for w in (0, 4):
        wait = np.stack([d(w, 3 * k + 1, 3 * k) for k in range(nk)], 1)
        comp = np.stack([d(w, 3 * (k + 1), 3 * k + 1) for k in range(nk - 1)], 1)
        print(f"consumer wave {w}: barrier wait {np.round(wait.mean(0)).astype(int).tolist()}")
        print(f"                 frags + MFMA {np.round(comp.mean(0)).astype(int).tolist()}")
        if nk * 3 <= 60:
            print(f"                 last slab {d(w, 60, 3 * (nk - 1) + 1).mean():.0f}  X wait {d(w, 61, 60).mean():.0f}  epilogue {d(w, 62, 61).mean():.0f}"
                  f"  tile total {d(w, 62, 0).mean():.0f}  slab mean {(wait[:, 1:].mean() + comp.mean()):.0f}")
    cyc, rt = d(0, 59, 58), d(0, 57, 56)
    print(f"whole kernel (wave 0): {cyc.mean():.0f} shader cycles in {rt.mean() / 100:.1f} us (s_memrealtime, 100 MHz) -> effective shader clock "
          f"{(cyc / rt).mean() * 0.1:.2f} GHz")
    for w in (8, 9):
        wait = np.stack([d(w, 3 * k + 1, 3 * k) for k in range(nk)], 1)
        iss = np.stack([d(w, 3 * k + 2, 3 * k + 1) for k in range(nk)], 1)
        land = np.stack([d(w, 3 * (k + 1), 3 * k + 2) for k in range(nk - 1)], 1)
        print(f"producer wave {w}: barrier wait {np.round(wait.mean(0)).astype(int).tolist()}")
        print(f"                  issue        {np.round(iss.mean(0)).astype(int).tolist()}")
        print(f"                  vmcnt wait   {np.round(land.mean(0)).astype(int).tolist()}")


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    else:
        run(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5] if len(sys.argv) > 5 else "bf16")
