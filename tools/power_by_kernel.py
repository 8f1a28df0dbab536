"""Board power / reported clock while ONE kernel family runs back to back (sysfs hwmon, as bench.py's power_probe): which launches of the
step sit at the 1400 W cap?   python tools/power_by_kernel.py   -> one line per kernel (profiles/r4_power_by_kernel.md)"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from boxdreamer_amd import hip_ops

dev = torch.device("cuda", 0)
h = bench._hwmon_of(dev)
assert h, "no hwmon node for cuda:0"
rd = lambda n: int(open(os.path.join(h, n)).read())


def probe(name, fn, flop=0.0, seconds=2.0):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            samples.append((time.perf_counter(), rd("power1_input") / 1e6, rd("freq1_input") / 1e6))
            time.sleep(0.005)
    th = threading.Thread(target=sampler, daemon=True)
    t0 = time.perf_counter(); th.start(); n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(8):
            fn()
        n += 8
        torch.cuda.synchronize()
    t1 = time.perf_counter(); stop.set(); th.join()
    late = [(w, f) for t, w, f in samples if t - t0 > 0.5 * (t1 - t0)]
    w = sum(x for x, _ in late) / len(late); f = sum(x for _, x in late) / len(late)
    us = (t1 - t0) / n * 1e6
    print(f"{name:34s} {us:8.1f} us/launch  {flop / us / 1e6 if flop else 0:7.0f} TF/s   {w:7.1f} W avg ({max(x for x, _ in late):.0f} max, cap {rd('power1_cap') / 1e6:.0f})   sclk reported {f:5.0f} MHz", flush=True)
    time.sleep(1.0)


def gemm_case(prec, M, N, K, act, resid):
    a = hip_ops.to_operand(torch.randn(M, K, device=dev), prec)
    wf = torch.randn(N, K, device=dev) * 0.05
    qe = hip_ops.f16c8_qexp(wf) if prec == "f16c8" else 0
    w = hip_ops.f16c8_encode(wf, qe, True) if prec == "f16c8" else hip_ops.to_operand(wf, prec)
    b = torch.randn(N, device=dev)
    r = torch.randn(M, N, device=dev) if resid else None
    o = hip_ops.gemm(a, w, b, prec=prec, act=act, out_f32=resid, resid=r, out=r, w_qexp=qe)
    return lambda: hip_ops.gemm(a, w, b, prec=prec, act=act, out_f32=resid, resid=r, out=o, w_qexp=qe)


print("idle", rd("power1_input") / 1e6, "W")
M = 49152
probe("gemm bf16 qkv (N 2304, K 768)", gemm_case("bf16", M, 2304, 768, 0, False), 2.0 * M * 2304 * 768)
probe("gemm bf16 fc1+gelu (N 3072, K 768)", gemm_case("bf16", M, 3072, 768, 1, False), 2.0 * M * 3072 * 768)
probe("gemm bf16 fc2+res (N 768, K 3072)", gemm_case("bf16", M, 768, 3072, 0, True), 2.0 * M * 768 * 3072)
probe("gemm bf16 proj+res (N 768, K 768)", gemm_case("bf16", M, 768, 768, 0, True), 2.0 * M * 768 * 768)
probe("gemm f16c8 qkv", gemm_case("f16c8", M, 2304, 768, 0, False), 2.0 * M * 2304 * 768)
probe("gemm f16c8 fc2+res", gemm_case("f16c8", M, 768, 3072, 0, True), 2.0 * M * 768 * 3072)
try:
    probe("gemm fp8 fc1+gelu", gemm_case("fp8", M, 3072, 768, 1, False), 2.0 * M * 3072 * 768)
except Exception as e:  # noqa: BLE001
    print("fp8 gemm probe skipped:", e)
for prec in ("bf16", "bf16x3"):
    for batch, seq, heads, hd in ((32, 1536, 8, 96), (192, 261, 12, 64)):
        if prec == "bf16x3" and seq == 1536:
            continue
        qkv = hip_ops.to_operand(torch.randn(batch * seq, 3 * heads * hd, device=dev), prec)
        probe(f"attention {prec} seq {seq} hd {hd}", lambda: hip_ops.attention(qkv, batch, seq, heads, hd, hd ** -0.5, prec=prec),
              4.0 * seq * seq * hd * heads * batch)
x = torch.randn(M, 768, device=dev); g = torch.ones(768, device=dev); bb = torch.zeros(768, device=dev)
probe("layernorm bf16 (49152 x 768)", lambda: hip_ops.layernorm(x, g, bb, 1e-6, prec="bf16"))
