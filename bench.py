#!/usr/bin/env python
"""bench.py -- poses/s of the corner-heatmap inference path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--prec bf16|fp16|bf16x3] [--batch B] [--views T]

One "step" = one pass of the hot path over one batch of synthetic input already resident in HBM:
DINOv2 ViT-B/14-reg encoder on all B*T crops -> BETR decoder -> corner decode (top-20 mean), plus for
N > 1 the RCCL all-gather of predicted corners.  Default workload = BASELINE.json configs[1]:
1 query + 5 refs, 224x224, batch 32 per GPU.  One process per GPU (torch.distributed.run), batch
sharded across ranks (independent samples -> weak scaling, no data-path collective besides the
corner gather).  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# peaks from /opt/skills/guides/MI355X_MICROARCH.md (dense, no sparsity)
PEAK_MFMA_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0, "bf16x3": 2500.0, "fp8": 5000.0}   # fp8 = MX-scaled dense peak
PEAK_HBM_GBS = 8000.0
DINO_FLOP_PER_IMAGE = 47_078_313_984          # BASELINE.md §4 / SURVEY.md §8(d)


def betr_flops(T: int) -> int:
    S = 256 * T
    return S * (4 * 768 ** 2 + 2 * 1568 * 768) + 12 * (S * 14_155_776 + 3072 * S * S) + 2 * 256 * 768 * 1568


def flops_per_pose(T: int) -> int:
    return T * DINO_FLOP_PER_IMAGE + betr_flops(T)


def build_models(prec, device):
    from boxdreamer_amd import synth
    from boxdreamer_amd.betr import BETR
    from boxdreamer_amd.encoder import DinoV2Wrapper
    enc = DinoV2Wrapper(None, {"model_type": "dinov2_vitb14_reg", "synthetic_seed": 4321, "hip_precision": prec})
    enc.to_device(device)
    dec = BETR(d_model=768, nhead=8, num_decoder_layers=12, decoder_only=True, patch_size=14, img_size=224,
               diff_emb=False, nvs_supervision=False, ray_supervision=True, use_mask=False, use_pretrained=True,
               patchify_rays=True, pose_representation="bb8", bbox_representation="heatmap", hip_precision=prec)
    dec.load_state_dict(synth.betr_state_dict(seed=1234, depth=12), strict=True)
    return enc, dec.to(device).eval()


def usable_cpus() -> int:
    """CPUs this process may actually use: min(affinity, cgroup cpu.max quota).  (The GPU box shows 256 logical
    CPUs but a 16-CPU cgroup quota; 256 torch threads there run 100x slower than 16.)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(T: int, budget_s: float = 20.0) -> dict:
    """The oracle (CPU restatement of the reference arithmetic, torch fp32) timed on this box's host cores on a
    bounded sample of the same workload: single poses (B=1) with T views, full depth."""
    from boxdreamer_amd import synth
    from oracle import boxdreamer_oracle as orc
    cores = usable_cpus()
    torch.set_num_threads(cores)
    bsd, dsd = synth.betr_state_dict(1234, 12), synth.dino_state_dict(4321, 12)
    data = synth.make_batch(seed=11, B=1, T=T)
    with torch.no_grad():
        orc.boxdreamer_forward(data, bsd, dsd)          # warm-up
        n, t0 = 0, time.perf_counter()
        while True:
            orc.boxdreamer_forward(data, bsd, dsd)
            n += 1
            dt = time.perf_counter() - t0
            if dt > budget_s or n >= 64:
                break
    return {"value": n / dt, "unit": "poses/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} single-pose forwards (B=1, T={T}, 224x224, fp32, full depth) of oracle/boxdreamer_oracle.py "
                      f"in {dt:.1f}s, torch {torch.__version__} CPU"}


def parity_probe(prec, T: int, device) -> dict:
    """Max-abs error of the heatmap logits vs the CPU oracle on one full-depth pose (outside the timed region)."""
    from boxdreamer_amd import hip_ops, synth
    from oracle import boxdreamer_oracle as orc
    enc, dec = build_models(prec, device)
    data = synth.make_batch(seed=11, B=1, T=T)
    mask = torch.zeros(1, T, dtype=torch.bool); mask[0, T - 1] = True
    img, bf = data["images"].to(device), data["bbox_feat"].to(device)
    heat = dec(bf, img, mask.to(device), enc.predict(img), None)
    _, _, idx = hip_ops.decode_topk(heat)
    with torch.no_grad():
        o = orc.boxdreamer_forward(data, synth.betr_state_dict(1234, 12), synth.dino_state_dict(4321, 12))
    same = (idx.cpu().long().sort(-1)[0] == o["topk_idx"].sort(-1)[0]).all(-1).float().mean().item()
    return {"mode": prec, "logits_max_abs_err": float((dec.last_logits.cpu() - o["logits"]).abs().max()),
            "heat_max_abs_err": float((heat.cpu() - o["heat"]).abs().max()),
            "top20_sets_equal_frac": same, "case": f"B=1,T={T} full depth vs CPU oracle (fp32)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--prec", default=os.environ.get("BOXDREAMER_HIP_PREC", "bf16"), choices=["bf16", "fp16", "bf16x3", "fp8"])
    ap.add_argument("--batch", type=int, default=32, help="samples per GPU")
    ap.add_argument("--views", type=int, default=6, help="T = refs + 1")
    ap.add_argument("--cache-refs", action="store_true",
                    help="'next' row f1: reference features encoded once outside the timed region; per step the encoder "
                         "sees only the query crops (different algorithmic FLOPs -> reported as its own metric)")
    ap.add_argument("--streams", type=int, default=1,
                    help="split the per-GPU batch into this many independent sub-batches, each on its own HIP stream "
                         "(samples are independent units; overlaps memory-bound phases of one with MFMA phases of another)")
    ap.add_argument("--graph", dest="graph", action="store_true", default=None,
                    help="replay the step from a captured HIP graph (default for the plain single-stream step)")
    ap.add_argument("--no-graph", dest="graph", action="store_false", help="launch every kernel from the host each step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-pnp", action="store_true", help="skip the PnP-inclusive side measurement (host PnP, SURVEY 8d / 8f3)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (MI355X); there is no CPU fallback for the product path")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    n_gpus = world

    from boxdreamer_amd import _lib, hip_ops, synth
    from boxdreamer_amd.dist import gather_corners
    lib = _lib.load()
    B, T, prec = args.batch, args.views, args.prec
    enc, dec = build_models(prec, device)
    nstream = max(1, args.streams)
    lanes = [(enc, dec, torch.cuda.current_stream(device))]
    for _ in range(nstream - 1):
        e2, d2 = build_models(prec, device)
        lanes.append((e2, d2, torch.cuda.Stream(device=device)))
    # synthetic batch: every rank gets its own shard (different seed), values pre-rounded to bf16 as the
    # reference dataset does; tensors are bf16 on device (the dataset's `precision`), resident before timing.
    one = synth.make_batch(seed=100 + rank, B=min(B, 4), T=T)
    reps = (B + one["images"].shape[0] - 1) // one["images"].shape[0]
    images = one["images"].repeat(reps, 1, 1, 1, 1)[:B].to(torch.bfloat16).to(device)
    bbox = one["bbox_feat"].repeat(reps, 1, 1, 1, 1)[:B].to(torch.bfloat16).to(device)
    mask = torch.zeros(B, T, dtype=torch.bool, device=device); mask[:, T - 1] = True

    cached = None
    if args.cache_refs:
        from boxdreamer_amd.cache import RefFeatureCache, merge_cached_features
        cache = RefFeatureCache(enc)
        qidx = torch.full((B,), T - 1, dtype=torch.long, device=device)
        cached = cache.place(cache.encode(images[:, : T - 1]), qidx, T)
        if max(1, args.streams) > 1:
            raise SystemExit("--cache-refs with --streams > 1 is not wired up")

    from boxdreamer_amd.dist import shard_range
    spans = [shard_range(B, i, nstream) for i in range(nstream)]
    kp_all = torch.empty((B, 8, 2), dtype=torch.float32, device=device)

    def run_span(e, d, lo, hi):
        if cached is not None:          # (single stream only: slicing would drop the attached operand-dtype copy)
            feats = merge_cached_features(e, images, cached[0], cached[1])
        else:
            feats = e.predict(images[lo:hi])
        heat = d(bbox[lo:hi], images[lo:hi], mask[lo:hi], feats, None)
        kp, kn, _ = hip_ops.decode_topk(heat, want_idx=False)
        kp_all[lo:hi] = kp

    # The step is ~300 launches; replaying it from one HIP graph removes the host launch cost and most inter-kernel
    # gaps (B = 32: 31.5 -> 30.5 ms).  HIP event records cannot be timed inside a captured graph (ROCm 7.2), so in graph
    # mode the per-launch durations behind `roofline` come from TRACE_STEPS un-graphed executions of the same step on the
    # same buffers immediately after the timed region; --no-graph measures them inside the timed region itself.
    cap = 4096
    graphed = None
    if args.graph is None:
        args.graph = nstream == 1 and not args.cache_refs
    if args.graph:
        if nstream > 1 or args.cache_refs:
            raise SystemExit("--graph is wired for the plain single-stream step")
        from boxdreamer_amd.graph import GraphedPath
        graphed = GraphedPath(enc, dec, B, T, 224, torch.bfloat16, device)
        graphed.set_inputs(images, bbox)

    def step():
        if graphed is not None:
            kp = graphed.replay()[1]
            return gather_corners(kp, world) if world > 1 else kp
        main = torch.cuda.current_stream(device)
        if nstream == 1:
            run_span(enc, dec, 0, B)
        else:
            for (e, d, st), (lo, hi) in zip(lanes, spans):
                if st is not main:
                    st.wait_stream(main)
                with torch.cuda.stream(st):
                    run_span(e, d, lo, hi)
            for _, _, st in lanes:
                if st is not main:
                    main.wait_stream(st)
        if world > 1:
            return gather_corners(kp_all, world)
        return kp_all

    for _ in range(args.warmup):
        step()
    # ---- timed region: EXACTLY K steps between barrier+sync pairs; launch trace active on rank 0
    if rank == 0 and graphed is None:
        _lib.check(lib.bd_trace_begin(cap), "bd_trace_begin")
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    recs = []
    TRACE_STEPS = 3
    if rank == 0 and graphed is not None:
        _lib.check(lib.bd_trace_begin(cap), "bd_trace_begin")
        t1 = time.perf_counter()
        for _ in range(TRACE_STEPS):
            run_span(enc, dec, 0, B)
        torch.cuda.synchronize()
        traced_wall_ms = (time.perf_counter() - t1) * 1e3
    if rank == 0:
        buf = (_lib.TraceRecord * cap)()
        n = lib.bd_trace_end(buf, cap)
        recs = [(buf[i].kind, buf[i].M, buf[i].N, buf[i].K, buf[i].ms) for i in range(n)]
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    assert out.shape[0] == B * world and torch.isfinite(out).all()

    if rank == 0:
        poses = B * world * args.steps
        value = poses / dt
        fpp = flops_per_pose(T) if not args.cache_refs else DINO_FLOP_PER_IMAGE + betr_flops(T)
        # dominant kernel = the MFMA GEMM (kind 0).  Algorithmic FLOPs per launch = 2*M*N*K of that launch;
        # achieved = sum(flops) / sum(duration) = mean flops per launch / mean launch duration.
        g = [(2.0 * m * n * k, ms) for kind, m, n, k, ms in recs if kind == 0]
        a = [(4.0 * m * n * n * k, ms) for kind, m, n, k, ms in recs if kind == 1]   # 4*S^2*d per (batch*head)
        passes = 3.0 if prec == "bf16x3" else 1.0
        traced_ms = traced_wall_ms if graphed is not None else dt * 1e3
        peak = PEAK_MFMA_TFLOPS[prec]
        gemm_tf = sum(f for f, _ in g) / max(sum(ms for _, ms in g), 1e-9) / 1e9 if g else 0.0
        attn_tf = sum(f for f, _ in a) / max(sum(ms for _, ms in a), 1e-9) / 1e9 if a else 0.0
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "gemm_traffic.json")
        if os.path.exists(tpath) and prec == "bf16" and B == 32 and T == 6:
            tj = json.load(open(tpath))
            traffic, traffic_src = tj["hbm_bytes_per_launch"], tj["source"]
            if tj.get("steps_profiled") and g:      # per kernel launch -> per bd_gemm call (the unit of `achieved`)
                calls_per_step = len(g) / (TRACE_STEPS if graphed is not None else args.steps)
                traffic = round(tj["hbm_bytes_per_launch"] * tj["launches"] / tj["steps_profiled"] / calls_per_step)
        roofline = {"bound": "mfma", "kernel": "gemm_kernel_glds (256x256 / 128x128 tiles, LDS-DMA operands, v_mfma_f32_32x32x16)",
                    "achieved": round(gemm_tf, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(gemm_tf / peak, 4),
                    "traffic": traffic, "traffic_source": traffic_src,
                    "algorithmic_flops_per_launch": round(sum(f for f, _ in g) / max(len(g), 1)),
                    "launches": len(g),
                    "events": ("HIP events around every launch of %d un-graphed executions of the step right after the timed "
                               "region (events cannot be timed inside a captured graph)" % TRACE_STEPS) if graphed is not None
                              else "HIP events around every launch inside the timed region", "avg_launch_ms": round(sum(ms for _, ms in g) / max(len(g), 1), 4),
                    "mfma_passes_per_algorithmic_flop": passes,
                    "attention_achieved": round(attn_tf, 2),
                    "attention_avg_launch_ms": round(sum(ms for _, ms in a) / max(len(a), 1), 4),
                    "gemm_time_frac_of_step": round(sum(ms for _, ms in g) / traced_ms, 4),
                    "attention_time_frac_of_step": round(sum(ms for _, ms in a) / traced_ms, 4),
                    "whole_path_achieved": round(value / world * fpp / 1e12, 2),
                    "whole_path_frac": round(value / world * fpp / 1e12 / peak, 4)}
        metric = "poses/s/GPU (5-ref, 224×224, bf16); heatmap max-abs err vs CPU ref"     # BASELINE.json's metric string
        if args.cache_refs:
            metric = "poses/s with reference features cached across queries (SURVEY 8f1; encoder on the query crop only)"
        line = {"metric": metric,
                "value": round(value, 2), "unit": "poses/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": {"bf16": "bf16", "fp16": "f16", "bf16x3": "bf16x3", "fp8": "fp8-e4m3 (Linears) + bf16 (attention)"}[prec],
                "data": "synthetic",
                "config": {"workload": f"configs[1]: 1 query + {T - 1} ref, 224x224, batch {B}/GPU, DINOv2 ViT-B/14-reg "
                                       f"+ BETR-12 + top-20 decode, random-init weights, inputs bf16 in HBM",
                           "global_batch": B * world, "views": T, "parallelism": f"dp{world}", "streams_per_gpu": nstream, "hip_graph": bool(args.graph),
                           "gflop_per_pose": round(fpp / 1e9, 2)},
                "poses_per_s_per_gpu": round(value / world, 2),
                "value_is": "whole-job aggregate over n_gpus (bench contract); the per-GPU figure of the metric is poses_per_s_per_gpu",
                "roofline": roofline}
        if not args.no_pnp:
            # PnP-inclusive rate (SURVEY 8d asks for it next to `value`, never as `value`): one D2H of the decoded corners
            # per batch + ONE batched host solve (boxdreamer_amd/pnp.py, row f3).  Serialised = no overlap at all;
            # overlapped = PnP of batch i on the host while the GPU runs batch i+1.
            from boxdreamer_amd.box_utils import solve_poses_host
            b3 = one["bbox_3d"].float().reshape(-1, 8, 3)
            b3 = b3[-1:].repeat(B, 1, 1).numpy()
            Kq = one["non_ndc_intrinsics"].float().reshape(-1, 3, 3)[-1:].repeat(B, 1, 1).numpy()
            tp = []
            for _ in range(3):
                t1 = time.perf_counter()
                kp_host = out[:B].float().cpu().numpy()
                solve_poses_host(kp_host, b3, Kq)
                tp.append(time.perf_counter() - t1)
            pnp_ms = sorted(tp)[1] * 1e3
            step_ms = dt / args.steps * 1e3
            # the same solve as a HIP kernel (one pose per thread, fp64): corners stay on the device
            from boxdreamer_amd.box_utils import solve_poses_device
            b3d, Kd = torch.from_numpy(b3).to(device), torch.from_numpy(Kq).to(device)
            for _ in range(2):
                solve_poses_device(out[:B], b3d, Kd)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(5):
                solve_poses_device(out[:B], b3d, Kd)
            torch.cuda.synchronize()
            gpu_pnp_ms = (time.perf_counter() - t1) / 5 * 1e3
            line["pnp_inclusive"] = {"pnp_ms_per_batch": round(pnp_ms, 2), "host": "numpy batched DLT + LM, 1 thread, parity vs OpenCV un-pinned",
                                     "gpu_pnp_ms_per_batch": round(gpu_pnp_ms, 2),
                                     "gpu_pnp_serialised_poses_per_s": round(B * world / ((step_ms + gpu_pnp_ms) / 1e3), 1),
                                     "gpu_pnp": "bd_solve_pnp, 1 pose per thread fp64, same stream (random-weight corners: LM runs all 30 iterations)",
                                     "serialised_poses_per_s": round(B * world / ((step_ms + pnp_ms) / 1e3), 1),
                                     "overlapped_poses_per_s": round(min(value, B * world / (pnp_ms / 1e3)), 1)}
        if not args.no_parity:
            line["parity"] = parity_probe(prec, T, device)
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(T)
            line["gpu_over_cpu"] = round(value / line["cpu_baseline"]["value"], 1)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
