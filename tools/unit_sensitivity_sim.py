"""Per-unit DEMOTION sensitivity of the default precision mode: what does ONE Linear cost when it leaves the F16C8
class for a single f16 MFMA pass (the third level of a single-f16 / F16C8 / split-f16 assignment)?

CPU emulation at full depth on top of the default mode's policy (`f16c8_qk16`: F16C8 Linears with the kernel's fixed scales, BETR's q, k
columns one f16 pass with f16 results), operands rounded exactly as the kernels round them, fp32 accumulation (oracle/numerics_sim.py).
For every Linear u of the path:

    e_demote[u] = max |logits(default policy, u as ONE f16 pass) - logits(fp32 oracle)|

next to e_default = max |logits(default policy) - logits(fp32 oracle)|.  Errors of independent units add roughly in quadrature, so the
"excess" sqrt(e_demote^2 - e_default^2) is what unit u spends of the budget; `work` is the unit's share of the Linear FLOPs of the path.
Build-container tool (TEST INFRASTRUCTURE: imports the oracle); writes profiles/r5_unit_sensitivity.json.

    python tools/unit_sensitivity_sim.py [--views 6] [--seeds 11,12] [--out profiles/r5_unit_sensitivity.json]
"""
import argparse
import json
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boxdreamer_amd import synth                                    # noqa: E402
from oracle import boxdreamer_oracle as orc, numerics_sim as ns    # noqa: E402

f16 = ns.make_linear("f16")
c8 = ns.make_linear("f16c8fix")


def unit_name(kind, n, shape):
    if kind == "qkv":
        return f"dino.{n}.qkv" if n < 12 else f"betr.{n - 12}.qkv"
    if kind == "proj768":
        if n < 12:
            return f"dino.{n}.proj"
        if n < 14:
            return "betr.adapter_fc1" if n == 12 else "betr.adapter_fc2"
        return f"betr.{n - 14}.proj"
    if kind in ("fc1", "fc2"):
        return f"dino.{n}.{kind}" if n < 12 else f"betr.{n - 12}.{kind}"
    return "betr.bbox_emb" if tuple(shape) == (768, 1568) else "betr.bbox_proj"


def policy(demote, seen):
    """default-mode policy with the units named in `demote` as ONE f16 pass; `seen` collects (name -> MACs) of every Linear call"""
    def fn(kind, n, x, w, b):
        name = unit_name(kind, n, w.shape)
        seen[name] = x.numel() // x.shape[-1] * w.shape[0] * w.shape[1]
        if name in demote:
            return f16(x, w, b)
        if kind == "qkv" and n >= 12:                   # BETR: q, k columns one f16 pass (f16 results), v F16C8
            y8, y16 = c8(x, w, b), f16(x, w, b)
            y = y8.clone()
            y[..., :1536] = y16[..., :1536].half().float()
            return y
        return None                                      # the run's default scheme (f16c8fix)
    return fn


def run(data, bsd, dsd, demote, seen):
    ns.POLICY["fn"] = policy(demote, seen)
    try:
        return ns.run("f16c8fix", data, bsd, dsd)
    finally:
        ns.POLICY.clear()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=6)
    ap.add_argument("--seeds", default="11,12")
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--out", default="profiles/r5_unit_sensitivity.json")
    ap.add_argument("--only", default="", help="comma-separated unit-name prefixes (debug)")
    ap.add_argument("--replan", action="store_true", help="re-derive the plans from the measurements already in --out")
    a = ap.parse_args()
    if a.replan:
        old = json.load(open(a.out))
        return finish(a, old["units"], old["e_default"], old["seeds"])
    torch.set_num_threads(a.threads)
    bsd, dsd = synth.betr_state_dict(1234, 12), synth.dino_state_dict(4321, 12)
    seeds = [int(s) for s in a.seeds.split(",")]
    rows, e_default, t0 = {}, [], time.time()
    for seed in seeds:
        data = synth.make_batch(seed=seed, B=1, T=a.views)
        with torch.no_grad():
            ref = orc.boxdreamer_forward(data, bsd, dsd)["logits"]
        seen = {}
        d0 = float((run(data, bsd, dsd, set(), seen)["logits"] - ref).abs().max())
        e_default.append(d0)
        total = float(sum(seen.values()))
        units = [u for u in seen if not a.only or any(u.startswith(p) for p in a.only.split(","))]
        print(f"seed {seed}: default policy {d0:.3e} vs fp32 oracle; {len(units)} units ({time.time() - t0:.0f}s)", flush=True)
        for u in units:
            e = float((run(data, bsd, dsd, {u}, {})["logits"] - ref).abs().max())
            r = rows.setdefault(u, {"work": round(seen[u] / total, 5), "e_demote": []})
            r["e_demote"].append(e)
            print(f"  {u:20s} work {r['work']:.4f}  e_demote {e:.3e}  excess {math.sqrt(max(e * e - d0 * d0, 0.0)):.3e}  ({time.time() - t0:.0f}s)", flush=True)
    finish(a, rows, e_default, seeds)


def finish(a, rows, e_default, seeds):
    d0 = max(e_default)
    for u, r in rows.items():
        r["e_demote_max"] = max(r["e_demote"])
        # what a demotion SAVES: the Linear's share of the Linear FLOPs that still runs as F16C8 -- BETR's q, k columns already are one
        # f16 pass in the default mode, only its v columns (a third) are left
        r["demotable_work"] = round(r["work"] / 3.0, 5) if (u.startswith("betr.") and u.endswith(".qkv")) else r["work"]
        r["excess"] = math.sqrt(max(r["e_demote_max"] ** 2 - d0 ** 2, 0.0))
    # greedy knapsack: demote as much work as possible while sqrt(d0^2 + sum excess^2) stays inside each budget
    order = sorted(rows, key=lambda u: rows[u]["excess"] ** 2 / max(rows[u]["demotable_work"], 1e-9))
    plans = {}
    for budget in (3e-4, 4e-4, 5e-4):
        acc, work, chosen = d0 * d0, 0.0, []
        for u in order:
            if acc + rows[u]["excess"] ** 2 <= budget * budget:
                acc += rows[u]["excess"] ** 2
                work += rows[u]["demotable_work"]
                chosen.append(u)
        # a demoted F16C8 Linear runs ~2.1x faster (445 -> 935 TF/s, profiles/r4_gemm_bench.txt); F16C8 Linears are ~75 % of the step
        plans[f"{budget:.0e}"] = {"units": chosen, "linear_work_demoted": round(work, 4), "predicted_error": round(math.sqrt(acc), 7),
                                  "predicted_step_gain": round(0.75 * work * (1 - 1 / 2.1), 4)}
    out = {"what": "per-unit cost of ONE f16 pass instead of F16C8 (CPU emulation at full depth, tools/unit_sensitivity_sim.py)",
           "views": a.views, "seeds": seeds, "e_default": e_default, "units": rows, "quadrature_plans": plans}
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(plans, indent=1))


if __name__ == "__main__":
    main()
