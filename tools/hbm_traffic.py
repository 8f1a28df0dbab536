"""Builds profiles/gemm_traffic.json from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) over bench.py.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out/f -- python bench.py --steps 2 --warmup 1 --no-graph ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d out/w -- python bench.py --steps 2 --warmup 1 --no-graph ...
    python tools/hbm_traffic.py out/f/<...>counter_collection.csv out/w/<...>counter_collection.csv profiles/gemm_traffic.json

Units / corrections as MI355X_MICROARCH.md (HBM section) prescribes: counters are KB; FETCH_SIZE reports half of the bytes of
wide coalesced reads on gfx950 -> x2; WRITE_SIZE as is."""
import collections
import csv
import json
import sys


def classify(name: str) -> str:
    for key, cls in (("gemm_kernel", "gemm"), ("attn_kernel", "attn"), ("layernorm_kernel", "layernorm"),
                     ("qk_rmsnorm", "rmsnorm")):
        if key in name:
            return cls
    return "other"


def read(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[classify(r["Kernel_Name"])].append(float(r["Counter_Value"]) * 1024.0)
    return acc


f, w, out = sys.argv[1:4]
fetch, write = read(f, "FETCH_SIZE"), read(w, "WRITE_SIZE")
per = {}
for cls in sorted(set(fetch) | set(write)):
    fl, wl = fetch.get(cls, []), write.get(cls, [])
    n = max(len(fl), len(wl), 1)
    fb, wb = 2.0 * sum(fl) / max(len(fl), 1), sum(wl) / max(len(wl), 1)
    per[cls] = {"launches": n, "fetch_bytes_per_launch": round(fb), "write_bytes_per_launch": round(wb),
                "hbm_bytes_per_launch": round(fb + wb)}
g = per.get("gemm", {"launches": 0, "fetch_bytes_per_launch": 0, "write_bytes_per_launch": 0, "hbm_bytes_per_launch": 0})
json.dump({"hbm_bytes_per_launch": g["hbm_bytes_per_launch"], "fetch_bytes_per_launch": g["fetch_bytes_per_launch"],
           "write_bytes_per_launch": g["write_bytes_per_launch"], "launches": g["launches"], "per_kernel_class": per,
           "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, --kernel-trace) over "
                     "`bench.py --steps 2 --warmup 1 --no-graph` (bf16, B=32, T=6); bytes = counter KB x 1024, FETCH_SIZE x2 "
                     "(gfx950 correction, MI355X_MICROARCH.md HBM section); mean over all gemm_kernel launches (tools/hbm_traffic.py)"},
          open(out, "w"), indent=1)
print(json.dumps(per, indent=1))
