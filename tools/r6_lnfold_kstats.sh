#!/bin/bash
# per-kernel time of the default mode's step with and without the LayerNorm fold (rocprofv3 --kernel-trace --stats, one lane, un-graphed)
R=$(cd "$(dirname "$0")/.." && pwd)
out=$R/gpurun_out/r6_lnfold_kstats; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for f in 0 1; do
  BOXDREAMER_HIP_LNFOLD=$f timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$f -- python $R/bench.py --prec f16c8_qk16 --batch 32 --in-flight 1 --lanes 1 --steps 5 --warmup 2 --no-graph --no-strict --no-fp8 --no-cpu-baseline --no-pnp --no-h2d --no-parity --no-inline-counters --no-power --no-latency --no-rccl-probe > /dev/null 2>&1
  echo "rocprof fold=$f rc $?"
  src=$(find $out/prof_$f -name "*kernel_stats.csv" | head -1); [ -n "$src" ] && cp $src $out/kernel_stats_fold$f.csv
  rm -rf $out/prof_$f
done
python - <<PY
import csv, re
def load(p):
    d = {}
    for r in csv.DictReader(open(p)):
        n = r["Name"]
        n = re.sub(r"void \(anonymous namespace\)::", "", n); n = re.sub(r"\(.*", "", n)
        d[n] = (int(r["Calls"]), int(r["TotalDurationNs"]))
    return d
a, b = load("$out/kernel_stats_fold0.csv"), load("$out/kernel_stats_fold1.csv")
keys = sorted(set(a) | set(b), key=lambda k: -(a.get(k, (0, 0))[1] + b.get(k, (0, 0))[1]))
ta = sum(v[1] for k, v in a.items() if not k.startswith("at::") and "rocclr" not in k); tb = sum(v[1] for k, v in b.items() if not k.startswith("at::") and "rocclr" not in k)
print(f"{'kernel':90s} {'calls0':>7s} {'ms0':>9s} {'calls1':>7s} {'ms1':>9s}")
for k in keys[:22]:
    ca, na = a.get(k, (0, 0)); cb, nb = b.get(k, (0, 0))
    print(f"{k[:90]:90s} {ca:7d} {na / 1e6:9.2f} {cb:7d} {nb / 1e6:9.2f}")
print("library kernels total ms:", ta / 1e6, tb / 1e6)
PY
