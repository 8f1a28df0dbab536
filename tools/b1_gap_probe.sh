#!/bin/bash
# How much of a B = 1 step (HIP graph replay) is spent BETWEEN kernels?  rocprofv3 kernel trace of the graphed bench loop; per replay: wall span of the
# step's kernels against the sum of their durations.
cd "$(dirname "$0")/.."
R=$(pwd); out=gpurun_out/b1gap; rm -rf $out; mkdir -p $out
for pr in ${1:-f16c8_qk16}; do
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$out/$pr -- python $R/bench.py --prec $pr --batch 1 --in-flight 1 --lanes 1 --steps 40 --warmup 10 --no-strict --no-fp8 --no-cpu-baseline --no-pnp --no-h2d --no-parity --no-inline-counters --no-power --no-trained-like --no-latency > /dev/null 2>&1 )
f=$(find $out/$pr -name "*kernel_trace.csv" | head -1)
python - "$f" $pr <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the timed replays are the tail of the trace: find the decode kernel (one per step) and cut steps at it
idx = [i for i, r in enumerate(rows) if "decode_kernel" in r["Kernel_Name"]]
steps = []
for a, b in zip(idx[-21:-1], idx[-20:]):
    seg = rows[a + 1:b + 1]
    span = int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
    gaps = [int(seg[i + 1]["Start_Timestamp"]) - int(seg[i]["End_Timestamp"]) for i in range(len(seg) - 1)]
    steps.append((span, busy, len(seg), sum(g for g in gaps if g > 0), sorted(gaps)[len(gaps) // 2]))
n = len(steps)
print(sys.argv[2], "replays", n, "kernels/step", steps[0][2], "span us", round(sum(s[0] for s in steps) / n / 1e3, 1), "sum of kernel durations us", round(sum(s[1] for s in steps) / n / 1e3, 1),
      "sum of positive gaps us", round(sum(s[3] for s in steps) / n / 1e3, 1), "median gap ns", steps[0][4])
PY
done
