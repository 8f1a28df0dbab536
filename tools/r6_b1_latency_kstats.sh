#!/bin/bash
# Round 6: kernel stats of the default mode one pose at a time WITH the opt-in latency forms (bench.py --latency-forms), un-graphed, 20 steps.
R=$(cd "$(dirname "$0")/.." && pwd); out=$R/gpurun_out/r6_b1_latency; mkdir -p $out
COMMON="--no-strict --no-fp8 --no-cpu-baseline --no-pnp --no-h2d --no-parity --no-inline-counters --no-power --no-latency --no-rccl-probe --no-facade --no-trained-like --sustained 0"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python $R/bench.py --latency-forms --prec f16c8_qk16 --batch 1 --in-flight 1 --lanes 1 --steps 20 --warmup 5 --no-graph $COMMON > /dev/null 2>&1
echo "rocprof rc $?"
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/r6_b1_latency_forms_kernel_stats.csv; rm -rf $out/prof
python - $out/r6_b1_latency_forms_kernel_stats.csv <<'PY'
import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:18]:
    n=re.sub(r"void \(anonymous namespace\)::","",r['Name'])
    print(f"  {n[:100]:100s} {int(r['Calls']):5d} {float(r['AverageNs'])/1e3:8.1f}us")
PY
