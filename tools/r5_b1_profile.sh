#!/bin/bash
# Round 5: where a B = 1 pose (the reference demo's per-frame call: 1 query + 5 references) spends its time -- kernel stats of both modes at batch 1.
cd "$(dirname "$0")/.."
R=$(pwd); out=gpurun_out/b1; mkdir -p $out
for pr in bf16 f16c8_qk16; do
  python bench.py --prec $pr --batch 1 --no-strict --no-fp8 --no-cpu-baseline --no-pnp --no-h2d --no-inline-counters --no-trained-like --no-power --steps 50 --warmup 10 2>/dev/null | grep '^{' > $out/line_$pr.json
  python -c "
import json; j=json.load(open('$out/line_$pr.json')); print('$pr B=1 graph: ms/pose', j['ms_per_step'], 'poses/s', j['value'])"
  ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof_$pr -- python $R/bench.py --prec $pr --batch 1 --in-flight 1 --lanes 1 --steps 20 --warmup 5 --no-graph --no-strict --no-fp8 --no-cpu-baseline --no-pnp --no-h2d --no-parity --no-inline-counters --no-power --no-trained-like > /dev/null 2>&1 )
  f=$(find $out/prof_$pr -name "*kernel_stats.csv" | head -1); cp $f $out/b1_${pr}_kernel_stats.csv
  python - $out/b1_${pr}_kernel_stats.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:14]:
    print(f"  {r['Name'][:100]:100s} {int(r['Calls']):5d} {float(r['AverageNs'])/1e3:8.1f}us {float(r['Percentage']):6.2f}%")
print('  total kernel ms per step', tot/1e6/25)
PY
done
