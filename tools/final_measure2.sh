#!/bin/bash
# Second final pass of round 4 (after the lanes + PnP changes): GPU suite, race screen incl. lanes, kernel stats, counters, default bench line.
tag=${1:-r4c}
out=gpurun_out/final_$tag; mkdir -p $out
R=$(pwd)
timeout 1500 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; echo "pytest rc $?" | tee $out/status; tail -3 $out/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $out/status
{ python tools/stress_determinism.py bf16 60 32 auto; python tools/stress_determinism.py f16c8_qk16 40 32 auto; python tools/stress_determinism.py bf16 30 16 auto;
  python tools/stress_determinism.py f16c8_qk16 20 32 3; python tools/stress_determinism.py fp8 20 32 2; } 2>&1 | grep -v amdgpu | tee $out/race_screen.txt
( cd /tmp && export TMPDIR=/tmp
  for pr in bf16 f16c8_qk16 fp8; do
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof_$pr -- python $R/bench.py --prec $pr --in-flight 1 --lanes 1 --steps 5 --warmup 2 --no-graph --no-strict --no-fp8 --no-cpu-baseline --no-pnp --no-h2d --no-parity > /dev/null 2>&1
    echo "rocprof $pr rc $?"
    f=$(find $R/$out/prof_$pr -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/$out/${tag}_bench_${pr}_kernel_stats.csv
  done )
for pr in bf16 f16c8_qk16; do timeout 900 python bench.py --measure-counters --prec $pr > $out/counters_$pr.log 2>&1; echo "counters $pr rc $?" | tee -a $out/status; done
timeout 900 python bench.py --measure-counters --prec fp8 --batch 64 > $out/counters_fp8.log 2>&1; echo "counters fp8 rc $?" | tee -a $out/status
cp profiles/counters_*.json $out/ 2>/dev/null
python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench default rc $?" | tee -a $out/status
python - <<PY
import json
j=json.load(open('$out/bench_default.json')); s=j['strict']; f=j.get('fp8',{})
print('bf16', j['value'], j.get('single_stream'), j['roofline']['achieved'], j['roofline']['traffic_over_algorithmic'], j['roofline']['mfma_busy'])
print('strict', s['value'], s.get('single_stream'), s['roofline']['achieved'], s['roofline']['traffic_over_algorithmic'], s['parity']['logits_max_abs_err'])
print('fp8', f.get('value'), f.get('roofline',{}).get('achieved'), f.get('roofline',{}).get('traffic_over_algorithmic'))
PY
