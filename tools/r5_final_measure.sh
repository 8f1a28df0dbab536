#!/bin/bash
# Round 5 evidence pass on one box: GPU suite (with durations), smoke, race screen, rocprofv3 kernel stats of the three benched modes,
# all counter groups per mode, the default bench line (driver's command) and the same line with the configs[3] leg.
tag=${1:-r5}
out=gpurun_out/final_$tag; mkdir -p $out
R=$(pwd)
timeout 1500 python -m pytest tests -m gpu -q --durations=40 > $out/pytest.log 2>&1; echo "pytest rc $?" | tee $out/status; tail -3 $out/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $out/status
{ timeout 600 python tools/stress_determinism.py bf16 40 32 auto; timeout 600 python tools/stress_determinism.py f16c8_qk16 30 32 auto; } 2>&1 | grep -v amdgpu | tee $out/race_screen.txt
( cd /tmp && export TMPDIR=/tmp
  for pr in bf16 f16c8_qk16 fp8; do
    b=32; [ $pr = fp8 ] && b=64
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof_$pr -- python $R/bench.py --prec $pr --batch $b --in-flight 1 --lanes 1 --steps 5 --warmup 2 --no-graph --no-strict --no-fp8 --no-cpu-baseline --no-pnp --no-h2d --no-parity --no-inline-counters --no-power --no-latency > /dev/null 2>&1
    echo "rocprof $pr rc $?"
    f=$(find $R/$out/prof_$pr -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/$out/${tag}_bench_${pr}_kernel_stats.csv
  done )
for pr in bf16 f16c8_qk16; do timeout 900 python bench.py --measure-counters --prec $pr > $out/counters_$pr.log 2>&1; echo "counters $pr rc $?" | tee -a $out/status; done
timeout 900 python bench.py --measure-counters --prec fp8 --batch 64 > $out/counters_fp8.log 2>&1; echo "counters fp8 rc $?" | tee -a $out/status
cp profiles/counters_*.json $out/ 2>/dev/null
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $out/bench_default.json 2> $out/bench_default.err; echo "bench default rc $?" | tee -a $out/status
python bench.py --config3 --no-strict --no-fp8 --no-cpu-baseline --no-pnp --no-h2d --no-inline-counters --no-trained-like --no-latency > $out/bench_config3.json 2> $out/bench_config3.err; echo "bench config3 rc $?" | tee -a $out/status
python bench.py --prec f16c8_qk16 --views 17 --no-strict --no-fp8 --no-cpu-baseline --no-pnp --no-h2d --no-inline-counters --no-trained-like --no-latency > $out/bench_T17_default.json 2> /dev/null; echo "bench T17 default rc $?" | tee -a $out/status
python - <<PY
import json
j=json.loads(open('$out/bench_default.json').read().strip().splitlines()[-1]); s=j['strict']; f=j.get('fp8',{})
print('bf16', j['value'], j.get('single_stream'), j['roofline']['achieved'], j['roofline']['traffic_over_algorithmic'], j['roofline']['mfma_busy'])
print('strict', s['value'], s.get('single_stream'), s['roofline']['achieved'], s['roofline']['traffic_over_algorithmic'], s['parity']['logits_max_abs_err'])
print('fp8', f.get('value'), f.get('roofline',{}).get('achieved'), f.get('roofline',{}).get('traffic_over_algorithmic'))
c=json.loads(open('$out/bench_config3.json').read().strip().splitlines()[-1]); print('config3', c.get('config3'))
PY
tail -4 $out/bench_default.err
