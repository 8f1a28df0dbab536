#!/bin/bash
# Round 6 evidence pass on one box: GPU suite (with durations), smoke, race screen, rocprofv3 kernel stats of the benched modes (one lane,
# un-graphed) at B = 32 and of both modes at B = 1, all counter groups per mode, the default bench line (driver's command).
# Usage: tools/r6_final_measure.sh [tag] [skip-list: comma separated of suite,race,kstats,b1,counters,bench]
tag=${1:-r6}; skip=",${2:-},"
out=gpurun_out/final_$tag; mkdir -p $out
R=$(pwd)
has() { case "$skip" in *",$1,"*) return 0;; esac; return 1; }
COMMON="--no-strict --no-fp8 --no-cpu-baseline --no-pnp --no-h2d --no-parity --no-inline-counters --no-power --no-latency --no-rccl-probe --no-facade --no-trained-like --sustained 0"
if ! has suite; then
  timeout 1800 python -m pytest tests -m gpu -q --durations=30 > $out/pytest.log 2>&1; echo "pytest rc $?" | tee $out/status; tail -3 $out/pytest.log
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $out/status
fi
if ! has race; then
  { timeout 600 python tools/stress_determinism.py bf16 30 32 auto; timeout 600 python tools/stress_determinism.py f16c8_qk16 30 32 auto; } 2>&1 | grep -v amdgpu | tee $out/race_screen.txt
fi
if ! has kstats; then
( cd /tmp && export TMPDIR=/tmp
  for pr in f16c8_qk16 bf16 fp8; do
    b=32; [ $pr = fp8 ] && b=64
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof_$pr -- python $R/bench.py --prec $pr --batch $b --in-flight 1 --lanes 1 --steps 5 --warmup 2 --no-graph $COMMON > /dev/null 2>&1
    echo "rocprof $pr rc $?"
    f=$(find $R/$out/prof_$pr -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/$out/${tag}_bench_${pr}_kernel_stats.csv
    rm -rf $R/$out/prof_$pr
  done )
fi
if ! has b1; then
( cd /tmp && export TMPDIR=/tmp
  for pr in f16c8_qk16 bf16; do
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof_b1_$pr -- python $R/bench.py --prec $pr --batch 1 --in-flight 1 --lanes 1 --steps 20 --warmup 5 --no-graph $COMMON > /dev/null 2>&1
    echo "rocprof b1 $pr rc $?"
    f=$(find $R/$out/prof_b1_$pr -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/$out/${tag}_b1_${pr}_kernel_stats.csv
    rm -rf $R/$out/prof_b1_$pr
  done )
fi
if ! has counters; then
  for pr in bf16 f16c8_qk16; do timeout 900 python bench.py --measure-counters --prec $pr > $out/counters_$pr.log 2>&1; echo "counters $pr rc $?" | tee -a $out/status; done
  cp profiles/counters_*.json $out/ 2>/dev/null
fi
if ! has bench; then
  ( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $out/bench_default.json 2> $out/bench_default.err; echo "bench default rc $?" | tee -a $out/status
  python - <<PY
import json
j=json.loads(open('$out/bench_default.json').read().strip().splitlines()[-1]); c=j['config']
print('value', j['value'], j['ms_per_step'], 'mode', c.get('value_mode'), 'err', c.get('value_logits_max_abs_err'), 'sets', c.get('value_top20_sets_equal_frac'))
print('bf16', c.get('bf16_value'), 'facade', c.get('facade_poses_per_s'), 'sustained', c.get('sustained_last_5s_poses_per_s'), 'one pose', c.get('one_pose_ms'), c.get('bf16_one_pose_ms'))
print('roofline', {k: j['roofline'].get(k) for k in ('achieved', 'frac', 'traffic', 'mfma_busy', 'traffic_over_algorithmic')})
PY
  tail -4 $out/bench_default.err
fi
