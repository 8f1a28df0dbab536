#!/bin/bash
# One pass over everything profiles/ quotes for a round (run on the GPU box from the repo root; writes under gpurun_out/final_<tag>/).
tag=${1:-r4b}
out=gpurun_out/final_$tag; mkdir -p $out
R=$(pwd)
timeout 3000 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; echo "pytest rc $?" | tee $out/status; tail -3 $out/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $out/status
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
for extra in "--prec f16x3" "--prec f16x3_attn_x3" "--prec f16c8_qkv16" "--prec f16c8" "--prec bf16x3" "--prec fp16" "--views 17" "--batch 1 --in-flight 1" "--cache-refs" "--prec f16c8_qk16 --views 17" "--prec fp8_mixed --batch 64 --in-flight 1" "--prec fp8 --batch 64 --in-flight 1"; do
  python bench.py $extra --no-strict --no-fp8 --no-cpu-baseline --no-pnp --no-h2d --no-trained-like 2>/dev/null | grep '^{' >> $out/bench_variants.jsonl; echo "bench $extra rc $?"
done
for pr in bf16 f16c8; do echo "== gemm_bench $pr"; python tools/gemm_bench.py $pr 2>&1 | grep -v amdgpu; done > $out/gemm_bench.txt
for pr in bf16 bf16x3; do echo "== attn_probe $pr"; python tools/attn_probe.py $pr 2>&1 | grep -v amdgpu; done > $out/attn_probe.txt
cp gpurun_out/strict_margin*.json gpurun_out/parity_report.json gpurun_out/calibration_outliers_g*.json gpurun_out/fp8_mixed_report.json $out/ 2>/dev/null
python - <<PY
import json
j=json.load(open('$out/bench_default.json')); s=j['strict']; f=j.get('fp8',{})
print('bf16', j['value'], j['single_stream'], j['roofline']['achieved'], j['roofline']['traffic_over_algorithmic'], j['roofline']['mfma_busy'])
print('strict', s['value'], s['single_stream'], s['roofline']['achieved'], s['roofline']['traffic_over_algorithmic'], s['roofline']['mfma_busy'], s['parity']['logits_max_abs_err'])
print('fp8', f.get('value'), f.get('roofline',{}).get('achieved'))
for l in open('$out/bench_variants.jsonl'):
    v=json.loads(l); print(v['dtype'][:40], '|', v['config']['workload'][:30], '|', v['value'], v.get('single_stream'), v['ms_per_step'], (v.get('parity') or {}).get('logits_max_abs_err'))
PY
