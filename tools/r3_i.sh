#!/bin/bash
mkdir -p gpurun_out/r3i
R=$(pwd)
( cd /tmp && export TMPDIR=/tmp
  for pr in bf16 f16c8_qk16; do
    timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3i/prof_$pr -- python $R/bench.py --prec $pr --in-flight 1 --steps 5 --warmup 2 --no-graph --no-strict --no-fp8 --no-cpu-baseline --no-pnp --no-h2d --no-parity > /dev/null 2>&1
    echo "rocprof $pr rc $?"
    f=$(find $R/gpurun_out/r3i/prof_$pr -name "*kernel_stats.csv" | head -1); cp $f $R/gpurun_out/r3i/r3_bench_${pr}_kernel_stats.csv
    cut -c1-200 $f | head -30
  done )
timeout 900 python -m pytest tests/test_gpu_path.py tests/test_gpu_facade.py -m gpu -q -k "views17_full or multi_round" 2>&1 | tail -3
timeout 900 python bench.py > gpurun_out/r3i/bench.json 2> gpurun_out/r3i/bench.err; echo "bench rc $?"
python -c "
import json; j=json.load(open('gpurun_out/r3i/bench.json')); s=j['strict']; f=j['fp8']
print('bf16', j['value'], j['single_stream'])
print('strict', s['value'], s['single_stream'], s['parity']['logits_max_abs_err'])
print('fp8', f['value'], f['ms_per_step'], f['roofline']['achieved'], f['parity']['logits_max_abs_err'], f['workload'][:60])"
