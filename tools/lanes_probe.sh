#!/bin/bash
# sub-batch lanes of one batch: poses/s over (mode, batch, lanes), one box (profiles/r4_subbatch_lanes.md)
mkdir -p gpurun_out/lanes
F="--no-strict --no-fp8 --no-cpu-baseline --no-pnp --no-h2d --no-trained-like --no-parity --steps 30 --warmup 6 --in-flight 1"
for pr in bf16 f16c8_qk16; do
for cfg in "--batch 32" "--batch 16" "--batch 8" "--batch 4" "--batch 64" "--batch 32 --views 17" "--batch 32 --views 2"; do
for l in 1 2 3 4; do
  python bench.py --prec $pr $cfg --lanes $l $F 2>/dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$pr', '$cfg', 'lanes $l', j['value'], j['ms_per_step'], j['config'].get('sub_batch_lanes'))"
done; done; done | tee gpurun_out/lanes/lanes.txt
