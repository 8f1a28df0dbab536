#!/bin/bash
# default mode, same box: sub-batch lane settings after the LayerNorm fold (the fold removed the HBM-bound launches that one lane overlapped with the other's GEMMs)
cd "$(dirname "$0")/.."
COMMON="--prec f16c8_qk16 --no-strict --sustained 0 --no-facade --no-trained-like --no-fp8 --no-latency --no-cpu-baseline --no-inline-counters --no-h2d --no-pnp --no-rccl-probe --no-parity --no-power --steps 20 --warmup 5"
for l in auto 1 2 3 4 2,3 3,2; do
  timeout 600 python bench.py $COMMON --lanes $l 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lanes $l', j['value'], j['ms_per_step'], j['config']['sub_batch_lanes'], j.get('value_single_stream'))"
done
