#!/bin/bash
# encoder / decoder lane counts separately (profiles/r4_subbatch_lanes.md)
mkdir -p gpurun_out/lanes
F="--no-strict --no-fp8 --no-cpu-baseline --no-pnp --no-h2d --no-trained-like --no-parity --steps 30 --warmup 6 --in-flight 1"
for pr in bf16 f16c8_qk16; do
for l in 1,1 2,1 1,2 2,2 2,2 1,1; do
  python bench.py --prec $pr --lanes $l $F 2>/dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$pr', 'lanes $l', j['value'], j['ms_per_step'])"
done; done | tee gpurun_out/lanes/lanes_enc_dec.txt
