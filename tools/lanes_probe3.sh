#!/bin/bash
# fp8 legs and mixed encoder / decoder lane counts (profiles/r4_subbatch_lanes.md)
F="--no-strict --no-fp8 --no-cpu-baseline --no-pnp --no-h2d --no-trained-like --no-parity --steps 30 --warmup 6 --in-flight 1"
run() { python bench.py $1 --lanes $2 $F 2>/dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$1', 'lanes $2', j['value'], j['ms_per_step'])"; }
for rep in 1 2; do
for cfg in "--prec fp8 --batch 64" "--prec fp8 --batch 32" "--prec fp8_mixed --batch 64"; do for l in 1 2; do run "$cfg" $l; done; done
done
for l in 2,2 4,2 3,2 4,1 2,2; do run "--prec bf16" $l; run "--prec f16c8_qk16" $l; done
