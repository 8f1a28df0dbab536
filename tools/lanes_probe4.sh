#!/bin/bash
# small batches: where should "auto" start using two lanes? (profiles/r4_subbatch_lanes.md)
F="--no-strict --no-fp8 --no-cpu-baseline --no-pnp --no-h2d --no-trained-like --no-parity --no-power --steps 40 --warmup 8 --in-flight 1"
run() { python bench.py $1 --lanes $2 $F 2>/dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$1', 'lanes $2', j['value'], j['ms_per_step'])"; }
for rep in 1 2; do for pr in bf16 f16c8_qk16; do for b in 2 4 6 8 12; do for l in 1 2; do run "--prec $pr --batch $b" $l; done; done; done; done
