#!/bin/bash
# whole step with the persistent GEMMs pinned to fewer CUs (A/B builds, profiles/r4_cu_limit_probe.diff), 1 and 2 sub-batch lanes
F="--no-strict --no-fp8 --no-cpu-baseline --no-pnp --no-h2d --no-trained-like --no-parity --steps 30 --warmup 6 --in-flight 1"
for pr in bf16 f16c8_qk16; do
for lib in default cu224 cu192 cu128; do
for l in 2 1; do
  if [ $lib = default ]; then unset BOXDREAMER_HIP_LIB; else export BOXDREAMER_HIP_LIB=tools/_probe/libbd_$lib.so; fi
  python bench.py --prec $pr --lanes $l $F 2>/dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$pr', '$lib', 'lanes $l', j['value'], j['ms_per_step'])"
done; done; done | tee gpurun_out/cu_limit_lanes.txt
