#!/bin/bash
# Round 5 same-box A/B: the F16C8 GEMM's SMALL form (128 x 96 tiles, 2 + 2 waves, two workgroups per CU) for launches with few tiles -- the default
# mode one pose at a time (B = 1 .. 4) -- against the previous build (tools/_probe/libbd_b1base.so: 256 x 192 tiles for every shape).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
echo "== F16C8 GEMM op tests + small-form bit-identity + batch invariance of the path"
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "f16c8" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_path.py tests/test_gpu_facade.py tests/test_gpu_lanes.py tests/test_gpu_promote.py -x -q -m gpu 2>&1 | tail -2
for rep in 1 2; do
  for v in base new; do
    if [ $v = new ]; then unset BOXDREAMER_HIP_LIB; else export BOXDREAMER_HIP_LIB=tools/_probe/libbd_b1base.so; fi
    for B in 1 2 4 8; do
      timeout 300 python bench.py --prec f16c8_qk16 --batch $B --no-strict --no-fp8 --no-cpu-baseline --no-pnp --no-h2d --no-inline-counters --no-trained-like --no-power --steps 40 --warmup 10 2>/dev/null | grep '^{' > /tmp/ab.json
      python -c "
import json; j=json.load(open('/tmp/ab.json')); print('$v rep $rep default mode B=$B: ms/step', j['ms_per_step'], 'poses/s', j['value'], 'err', (j.get('parity') or {}).get('logits_max_abs_err'))"
    done
  done
done
for v in base new; do
  if [ $v = new ]; then unset BOXDREAMER_HIP_LIB; else export BOXDREAMER_HIP_LIB=tools/_probe/libbd_b1base.so; fi
  timeout 600 python bench.py --prec f16c8_qk16 --no-strict --no-fp8 --no-cpu-baseline --no-pnp --no-h2d --no-inline-counters --no-trained-like --no-power --steps 10 --warmup 3 2>/dev/null | grep '^{' > /tmp/ab.json
  python -c "
import json; j=json.load(open('/tmp/ab.json')); print('$v default mode B=32: poses/s', j['value'], 'one lane', j.get('value_single_stream'))"
done
