#!/bin/bash
# Round 5 same-box A/B: the F16C8 GEMM's e4m3 conversions of A fed with the previous slab's registers as their "old" operand (no zeroing moves:
# 8 VALU instructions fewer per slab and wave) against the previous build (tools/_probe/libbd_base.so).  Bit-identical by construction.
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "f16c8" 2>&1 | tail -1
for rep in 1 2 3; do
  for v in base new; do
    if [ $v = new ]; then unset BOXDREAMER_HIP_LIB; else export BOXDREAMER_HIP_LIB=tools/_probe/libbd_base.so; fi
    echo "== $v f16c8 rep $rep"; timeout 300 python tools/gemm_bench.py f16c8 2>&1 | grep -E "weighted"
  done
done
for rep in 1 2; do
  for v in base new; do
    if [ $v = new ]; then unset BOXDREAMER_HIP_LIB; else export BOXDREAMER_HIP_LIB=tools/_probe/libbd_base.so; fi
    timeout 600 python bench.py --prec f16c8_qk16 --no-strict --no-fp8 --no-cpu-baseline --no-pnp --no-h2d --no-inline-counters --no-trained-like --no-latency --steps 10 --warmup 3 2>/dev/null | grep '^{' > /tmp/ab.json
    python -c "
import json; j=json.load(open('/tmp/ab.json')); print('$v rep $rep default step: poses/s', j['value'], 'one lane', j.get('value_single_stream'), 'err', (j.get('parity') or {}).get('logits_max_abs_err'))"
  done
done
