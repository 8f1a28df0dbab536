#!/bin/bash
# Round 5 same-box A/B: staggered consumer groups in the F16C8 persistent GEMM (prestag = the lock-step build) -- every step under its own timeout
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
A=${1:-prestag}
echo "== op tests (f16c8 + tile-shape independence)"; timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "f16c8 or tile_shape or sparse_last_round or promot" 2>&1 | tail -3
for rep in 1 2; do
  for v in $A default; do
    if [ $v = default ]; then unset BOXDREAMER_HIP_LIB; else export BOXDREAMER_HIP_LIB=tools/_probe/libbd_$v.so; fi
    echo "== $v rep $rep"; timeout 300 python tools/gemm_bench.py f16c8 2>&1 | grep -E "TF/s"
  done
done
unset BOXDREAMER_HIP_LIB
for rep in 1 2; do
  for v in $A default; do
    if [ $v = default ]; then unset BOXDREAMER_HIP_LIB; else export BOXDREAMER_HIP_LIB=tools/_probe/libbd_$v.so; fi
    timeout 600 python bench.py --prec f16c8_qk16 --no-strict --no-fp8 --no-cpu-baseline --no-pnp --no-h2d --no-inline-counters --no-trained-like --steps 10 --warmup 3 2>/dev/null | grep '^{' > /tmp/ab.json
    python -c "
import json; j=json.load(open('/tmp/ab.json')); p=j.get('power') or {}; print('$v rep $rep default-mode step: poses/s', j['value'], 'ms', j['ms_per_step'], 'one lane', j.get('value_single_stream'), 'gemm TF/s', j['roofline']['achieved'], 'err', j.get('logits_max_abs_err'), 'W', p.get('avg_w'), 'MHz', p.get('sclk_reported_mhz_avg'))"
  done
done
