#!/bin/bash
# Round 5 same-box A/B: F16C8 fc1 epilogue GELU forms (base = round 4's scalar erf; default build = packed erf; timing probes: none / exp-fit)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for rep in 1 2; do
  for v in base default gelu_none gelu_fast; do
    if [ $v = default ]; then unset BOXDREAMER_HIP_LIB; else export BOXDREAMER_HIP_LIB=tools/_probe/libbd_$v.so; fi
    echo "== $v rep $rep"; python tools/gemm_bench.py f16c8 2>&1 | grep -E "fc1|qkv  |weighted"
  done
done
unset BOXDREAMER_HIP_LIB
for rep in 1 2; do
  for v in base default; do
    if [ $v = default ]; then unset BOXDREAMER_HIP_LIB; else export BOXDREAMER_HIP_LIB=tools/_probe/libbd_$v.so; fi
    python bench.py --prec f16c8_qk16 --no-strict --no-fp8 --no-cpu-baseline --no-pnp --no-h2d --no-power --no-inline-counters --no-trained-like --steps 10 --warmup 3 2>/dev/null | grep '^{' > /tmp/ab.json
    python -c "
import json; j=json.load(open('/tmp/ab.json')); print('$v rep $rep default-mode step: poses/s', j['value'], 'ms', j['ms_per_step'], 'one lane', j.get('value_single_stream'), 'gemm TF/s', j['roofline']['achieved'], 'err', j.get('logits_max_abs_err'))"
  done
done
