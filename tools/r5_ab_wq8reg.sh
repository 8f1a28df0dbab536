#!/bin/bash
# Round 5 same-box A/B: the F16C8 GEMM deriving W's e4m3 image q8 in registers from the f16 W fragments (as it already does for A)
# instead of reading it from LDS (tools/_probe/libbd_wq8reg.so, -DBD_EXP_WQ8REG): 15 instead of 18 ds_read_b128 per wave and slab.
# Stage 1 of the experiment: the W plane still CARRIES q8 (DMA unchanged), so this prices the LDS reads alone.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
echo "== f16c8 op tests with the variant"
BOXDREAMER_HIP_LIB=tools/_probe/libbd_wq8reg.so timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "f16c8" 2>&1 | tail -2
for rep in 1 2; do
  for v in default wq8reg; do
    if [ $v = default ]; then unset BOXDREAMER_HIP_LIB; else export BOXDREAMER_HIP_LIB=tools/_probe/libbd_$v.so; fi
    echo "== $v f16c8 rep $rep"; timeout 300 python tools/gemm_bench.py f16c8 2>&1 | grep -E "qkv|proj|fc1|fc2|weighted"
  done
done
for rep in 1 2; do
  for v in default wq8reg; do
    if [ $v = default ]; then unset BOXDREAMER_HIP_LIB; else export BOXDREAMER_HIP_LIB=tools/_probe/libbd_$v.so; fi
    timeout 600 python bench.py --prec f16c8_qk16 --no-strict --no-fp8 --no-cpu-baseline --no-pnp --no-h2d --no-inline-counters --no-trained-like --steps 10 --warmup 3 2>/dev/null | grep '^{' > /tmp/ab.json
    python -c "
import json; j=json.load(open('/tmp/ab.json')); p=j.get('power') or {}; print('$v rep $rep default step: poses/s', j['value'], 'ms', j['ms_per_step'], 'one lane', j.get('value_single_stream'), 'gemm TF/s', j['roofline']['achieved'], 'err', (j.get('parity') or {}).get('logits_max_abs_err'), 'W', p.get('avg_w'))"
  done
done
