#!/bin/bash
# Round 5 same-box A/B: the slab's first fragment reads in the order a0, b0, a1, b1, b2 (pinned) behind an explicit lgkmcnt(0) after the slab
# barrier, so that hipcc's wait model lets the slab's FIRST MFMA go after two reads instead of all five (tools/_probe/libbd_pinfirst.so,
# -DBD_EXP_PIN_FIRST; both the 16-bit and the F16C8 persistent GEMM).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
V=${1:-pinfirst}
echo "== gemm op tests with the variant"
BOXDREAMER_HIP_LIB=tools/_probe/libbd_$V.so timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm" 2>&1 | tail -2
for rep in 1 2; do
  for v in default $V; do
    if [ $v = default ]; then unset BOXDREAMER_HIP_LIB; else export BOXDREAMER_HIP_LIB=tools/_probe/libbd_$v.so; fi
    for pr in bf16 f16c8; do echo "== $v $pr rep $rep"; timeout 300 python tools/gemm_bench.py $pr 2>&1 | grep -E "qkv|proj|fc1|fc2|weighted"; done
  done
done
for rep in 1 2; do
  for v in default $V; do
    if [ $v = default ]; then unset BOXDREAMER_HIP_LIB; else export BOXDREAMER_HIP_LIB=tools/_probe/libbd_$v.so; fi
    for pr in bf16 f16c8_qk16; do
    timeout 600 python bench.py --prec $pr --no-strict --no-fp8 --no-cpu-baseline --no-pnp --no-h2d --no-inline-counters --no-trained-like --steps 10 --warmup 3 2>/dev/null | grep '^{' > /tmp/ab.json
    python -c "
import json; j=json.load(open('/tmp/ab.json')); p=j.get('power') or {}; print('$v rep $rep $pr step: poses/s', j['value'], 'ms', j['ms_per_step'], 'one lane', j.get('value_single_stream'), 'gemm TF/s', j['roofline']['achieved'], 'err', (j.get('parity') or {}).get('logits_max_abs_err'), 'W', p.get('avg_w'))"
    done
  done
done
