#!/bin/bash
mkdir -p gpurun_out/r3j
BOXDREAMER_HIP_LIB=tools/_probe/libbd_res.so timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "attention" 2>&1 | tail -3
for i in 1 2; do
  for pr in bf16 fp16; do
    echo "default $pr: $(python tools/attn_probe.py $pr 2>/dev/null | head -4 | tr '\n' ';')"
    echo "res     $pr: $(BOXDREAMER_HIP_LIB=tools/_probe/libbd_res.so python tools/attn_probe.py $pr 2>/dev/null | head -4 | tr '\n' ';')"
  done
done | tee gpurun_out/r3j/attn_res.txt
show() { python -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('$1', j['value'], j.get('single_stream'), 'attn frac', r['attention_time_frac_of_step'], r['attention_achieved'])"; }
for i in 1 2; do
  python bench.py --no-strict --no-fp8 --no-cpu-baseline --no-pnp --no-h2d --no-parity 2>/dev/null | grep '^{' | show "bf16 default"
  BOXDREAMER_HIP_LIB=tools/_probe/libbd_res.so python bench.py --no-strict --no-fp8 --no-cpu-baseline --no-pnp --no-h2d --no-parity 2>/dev/null | grep '^{' | show "bf16 res    "
done | tee -a gpurun_out/r3j/attn_res.txt
