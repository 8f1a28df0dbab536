#!/bin/bash
# Round 5 same-box A/B: DINOv2's split-bf16 attention (attn_kernel<bf16, 2, 64, 3>) double-buffered (tools/_probe/libbd_x3nbuf2.so, -DBD_TMP_X3_NBUF2)
cd "$(dirname "$0")/.."
echo "== attention op tests with the variant"; BOXDREAMER_HIP_LIB=tools/_probe/libbd_x3nbuf2.so timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "attention" 2>&1 | tail -2
for rep in 1 2; do
  for v in default x3nbuf2; do
    if [ $v = default ]; then unset BOXDREAMER_HIP_LIB; else export BOXDREAMER_HIP_LIB=tools/_probe/libbd_$v.so; fi
    echo "== $v rep $rep"; timeout 300 python tools/attn_probe.py bf16x3 2>&1 | grep -E "seq 261|prefix|seq 256"
  done
done
for rep in 1 2; do
  for v in default x3nbuf2; do
    if [ $v = default ]; then unset BOXDREAMER_HIP_LIB; else export BOXDREAMER_HIP_LIB=tools/_probe/libbd_$v.so; fi
    timeout 600 python bench.py --prec f16c8_qk16 --no-strict --no-fp8 --no-cpu-baseline --no-pnp --no-h2d --no-inline-counters --no-trained-like --no-power --steps 10 --warmup 3 2>/dev/null | grep '^{' > /tmp/ab.json
    python -c "
import json; j=json.load(open('/tmp/ab.json')); print('$v rep $rep default-mode step: poses/s', j['value'], 'ms', j['ms_per_step'], 'one lane', j.get('value_single_stream'), 'attn TF/s', j['roofline']['attention_achieved'], 'err', j.get('logits_max_abs_err'))"
  done
done
