#!/bin/bash
mkdir -p gpurun_out/r3c
timeout 2000 python -m pytest tests/test_gpu_path.py tests/test_gpu_ops.py -m gpu -q -k "range_stress or outliers or margin or f16c8 or fused_qk or odd_head or default_precision or tile_shape or qk16" > gpurun_out/r3c/pytest.log 2>&1; echo "pytest rc $?"
grep -E "range stress|outliers gain|strict margin|passed|failed|^FAILED" gpurun_out/r3c/pytest.log | cut -c1-700
for i in 1 2; do
  BOXDREAMER_HIP_LIB=tools/_probe/libbd_r3base.so python bench.py --prec f16c8_qkv16 --no-cpu-baseline --no-pnp --no-h2d --no-parity 2>/dev/null | grep '^{' | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('base qkv16', j['value'], j['single_stream'], j['roofline']['achieved'])"
  python bench.py --prec f16c8_qkv16 --no-cpu-baseline --no-pnp --no-h2d --no-parity 2>/dev/null | grep '^{' | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('new  qkv16', j['value'], j['single_stream'], j['roofline']['achieved'])"
  python bench.py --prec f16c8_qk16 --no-cpu-baseline --no-pnp --no-h2d 2>/dev/null | grep '^{' | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('new  qk16 ', j['value'], j['single_stream'], j['roofline']['achieved'], j['parity']['logits_max_abs_err'], j['parity']['top20_sets_equal_frac'])"
done
cp gpurun_out/strict_margin*.json gpurun_out/parity_report.json gpurun_out/r3c/ 2>/dev/null
