#!/bin/bash
mkdir -p gpurun_out/r3b
tools/_probe/fp8_sat_probe > gpurun_out/r3b/fp8_sat_probe.txt 2>&1; cat gpurun_out/r3b/fp8_sat_probe.txt
timeout 1500 python -m pytest tests/test_gpu_path.py tests/test_gpu_ops.py -m gpu -x -q -k "range_stress or outliers or margin or f16c8 or fused_qk or odd_head or default_precision" > gpurun_out/r3b/pytest.log 2>&1; echo "pytest rc $?"
grep -E "range stress|outliers gain|strict margin|passed|failed" gpurun_out/r3b/pytest.log | cut -c1-400
for i in 1 2; do
  BOXDREAMER_HIP_LIB=tools/_probe/libbd_r3base.so python bench.py --prec f16c8_qkv16 --no-cpu-baseline --no-pnp --no-h2d --no-parity 2>/dev/null | grep '^{' | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('base', j['value'], j['single_stream'], j['roofline']['achieved'])"
  python bench.py --prec f16c8_qkv16 --no-cpu-baseline --no-pnp --no-h2d --no-parity 2>/dev/null | grep '^{' | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('new ', j['value'], j['single_stream'], j['roofline']['achieved'])"
done
cp gpurun_out/strict_margin.json gpurun_out/parity_report.json gpurun_out/r3b/ 2>/dev/null
