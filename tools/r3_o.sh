#!/bin/bash
timeout 1500 python -m pytest tests/test_gpu_path.py -m gpu -q -k "range_stress or outliers or default_precision" 2>&1 | tail -3
python bench.py 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); s=j['strict']; f=j['fp8']
print('bf16', j['value'], j['single_stream'], j['roofline']['traffic'], j['roofline']['mfma_busy'] is not None)
print('strict', s['value'], s['single_stream'], s['parity']['logits_max_abs_err'], s['roofline']['traffic'])
print('fp8', f['value']); print('cpu', j['cpu_baseline']['value'])"
