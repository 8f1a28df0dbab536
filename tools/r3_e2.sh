#!/bin/bash
mkdir -p gpurun_out/r3e
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "fused_layernorm" 2>&1 | tail -2
timeout 1500 python -m pytest tests/test_gpu_path.py -m gpu -q -k "full_size or (golden and full_T6)" 2>&1 | tail -2
show() { python -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('$1', j['value'], j.get('single_stream'), 'gemm', r['achieved'], r['gemm_time_frac_of_step'], 'attn', r['attention_time_frac_of_step'])"; }
for i in 1 2; do
  python bench.py --no-strict --no-cpu-baseline --no-pnp --no-h2d --no-parity 2>/dev/null | grep '^{' | show "bf16 fused  "
  BOXDREAMER_HIP_LIB=tools/_probe/libbd_nolnf.so python bench.py --no-strict --no-cpu-baseline --no-pnp --no-h2d --no-parity 2>/dev/null | grep '^{' | show "bf16 nofuse "
done
python bench.py --prec f16c8_qk16 --no-cpu-baseline --no-pnp --no-h2d --no-parity 2>/dev/null | grep '^{' | show "qk16 fused  "
BOXDREAMER_HIP_LIB=tools/_probe/libbd_nolnf.so python bench.py --prec f16c8_qk16 --no-cpu-baseline --no-pnp --no-h2d --no-parity 2>/dev/null | grep '^{' | show "qk16 nofuse "
