#!/bin/bash
mkdir -p gpurun_out/r3d
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "attention_f16c8 or test_attention" > gpurun_out/r3d/pytest_attn.log 2>&1; echo "pytest attn rc $?"; tail -3 gpurun_out/r3d/pytest_attn.log
BOXDREAMER_HIP_LIB=tools/_probe/libbd_nosat.so timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "attention_f16c8" > gpurun_out/r3d/pytest_attn_nosat.log 2>&1; echo "pytest attn nosat rc $?"; tail -3 gpurun_out/r3d/pytest_attn_nosat.log
# w4 variant: GEMM correctness (bf16 / fp16 paths) then micro-bench and step
BOXDREAMER_HIP_LIB=tools/_probe/libbd_w4.so timeout 1200 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "gemm and not f16c8" > gpurun_out/r3d/pytest_w4.log 2>&1; echo "pytest w4 rc $?"; tail -3 gpurun_out/r3d/pytest_w4.log
for i in 1 2; do
  python tools/gemm_bench.py bf16 2>&1 | grep -v amdgpu | tr '\n' ';' ; echo
  BOXDREAMER_HIP_LIB=tools/_probe/libbd_w4.so python tools/gemm_bench.py bf16 2>&1 | grep -v amdgpu | tr '\n' ';'; echo
done > gpurun_out/r3d/gemm_bench_w4.txt 2>&1
cat gpurun_out/r3d/gemm_bench_w4.txt
show() { python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$1', j['value'], j.get('single_stream'), j['roofline']['achieved'])"; }
for i in 1 2; do
  python bench.py --no-strict --no-cpu-baseline --no-pnp --no-h2d --no-parity 2>/dev/null | grep '^{' | show "bf16 default"
  BOXDREAMER_HIP_LIB=tools/_probe/libbd_w4.so python bench.py --no-strict --no-cpu-baseline --no-pnp --no-h2d --no-parity 2>/dev/null | grep '^{' | show "bf16 w4     "
  python bench.py --prec f16c8_qkv16 --no-cpu-baseline --no-pnp --no-h2d --no-parity 2>/dev/null | grep '^{' | show "strict sat  "
  BOXDREAMER_HIP_LIB=tools/_probe/libbd_nosat.so python bench.py --prec f16c8_qkv16 --no-cpu-baseline --no-pnp --no-h2d --no-parity 2>/dev/null | grep '^{' | show "strict nosat"
done
