#!/bin/bash
mkdir -p gpurun_out/r3e
timeout 1500 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "fused_layernorm" > gpurun_out/r3e/pytest_lnf_op.log 2>&1; echo "pytest lnf op rc $?"; tail -4 gpurun_out/r3e/pytest_lnf_op.log
timeout 2400 python -m pytest tests/test_gpu_path.py -m gpu -q -k "full_size or golden or views17 or batch_independence or margin" > gpurun_out/r3e/pytest_path.log 2>&1; echo "pytest path rc $?"; tail -6 gpurun_out/r3e/pytest_path.log
show() { python -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('$1', j['value'], j.get('single_stream'), 'gemm', r['achieved'], r['gemm_time_frac_of_step'], 'attn', r['attention_time_frac_of_step'])"; }
for i in 1 2; do
  python bench.py --no-strict --no-cpu-baseline --no-pnp --no-h2d --no-parity 2>/dev/null | grep '^{' | show "bf16 fused  "
  BOXDREAMER_HIP_LIB=tools/_probe/libbd_nolnf.so python bench.py --no-strict --no-cpu-baseline --no-pnp --no-h2d --no-parity 2>/dev/null | grep '^{' | show "bf16 nofuse "
  python bench.py --prec f16c8_qk16 --no-cpu-baseline --no-pnp --no-h2d --no-parity 2>/dev/null | grep '^{' | show "qk16 fused  "
  BOXDREAMER_HIP_LIB=tools/_probe/libbd_nolnf.so python bench.py --prec f16c8_qk16 --no-cpu-baseline --no-pnp --no-h2d --no-parity 2>/dev/null | grep '^{' | show "qk16 nofuse "
done
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/r3e/prof_bf16 -- python $OLDPWD/bench.py --prec bf16 --in-flight 1 --steps 5 --warmup 2 --no-graph --no-strict --no-cpu-baseline --no-pnp --no-h2d --no-parity > /dev/null 2>&1; echo "rocprof rc $?" )
find gpurun_out/r3e/prof_bf16 -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -12 {} | cut -c1-160'
