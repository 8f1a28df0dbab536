#!/bin/bash
# Round 5 same-box A/B: barrier-free main loop (LDS counters) in the F16C8 persistent GEMM (tools/_probe/libbd_flagsync.so, -DBD_TMP_FLAGSYNC)
cd "$(dirname "$0")/.."
V=tools/_probe/libbd_flagsync.so
echo "== one GEMM first (hang guard: 90 s)"; BOXDREAMER_HIP_LIB=$V timeout 90 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "test_gemm_f16c8 and 7500" 2>&1 | tail -2 || { echo "first GEMM failed / timed out"; exit 1; }
echo "== op tests with the variant"; BOXDREAMER_HIP_LIB=$V timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "f16c8 or tile_shape or sparse_last_round or promot" 2>&1 | tail -3
echo "== whole-path tests with the variant"; BOXDREAMER_HIP_LIB=$V timeout 600 python -m pytest tests/test_gpu_path.py -x -q -m gpu -k "full_T6-f16c8_qk16 or full_size_properties and f16c8" 2>&1 | tail -3
for rep in 1 2; do
  for v in default flagsync; do
    if [ $v = default ]; then unset BOXDREAMER_HIP_LIB; else export BOXDREAMER_HIP_LIB=tools/_probe/libbd_$v.so; fi
    echo "== $v rep $rep"; timeout 300 python tools/gemm_bench.py f16c8 2>&1 | grep -E "TF/s"
  done
done
for rep in 1 2; do
  for v in default flagsync; do
    if [ $v = default ]; then unset BOXDREAMER_HIP_LIB; else export BOXDREAMER_HIP_LIB=tools/_probe/libbd_$v.so; fi
    timeout 600 python bench.py --prec f16c8_qk16 --no-strict --no-fp8 --no-cpu-baseline --no-pnp --no-h2d --no-inline-counters --no-trained-like --steps 10 --warmup 3 2>/dev/null | grep '^{' > /tmp/ab.json
    python -c "
import json; j=json.load(open('/tmp/ab.json')); p=j.get('power') or {}; print('$v rep $rep default-mode step: poses/s', j['value'], 'ms', j['ms_per_step'], 'one lane', j.get('value_single_stream'), 'gemm TF/s', j['roofline']['achieved'], 'err', j.get('logits_max_abs_err'), 'W', p.get('avg_w'), 'MHz', p.get('sclk_reported_mhz_avg'))"
  done
done
unset BOXDREAMER_HIP_LIB
echo "== race screen with the variant"; BOXDREAMER_HIP_LIB=$V timeout 600 python tools/stress_determinism.py f16c8_qk16 30 32 auto 2>&1 | grep -v amdgpu | tail -2
