#!/bin/bash
# Round 5 same-box A/B: the 16-bit GEMM's persistent-kernel occupancy estimate counting the tiles of ALL sub-batch lanes that enqueue the launch
# side by side (bd_concurrent_launches) against the previous build (tools/_probe/libbd_base.so), bf16 over batch sizes.
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_gpu_lanes.py -x -q -m gpu 2>&1 | tail -1
for rep in 1 2; do
  for B in 4 6 8 10 12 16 24 32; do
    for v in base new; do
      if [ $v = new ]; then unset BOXDREAMER_HIP_LIB; else export BOXDREAMER_HIP_LIB=tools/_probe/libbd_base.so; fi
      timeout 300 python bench.py --prec bf16 --batch $B --no-strict --no-fp8 --no-cpu-baseline --no-pnp --no-h2d --no-inline-counters --no-trained-like --no-latency --no-power --no-parity --steps 20 --warmup 5 2>/dev/null | grep '^{' > /tmp/ab.json
      python -c "
import json; j=json.load(open('/tmp/ab.json')); print('$v rep $rep bf16 B=$B: ms/step', j['ms_per_step'], 'poses/s', j['value'], 'lanes', j['config']['sub_batch_lanes'])"
    done
  done
done
