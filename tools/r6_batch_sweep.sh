#!/bin/bash
# Round 6: the default mode over batch sizes (HIP graph, auto lanes): ms per step / per pose; B = 1, 2 also with the opt-in latency forms.
cd "$(dirname "$0")/.."
out=gpurun_out/r6_batch_sweep.txt; : > $out
COMMON="--prec f16c8_qk16 --no-strict --no-fp8 --no-cpu-baseline --no-pnp --no-h2d --no-inline-counters --no-trained-like --no-power --no-latency --no-rccl-probe --no-facade --no-parity --sustained 0 --steps 30 --warmup 8"
for b in 1 2 3 4 6 8 12 16 24 32; do
  for lat in "" "--latency-forms"; do
    [ -n "$lat" ] && [ $b -gt 2 ] && continue
    python bench.py $COMMON --batch $b $lat 2>/dev/null | grep '^{' | python -c "
import json,sys; j=json.loads(sys.stdin.read()); b=$b
print('f16c8_qk16 B=%d %s lanes %s ms/step %.3f ms/pose %.3f poses/s %.1f' % (b, 'latency-forms' if '$lat' else 'default-forms', j['config'].get('sub_batch_lanes'), j['ms_per_step'], j['ms_per_step']/b, j['value']))" | tee -a $out
  done
done
