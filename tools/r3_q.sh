#!/bin/bash
t0=$(date +%s)
timeout 900 python bench.py --gpus 2 --backend gloo --single-device-test --steps 3 --warmup 1 --no-cpu-baseline --no-h2d --no-pnp --dist-timeout 300 > gpurun_out/r3q_out.txt 2> gpurun_out/r3q_err.txt; rc=$?
echo "rc $rc after $(( $(date +%s) - t0 )) s"
grep '^{' gpurun_out/r3q_out.txt | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print({k: j.get(k) for k in ('n_gpus','value','ms_per_step','per_rank_ms_per_step','corner_allgather_ms','error')}); print('strict', (j.get('strict') or {}).get('value'), 'lanes', j['config']['batches_in_flight'], j['config']['global_batch'])"
tail -3 gpurun_out/r3q_err.txt | cut -c1-300
