#!/bin/bash
# Same-box A/B of bench.py between the default library and tools/_probe/libbd_<name>.so, alternating, `reps` times.
#   tools/ab_bench.sh <name> [reps]      (box-to-box variance is 5-10 %, so only same-box deltas mean anything)
name=$1; reps=${2:-2}
show() { python -c "
import json,sys
j=json.loads(open('$1').read())
s=j.get('strict') or {}
print('$2', 'poses/s', j['value'], 'ms', j['ms_per_step'], 'gemm TF/s', j['roofline']['achieved'], 'strict', s.get('value'), 'strict gemm', (s.get('roofline') or {}).get('achieved'))"; }
for i in $(seq $reps); do
  python bench.py --no-cpu-baseline --no-pnp --no-h2d 2>/dev/null | grep '^{' > /tmp/ab_a.json; show /tmp/ab_a.json default
  BOXDREAMER_HIP_LIB=tools/_probe/libbd_${name}.so python bench.py --no-cpu-baseline --no-pnp --no-h2d 2>/dev/null | grep '^{' > /tmp/ab_b.json; show /tmp/ab_b.json $name
done
