#!/bin/bash
# Round 6: the trained-like leg (outlier weights, load-time promotion) with the fold / 3-byte stream switched at pack time: how many units the
# self-check promotes and what the promoted mode delivers.
cd "$(dirname "$0")/.."
COMMON="--prec f16c8_qk16 --no-fp8 --no-latency --no-cpu-baseline --no-inline-counters --no-h2d --no-pnp --no-rccl-probe --no-facade --sustained 0 --no-power --steps 20 --warmup 5"
for cfg in "1 1" "1 0" "0 0"; do
  set -- $cfg
  BOXDREAMER_HIP_LNFOLD=$1 BOXDREAMER_HIP_RESID3=$2 timeout 600 python bench.py $COMMON 2>/dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); t=j['strict_trained_like']; c=t['calibration']
print('fold=$1 resid3=$2 plain', j['value'], '| trained-like', t['value'], 'promoted', c['promoted_units'], 'of', c['units'], 'work frac', c['promoted_work_frac'], 'self-check', c['self_check_unpromoted_max_abs_dlogits'], '->', c['self_check_final'], 'vs oracle', (t.get('parity') or {}).get('logits_max_abs_err'))"
done
