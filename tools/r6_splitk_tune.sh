#!/bin/bash
# Round 6: split-K factors of the default mode one pose at a time (B = 1, 2, 4): ms per step for deep (fc2) x flat (proj) factors.
# (BD_SK_DEEP / BD_SK_FLAT were read by the TUNING build of gemm_f16c8.hip only -- profiles/r6_split_k.md; the shipped library has no
# environment switches, so every row now measures the library's own rule.)
cd "$(dirname "$0")/.."
out=gpurun_out/r6_splitk; mkdir -p $out
COMMON="--no-strict --no-fp8 --no-cpu-baseline --no-pnp --no-h2d --no-inline-counters --no-trained-like --no-power --no-latency --no-rccl-probe --no-facade --sustained 0"
for b in ${BATCHES:-1 2 4}; do
  for cfg in ${CFGS:-"1 1" "2 1" "3 1" "4 1" "3 2" "4 2" "4 3" "4 4" "0 0"}; do
    set -- $cfg
    r=$(BD_SK_DEEP=$1 BD_SK_FLAT=$2 python bench.py --latency-forms --prec f16c8_qk16 --batch $b --steps 50 --warmup 10 $COMMON 2>/dev/null | grep '^{' | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['config'].get('value_logits_max_abs_err'))")
    echo "B=$b deep=$1 flat=$2 : $r" | tee -a $out/tune.txt
  done
done
