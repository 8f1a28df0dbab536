#!/bin/bash
# same-box A/B of the LayerNorm fold (default mode): BOXDREAMER_HIP_LNFOLD=0 keeps every bd_layernorm launch
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
COMMON="--prec f16c8_qk16 --no-strict --sustained 0 --no-facade --no-trained-like --no-fp8 --no-latency --no-cpu-baseline --no-inline-counters --no-h2d --no-pnp --no-rccl-probe --steps 20 --warmup 5"
for rep in 1 2; do
  for f in 0 1; do
    env ${AB_VAR:-BOXDREAMER_HIP_LNFOLD}=$f timeout 600 python bench.py $COMMON > gpurun_out/r6_ab_lnfold_${f}_${rep}.json 2> gpurun_out/r6_ab_lnfold_${f}_${rep}.err
    python - <<PY
import json
try:
    j = json.loads(open("gpurun_out/r6_ab_lnfold_${f}_${rep}.json").read().strip().splitlines()[-1])
    print("${AB_VAR:-LNFOLD}=${f} rep=${rep}", j["value"], j["ms_per_step"], j.get("single_stream", {}).get("value"), j.get("logits_max_abs_err"), j.get("parity", {}).get("top20_sets_equal_frac"), j.get("power", {}).get("mean_w"))
except Exception as e:
    print("fold=${f} rep=${rep} FAILED", e)
PY
  done
done
