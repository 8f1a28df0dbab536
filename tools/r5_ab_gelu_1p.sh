#!/bin/bash
# Round 5 timing probe: what the fc1 activation costs in the one-pass / e4m3 classes (identity instead of the packed-polynomial GELU: WRONG results)
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for v in default gelu1p_none; do
    if [ $v = default ]; then unset BOXDREAMER_HIP_LIB; else export BOXDREAMER_HIP_LIB=tools/_probe/libbd_$v.so; fi
    for pr in fp8 bf16; do echo "== $v $pr rep $rep"; timeout 300 python tools/gemm_bench.py $pr 2>&1 | grep -E "qkv  |fc1|fc2  |weighted"; done
  done
done
