#!/bin/bash
# One pass over everything profiles/ quotes for a round (run on the GPU box from the repo root; writes under gpurun_out/final_<tag>/).
tag=${1:-r2}
out=gpurun_out/final_$tag; mkdir -p $out
R=$(pwd)
( cd /tmp && export TMPDIR=/tmp
  for pr in bf16 f16c8_qkv16 fp8; do
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof_$pr -- python $R/bench.py --prec $pr --in-flight 1 --steps 5 --warmup 2 --no-graph --no-strict --no-cpu-baseline --no-pnp --no-h2d --no-parity > /dev/null 2>&1
    echo "rocprof $pr rc $?"
  done )
python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench default rc $?"
for extra in "--prec fp8 --no-strict" "--prec fp16 --no-strict" "--views 17 --no-strict" "--batch 1 --no-strict --in-flight 1" "--prec bf16x3 --no-strict" "--prec f16c8 --no-strict" "--prec bf16x3_qkv16 --no-strict" "--cache-refs --no-strict"; do
  python bench.py $extra --no-cpu-baseline --no-pnp --no-h2d 2>/dev/null | grep '^{' >> $out/bench_variants.jsonl; echo "bench $extra rc $?"
done
for pr in bf16 fp16 f16c8 bf16x3 fp8; do echo "== gemm_bench $pr"; python tools/gemm_bench.py $pr 2>&1 | grep -v amdgpu; done > $out/gemm_bench.txt
for pr in bf16 fp16 bf16x3; do echo "== attn_probe $pr"; python tools/attn_probe.py $pr 2>&1 | grep -v amdgpu; done > $out/attn_probe.txt
echo done
