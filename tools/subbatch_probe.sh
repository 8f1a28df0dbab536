mkdir -p gpurun_out/s2b
F="--no-strict --no-fp8 --no-cpu-baseline --no-pnp --no-h2d --no-trained-like --no-parity --steps 40 --warmup 6"
for pr in bf16 f16c8_qk16; do
for cfg in "--batch 32 --in-flight 1" "--batch 16 --in-flight 2" "--batch 8 --in-flight 4" "--batch 16 --in-flight 1" "--batch 64 --in-flight 1" "--batch 32 --in-flight 2" "--batch 16 --in-flight 4"; do
  python bench.py --prec $pr $cfg $F 2>/dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$pr', '$cfg', j['value'], j.get('value_single_stream'), j['ms_per_step'])"
done; done | tee gpurun_out/s2b/subbatch.txt
