#!/bin/bash
F="--no-strict --no-fp8 --no-cpu-baseline --no-pnp --no-h2d --no-trained-like --no-parity --no-power --steps 40 --warmup 8"
for lib in default ln_nt0 ln_pipe8 ln_pipe8nt0 ln_pipe4 ln_pipe16; do
  if [ $lib = default ]; then unset BOXDREAMER_HIP_LIB; else export BOXDREAMER_HIP_LIB=tools/_probe/libbd_$lib.so; fi
  echo "== $lib"; python tools/ln_probe.py bf16 2>&1 | grep -v amdgpu; python tools/ln_probe.py f16c8 2>&1 | grep -v amdgpu | head -1
  for pr in bf16 f16c8_qk16; do python bench.py --prec $pr $F 2>/dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$lib $pr', j['value'], j.get('value_single_stream'))"; done
done
