#!/bin/bash
# Round 5 same-box A/B: the GENERIC attention kernel's (DINOv2, ragged ranges) K / V tile loads as bounds-checked buffer loads (default build) against
# the per-lane-clamped flat loads (tools/_probe/libbd_ppbuf.so = the previous commit)
cd "$(dirname "$0")/.."
echo "== attention op tests"; timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "attention" 2>&1 | tail -2
echo "== whole-path + facade tests"; timeout 900 python -m pytest tests/test_gpu_path.py tests/test_gpu_facade.py tests/test_gpu_lanes.py -x -q -m gpu 2>&1 | tail -2
python tools/attn_hash.py > /tmp/h_new.txt 2>&1; BOXDREAMER_HIP_LIB=tools/_probe/libbd_ppbuf.so python tools/attn_hash.py > /tmp/h_old.txt 2>&1; if diff -q /tmp/h_new.txt /tmp/h_old.txt > /dev/null; then echo "attn_hash: outputs BIT-IDENTICAL over $(wc -l < /tmp/h_new.txt) cases"; else echo "attn_hash: DIFFER"; diff /tmp/h_new.txt /tmp/h_old.txt | head; fi
for rep in 1 2; do
  for v in ppbuf default; do
    if [ $v = default ]; then unset BOXDREAMER_HIP_LIB; else export BOXDREAMER_HIP_LIB=tools/_probe/libbd_$v.so; fi
    for pr in bf16 bf16x3; do echo "== $v $pr rep $rep"; timeout 300 python tools/attn_probe.py $pr 2>&1 | grep -E "seq 261|prefix|seq 256|seq 320"; done
  done
done
for rep in 1 2; do
  for v in ppbuf default; do
    if [ $v = default ]; then unset BOXDREAMER_HIP_LIB; else export BOXDREAMER_HIP_LIB=tools/_probe/libbd_$v.so; fi
    timeout 600 python bench.py --no-fp8 --no-cpu-baseline --no-pnp --no-h2d --no-inline-counters --no-trained-like --no-power --steps 10 --warmup 3 2>/dev/null | grep '^{' > /tmp/ab.json
    python -c "
import json; j=json.load(open('/tmp/ab.json')); s=j['strict']; print('$v rep $rep: bf16', j['value'], 'one lane', j.get('value_single_stream'), 'attn TF/s', j['roofline']['attention_achieved'], '| default', s['value'], 'one lane', s['single_stream']['value'], 'attn', s['roofline']['attention_achieved'], 'err', s['parity']['logits_max_abs_err'])"
  done
done
