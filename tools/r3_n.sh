#!/bin/bash
for v in res res1 res2 res3; do
  echo "$v: $(BOXDREAMER_HIP_LIB=tools/_probe/libbd_$v.so python tools/attn_probe.py bf16 2>/dev/null | head -2 | tr '\n' ';')"
done
