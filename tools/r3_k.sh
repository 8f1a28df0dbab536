#!/bin/bash
BOXDREAMER_HIP_LIB=tools/_probe/libbd_res.so timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "attention" 2>&1 | tail -2
for i in 1 2 3; do
  echo "default bf16: $(python tools/attn_probe.py bf16 2>/dev/null | head -2 | tr '\n' ';')"
  echo "res     bf16: $(BOXDREAMER_HIP_LIB=tools/_probe/libbd_res.so python tools/attn_probe.py bf16 2>/dev/null | head -2 | tr '\n' ';')"
done
