#!/bin/bash
# Round 5: the F16C8 GEMM's forms over small row counts, same box: large only (tools/_probe/libbd_b1base.so), small 128 x 192 forced
# (libbd_c8s192.so: -DBD_C8_FORCE_SMALL), small 128 x 96 forced (libbd_c8s96.so: -DBD_C8_FORCE_SMALL -DBD_C8_SMALL_96).  Microseconds per launch.
cd "$(dirname "$0")/.."
for v in b1base c8s192 c8s96; do
  echo "== $v"
  BOXDREAMER_HIP_LIB=tools/_probe/libbd_$v.so timeout 600 python tools/c8_form_sweep.py 2>&1 | grep -v amdgpu
done
