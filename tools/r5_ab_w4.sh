#!/bin/bash
# Round 5 same-box A/B: fc2 (K = 3072, fp32 residual) on 4 consumer waves of 128 x 96 (tools/_probe/libbd_w4.so, -DBD_PC_WAVES4)
# against the shipped 8 x (64 x 96) -- rates, bit-identity of the rows, the bf16 step, and PMC counters of the fc2 launch for both forms.
cd "$(dirname "$0")/.."
R=$(pwd); mkdir -p gpurun_out
echo "== op tests with the w4 library (rows bit-identical across tile shapes)"
BOXDREAMER_HIP_LIB=tools/_probe/libbd_w4.so timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "tile_shape or sparse_last_round or test_gemm_block_sized or test_gemm_epilogues" 2>&1 | tail -2
for rep in 1 2; do
  for v in default w4; do
    if [ $v = default ]; then unset BOXDREAMER_HIP_LIB; else export BOXDREAMER_HIP_LIB=tools/_probe/libbd_$v.so; fi
    for pr in bf16 fp16; do echo "== $v $pr rep $rep"; timeout 300 python tools/gemm_bench.py $pr 2>&1 | grep -E "fc2|proj|weighted"; done
  done
done
for rep in 1 2; do
  for v in default w4; do
    if [ $v = default ]; then unset BOXDREAMER_HIP_LIB; else export BOXDREAMER_HIP_LIB=tools/_probe/libbd_$v.so; fi
    timeout 600 python bench.py --no-strict --no-fp8 --no-cpu-baseline --no-pnp --no-h2d --no-inline-counters --no-trained-like --steps 10 --warmup 3 2>/dev/null | grep '^{' > /tmp/ab.json
    python -c "
import json; j=json.load(open('/tmp/ab.json')); p=j.get('power') or {}; print('$v rep $rep bf16 step: poses/s', j['value'], 'ms', j['ms_per_step'], 'one lane', j.get('value_single_stream'), 'gemm TF/s', j['roofline']['achieved'], 'err', j.get('logits_max_abs_err'), 'W', p.get('avg_w'))"
  done
done
cd /tmp; export TMPDIR=/tmp
for v in default w4; do
  if [ $v = default ]; then unset BOXDREAMER_HIP_LIB; else export BOXDREAMER_HIP_LIB=$R/tools/_probe/libbd_$v.so; fi
  for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
    d=/tmp/pmc_${v}_$(echo $grp | cut -c1-12 | tr ' ' '_'); rm -rf $d
    timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $d -- python $R/tools/gemm_fc2_once.py 49152 bf16 20 > /dev/null 2>&1
    f=$(find $d -name "*counter_collection.csv" | head -1)
    python - "$f" "$v" <<'PY'
import csv, sys, collections
f, v = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(float); n = collections.defaultdict(int); dur = 0.0; seen=set()
for r in csv.DictReader(open(f)):
    if "gemm_kernel_pc" not in r["Kernel_Name"]: continue
    acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    if r["Dispatch_Id"] not in seen:
        seen.add(r["Dispatch_Id"]); dur += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
k = max(n.values()) if n else 1
print(v, "fc2 per launch:", {c: round(acc[c] / k) for c in acc}, "us per launch under the counters", round(dur / max(len(seen),1) / 1e3, 1))
PY
  done
done
