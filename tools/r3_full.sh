#!/bin/bash
# full GPU validation + counters + default bench (round-3 state)
tag=${1:-r3full}
mkdir -p gpurun_out/$tag
timeout 3000 python -m pytest tests -m gpu -q > gpurun_out/$tag/pytest.log 2>&1; echo "pytest rc $?" | tee gpurun_out/$tag/status
tail -8 gpurun_out/$tag/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --measure-counters --prec bf16 > gpurun_out/$tag/counters_bf16.log 2>&1; echo "counters bf16 rc $?" | tee -a gpurun_out/$tag/status
timeout 900 python bench.py --measure-counters --prec f16c8_qk16 > gpurun_out/$tag/counters_strict.log 2>&1; echo "counters strict rc $?" | tee -a gpurun_out/$tag/status
cp profiles/counters_*.json gpurun_out/$tag/ 2>/dev/null
timeout 900 python bench.py > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err; echo "bench rc $?" | tee -a gpurun_out/$tag/status
cp gpurun_out/strict_margin*.json gpurun_out/parity_report.json gpurun_out/$tag/ 2>/dev/null
python -c "
import json; j=json.load(open('gpurun_out/$tag/bench.json')); s=j['strict']
print('bf16', j['value'], j['single_stream'], j['roofline']['achieved'], j['roofline']['traffic_over_algorithmic'], j['roofline']['mfma_busy'])
print('strict', s['value'], s['single_stream'], s['roofline']['achieved'], s['roofline']['traffic_over_algorithmic'], s['roofline']['mfma_busy'], s['parity']['logits_max_abs_err'])"
