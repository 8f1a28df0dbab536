#!/bin/bash
# round-3 first GPU pass: tests, default bench, counter passes for both benched modes
mkdir -p gpurun_out/r3a
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r3a/pytest.log 2>&1; echo "pytest rc $?" | tee -a gpurun_out/r3a/status
tail -5 gpurun_out/r3a/pytest.log
timeout 900 python bench.py --measure-counters --prec bf16 > gpurun_out/r3a/counters_bf16.log 2>&1; echo "counters bf16 rc $?" | tee -a gpurun_out/r3a/status
timeout 900 python bench.py --measure-counters --prec f16c8_qkv16 > gpurun_out/r3a/counters_strict.log 2>&1; echo "counters strict rc $?" | tee -a gpurun_out/r3a/status
cp profiles/counters_*.json gpurun_out/r3a/ 2>/dev/null
timeout 900 python bench.py > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err; echo "bench rc $?" | tee -a gpurun_out/r3a/status
cp gpurun_out/strict_margin.json gpurun_out/parity_report.json gpurun_out/r3a/ 2>/dev/null
tail -c 600 gpurun_out/r3a/counters_strict.log
