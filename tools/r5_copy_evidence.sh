#!/bin/bash
# Copy one tools/r5_final_measure.sh pass (gpurun_out/final_<tag>/) into profiles/ as the round's evidence of record.
tag=$1; o=gpurun_out/final_$tag
for pr in bf16 f16c8_qk16 fp8; do cp $o/${tag}_bench_${pr}_kernel_stats.csv profiles/r5_bench_${pr}_kernel_stats.csv; done
cp $o/counters_*.json profiles/; cp $o/bench_default.json profiles/r5_bench_default.json; cp $o/pytest.log profiles/r5_gpu_suite.log
grep "runs against" $o/race_screen.txt > /tmp/rs.txt; (cat /tmp/rs.txt; grep -A20 "^-- small batches" profiles/r5_race_screen.txt) > /tmp/rs2.txt; cp /tmp/rs2.txt profiles/r5_race_screen.txt
cp gpurun_out/strict_margin_f16c8_qk16.json profiles/r5_strict_margin_f16c8_qk16.json; cp gpurun_out/parity_report.json profiles/r5_parity_report.json
for g in 0.25 0.5 0.75; do cp gpurun_out/calibration_outliers_g$g.json profiles/r5_calibration_outliers_g$g.json; done
(tail -1 $o/bench_config3.json; tail -1 $o/bench_T17_default.json) > profiles/r5_bench_lines.jsonl
tail -1 $o/pytest.log
