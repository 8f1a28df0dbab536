#!/bin/bash
# a 2-rank launch on a 1-GPU box: rank 1 cannot take cuda:1 -> the launch must end quickly with ONE JSON line carrying `error`, rc != 0
t0=$(date +%s)
timeout 600 python bench.py --gpus 2 --steps 2 --warmup 1 --dist-timeout 120 > gpurun_out/r3p_out.txt 2> gpurun_out/r3p_err.txt; rc=$?
t1=$(date +%s)
echo "rc $rc after $((t1 - t0)) s"
grep '^{' gpurun_out/r3p_out.txt | cut -c1-400
tail -5 gpurun_out/r3p_err.txt | cut -c1-300
