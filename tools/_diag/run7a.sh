#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
rm -f gpurun_out/parity_record.txt
timeout -k 10 600 python -m pytest tests -q -m gpu --durations=10 > gpurun_out/tests_b.log 2>&1; echo "pytest rc=$?" >> gpurun_out/tests_b.log
tail -25 gpurun_out/tests_b.log
timeout -k 10 400 bash tools/ab_bench.sh tools/_diag/libcmunet_r02.so > gpurun_out/r03_ab_vs_r02.log 2>&1
cat gpurun_out/r03_ab_vs_r02.log
