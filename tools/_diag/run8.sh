#!/bin/bash
# r03 profiles: kernel-trace stats + HBM PMC passes of the default bench, stats of the other workloads, SQ counters of the conv kernels
mkdir -p gpurun_out
bash tools/profile_round.sh r03 > gpurun_out/r03_profile_round.log 2>&1; tail -2 gpurun_out/r03_profile_round.log | cut -c1-300
bash tools/profile_workloads.sh r03 > gpurun_out/r03_profile_workloads.log 2>&1; tail -4 gpurun_out/r03_profile_workloads.log | cut -c1-200
bash tools/pmc_conv.sh r03 > gpurun_out/r03_pmc_conv_raw.txt 2>&1; tail -12 gpurun_out/r03_pmc_conv_raw.txt | cut -c1-300
python bench.py > gpurun_out/r03_bench_default.json.log 2>gpurun_out/r03_bench_default.err; tail -c 700 gpurun_out/r03_bench_default.json.log
# keep the merged output small: only the stats / counter csv files
find gpurun_out/prof_r03* gpurun_out/pmc*_r03* -type f ! -name "*stats*.csv" ! -name "*counter_collection.csv" -delete 2>/dev/null
du -sh gpurun_out
