#!/bin/bash
# r03 profiles at HEAD
mkdir -p gpurun_out
bash tools/profile_round.sh r03 > gpurun_out/r03_profile_round.log 2>&1; tail -2 gpurun_out/r03_profile_round.log | cut -c1-300
bash tools/profile_workloads.sh r03 > gpurun_out/r03_profile_workloads.log 2>&1; tail -4 gpurun_out/r03_profile_workloads.log | cut -c1-200
bash tools/pmc_conv.sh r03 > gpurun_out/r03_pmc_conv_raw.txt 2>&1; tail -3 gpurun_out/r03_pmc_conv_raw.txt | cut -c1-200
python bench.py > gpurun_out/r03_bench_default.json.log 2>gpurun_out/r03_bench_default.err; tail -c 400 gpurun_out/r03_bench_default.json.log
find gpurun_out/prof_r03* gpurun_out/pmc*_r03* -type f ! -name "*stats*.csv" ! -name "*counter_collection.csv" -delete 2>/dev/null
timeout -k 10 500 bash tools/bench_all.sh r03 > gpurun_out/r03_bench_all.log 2>&1
grep -A1 "^== " gpurun_out/r03_bench_all.log | grep -v "^--" | cut -c1-300
{
for f in 1 0; do
  CMU_EMA_FUSE=$f timeout -k 10 200 python bench.py --workload joint --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-events 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('joint EMA_FUSE=$f ms/step', j['ms_per_step'], 'img/s', j['value'], 'loss', j['config']['loss'])"
done
for f in 1 0; do
  CMU_SPARK_POOL_FUSE=$f timeout -k 10 200 python bench.py --workload spark --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('spark SPARK_POOL_FUSE=$f ms/step', j['ms_per_step'], 'img/s', j['value'], 'loss', j['config']['loss'])"
done
CMU_SPARK_TILES=0 timeout -k 10 200 python bench.py --workload spark --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('spark dense encoder (CMU_SPARK_TILES=0) ms/step', j['ms_per_step'], 'img/s', j['value'])"
for r in 0 1; do
  if [ $r = 1 ]; then export WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 CMU_DP_REHEARSE=1; fi
  for wl in joint moco; do
  timeout -k 10 200 python bench.py --workload $wl --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-events 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl rccl_one_rank_group=$r ms/step', j['ms_per_step'], 'img/s', j['value'], 'loss', j['config']['loss'])"
  done
done
unset WORLD_SIZE RANK LOCAL_RANK MASTER_ADDR MASTER_PORT CMU_DP_REHEARSE
timeout -k 10 300 python bench.py --dtype f32 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('recon f32 ms/step', j['ms_per_step'], 'img/s', j['value'])"
timeout -k 10 600 python tools/chain_config4.py 2>/dev/null | tail -1
} > gpurun_out/r03_ab_misc.log 2>&1
cat gpurun_out/r03_ab_misc.log | cut -c1-400
