#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
{
for i in 1 2 3; do
  for lib in tools/_diag/libcmunet_r02.so ""; do
    tag=${lib:-tree}
    if [ -n "$lib" ]; then export CMU_POOL_FUSE=0; else unset CMU_POOL_FUSE; fi
    CMU_LIB_PATH=$lib timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --all-kernel-events 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = j.get('kernel_ms_per_step', {})
print('$tag'.rjust(32), 'ms/step %.2f' % j['ms_per_step'], 'img/s %.1f' % j['value'], 'frac %.4f' % j['roofline']['frac'], ' '.join('%s %.2f' % (n.replace('cmu_', ''), k[n]) for n in sorted(k, key=lambda n: -k[n])[:6]))"
  done
done
unset CMU_POOL_FUSE
} > gpurun_out/r03_ab_vs_r02.log 2>&1
cat gpurun_out/r03_ab_vs_r02.log
timeout -k 10 500 bash tools/bench_all.sh r03 > gpurun_out/r03_bench_all.log 2>&1
tail -30 gpurun_out/r03_bench_all.log | cut -c1-400
{
for f in 1 0 1 0; do
  CMU_EMA_FUSE=$f timeout -k 10 200 python bench.py --workload joint --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-events 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('joint EMA_FUSE=$f ms/step', j['ms_per_step'], 'img/s', j['value'], 'loss', j['config']['loss'])"
done
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
cat gpurun_out/r03_ab_misc.log
