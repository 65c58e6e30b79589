#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
L=cmunet_amd/csrc/libcmunet_hip.so
timeout -k 10 600 python -m pytest tests/test_gpu_fwd_ops.py tests/test_gpu_bwd_ops.py tests/test_gpu_fullsize.py -x -q -m gpu -k "persistent or conv3x3 or dgrad" > gpurun_out/r15_tests.log 2>&1 || { tail -40 gpurun_out/r15_tests.log; exit 1; }
tail -3 gpurun_out/r15_tests.log
{
for w in 1 0 1 0; do
  CMU_CONV_SPREAD=$w CMU_SWEEP_DT=1 timeout -k 10 120 python tools/igemm_stamps.py $L 512 64 64 2>/dev/null | grep layer | sed "s/^/SPREAD=$w /"
done
for d in zero_both; do
  for w in 1 0; do
  CMU_CONV_SPREAD=$w CMU_SWEEP_DT=1 CMU_SWEEP_DATA=$d timeout -k 10 120 python tools/igemm_stamps.py $L 512 64 64 2>/dev/null | grep layer | sed "s/^/SPREAD=$w /"
  done
done
} > gpurun_out/r15_sweep.log 2>&1
cat gpurun_out/r15_sweep.log
for i in 1 2 3; do
for w in 0 1; do
  CMU_CONV_SPREAD=$w timeout -k 10 200 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --all-kernel-events 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = j.get('kernel_ms_per_step', {})
print('recon SPREAD=$w', 'ms/step %.2f' % j['ms_per_step'], 'loss', j['config']['loss'], ' '.join('%s %.3f' % (n.replace('cmu_', ''), k[n]) for n in sorted(k, key=lambda n: -k[n])[:3]))"
done
done > gpurun_out/r15_ab.log 2>&1
cat gpurun_out/r15_ab.log
