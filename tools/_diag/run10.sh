#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_sparse_tiles.py -q -m gpu -x -k "gather or spark_step" > gpurun_out/r10_tests.log 2>&1 || { tail -40 gpurun_out/r10_tests.log; exit 1; }
tail -3 gpurun_out/r10_tests.log
for f in 0 256 0 256 128; do
  CMU_GATHER_NB=$f timeout -k 10 200 python bench.py --workload spark --steps 10 --warmup 3 --no-cpu-baseline --all-kernel-events 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = j.get('kernel_ms_per_step', {})
print('GATHER_NB=$f (0 = heuristic)', 'ms/step %.2f' % j['ms_per_step'], 'loss', j['config']['loss'], ' '.join('%s %.2f' % (n.replace('cmu_', ''), k[n]) for n in sorted(k, key=lambda n: -k[n]) if 'rows' in n or 'conv3x3' in n))"
done > gpurun_out/r10_gather_nb.log 2>&1
cat gpurun_out/r10_gather_nb.log
