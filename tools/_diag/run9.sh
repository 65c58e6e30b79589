#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_sparse_tiles.py tests/test_gpu_pretrain.py tests/test_gpu_fullsize.py tests/test_gpu_dataparallel.py -q -m gpu -x -k "spark or masked_pools or tile" > gpurun_out/r9_tests.log 2>&1 || { tail -40 gpurun_out/r9_tests.log; exit 1; }
tail -3 gpurun_out/r9_tests.log
for f in 1 0 1 0; do
  CMU_SPARK_POOL_FUSE=$f timeout -k 10 200 python bench.py --workload spark --steps 10 --warmup 3 --no-cpu-baseline --all-kernel-events 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = j.get('kernel_ms_per_step', {})
print('SPARK_POOL_FUSE=$f', 'ms/step %.2f' % j['ms_per_step'], 'loss', j['config']['loss'], ' '.join('%s %.2f' % (n.replace('cmu_', ''), k[n]) for n in sorted(k, key=lambda n: -k[n]) if 'pool' in n or 'mask' in n))"
done > gpurun_out/r9_spark_poolfuse.log 2>&1
cat gpurun_out/r9_spark_poolfuse.log
