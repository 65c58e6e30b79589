#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
rm -f gpurun_out/parity_record.txt
timeout -k 10 700 python -m pytest tests -q -m gpu --durations=5 > gpurun_out/tests_c.log 2>&1; echo "pytest rc=$?" >> gpurun_out/tests_c.log
tail -12 gpurun_out/tests_c.log
timeout -k 10 300 python __graft_entry__.py smoke 2>&1 | tail -1
{
for i in 1 2 3; do
  for lib in tools/_diag/libcmunet_r02.so ""; do
    tag=${lib:-tree}
    if [ -n "$lib" ]; then export CMU_POOL_FUSE=0; else unset CMU_POOL_FUSE; fi
    CMU_LIB_PATH=$lib timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --all-kernel-events 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = j.get('kernel_ms_per_step', {})
print('$tag'.rjust(32), 'ms/step %.2f' % j['ms_per_step'], 'img/s %.1f' % j['value'], 'frac %.4f' % j['roofline']['frac'], ' '.join('%s %.2f' % (n.replace('cmu_', ''), k[n]) for n in sorted(k, key=lambda n: -k[n])[:8]))"
  done
done
unset CMU_POOL_FUSE
} > gpurun_out/r03_ab_vs_r02.log 2>&1
cat gpurun_out/r03_ab_vs_r02.log
