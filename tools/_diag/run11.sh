#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_sparse_tiles.py tests/test_gpu_fwd_ops.py tests/test_gpu_bwd_ops.py tests/test_gpu_fullsize.py -q -m gpu -x -k "gather or convT or spark_step or conv_rows" > gpurun_out/r11_tests.log 2>&1 || { tail -40 gpurun_out/r11_tests.log; exit 1; }
tail -3 gpurun_out/r11_tests.log
for i in 1 2; do
for lib in tools/_diag/libcmunet_2bar.so ""; do
  for wl in spark recon; do
  CMU_LIB_PATH=$lib timeout -k 10 200 python bench.py --workload $wl --steps 12 --warmup 4 --no-cpu-baseline --all-kernel-events 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = j.get('kernel_ms_per_step', {})
print('$wl ${lib:-one-barrier(tree)}'.ljust(50), 'ms/step %.2f' % j['ms_per_step'], ' '.join('%s %.2f' % (n.replace('cmu_', ''), k[n]) for n in sorted(k, key=lambda n: -k[n]) if 'rows' in n or 'convT2x2_fwd' in n or 'convT2x2_dgrad' in n))"
  done
done
done > gpurun_out/r11_onebarrier.log 2>&1
cat gpurun_out/r11_onebarrier.log
