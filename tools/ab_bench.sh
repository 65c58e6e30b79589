#!/bin/bash
# same-box A/B of two builds of the library: bash tools/ab_bench.sh <base.so> [bench.py args ...]
# (boxes of the pool differ by +-4 %; alternating runs on one box separate a 1 % kernel change from that)
BASE=$1; shift
for i in 1 2 3; do
  for lib in "$BASE" ""; do
    tag=${lib:-tree}
    CMU_LIB_PATH=$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline --all-kernel-events "$@" 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = j.get('kernel_ms_per_step', {})
print('$tag'.rjust(20), 'ms/step %.2f' % j['ms_per_step'], 'frac %.4f' % j['roofline']['frac'], ' '.join('%s %.2f' % (n.replace('cmu_', ''), k[n]) for n in sorted(k, key=lambda n: -k[n])[:4]))"
  done
done
