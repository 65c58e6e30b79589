#!/bin/bash
# same-box A/B of two builds of the library: bash tools/ab_bench.sh <base.so> [bench.py args ...]
# (boxes of the pool differ by +-4 %; alternating runs on one box separate a 1 % kernel change from that)
BASE=$1; shift
for i in 1 2 3; do
  for lib in "$BASE" ""; do
    tag=${lib:-tree}
    # (stderr of every run is kept -- gpurun_out/ab_bench_<tag>_<i>.err -- and a run that prints no JSON line is reported with its last
    # stderr line instead of a parser traceback: round 4's g21 A/B lost the reason of a silent variant)
    mkdir -p gpurun_out
    CMU_LIB_PATH=$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline --all-kernel-events "$@" 2>gpurun_out/ab_bench_$(basename ${tag})_$i.err | python -c "
import sys, json
lines = [l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')]
if not lines:
    print('$tag'.rjust(20), 'NO JSON LINE -- last stderr line:', (open('gpurun_out/ab_bench_$(basename ${tag})_$i.err').read().strip().splitlines() or ['(empty)'])[-1][:200]); sys.exit(0)
j = json.loads(lines[-1]); k = j.get('kernel_ms_per_step', {})
print('$tag'.rjust(20), 'ms/step %.2f' % j['ms_per_step'], 'frac %.4f' % j['roofline']['frac'], ' '.join('%s %.2f' % (n.replace('cmu_', ''), k[n]) for n in sorted(k, key=lambda n: -k[n])[:4]))"
  done
done
