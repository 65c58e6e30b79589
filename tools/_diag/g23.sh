#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for i in 1 2 3; do
for v in 1 0; do
CMU_POOL_FUSE2=$v python bench.py --workload joint --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('joint fuse2=$v', d['value'], d['ms_per_step'], d['config'].get('loss'))"
done
done
