#!/bin/bash
# Usage (on the GPU box, from the repo root):  bash tools/profile_workloads.sh r02
# rocprofv3 kernel-trace stats of the three other bench workloads (BASELINE configs 3-5); copy the kernel_stats.csv files into profiles/.
set -e
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for WL in moco joint spark; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_$WL -- python3 $R/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events > $R/gpurun_out/prof_${TAG}_$WL.log 2>&1
  tail -1 $R/gpurun_out/prof_${TAG}_$WL.log | cut -c1-300
done
echo profiled workloads $TAG
