#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for WL in recon moco joint; do
  rm -rf $R/gpurun_out/tl_$WL
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_$WL -- python3 $R/bench.py --workload $WL --steps 2 --warmup 2 --no-cpu-baseline --no-kernel-events > $R/gpurun_out/tl_$WL.log 2>&1 || exit 1
  python3 $R/tools/step_timeline.py $R/gpurun_out/tl_$WL > $R/gpurun_out/timeline_$WL.txt || exit 1
  rm -rf $R/gpurun_out/tl_$WL
done
echo done
