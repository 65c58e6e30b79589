#!/bin/bash
# Usage (on the GPU box, from the repo root):  bash tools/profile_round.sh r01
# Writes the rocprofv3 kernel-trace stats of the default bench command and one PMC pass (HBM traffic of the
# dominant kernel) under gpurun_out/; copy the summaries into profiles/ afterwards.
set -e
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events > $R/gpurun_out/prof_$TAG.log 2>&1
tail -1 $R/gpurun_out/prof_$TAG.log | cut -c1-400
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch_$TAG -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events > $R/gpurun_out/pmc_fetch_$TAG.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write_$TAG -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events > $R/gpurun_out/pmc_write_$TAG.log 2>&1
echo profiled $TAG
