#!/bin/bash
# full GPU suite -> gpurun_out/tests_<tag>.log
set -o pipefail
mkdir -p gpurun_out
tag=${1:-x}
timeout -k 10 1000 python -m pytest tests -q -m gpu -x --durations=15 > gpurun_out/tests_$tag.log 2>&1
rc=$?
tail -40 gpurun_out/tests_$tag.log
exit $rc
