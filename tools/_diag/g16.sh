#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_sparse_tiles.py -x -q -k "cells" > gpurun_out/sp_tests.log 2>&1 || { tail -30 gpurun_out/sp_tests.log; exit 1; }
tail -2 gpurun_out/sp_tests.log
bash tools/_diag/g14.sh
