#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_bwd_ops.py tests/test_gpu_fullsize.py -x -q -k "convT" > gpurun_out/wgt_tests.log 2>&1 || { tail -30 gpurun_out/wgt_tests.log; exit 1; }
tail -2 gpurun_out/wgt_tests.log
for v in 1 0 1 0; do echo "== CMU_WGT2_NX256=$v"; CMU_WGT2_NX256=$v python tools/convt_sweep.py f16 2>&1 | grep wgrad; done
for i in 1 2 3; do
for v in 1 0; do
CMU_WGT2_NX256=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('recon nx256=$v', d['value'], d['ms_per_step'], d['config'].get('loss'))"
done
done
