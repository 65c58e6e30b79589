#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_fwd_ops.py tests/test_gpu_bwd_ops.py -x -q -k "random_shapes or conv3x3_fwd or conv3x3_wgrad" > gpurun_out/relu_tests.log 2>&1 || { tail -30 gpurun_out/relu_tests.log; exit 1; }
tail -2 gpurun_out/relu_tests.log
timeout -k 10 1000 python -m pytest tests/test_gpu_pretrain.py tests/test_gpu_fullsize.py tests/test_gpu_dataparallel.py -x -q -k "joint or cmunet or CM" > gpurun_out/joint_tests.log 2>&1 || { tail -40 gpurun_out/joint_tests.log; exit 1; }
tail -2 gpurun_out/joint_tests.log
for i in 1 2 3; do
for v in 1 0; do
CMU_SHARED_SKIPS=$v python bench.py --workload joint --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('joint shared_skips=$v', d['value'], d['ms_per_step'], d['config'].get('loss'))"
done
done
