#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_sparse_tiles.py -x -q > gpurun_out/sp_tests.log 2>&1 || { tail -30 gpurun_out/sp_tests.log; exit 1; }
tail -3 gpurun_out/sp_tests.log
timeout -k 10 900 python -m pytest tests/test_gpu_pretrain.py tests/test_gpu_fullsize.py tests/test_gpu_bwd_ops.py tests/test_gpu_model.py -x -q -k "spark or SparK or lamb or LAMB" > gpurun_out/sp_tests2.log 2>&1 || { tail -30 gpurun_out/sp_tests2.log; exit 1; }
tail -3 gpurun_out/sp_tests2.log
for i in 1 2; do
python bench.py --workload spark --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('spark cells', d['value'], d['ms_per_step'])"
CMU_SPARK_CELLS=0 CMU_SPARK_C1_TILES=0 python bench.py --workload spark --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('spark pixel-form', d['value'], d['ms_per_step'])"
done
