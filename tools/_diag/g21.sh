#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for i in 1 2; do
python bench.py --workload spark --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('spark', d['value'], d['ms_per_step'])"
CMU_SPARK_TILES=0 python bench.py --workload spark --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('spark CMU_SPARK_TILES=0', d['value'], d['ms_per_step'])"
CMU_LIB_PATH=tools/_diag/libcmunet_r03.so CMU_SPARK_C1_TILES=0 python bench.py --workload spark --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('spark r03 library', d['value'], d['ms_per_step'])"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('recon', d['value'], d['ms_per_step'])"
done
bash tools/profile_workloads.sh r04 2>&1 | tail -4
bash tools/_diag/g14.sh
