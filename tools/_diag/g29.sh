#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for wl in recon moco joint spark; do
for i in 1 2; do
python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
done
CMU_LIB_PATH=tools/_diag/libcmunet_r03.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('recon r03 library', d['value'], d['ms_per_step'], d['roofline']['frac'])"
