python -m pytest tests/test_gpu_pretrain.py tests/test_gpu_dataparallel.py tests/test_gpu_sparse_tiles.py tests/test_gpu_fullsize.py -x -q -k "spark or arena or overlapped" > gpurun_out/t11.log 2>&1; tail -3 gpurun_out/t11.log
CMU_DIST_BACKEND=gloo CMU_SINGLE_DEVICE=1 python bench.py --gpus 2 --workload spark --batch 8 --size 256 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events 2>&1 | python -c "
import sys, json
ls=[l for l in sys.stdin if l.startswith('{')]
j=json.loads(ls[-1]) if ls else {}
print('spark 2 ranks', j.get('n_gpus'), j.get('value'), j.get('rccl'))"
for i in 1 2; do python bench.py --workload spark --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events 2>/dev/null | python -c "
import sys, json
j = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('spark', j['value'], j['ms_per_step'])"; done
