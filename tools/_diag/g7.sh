python -m pytest tests/test_gpu_sparse_tiles.py tests/test_gpu_pretrain.py -x -q -k "spark or tile or pixel or rows or pools" > gpurun_out/t7.log 2>&1; tail -3 gpurun_out/t7.log
for e in "" "CMU_SPARK_TILES=0" "" "CMU_SPARK_TILES=0"; do env $e python bench.py --workload spark --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events 2>/dev/null | python -c "
import sys, json
j = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('spark [$e]', j['value'], j['ms_per_step'])"; done
