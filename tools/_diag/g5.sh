python -m pytest tests/test_gpu_sparse_tiles.py tests/test_gpu_bwd_ops.py -x -q > gpurun_out/t5.log 2>&1; tail -3 gpurun_out/t5.log
for L in tools/_diag/libcmunet_r03.so "" tools/_diag/libcmunet_r03.so ""; do CMU_LIB_PATH=$L python bench.py --workload spark --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events 2>/dev/null | python -c "
import sys, json
j = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('spark ${L:-tree}', j['value'], j['ms_per_step'])"; done > gpurun_out/w5.log 2>&1; cat gpurun_out/w5.log
python tools/igemm3p_stamps.py 512 64 64 > gpurun_out/stamps_64_64.txt 2>&1; head -30 gpurun_out/stamps_64_64.txt
python tools/igemm3p_stamps.py 512 128 64 > gpurun_out/stamps_128_64.txt 2>&1
python tools/igemm3p_stamps.py 256 128 128 > gpurun_out/stamps_128_128.txt 2>&1
bash tools/pmc_conv.sh r04a > gpurun_out/pmc_conv_r04a.txt 2>&1; tail -30 gpurun_out/pmc_conv_r04a.txt | cut -c1-400
