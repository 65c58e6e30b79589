python -m pytest tests/test_gpu_skinny.py tests/test_gpu_bwd_ops.py tests/test_gpu_fwd_ops.py -x -q > gpurun_out/t2.log 2>&1; tail -4 gpurun_out/t2.log
( for L in tools/_diag/libcmunet_r03.so ""; do echo "== lib ${L:-tree}"; CMU_LIB_PATH=$L python tools/skinny_bench.py 32 262144 1536; CMU_LIB_PATH=$L python tools/skinny_bench.py 32 50176 1536; done; python tools/skinny_bench.py 256 50176 1536; python tools/skinny_bench.py 64 50176 1536 ) > gpurun_out/sk2.log 2>&1
bash tools/ab_bench.sh tools/_diag/libcmunet_r03.so > gpurun_out/ab_r04a.log 2>&1; cat gpurun_out/ab_r04a.log
