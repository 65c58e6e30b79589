python -m pytest tests/test_gpu_pretrain.py -x -q -k "joint_step_reference_geometry or gate_forced" > gpurun_out/t3a.log 2>&1; tail -15 gpurun_out/t3a.log
python -m pytest tests/test_gpu_fullsize.py -x -q -k "spark" > gpurun_out/t3b.log 2>&1; tail -8 gpurun_out/t3b.log
python -m pytest tests/test_gpu_skinny.py -x -q > gpurun_out/t3c.log 2>&1; tail -3 gpurun_out/t3c.log
( python tools/skinny_bench.py 32 262144 1536; python tools/skinny_bench.py 32 50176 1536 ) > gpurun_out/sk3.log 2>&1; grep wgrad gpurun_out/sk3.log | cut -c1-200
bash tools/ab_bench.sh tools/_diag/libcmunet_r03.so > gpurun_out/ab_r04b.log 2>&1; cat gpurun_out/ab_r04b.log
