python -m pytest tests/test_gpu_fwd_ops.py tests/test_gpu_bwd_ops.py -q -k "conv3x3" > gpurun_out/m16_t1.log 2>&1; tail -15 gpurun_out/m16_t1.log | cut -c1-300
L=contrastive-masked-unet_amd/csrc/libcmunet_hip.so
for s in 0 1; do echo "CMU_CONV_MFMA16=$s"; for cfg in "256 128 128" "128 256 256" "64 512 512" "32 1024 1024" "64 1024 512" "512 64 128"; do CMU_SWEEP_DT=1 CMU_CONV_MFMA16=$s python tools/igemm_stamps.py $L $cfg 2>/dev/null | grep layer; done; done
