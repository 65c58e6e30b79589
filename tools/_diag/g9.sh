export CMU_SWEEP_DT=1
for L in cmunet_amd/csrc/libcmunet_hip.so tools/_diag/libcmunet_mockd.so cmunet_amd/csrc/libcmunet_hip.so tools/_diag/libcmunet_mockd.so; do
  echo "== $L"
  for cfg in "512 64 64" "512 128 64" "256 64 128" "256 128 128" "256 256 128" "128 256 256" "64 512 512" "32 1024 1024"; do
    python tools/igemm_stamps.py $L $cfg 2>/dev/null | grep layer
  done
done
