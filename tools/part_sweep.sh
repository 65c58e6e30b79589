#!/bin/bash
# 3x3 conv entry on the levels of a 224 x 224 input (partial 16 x 32 tiles):  bash tools/part_sweep.sh <batch> <dt> [ENV=VAL ...]
L=cmunet_amd/csrc/libcmunet_hip.so
export CMU_SWEEP_B=${1:-128} CMU_SWEEP_DT=${2:-1}
shift 2
for kv in "$@"; do export "$kv"; done
for cfg in "112 128 128" "112 64 128" "56 256 256" "56 128 256" "28 512 512" "28 1024 512" "56 512 256" "112 256 128"; do
  python3 tools/igemm_stamps.py $L $cfg | grep layer
done
