#!/bin/bash
# 3x3 conv entry on shapes whose last 16 x 32 tile is half empty:  bash tools/slim_sweep.sh <batch> <dt> [ENV=VAL ...]
L=cmunet_amd/csrc/libcmunet_hip.so
export CMU_SWEEP_B=${1:-16} CMU_SWEEP_DT=${2:-0}
shift 2
for kv in "$@"; do export "$kv"; done
for cfg in "16 1024 1024" "16 512 1024" "14 1024 1024" "48 256 256" "112 128 128" "80 256 256"; do
  python3 tools/igemm_stamps.py $L $cfg | grep layer
done
