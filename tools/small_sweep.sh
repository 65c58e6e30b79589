#!/bin/bash
# per-layer time of the 3x3 conv entry on SMALL-batch shapes (finetuning: 256 x 256 inputs, bs 6 ... 32):
#   bash tools/small_sweep.sh <batch> <dt: 0 f32 | 1 f16 | 2 bf16> [ENV=VAL ...]     e.g.  bash tools/small_sweep.sh 6 0 CMU_CONV_NARROW=0
L=cmunet_amd/csrc/libcmunet_hip.so
export CMU_SWEEP_B=${1:-6} CMU_SWEEP_DT=${2:-0}
shift 2
for kv in "$@"; do export "$kv"; done
for cfg in "256 64 64" "128 128 128" "64 256 256" "32 512 512" "16 1024 1024" "16 512 1024" "32 1024 512" "64 512 256" "128 256 128"; do
  python3 tools/igemm_stamps.py $L $cfg | grep layer
done
