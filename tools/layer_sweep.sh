#!/bin/bash
# per-layer throughput of the 3x3 conv entry on the bench shapes: bash tools/layer_sweep.sh [ENV=VAL ...]
L=cmunet_amd/csrc/libcmunet_hip.so
for kv in "$@"; do export "$kv"; done
for cfg in "512 64 64" "256 128 128" "128 256 256" "64 512 512" "32 1024 1024" "64 1024 512" "256 256 128" "512 128 64"; do
  python tools/igemm_stamps.py $L $cfg | grep layer
done
