#!/bin/bash
L=${1:-cmunet_amd/csrc/libcmunet_hip.so}
for cfg in "512 64 64" "256 128 128" "128 256 256" "64 512 512" "32 1024 1024" "64 1024 512" "256 256 128" "512 128 64"; do
  python tools/wgrad_stamps.py $L $cfg | grep wgrad
done
