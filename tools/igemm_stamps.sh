#!/bin/bash
# builds the diagnostic library (in-kernel s_memtime stamps, -DCMU_IG_STAMPS) next to the product one; never shipped
cd "$(dirname "$0")/../cmunet_amd/csrc" && mkdir -p ../../tools/_diag && \
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DCMU_IG_STAMPS -shared -o ../../tools/_diag/libcmunet_stamps.so conv_igemm.hip conv_wgrad.hip elementwise.hip
