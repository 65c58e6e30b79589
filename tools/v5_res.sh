#!/bin/bash
# resource usage of the conv_igemm5 instantiations + the ISA of one of them in /tmp/v5 (no GPU needed)
mkdir -p /tmp/v5
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage -save-temps=obj -c /root/repo/cmunet_amd/csrc/conv_igemm.hip -o /tmp/v5/conv_igemm.o "$@" 2> /tmp/v5/build.log
grep -v "^remark" /tmp/v5/build.log | head -20
grep -A 8 "conv_igemm5_kernelI9F16" /tmp/v5/build.log | grep "Name\|VGPRs\|Scratch" | sed -e 's/remark: .*: //'
