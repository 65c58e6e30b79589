#!/bin/bash
# same-box timing of library variants through tools/v5_check.py: bash tools/v5_variants.sh lib1.so lib2.so ...  ("-" = the tree's library)
for l in "$@"; do
  [ "$l" = "-" ] && l=cmunet_amd/csrc/libcmunet_hip.so
  echo "== $l"
  python tools/v5_check.py $l 2>&1 | grep -v amdgpu.ids | sed -e 's/\[conv_igemm[0-9a-z_]*\]//g' -e 's/| y relL2.*//'
done
