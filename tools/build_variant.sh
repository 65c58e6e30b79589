#!/bin/bash
# Variant build of the library for same-box A/B runs (CMU_LIB_PATH / tools/ab_lib.sh):
#   bash tools/build_variant.sh NAME "file1.hip file2.hip" "-DFOO=1 -DBAR=2"
# recompiles the listed sources of cmunet_amd/csrc with the extra flags, links them with the tree's other objects into
# tools/_diag/libcmunet_NAME.so (git-ignored, travels to the GPU box with gpurun).  Run `make -C cmunet_amd/csrc` first.
set -e
NAME=$1; FILES=$2; FLAGS=$3
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/cmunet_amd/csrc
T=$(mktemp -d)
objs=""
for f in $C/*.hip; do
  b=$(basename $f .hip)
  if echo " $FILES " | grep -q " $b.hip "; then
    extra=""; [ "$b" = "augment" ] && extra="-ffp-contract=off"
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable $extra $FLAGS -I$C -c $f -o $T/$b.o &
    objs="$objs $T/$b.o"
  else
    objs="$objs $C/$b.o"
  fi
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/_diag/libcmunet_$NAME.so $objs
rm -rf $T
echo built tools/_diag/libcmunet_$NAME.so
