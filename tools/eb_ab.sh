#!/bin/bash
# same-box A/B of the element-wise passes: bash tools/eb_ab.sh lib1.so lib2.so ... ("-" = the product library) -> gpurun_out/eb_<name>.log
mkdir -p gpurun_out
for L in "$@"; do
  if [ "$L" = "-" ]; then unset CMU_LIB_PATH; n=tree; else export CMU_LIB_PATH=$PWD/$L; n=$(basename $L .so); fi
  python tools/elem_bench.py > gpurun_out/eb_$n.log 2>&1 || { echo "FAILED $L"; tail -5 gpurun_out/eb_$n.log; }
  echo "== $n"; cat gpurun_out/eb_$n.log
done
