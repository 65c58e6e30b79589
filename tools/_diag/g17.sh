#!/bin/bash
for L in "" tools/_diag/libcmunet_cells_x0.so tools/_diag/libcmunet_cells_x9.so tools/_diag/libcmunet_cells_x10.so; do
  echo "== ${L:-product (XLOG 8)}"
  CMU_LIB_PATH=$L python tools/cells_bench.py || exit 1
done
