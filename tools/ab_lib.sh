#!/bin/bash
# same-box A/B of library builds: bash tools/ab_lib.sh lib1.so lib2.so ...  ("-" = the product library); prints the per-entry ms of a short bench
for L in "$@"; do
  if [ "$L" = "-" ]; then unset CMU_LIB_PATH; else export CMU_LIB_PATH=$PWD/$L; fi
  python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k = d['kernel_ms_per_step']
print('$L', d['value'], ' '.join(f'{n[4:]}={k[n]}' for n in ('cmu_bnrelu_maxpool_fwd', 'cmu_conv1x1_head_fwd', 'cmu_conv3x3_c1_fwd', 'cmu_conv3x3_c1_wgrad_bn', 'cmu_bn_bwd_apply', 'cmu_maxpool_bwd', 'cmu_conv1x1_head_bwd', 'cmu_conv3x3_wgrad', 'cmu_convT2x2_wgrad')))"
done
