#!/bin/bash
# A/B of the 3x3 conv kernel variants inside one gpurun call (same box, back to back)
for v in 0 1 2; do
  CMU_IG2_VARIANT=$v python bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/ab_v$v.log 2>&1
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab_v$v.log").read().strip().splitlines()[-1])
print("variant $v", d["value"], "img/s", d["kernel_ms_per_step"]["cmu_conv3x3_fwd"], "ms conv3x3", d["roofline"]["achieved"], "TF")
PY
done
