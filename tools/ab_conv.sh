#!/bin/bash
# A/B of the 3x3 conv kernels inside one gpurun call (same box, back to back)
for cfg in "$@"; do
  env $cfg python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/ab.log 2>&1
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab.log").read().strip().splitlines()[-1])
print("$cfg:", d["value"], "img/s", d["kernel_ms_per_step"]["cmu_conv3x3_fwd"], "ms conv3x3", d["roofline"]["achieved"], "TF")
PY
done
