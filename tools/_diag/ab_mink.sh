for i in 1 2 3; do
for cfg in "CMU_X=0" "CMU_V5_MIN_K=64 CMU_V5_MIN_K_BST=128" "CMU_V5_MIN_K_BST=128" "CMU_V5_MIN_K=64"; do
  env $cfg python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab.log 2> gpurun_out/ab.err
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/ab.log") if l.startswith("{")][-1])
k=d["kernel_ms_per_step"]
print("%-44s %.3f ms/step frac %.4f fwd %.2f dgrad_bn %.2f" % ("$cfg", d["ms_per_step"], d["roofline"]["frac"], k["cmu_conv3x3_fwd"], k["cmu_conv3x3_dgrad_bn"]))
PY
done; done
