#!/bin/bash
# Usage (on the GPU box, from the repo root):  bash tools/bench_all.sh r02
# The four bench workloads (BASELINE configs 2-5) at f16, the bf16 line of the default workload, the SparK A/B of tile skipping
# and a two-rank rehearsal of the self-launching data-parallel path (gloo, both ranks on the one GPU).
TAG=${1:-r02}
O=gpurun_out/bench_$TAG
mkdir -p $O
python3 bench.py --steps 20 --warmup 5 --all-kernel-events > $O/recon_f16.json 2> $O/recon_f16.err && tail -c 600 $O/recon_f16.json | cut -c1-300
python3 bench.py --steps 20 --warmup 5 --dtype bf16 --no-cpu-baseline > $O/recon_bf16.json 2> $O/recon_bf16.err
python3 bench.py --workload moco --steps 10 --warmup 3 --all-kernel-events > $O/moco_f16.json 2> $O/moco_f16.err
python3 bench.py --workload joint --steps 5 --warmup 2 --all-kernel-events > $O/joint_f16.json 2> $O/joint_f16.err
python3 bench.py --workload spark --steps 10 --warmup 3 --all-kernel-events > $O/spark_f16.json 2> $O/spark_f16.err
CMU_SPARK_TILES=0 python3 bench.py --workload spark --steps 10 --warmup 3 --no-cpu-baseline --all-kernel-events > $O/spark_f16_dense.json 2> $O/spark_f16_dense.err
CMU_DIST_BACKEND=gloo CMU_SINGLE_DEVICE=1 python3 bench.py --gpus 2 --steps 3 --warmup 1 --batch 8 > $O/recon_2rank_gloo.json 2> $O/recon_2rank_gloo.err
for f in $O/*.json; do echo "== $f"; python3 - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d.get("roofline", {})
    print(d["config"]["workload"], d["dtype"], "n_gpus", d["n_gpus"], "->", d["value"], d["unit"], f"({d['ms_per_step']} ms/step)", "dominant", r.get("kernel"), r.get("achieved"), "frac", r.get("frac"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
except Exception as e:
    print("no JSON line:", e)
PY
done
tail -3 $O/*.err | cut -c1-300
