#!/bin/bash
# PMC snapshot of the conv kernels (one step): usage  bash tools/pmc_conv.sh <tag> [env assignments...]
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/pmcA_$TAG -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events > $R/gpurun_out/pmcA_$TAG.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmcB_$TAG -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events > $R/gpurun_out/pmcB_$TAG.log 2>&1
python3 - <<PY
import csv, glob, collections
for d in ("pmcA_$TAG", "pmcB_$TAG"):
    f = glob.glob("$R/gpurun_out/%s/*/*_counter_collection.csv" % d)[0]
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        e = disp.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"], "grid": int(r["Grid_Size"]), "wg": int(r["Workgroup_Size"]), "c": {}})
        e["c"][r["Counter_Name"]] = float(r["Counter_Value"])
    items = list(disp.values())
    idx = [i for i, e in enumerate(items) if e["name"].startswith("adam_kernel")]
    step = items[idx[-2] + 1: idx[-1] + 1]
    agg = collections.OrderedDict()
    for e in step:
        k = e["name"].split("(")[0]
        if "conv_igemm" not in k and "conv_wgrad" not in k: continue
        a = agg.setdefault(k, collections.Counter())
        a["launches"] += 1
        for c, v in e["c"].items(): a[c] += v
    for k, a in agg.items():
        print(d, k, {c: int(v) for c, v in a.items()})
PY
