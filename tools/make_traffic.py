"""Turn the rocprofv3 outputs of tools/profile_round.sh <tag> (gpurun_out/) into the committed summaries:
profiles/<tag>_kernel_stats.csv, profiles/<tag>_hbm_traffic_per_step.csv and profiles/traffic.json.
HBM bytes per the MI355X guide: FETCH_SIZE / WRITE_SIZE are in KB (x1024); FETCH_SIZE is doubled on gfx950."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
dtype = sys.argv[2] if len(sys.argv) > 2 else "f16"          # storage dtype of the profiled bench run
trait = {"f16": "<F16Traits", "bf16": "<BF16Traits", "f32": "<F32Traits"}[dtype]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go = os.path.join(root, "gpurun_out")


def one_step(pattern):
    f = max(glob.glob(os.path.join(go, pattern, "*", "*_counter_collection.csv")), key=os.path.getmtime)   # newest run
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        e = disp.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"], "c": {}})
        e["c"][r["Counter_Name"]] = float(r["Counter_Value"])
    items = list(disp.values())
    idx = [i for i, e in enumerate(items) if e["name"].startswith("adam_kernel")]
    return items[idx[-2] + 1: idx[-1] + 1]          # the launches of the last step


def short(n):
    return n.split("(")[0].replace("void ", "").strip()


fetch, write, launches = collections.Counter(), collections.Counter(), collections.Counter()
for e in one_step(f"pmc_fetch_{tag}"):
    fetch[short(e["name"])] += e["c"].get("FETCH_SIZE", 0.0) * 1024 * 2
    launches[short(e["name"])] += 1
for e in one_step(f"pmc_write_{tag}"):
    write[short(e["name"])] += e["c"].get("WRITE_SIZE", 0.0) * 1024
rows = sorted(launches, key=lambda k: -(fetch[k] + write[k]))
with open(os.path.join(root, "profiles", f"{tag}_hbm_traffic_per_step.csv"), "w") as f:
    f.write("kernel,launches_per_step,fetch_GB_x2_corrected,write_GB\n")
    for k in rows:
        f.write(f"\"{k}\",{launches[k]},{fetch[k] / 1e9:.3f},{write[k] / 1e9:.3f}\n")
tj = {}
for k in rows:
    base = k.split("<")[0]
    if base in ("conv_igemm5_kernel", "conv_igemm3p_kernel", "conv_igemm3_kernel", "conv_igemm_kernel", "conv_wgrad2_kernel", "conv_wgrad2s_kernel", "conv_wgrad_kernel", "conv_wgradT2_kernel", "conv_gemm_kernel", "conv_gemm_s_kernel") and trait in k:
        d = tj.setdefault(base, {"launches_per_step": 0, "fetch": 0.0, "write": 0.0, "instances": []})
        d["launches_per_step"] += launches[k]
        d["fetch"] += fetch[k]
        d["write"] += write[k]
        d["instances"].append(k)
out = {}
for base, d in tj.items():
    n = d["launches_per_step"]
    out[base] = {"instances": d["instances"], "launches_per_step": n, "hbm_bytes_per_launch": (d["fetch"] + d["write"]) / n,
                 "fetch_bytes_per_launch": d["fetch"] / n, "write_bytes_per_launch": d["write"] / n,
                 "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on `bench.py --steps 1 --warmup 1`, KB*1024, FETCH_SIZE "
                           f"doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B); {tag}"}
out["_command"] = {"dtype": dtype, "workload": "recon", "tag": tag,
                   "command": "python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events (rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE)"}
json.dump(out, open(os.path.join(root, "profiles", "traffic.json"), "w"), indent=1)
st = glob.glob(os.path.join(go, f"prof_{tag}", "*", "*_kernel_stats.csv"))
if st:
    shutil.copy(max(st, key=os.path.getmtime), os.path.join(root, "profiles", f"{tag}_kernel_stats.csv"))
print("wrote profiles/ for", tag, "-", ", ".join(f"{k}: {v['hbm_bytes_per_launch'] / 1e9:.2f} GB/launch x {v['launches_per_step']}" for k, v in out.items() if k != "_command"))
