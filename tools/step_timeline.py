#!/usr/bin/env python3
"""Ordered kernel timeline of ONE step from a rocprofv3 --kernel-trace CSV.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --workload spark --steps 2 --warmup 2 ...
    python tools/step_timeline.py gpurun_out/tl [marker-substring]

The step is the span between the last two launches of the marker kernel (default: the optimiser's last kernel of the workload,
guessed from the trace).  Prints start offset (us), duration (us), gap to the previous kernel's end, grid, and a short name; then
the per-name totals of that step.
"""
import csv
import glob
import re
import sys
from collections import OrderedDict

d = sys.argv[1]
files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
if not files:
    sys.exit("no *kernel_trace.csv under " + d)
rows = []
for f in files:
    with open(f) as fh:
        rows += list(csv.DictReader(fh))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
marker = sys.argv[2] if len(sys.argv) > 2 else None
if marker is None:
    for m in ("lamb_apply_kernel", "adam_ema_kernel", "adam_kernel", "sgd_kernel"):
        if any(m in n for n in names):
            marker = m
            break
idx = [i for i, n in enumerate(names) if marker in n]
if len(idx) < 2:
    sys.exit("marker %r seen %d times" % (marker, len(idx)))
a, b = idx[-2] + 1, idx[-1] + 1
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"])


def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", n)
    n = n.replace("F16Traits", "F16").replace("BF16Traits", "BF16").replace("F32Traits", "F32")
    return n[:70]


prev_end = t0
tot = OrderedDict()
busy = 0
print(f"# step = kernels {a}..{b - 1} of {len(rows)} (marker {marker}); {len(step)} launches")
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    g = "x".join(str(int(r[k]) // max(1, int(r[w]))) for k, w in (("Grid_Size_X", "Workgroup_Size_X"), ("Grid_Size_Y", "Workgroup_Size_Y")) if k in r and w in r)
    nm = short(r["Kernel_Name"])
    print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:9.1f} gap {(s - prev_end) / 1e3:7.1f}  {g:>10s}  {nm}")
    prev_end = max(prev_end, e)
    busy += e - s
    k = tot.setdefault(nm, [0, 0])
    k[0] += 1
    k[1] += e - s
span = prev_end - t0
print(f"# span {span / 1e6:.3f} ms, kernel time {busy / 1e6:.3f} ms, idle {(span - busy) / 1e6:.3f} ms")
for nm, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"# {t / 1e6:8.3f} ms {c:4d} x {nm}")
