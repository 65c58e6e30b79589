#!/usr/bin/env python3
"""The sparse encoder's patch-organised element-wise passes (csrc/sparse_elem.hip) stand-alone at the five level shapes of the SparK
bench step (bs 32, 512 x 512, 32 x 32 patch map, 25 % active): BatchNorm-backward apply (full zeros / border frames), mask select,
max-pool backward, mask-token sum -- against the pixel-organised kernels.  us per launch.

    python tools/cells_bench.py [f16|bf16|f32] [batch] [size]          (CMU_LIB_PATH selects a library build: same-box A/B)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cmunet_amd import ops  # noqa: E402
from cmunet_amd.ops import Act  # noqa: E402

dt = sys.argv[1] if len(sys.argv) > 1 else "f16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
S = int(sys.argv[3]) if len(sys.argv) > 3 else 512
f = S // 16
dev = torch.device("cuda")
tdt = ops.TORCH_DT[ops.dt_code(dt)]
torch.manual_seed(0)
g = torch.Generator().manual_seed(1)
act = torch.zeros(B, f * f, dtype=torch.uint8)
for b in range(B):
    act[b, torch.randperm(f * f, generator=g)[: f * f // 4]] = 1
act = act.view(B, f, f).to(dev)


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print(f"{'level':14s} {'apply':>8s} {'(ring)':>8s} {'(pixel)':>8s} {'select':>8s} {'(ring)':>8s} {'(pixel)':>8s} {'pool':>8s} {'(pixel)':>8s} {'tokens':>8s} {'(pixel)':>8s}")
tot = [0.0] * 10
for lvl, C in enumerate((64, 128, 256, 512, 1024)):
    H = S >> lvl
    y = Act(torch.randn(B, H, H, C, device=dev).to(tdt), 0, C, torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1, 0)
    dA = Act(torch.randn(B, H, H, C, device=dev).to(tdt))
    out = ops.new_act(B, H, H, C, dt, dev)
    mean, invstd, coef = torch.randn(C, device=dev) * 0.1, torch.rand(C, device=dev) + 0.5, torch.randn(2, C, device=dev) * 0.01
    r = [timeit(lambda: ops.bn_bwd_apply_masked(dA, y, mean, invstd, coef, out, act)),
         timeit(lambda: ops.bn_bwd_apply_masked(dA, y, mean, invstd, coef, out, act, ring=True)),
         timeit(lambda: ops.bn_bwd_apply_masked(dA, y, mean, invstd, coef, out, act, cells=False)),
         timeit(lambda: ops.mask_select(y, act, out, relu=True)),
         timeit(lambda: ops.mask_select(y, act, out, relu=True, ring=True)),
         timeit(lambda: ops.mask_select(y, act, out, relu=True, cells=False))]
    if H // f >= 2:
        dP = Act(torch.randn(B, H // 2, H // 2, C, device=dev).to(tdt))
        r += [timeit(lambda: ops.maxpool_bwd_masked(dP, dA, y, out, act)), timeit(lambda: ops.maxpool_bwd_masked(dP, dA, y, out, act, cells=False))]
    else:
        r += [0.0, 0.0]
    tok = torch.empty(C, device=dev)
    r += [timeit(lambda: ops.cells_channel_sum(dA, act, tok, invert=True)), timeit(lambda: ops.masked_channel_stats(dA, act, invert=True))]
    print(f"{C:5d} @ {H:4d}   " + " ".join(f"{v:8.1f}" for v in r), flush=True)
    tot = [a + b for a, b in zip(tot, r)]
    del y, dA, out
print(f"{'sum':14s} " + " ".join(f"{v:8.1f}" for v in tot))
