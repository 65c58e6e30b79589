#!/usr/bin/env python3
"""Per-entry timing of the HBM-bound passes of the default bench step at their bench shapes (bs 32, 512 x 512, f16 by default):
first-layer conv forward / weight gradient, 1x1 head forward / backward, max-pool forward / backward (sums-only and apply
forms) at the four pooled levels, the BatchNorm-backward apply pass.  Prints ms per launch and algorithmic TB/s.

    python tools/elem_bench.py [f16|bf16|f32] [batch] [size]        (CMU_LIB_PATH selects a library build: same-box A/B)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cmunet_amd import _lib, ops  # noqa: E402
from cmunet_amd.ops import Act  # noqa: E402

dt = sys.argv[1] if len(sys.argv) > 1 else "f16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
S = int(sys.argv[3]) if len(sys.argv) > 3 else 512
dev = torch.device("cuda")
tdt = ops.TORCH_DT[ops.dt_code(dt)]
es = torch.empty(0, dtype=tdt).element_size()
lib = _lib.lib()
torch.manual_seed(0)


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
    return e0.elapsed_time(e1) / n


def act(H, C, scale=True):
    a = Act(torch.randn(B, H, H, C, device=dev).to(tdt))
    if scale:
        a = a.with_transform(torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1, 0)
    return a


def report(name, ms, nbytes):
    print(f"{name:34s} {ms:8.4f} ms  {nbytes / 1e9:7.3f} GB  {nbytes / ms / 1e9:6.2f} TB/s", flush=True)


rows = []
# ---- first layer -----------------------------------------------------------------------------------------------------------
x = torch.randn(B, S, S, device=dev)
mask = (torch.rand(1, S // 16, S // 16, device=dev) < 0.6).to(torch.uint8).repeat_interleave(16, 1).repeat_interleave(16, 2).contiguous()
w1 = torch.randn(64, 1, 3, 3, device=dev) * 0.3
y1 = ops.new_act(B, S, S, 64, dt, dev)
npx = B * S * S


def c1_fwd():
    st = ops.new_stats(B, S, S, 64, dev)
    ops.conv3x3_c1_fwd(x, w1, y1, st, mask, False)


report("conv3x3_c1_fwd", timeit(c1_fwd), npx * (4 + 64 * es))
c1_fwd()
y1t = y1.with_transform(torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.1, 0)
mean, invstd = torch.randn(64, device=dev) * 0.1, torch.rand(64, device=dev) + 0.5
coef = torch.randn(2, 64, device=dev) * 0.01
dA1 = act(S, 64, False)
dW1 = torch.empty(64, 1, 3, 3, device=dev)
ws1 = torch.empty(lib.cmu_conv3x3_c1_wgrad_ws_bytes(B, S, S, 64), dtype=torch.uint8, device=dev)
report("conv3x3_c1_wgrad_bn", timeit(lambda: ops.conv3x3_c1_wgrad_bn(x, dA1, y1t, y1t.scale, y1t.shift, mean, invstd, coef, dW1, ws1, mask, False)),
       npx * (4 + 2 * 64 * es))
if getattr(lib, "cmu_conv3x3_c1_wgrad_bn_w", None) is not None:
    report("conv3x3_c1_wgrad_bn (recomputed y)", timeit(lambda: ops.conv3x3_c1_wgrad_bn(x, dA1, y1t, y1t.scale, y1t.shift, mean, invstd, coef, dW1, ws1, mask, False, w=w1)),
           npx * (4 + 64 * es))

# ---- head ------------------------------------------------------------------------------------------------------------------
wl = torch.randn(2, 64, device=dev) * 0.1
bl = torch.zeros(2, device=dev)
logits = torch.empty(B, 2, S, S, device=dev)
report("conv1x1_head_fwd", timeit(lambda: ops.conv1x1_head_fwd(y1t, wl, bl, logits)), npx * (64 * es + 8))
dl = torch.randn(B, 2, S, S, device=dev) * 1e-3
dWl, dbl = torch.empty(2, 64, device=dev), torch.empty(2, device=dev)
wsh = torch.empty(lib.cmu_conv1x1_head_bwd_ws_bytes(B, S, S, 64, 2), dtype=torch.uint8, device=dev)
bnws = torch.empty(lib.cmu_bn_bwd_ws_bytes(64), dtype=torch.uint8, device=dev)
report("conv1x1_head_bwd (sums only)", timeit(lambda: ops.conv1x1_head_bwd(dl, y1t, wl, None, dWl, dbl, wsh, mean, invstd, bnws)), npx * (64 * es + 8))
dY1 = ops.new_act(B, S, S, 64, dt, dev)
report("conv1x1_head_bn_apply", timeit(lambda: ops.conv1x1_head_bn_apply(dl, y1t, wl, mean, invstd, coef, dY1)), npx * (2 * 64 * es + 8))
report("bn_bwd_apply 64@%d (in place)" % S, timeit(lambda: ops.bn_bwd_apply(dA1, y1t, mean, invstd, coef, dA1)), npx * 3 * 64 * es)
del dY1

# ---- pools -----------------------------------------------------------------------------------------------------------------
tot = {"fwd": 0.0, "bwd1": 0.0, "bwd2": 0.0}
for lvl, C in enumerate((64, 128, 256, 512)):
    H = S >> lvl
    y = act(H, C)
    mu, isd, cf = torch.randn(C, device=dev) * 0.1, torch.rand(C, device=dev) + 0.5, torch.randn(2, C, device=dev) * 0.01
    pooled = ops.new_act(B, H // 2, H // 2, C, dt, dev)
    n = B * H * H * C * es
    t = timeit(lambda: ops.bnrelu_maxpool_fwd(y, pooled))
    tot["fwd"] += t
    report(f"bnrelu_maxpool_fwd {C}@{H}", t, n * 1.25)
    dP = Act(torch.randn(B, H // 2, H // 2, C, device=dev).to(tdt))
    dS = Act(torch.randn(B, H, H, 2 * C, device=dev).to(tdt), C, C)        # right half of a concat gradient
    ws = torch.empty(lib.cmu_bn_bwd_ws_bytes(C), dtype=torch.uint8, device=dev)
    t = timeit(lambda: ops.maxpool_bwd(dP, dS, y, None, mu, isd, ws))
    tot["bwd1"] += t
    report(f"maxpool_bwd sums {C}@{H}", t, n * 2.25)
    dY = ops.new_act(B, H, H, C, dt, dev)
    t = timeit(lambda: ops.maxpool_bwd_apply(dP, dS, y, mu, isd, cf, dY))
    tot["bwd2"] += t
    report(f"maxpool_bwd_apply {C}@{H}", t, n * 3.25)
    del y, pooled, dP, dS, dY
print("pool totals (ms): " + " ".join(f"{k}={v:.4f}" for k, v in tot.items()), flush=True)
