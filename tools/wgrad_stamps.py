"""Diagnostic: per-layer throughput of cmu_conv3x3_wgrad and (with the -DCMU_IG_STAMPS build) its in-kernel timeline."""
import ctypes
import sys

import numpy as np
import torch

import os
LIB = sys.argv[1]
B = int(os.environ.get("CMU_SWEEP_B", "32"))
DT = int(os.environ.get("CMU_SWEEP_DT", "2"))           # 0: f32, 1: f16, 2: bf16
TDT = {0: torch.float32, 1: torch.float16, 2: torch.bfloat16}[DT]
H = W = int(sys.argv[2]); Cin = int(sys.argv[3]); Cout = int(sys.argv[4])
dev = torch.device("cuda:0")
lib = ctypes.CDLL(LIB)
lib.cmu_conv3x3_wgrad_ws_bytes.restype = ctypes.c_int64
lib.cmu_last_error.restype = ctypes.c_char_p
vp, i64 = ctypes.c_void_p, ctypes.c_int64
torch.manual_seed(0)
x = torch.randn(B, H, W, Cin, device=dev).to(TDT)
dy = torch.randn(B, H, W, Cout, device=dev).to(TDT)
sc = torch.rand(Cin, device=dev) + 0.5
if os.environ.get("CMU_SWEEP_DATA", "") == "zero_both":      # operands that do not toggle the multipliers (power limit vs schedule)
    x.zero_(); dy.zero_()
sh = torch.randn(Cin, device=dev) * 0.1
dW = torch.empty(Cout, Cin, 3, 3, device=dev)
ws = torch.empty(lib.cmu_conv3x3_wgrad_ws_bytes(B, H, W, Cin, Cout, DT), dtype=torch.uint8, device=dev)


def run():
    rc = lib.cmu_conv3x3_wgrad(vp(x.data_ptr()), i64(Cin), vp(sc.data_ptr()), vp(sh.data_ptr()), 0, vp(dy.data_ptr()), i64(Cout),
                               vp(dW.data_ptr()), B, H, W, Cin, Cout, DT, vp(ws.data_ptr()), vp(0))
    assert rc == 0, lib.cmu_last_error()


for _ in range(10):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"wgrad {Cin}->{Cout} @ {H}x{W} B={B}{(' f32', ' f16', '')[DT]}: {ms:.3f} ms  {2.0 * B * H * W * Cin * Cout * 9 / ms / 1e9:.0f} TFLOP/s (incl. reduce)")
if not hasattr(lib, "cmu_debug_wg_stamps"):
    sys.exit(0)
buf = np.zeros(64 * 16 * 8, dtype=np.uint64)
assert lib.cmu_debug_wg_stamps(buf.ctypes.data_as(vp)) == 0
names = sys.argv[5].split(",") if len(sys.argv) > 5 else ["mfma", "barrierA", "stage", "barrierB"]
if (DT != 0 and Cout % 128 == 0 and Cin % 64 == 0) or (DT == 0 and Cout % 64 == 0 and Cin % 64 == 0):      # wide kernels (conv_wgrad2.inc / conv_wgrad2f.inc): waves 0 and 4 of 32 workgroups stamp through LDS
    st2 = buf.reshape(32, 2, 16, 8).astype(np.int64)
    for grp in range(2):
        stg = st2[:, grp]
        stg = stg[stg[:, 0, 0] > 0][:, :15]
        nst = int((stg[0, :, 0] > 0).sum())
        print(f"wave {4 * grp}: workgroups {stg.shape[0]}, tiles stamped {nst}")
        for k, nm in enumerate(names):
            d = stg[:, 1:nst - 1, k + 1] - stg[:, 1:nst - 1, k]
            print(f"  {nm:11s} avg {d.mean():8.0f} cyc (min {d.min()}, max {d.max()})")
        print(f"  per-tile total {(stg[:, 2:nst - 1, 0] - stg[:, 1:nst - 2, 0]).mean():.0f} cyc")
    clk = st2[:, 0, 15]                      # slot 15 of wave 0: (memtime, memrealtime) before and after the loop
    clk = clk[clk[:, 0] > 0]
    if clk.shape[0]:
        mhz = (clk[:, 2] - clk[:, 0]) / np.maximum(clk[:, 3] - clk[:, 1], 1) * 100.0
        print(f"  shader clock over the loop: {mhz.mean():.0f} MHz (s_memtime / s_memrealtime at 100 MHz)")
    sys.exit(0)
st = buf.reshape(64, 16, 8).astype(np.int64)
st = st[st[:, 0, 0] > 0]
nst = int((st[0, :, 0] > 0).sum()) - 1
print("blocks", st.shape[0], "tiles/block stamped", nst + 1)
for k, nm in enumerate(["compute", "store_tile", "barrier", "load_issue"]):
    d = st[:, 1:nst, k + 1] - st[:, 1:nst, k]
    print(f"  {nm:11s} avg {d.mean():8.0f} cyc (min {d.min()}, max {d.max()})")
print(f"  per-tile total {(st[:, 2:nst, 0] - st[:, 1:nst - 1, 0]).mean():.0f} cyc")
