"""Diagnostic: in-kernel s_memtime timeline of the 3x3 implicit-GEMM kernel (build with -DCMU_IG_STAMPS, see
tools/igemm_stamps.sh).  Prints the average cycles wave 0 spends in each phase of a K stage."""
import ctypes
import sys

import numpy as np
import torch

LIB = sys.argv[1] if len(sys.argv) > 1 else "tools/_diag/libcmunet_stamps.so"
B, H, W, Cin, Cout = 32, 128, 128, 256, 256
if len(sys.argv) > 2:
    H = W = int(sys.argv[2]); Cin = int(sys.argv[3]); Cout = int(sys.argv[4])
NPH = int(sys.argv[5]) if len(sys.argv) > 5 else 6
NST = int(sys.argv[6]) if len(sys.argv) > 6 else 16

dev = torch.device("cuda:0")
lib = ctypes.CDLL(LIB)
lib.cmu_pack_conv3x3_elems.restype = ctypes.c_int64
lib.cmu_last_error.restype = ctypes.c_char_p
vp = ctypes.c_void_p
i64 = ctypes.c_int64

import os
# CMU_SWEEP_DT=1: f16 (default 2: bf16).  CMU_SWEEP_DATA=zero_w | zero_x | zero_both: the same launch on operands that do not
# toggle the multipliers -- the shader clock under an MFMA load depends on the data (csrc/probe.hip), so the time these take
# against the normal run separates the power limit from the kernel's own stalls
DT = int(os.environ.get("CMU_SWEEP_DT", "2"))           # 0: f32, 1: f16, 2: bf16
TDT = {0: torch.float32, 1: torch.float16, 2: torch.bfloat16}[DT]
B = int(os.environ.get("CMU_SWEEP_B", B))               # batch (small-batch finetuning shapes: tools/small_sweep.sh)
DATA = os.environ.get("CMU_SWEEP_DATA", "normal")
torch.manual_seed(0)
x = torch.randn(B, H, W, Cin, device=dev).to(TDT)
w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05
sc = torch.rand(Cin, device=dev) + 0.5
sh = torch.randn(Cin, device=dev) * 0.1
if DATA in ("zero_x", "zero_both"):
    x.zero_(); sc.fill_(1.0); sh.zero_()
if DATA in ("zero_w", "zero_both"):
    w.zero_()
n = lib.cmu_pack_conv3x3_elems(Cin, Cout, DT, 0)
wp = torch.empty(n, dtype=TDT, device=dev)
rc = lib.cmu_pack_conv3x3(vp(w.data_ptr()), vp(wp.data_ptr()), Cin, Cout, DT, 0, vp(0))
assert rc == 0, lib.cmu_last_error()
y = torch.empty(B, H, W, Cout, dtype=TDT, device=dev)
ntiles = lib.cmu_conv_ntiles(B, H, W)
stats = torch.empty(ntiles * 2 * Cout, device=dev)


def run():
    rc = lib.cmu_conv3x3_fwd(vp(x.data_ptr()), i64(Cin), vp(sc.data_ptr()), vp(sh.data_ptr()), 0, vp(wp.data_ptr()), vp(y.data_ptr()),
                             i64(Cout), vp(stats.data_ptr()), B, H, W, Cin, Cout, DT, vp(0))
    assert rc == 0, lib.cmu_last_error()


for _ in range(20):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
fl = 2.0 * B * H * W * Cin * Cout * 9
print(f"layer {Cin}->{Cout} @ {H}x{W} B={B}{'' if DATA == 'normal' else ' [' + DATA + ']'}{(' f32', ' f16', '')[DT]}: {ms:.3f} ms  {fl / ms / 1e9:.0f} TFLOP/s")

if not hasattr(lib, "cmu_debug_ig_stamps"):
    sys.exit(0)
buf = np.zeros(64 * NST * 8, dtype=np.uint64)
rc = lib.cmu_debug_ig_stamps(buf.ctypes.data_as(ctypes.c_void_p))
assert rc == 0
GROUPS = int(sys.argv[9]) if len(sys.argv) > 9 else 1    # wide kernel: 2 (waves 0 and 4 stamp), 32 blocks
if GROUPS == 2:
    st2 = buf.reshape(32, 2, NST, 8).astype(np.int64)
    for grp in range(2):
        stg = st2[:, grp]
        stg = stg[stg[:, 0, 0] > 0][:, :min(15, int(sys.argv[8]))]
        print(f"wave {4 * grp}: blocks {stg.shape[0]}")
        nm = sys.argv[7].split(",")
        for k in range(NPH - 1):
            d = stg[:, 1:, k + 1] - stg[:, 1:, k]
            print(f"  {nm[k]:12s} avg {d.mean():8.0f} cyc  (min {d.min()}, max {d.max()})")
        print(f"  per-step total avg {(stg[:, 2:, 0] - stg[:, 1:-1, 0]).mean():.0f} cyc")
    ck = st2[:, 0, 15, :4]
    ck = ck[ck[:, 0] > 0]
    if len(ck):
        mhz = (ck[:, 2] - ck[:, 0]) / np.maximum(ck[:, 3] - ck[:, 1], 1) * 100.0
        print(f"shader clock inside the main loop: median {np.median(mhz):.0f} MHz (min {mhz.min():.0f}, max {mhz.max():.0f}); "
              f"dense bf16 peak at that clock: {2500.0 * np.median(mhz) / 2400.0:.0f} TFLOP/s")
    sys.exit(0)
st = buf.reshape(64, NST, 8).astype(np.int64)
names = sys.argv[7].split(",") if len(sys.argv) > 7 else ["barrier1", "store_stage", "barrier2", "load_issue", "mfma_phase", "loop"]
ns = min(NST, int(sys.argv[8]) if len(sys.argv) > 8 else (Cin * 2 + 63) // 64)
valid = st[:, 0, 0] > 0
st = st[valid][:, :ns]
print("blocks sampled", st.shape[0], "stages", ns)
for k in range(NPH - 1):
    d = st[:, 1:, k + 1] - st[:, 1:, k]
    print(f"  {names[k]:12s} avg {d.mean():8.0f} cyc  (min {d.min()}, max {d.max()})")
per_stage = st[:, 2:, 0] - st[:, 1:-1, 0]
print(f"  per-stage total avg {per_stage.mean():.0f} cyc")
