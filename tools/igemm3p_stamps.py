"""Diagnostic: s_memtime timeline of the persistent wide conv kernel around a tile boundary (build the stamps library with
tools/igemm_stamps.sh).  Usage: python tools/igemm3p_stamps.py [H Cin Cout].  Prints, for waves 0 and 4, the cycles between the
stamps of 12 consecutive iterations starting two slices before the first tile's last slice."""
import ctypes
import sys

import numpy as np
import torch

import os
LIB = os.environ.get("CMU_STAMPS_LIB", "tools/_diag/libcmunet_stamps.so")
B, H, Cin, Cout = 32, 256, 128, 128
if len(sys.argv) > 3:
    H, Cin, Cout = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
W = H
lib = ctypes.CDLL(LIB)
lib.cmu_pack_conv3x3_elems.restype = ctypes.c_int64
lib.cmu_last_error.restype = ctypes.c_char_p
vp, i64 = ctypes.c_void_p, ctypes.c_int64
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.randn(B, H, W, Cin, device=dev).to(torch.bfloat16)
w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05
sc, sh = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.1
wp = torch.empty(lib.cmu_pack_conv3x3_elems(Cin, Cout, 2, 0), dtype=torch.bfloat16, device=dev)
assert lib.cmu_pack_conv3x3(vp(w.data_ptr()), vp(wp.data_ptr()), Cin, Cout, 2, 0, vp(0)) == 0
y = torch.empty(B, H, W, Cout, dtype=torch.bfloat16, device=dev)
stats = torch.empty(lib.cmu_conv_ntiles(B, H, W) * 2 * Cout, device=dev)
for _ in range(5):
    rc = lib.cmu_conv3x3_fwd(vp(x.data_ptr()), i64(Cin), vp(sc.data_ptr()), vp(sh.data_ptr()), 0, vp(wp.data_ptr()), vp(y.data_ptr()),
                             i64(Cout), vp(stats.data_ptr()), B, H, W, Cin, Cout, 2, vp(0))
    assert rc == 0, lib.cmu_last_error()
torch.cuda.synchronize()
buf = np.zeros(64 * 16 * 8, dtype=np.uint64)
assert lib.cmu_debug_ig_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
clk = buf.reshape(32, 256)[:4, 200:203].astype(np.int64)
clk = clk[clk[:, 1] > 0]
if len(clk):
    print(f"shader clock over the workgroup's stream: {(clk[:, 0] / clk[:, 1]).mean() * 100:.0f} MHz "
          f"({clk[:, 0].mean() / clk[:, 2].mean():.0f} cycles per position, {clk[:, 2].mean():.0f} positions)")
NIT = 12
st = buf.reshape(32, 256)[:4, :2 * NIT * 8].reshape(4, 2, NIT, 8).astype(np.int64)
nsl = Cin // 16
names = ["mfma", "barrier1", "store_regs", "slab+epilogue", "issue", "barrier2"]
for grp in range(2):
    print(f"wave {4 * grp} (layer {Cin}->{Cout} @ {H}, {nsl} slices per tile; iteration index relative to the tile's last slice)")
    for it in range(NIT):
        rows = st[:, grp, it]
        rows = rows[rows[:, 0] > 0]
        if not len(rows):
            continue
        d = (rows[:, 1:7] - rows[:, 0:6]).mean(0)
        rel = it - 2
        print(f"  it {rel:+3d}{' (last slice)' if rel % nsl == 0 else '             '}: " + "  ".join(f"{n} {v:6.0f}" for n, v in zip(names, d))
              + f"   total {(rows[:, 6] - rows[:, 0]).mean():6.0f}"
              + (f"   [st7 - st3 {(rows[:, 7] - rows[:, 3]).mean():6.0f}]" if (rows[:, 7] > rows[:, 3]).all() else ""))
