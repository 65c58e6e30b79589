"""Diagnostic: s_memtime timeline of conv_igemm6 (csrc/conv_igemm6.inc) over four consecutive pairs of positions.
Build: bash tools/build_variant.sh stamps6 "conv_igemm.hip" "-DCMU_IG_STAMPS"; run: python tools/v6_stamps.py tools/_diag/libcmunet_stamps6.so H Cin Cout [fwd_tf|fwd]
Prints per recorded workgroup (0 and 256: most likely one CU; 61 and 317) and pair, in cycles of the 100 MHz s_memtime clock x 24 (~ shader cycles at 2.4 GHz):
phase A | phase B | barrier A | DMA issue | epilogue | wait for the DMA | transform | barrier B + slab."""
import ctypes
import os
import sys

import numpy as np
import torch

LIB = sys.argv[1]
H = W = int(sys.argv[2]); Cin = int(sys.argv[3]); Cout = int(sys.argv[4])
MODE = sys.argv[5] if len(sys.argv) > 5 else "fwd"
B = int(os.environ.get("CMU_SWEEP_B", 32))
dev = torch.device("cuda:0")
lib = ctypes.CDLL(LIB)
lib.cmu_pack_conv3x3_elems.restype = ctypes.c_int64
lib.cmu_last_error.restype = ctypes.c_char_p
lib.cmu_last_kernel.restype = ctypes.c_char_p
vp, i64 = ctypes.c_void_p, ctypes.c_int64
torch.manual_seed(0)
x = torch.randn(B, H, W, Cin, device=dev).half()
w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05
sc = torch.rand(Cin, device=dev) + 0.5
sh = torch.randn(Cin, device=dev) * 0.1
n = lib.cmu_pack_conv3x3_elems(Cin, Cout, 1, 0)
wp = torch.empty(n, dtype=torch.float16, device=dev)
assert lib.cmu_pack_conv3x3(vp(w.data_ptr()), vp(wp.data_ptr()), Cin, Cout, 1, 0, vp(0)) == 0
y = torch.empty(B, H, W, Cout, dtype=torch.float16, device=dev)
stats = torch.empty(lib.cmu_conv_ntiles(B, H, W) * 2 * Cout, device=dev)
tf = MODE == "fwd_tf"
lib.cmu_set_dispatch_override(b"CMU_CONV_V6", 1)
for _ in range(3):
    rc = lib.cmu_conv3x3_fwd(vp(x.data_ptr()), i64(Cin), vp(sc.data_ptr() if tf else 0), vp(sh.data_ptr() if tf else 0), 0, vp(wp.data_ptr()), vp(y.data_ptr()),
                             i64(Cout), vp(stats.data_ptr()), B, H, W, Cin, Cout, 1, vp(0))
    assert rc == 0, lib.cmu_last_error()
torch.cuda.synchronize()
print("kernel:", lib.cmu_last_kernel().decode())
buf = np.zeros(64 * 16 * 8, dtype=np.uint64)
assert lib.cmu_debug_ig_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
names = ["phaseA", "phaseB", "barA", "dma", "epi", "waitdma", "transf", "barB+slab"]
print("s_memtime ticks (100 MHz): x24 ~ cycles at 2.4 GHz")
for slot, blk in enumerate((0, 256, 61, 317)):
    t = buf[slot * 32:(slot + 1) * 32].astype(np.int64).reshape(4, 8)
    print(f"workgroup {blk}: pair start offsets (ticks from the first): {[int(v) for v in t[:, 0] - t[0, 0]]}")
    for pi in range(4):
        d = [int(t[pi, k + 1] - t[pi, k]) for k in range(7)]
        last = int(t[pi + 1, 0] - t[pi, 7]) if pi < 3 else -1
        print(f"   pair {8 + pi}: " + "  ".join(f"{nm} {v:5d}" for nm, v in zip(names, d + [last])))
