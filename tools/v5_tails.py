"""Diagnostic: start / end time of every workgroup of one persistent conv_igemm5 launch (s_memrealtime, 100 MHz): how long do the first finishers idle?
Build: bash tools/build_variant.sh tails5 "conv_igemm.hip" "-DCMU_IG_STAMPS -DCMU_IG_TAILS"; run: python tools/v5_tails.py tools/_diag/libcmunet_tails5.so H Cin Cout [fwd|fwd_tf]"""
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
for rep in range(4):
    rc = lib.cmu_conv3x3_fwd(vp(x.data_ptr()), i64(Cin), vp(sc.data_ptr() if tf else 0), vp(sh.data_ptr() if tf else 0), 0, vp(wp.data_ptr()), vp(y.data_ptr()),
                             i64(Cout), vp(stats.data_ptr()), B, H, W, Cin, Cout, 1, vp(0))
    assert rc == 0, lib.cmu_last_error()
    torch.cuda.synchronize()
    buf = np.zeros(64 * 16 * 8, dtype=np.uint64)
    assert lib.cmu_debug_ig_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    g = 256
    st, en, nm = buf[3072:3072 + g].astype(np.int64), buf[1024:1024 + g].astype(np.int64), buf[2048:2048 + g].astype(np.int64)
    t0 = st.min()
    dur = (en - st) / 100.0          # us
    endt = (en - t0) / 100.0
    print(f"{lib.cmu_last_kernel().decode()} {Cin}->{Cout}@{H} {MODE} rep {rep}: launch {endt.max():.1f} us; workgroup durations min {dur.min():.1f} / median {np.median(dur):.1f} / max {dur.max():.1f} us; "
          f"first finisher idles {endt.max() - endt.min():.1f} us ({100 * (endt.max() - endt.min()) / endt.max():.1f} %), mean idle {np.mean(endt.max() - endt):.1f} us "
          f"({100 * np.mean(endt.max() - endt) / endt.max():.1f} % of the launch); start spread {(st.max() - t0) / 100.0:.1f} us; items per workgroup {nm.min()}-{nm.max()}")
    xcd = np.arange(g) % 8
    print("   mean end per XCD (us):", " ".join(f"{endt[xcd == k].mean():.1f}" for k in range(8)))
