"""Diagnostic: s_memtime timeline of the 16x16x32 persistent conv kernel (csrc/conv_igemm5.inc) around an item boundary.
Build: bash tools/build_variant.sh stamps5 "conv_igemm.hip" "-DCMU_IG_STAMPS"; run: python tools/v5_stamps.py tools/_diag/libcmunet_stamps5.so H Cin Cout [mode]
mode: fwd_tf (default) | fwd | dgrad_bn.  Prints, for waves 0 (first half) and 4 (second half), cycles per phase of 16 consecutive positions:
tap0, tap1 = the first two taps of the MFMA phase (tap 1 waits for everything issued before its weights: the halo chunks, the epilogue's stores),
taps2-8, barrier (second half only), stage (registers -> LDS, slab, next halo loads), epilogue, tail (first half: its barrier)."""
import ctypes
import os
import sys

import numpy as np
import torch

LIB = sys.argv[1]
H = W = int(sys.argv[2]); Cin = int(sys.argv[3]); Cout = int(sys.argv[4])
MODE = sys.argv[5] if len(sys.argv) > 5 else "fwd_tf"
B = int(os.environ.get("CMU_SWEEP_B", 32))
DT = 1
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
ntiles = lib.cmu_conv_ntiles(B, H, W)


def pack(flip):
    n = lib.cmu_pack_conv3x3_elems(Cin, Cout, DT, flip)
    wp = torch.empty(n, dtype=torch.float16, device=dev)
    assert lib.cmu_pack_conv3x3(vp(w.data_ptr()), vp(wp.data_ptr()), Cin, Cout, DT, flip, vp(0)) == 0
    return wp


if MODE in ("fwd", "fwd_tf"):
    wp = pack(0)
    y = torch.empty(B, H, W, Cout, dtype=torch.float16, device=dev)
    stats = torch.empty(ntiles * 2 * Cout, device=dev)
    tf = MODE == "fwd_tf"

    def run():
        rc = lib.cmu_conv3x3_fwd(vp(x.data_ptr()), i64(Cin), vp(sc.data_ptr() if tf else 0), vp(sh.data_ptr() if tf else 0), 0, vp(wp.data_ptr()), vp(y.data_ptr()),
                                 i64(Cout), vp(stats.data_ptr()), B, H, W, Cin, Cout, DT, vp(0))
        assert rc == 0, lib.cmu_last_error()
    K = Cin
else:
    wp = pack(1)
    dy = torch.randn(B, H, W, Cout, device=dev).half()
    yraw = torch.randn(B, H, W, Cin, device=dev).half()
    mean = torch.randn(Cin, device=dev) * 0.1
    invstd = torch.rand(Cin, device=dev) + 0.5
    dx = torch.empty(B, H, W, Cin, dtype=torch.float16, device=dev)
    bst = torch.empty(ntiles * 2 * Cin, device=dev)

    def run():
        rc = lib.cmu_conv3x3_dgrad_bn(vp(dy.data_ptr()), i64(Cout), vp(wp.data_ptr()), vp(dx.data_ptr()), i64(Cin), vp(yraw.data_ptr()), i64(Cin), vp(sc.data_ptr()),
                                      vp(sh.data_ptr()), vp(mean.data_ptr()), vp(invstd.data_ptr()), vp(bst.data_ptr()), B, H, W, Cout, Cin, DT, vp(0))
        assert rc == 0, lib.cmu_last_error()
    K = Cout
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
print(f"{MODE} {Cin}->{Cout} @ {H}: {ms:.3f} ms  {2.0 * B * H * W * Cin * Cout * 9 / ms / 1e9:.0f} TFLOP/s  [{lib.cmu_last_kernel().decode()}]")
buf = np.zeros(64 * 16 * 8, dtype=np.uint64)
assert lib.cmu_debug_ig_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
nsl = K // 32
NIT = 16
for slot in range(2):
    st = buf[slot * 256:slot * 256 + 256].astype(np.int64).reshape(2, NIT, 8)
    for grp in range(2):
        print(f"workgroup slot {slot}, wave {4 * grp} ({'second' if grp else 'first'} half); position index relative to an item's last (nsl = {nsl})")
        s = st[grp]
        for it in range(NIT):
            t = s[it]
            if t[0] == 0:
                continue
            rel = (2 * nsl - 3 + it) % nsl - (nsl - 1)
            nxt = s[it + 1][0] if it + 1 < NIT and s[it + 1][0] else t[7]
            if os.environ.get("V5_STAMPS_EPI"):      # build with -DCMU_IG_STAMPS -DCMU_IG_STAMPS_EPI: slots 1, 2 are taken INSIDE the epilogue
                if rel == 0:
                    print(f"  pos {rel:+3d} (last): mfma {t[3] - t[0]:6d}  barrier {t[4] - t[3]:6d}  stage {t[5] - t[4]:6d}  epilogue: part A (statistics / constants) {t[1] - t[5]:6d}"
                          f"  part B (conversion + stores / rows) {t[2] - t[1]:6d}  part C (rest: folds, clearing) {t[6] - t[2]:6d}  tail {t[7] - t[6]:6d}")
                continue
            print(f"  pos {rel:+3d}{' (last)' if rel == 0 else '       '}: tap0 {t[1] - t[0]:6d}  tap1 {t[2] - t[1]:6d}  taps2-8 {t[3] - t[2]:6d}  barrier {t[4] - t[3]:6d}  stage {t[5] - t[4]:6d}"
                  f"  epilogue {t[6] - t[5]:6d}  tail {t[7] - t[6]:6d}  | position total {nxt - t[0]:6d}")
