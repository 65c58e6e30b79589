"""Per-layer cost of the fused BN-backward statistics epilogue: cmu_conv3x3_fwd (flipped pack) vs cmu_conv3x3_dgrad_bn."""
import ctypes
import sys

import torch

lib = ctypes.CDLL(sys.argv[1] if len(sys.argv) > 1 else "cmunet_amd/csrc/libcmunet_hip.so")
lib.cmu_pack_conv3x3_elems.restype = ctypes.c_int64
lib.cmu_last_error.restype = ctypes.c_char_p
vp, i64 = ctypes.c_void_p, ctypes.c_int64
dev = torch.device("cuda:0")
B = 32
for H, C in ((512, 64), (256, 128), (128, 256), (64, 512), (32, 1024)):
    dy = torch.randn(B, H, H, C, device=dev).to(torch.bfloat16)
    y = torch.randn(B, H, H, C, device=dev).to(torch.bfloat16)
    w = torch.randn(C, C, 3, 3, device=dev) * 0.05
    wp = torch.empty(lib.cmu_pack_conv3x3_elems(C, C, 2, 1), dtype=torch.bfloat16, device=dev)
    assert lib.cmu_pack_conv3x3(vp(w.data_ptr()), vp(wp.data_ptr()), C, C, 2, 1, vp(0)) == 0
    dx = torch.empty(B, H, H, C, dtype=torch.bfloat16, device=dev)
    sc, sh, mu, iv = (torch.rand(C, device=dev) + 0.5 for _ in range(4))
    slab = torch.empty(lib.cmu_conv_ntiles(B, H, H) * 2 * C, device=dev)
    filler = torch.empty(512 << 20, dtype=torch.uint8, device=dev)   # evicts L2 / MALL between calls

    def plain():
        assert lib.cmu_conv3x3_fwd(vp(dy.data_ptr()), i64(C), vp(0), vp(0), 0, vp(wp.data_ptr()), vp(dx.data_ptr()), i64(C), vp(0), B, H, H,
                                   C, C, 2, vp(0)) == 0, lib.cmu_last_error()

    def fused():
        assert lib.cmu_conv3x3_dgrad_bn(vp(dy.data_ptr()), i64(C), vp(wp.data_ptr()), vp(dx.data_ptr()), i64(C), vp(y.data_ptr()), i64(C),
                                        vp(sc.data_ptr()), vp(sh.data_ptr()), vp(mu.data_ptr()), vp(iv.data_ptr()), vp(slab.data_ptr()), B,
                                        H, H, C, C, 2, vp(0)) == 0, lib.cmu_last_error()
    res = []
    for fn in (plain, fused):
        ts = []
        for _ in range(6):
            filler.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        res.append(sorted(ts)[len(ts) // 2])
    print(f"dgrad {C}->{C} @ {H}x{H}: plain {res[0]:.3f} ms, with BN-backward statistics {res[1]:.3f} ms (+{res[1] - res[0]:.3f})")
