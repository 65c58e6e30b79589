"""Per-level timing of cmu_convT2x2_fwd and cmu_convT2x2_dgrad on the bench shapes (low-res H=W, Cin -> Cout)."""
import ctypes
import sys

import torch

lib = ctypes.CDLL(sys.argv[1] if len(sys.argv) > 1 else "cmunet_amd/csrc/libcmunet_hip.so")
lib.cmu_pack_convT2x2_elems.restype = ctypes.c_int64
lib.cmu_last_error.restype = ctypes.c_char_p
lib.cmu_last_kernel.restype = ctypes.c_char_p
vp, i64 = ctypes.c_void_p, ctypes.c_int64
dev = torch.device("cuda:0")
B = 32
tot = [0.0, 0.0]
for H, Cin, Cout in ((32, 1024, 512), (64, 512, 256), (128, 256, 128), (256, 128, 64)):
    x = torch.randn(B, H, H, Cin, device=dev).to(torch.bfloat16)
    out = torch.empty(B, 2 * H, 2 * H, 2 * Cout, dtype=torch.bfloat16, device=dev)     # left half of a concat buffer
    dx = torch.empty(B, H, H, Cin, dtype=torch.bfloat16, device=dev)
    w = torch.randn(Cin, Cout, 2, 2, device=dev) * 0.05
    bias = torch.randn(Cout, device=dev)
    sc, sh = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.1
    wp = [torch.empty(lib.cmu_pack_convT2x2_elems(Cin, Cout, 2, m), dtype=torch.bfloat16, device=dev) for m in (0, 1)]
    for m in (0, 1):
        assert lib.cmu_pack_convT2x2(vp(w.data_ptr()), vp(wp[m].data_ptr()), Cin, Cout, 2, m, vp(0)) == 0

    def fwd():
        assert lib.cmu_convT2x2_fwd(vp(x.data_ptr()), i64(Cin), vp(sc.data_ptr()), vp(sh.data_ptr()), 0, vp(wp[0].data_ptr()), vp(bias.data_ptr()),
                                    vp(out.data_ptr()), i64(2 * Cout), B, H, H, Cin, Cout, 2, vp(0)) == 0, lib.cmu_last_error()

    def dgrad():
        assert lib.cmu_convT2x2_dgrad(vp(out.data_ptr()), i64(2 * Cout), vp(wp[1].data_ptr()), vp(dx.data_ptr()), i64(Cin), B, H, H, Cin, Cout, 2,
                                      vp(0)) == 0, lib.cmu_last_error()
    res = []
    for fn in (fwd, dgrad):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res.append((e0.elapsed_time(e1) / 20, lib.cmu_last_kernel().decode()))
    fl = 2.0 * B * H * H * Cin * Cout * 4
    tot[0] += res[0][0]; tot[1] += res[1][0]
    print(f"convT {Cin}->{Cout} @ {H}x{H}: fwd {res[0][0]:.3f} ms {fl / res[0][0] / 1e9:.0f} TFLOP/s [{res[0][1]}]   dgrad {res[1][0]:.3f} ms "
          f"{fl / res[1][0] / 1e9:.0f} TFLOP/s [{res[1][1]}]")
print(f"total fwd {tot[0]:.3f} ms, dgrad {tot[1]:.3f} ms")
