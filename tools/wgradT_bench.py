"""Per-layer timing of cmu_convT2x2_wgrad on the bench shapes (low-res H=W, Cin -> Cout)."""
import ctypes
import sys

import torch

lib = ctypes.CDLL(sys.argv[1] if len(sys.argv) > 1 else "cmunet_amd/csrc/libcmunet_hip.so")
lib.cmu_convT2x2_wgrad_ws_bytes.restype = ctypes.c_int64
lib.cmu_last_error.restype = ctypes.c_char_p
vp, i64 = ctypes.c_void_p, ctypes.c_int64
dev = torch.device("cuda:0")
B = 32
tot = 0.0
for H, Cin, Cout in ((32, 1024, 512), (64, 512, 256), (128, 256, 128), (256, 128, 64)):
    x = torch.randn(B, H, H, Cin, device=dev).to(torch.bfloat16)
    do = torch.randn(B, 2 * H, 2 * H, 2 * Cout, device=dev).to(torch.bfloat16)   # left half of a concat-gradient buffer
    sc, sh = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.1
    dW, db = torch.empty(Cin, Cout, 2, 2, device=dev), torch.empty(Cout, device=dev)
    ws = torch.empty(lib.cmu_convT2x2_wgrad_ws_bytes(B, H, H, Cin, Cout, 2), dtype=torch.uint8, device=dev)

    def run():
        rc = lib.cmu_convT2x2_wgrad(vp(x.data_ptr()), i64(Cin), vp(sc.data_ptr()), vp(sh.data_ptr()), 0, vp(do.data_ptr()), i64(2 * Cout),
                                    vp(dW.data_ptr()), vp(db.data_ptr()), B, H, H, Cin, Cout, 2, vp(ws.data_ptr()), vp(0))
        assert rc == 0, lib.cmu_last_error()
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    tot += ms
    gb = (x.numel() * 2 + do.numel()) / 1e9      # X once + the used half of dOut once
    print(f"convT wgrad {Cin}->{Cout} @ {H}x{H}: {ms:.3f} ms  {2.0 * B * H * H * Cin * Cout * 4 / ms / 1e9:.0f} TFLOP/s  {gb / ms:.2f} TB/s algorithmic")
print(f"total {tot:.3f} ms")
