"""Per-level time of the ConvTranspose2d 2x2 entries on the bench shapes (bs 32, UNet 64x5 at 512^2), against the
HBM floor (algorithmic bytes / 6.3 TB/s) and the MFMA rate: python tools/convt_sweep.py [f16|bf16]."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from cmunet_amd import ops  # noqa: E402

dt = sys.argv[1] if len(sys.argv) > 1 else "f16"
tdt = ops.TORCH_DT[ops.dt_code(dt)]
B = 32


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for H, Cin in ((256, 128), (128, 256), (64, 512), (32, 1024)):
    Cout = Cin // 2
    x = ops.Act(torch.randn(B, H, H, Cin, device="cuda").to(tdt))
    sc, sh = torch.rand(Cin, device="cuda") + 0.5, torch.randn(Cin, device="cuda") * 0.1
    xt = x.with_transform(sc, sh, 0)
    w = torch.randn(Cin, Cout, 2, 2, device="cuda") * 0.05
    bias = torch.randn(Cout, device="cuda")
    wp, wpd = ops.pack_convT2x2(w, dt, 0), ops.pack_convT2x2(w, dt, 1)
    cat = torch.empty(B, 2 * H, 2 * H, 2 * Cout, dtype=tdt, device="cuda")          # the decoder's concat buffer: left half
    out = ops.Act(cat, 0, Cout)
    dcat = torch.randn(B, 2 * H, 2 * H, 2 * Cout, device="cuda").to(tdt)
    dout = ops.Act(dcat, 0, Cout)
    dx = ops.new_act(B, H, H, Cin, dt, "cuda")
    mean, invstd = torch.zeros(Cin, device="cuda"), torch.ones(Cin, device="cuda")
    bst = ops.new_stats(B, H, H, Cin, "cuda")
    dW, db = torch.empty_like(w), torch.empty_like(bias)
    from cmunet_amd import _lib
    ws = torch.empty(_lib.lib().cmu_convT2x2_wgrad_ws_bytes(B, H, H, Cin, Cout, ops.dt_code(dt)), dtype=torch.uint8, device="cuda")
    es = 2 if dt != "f32" else 4
    vol_in, vol_out = B * H * H * Cin * es, B * 4 * H * H * Cout * es
    gf = 2.0 * 4 * Cin * Cout * B * H * H / 1e9
    for name, fn, nbytes in (("fwd", lambda: ops.convT2x2_fwd(xt, wp, bias, out), vol_in + vol_out),
                             ("dgrad_bn", lambda: ops.convT2x2_dgrad_bn(dout, wpd, dx, xt, mean, invstd, bst), vol_out + 2 * vol_in),
                             ("wgrad", lambda: ops.convT2x2_wgrad(xt, dout, dW, db, ws), vol_in + vol_out)):
        ms = timed(fn)
        print(f"convT {Cin:4d}->{Cout:3d} @{H:3d}  {name:8s} {ms:6.3f} ms   {gf / ms:7.1f} TFLOP/s   {nbytes / ms / 1e9:5.2f} TB/s algorithmic "
              f"(HBM floor {nbytes / 6.3e9:.3f} ms)")
