"""Round 5: the 16x16x32 persistent conv kernel (csrc/conv_igemm5.inc) against the 32x32x16 kernels on the same tensors, in one process
(cmu_set_dispatch_override("CMU_CONV_V5", 0 | 1)): outputs / statistics / BatchNorm-backward sums (relative L2, max abs) and ms per launch.
    python tools/v5_check.py [lib.so] [quick]
"""
import ctypes
import os
import sys

import torch

LIB = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".so") else "cmunet_amd/csrc/libcmunet_hip.so"
QUICK = "quick" in sys.argv
dev = torch.device("cuda:0")
lib = ctypes.CDLL(LIB)
lib.cmu_pack_conv3x3_elems.restype = ctypes.c_int64
lib.cmu_last_error.restype = ctypes.c_char_p
lib.cmu_last_kernel.restype = ctypes.c_char_p
vp, i64 = ctypes.c_void_p, ctypes.c_int64
DT = int(os.environ.get("CMU_SWEEP_DT", "1"))
TDT = {1: torch.float16, 2: torch.bfloat16}[DT]
B = int(os.environ.get("CMU_SWEEP_B", 32))
# round 6: V_SWITCH=CMU_CONV_V6 compares conv_igemm6 (64-channel items) with what the shape ran on before; V5_MODES="fwd_tf,fwd" picks the forms
SWITCH = os.environ.get("V_SWITCH", "CMU_CONV_V5").encode()
MODES = tuple(os.environ.get("V5_MODES", "fwd_tf,fwd,dgrad_bn").split(","))


def override(v):
    rc = lib.cmu_set_dispatch_override(SWITCH, v)
    assert rc == 0, lib.cmu_last_error()


def pack(w, flip):
    Cout, Cin = w.shape[0], w.shape[1]
    n = lib.cmu_pack_conv3x3_elems(Cin, Cout, DT, flip)
    wp = torch.empty(n, dtype=TDT, device=dev)
    rc = lib.cmu_pack_conv3x3(vp(w.data_ptr()), vp(wp.data_ptr()), Cin, Cout, DT, flip, vp(0))
    assert rc == 0, lib.cmu_last_error()
    return wp


def timeit(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item(), (a - b).abs().max().item()


def case(H, Cin, Cout, mode):
    torch.manual_seed(1)
    W = H
    x = torch.randn(B, H, W, Cin, device=dev).to(TDT)
    w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05
    ntiles = lib.cmu_conv_ntiles(B, H, W)
    res = {}
    sc = torch.rand(Cin, device=dev) + 0.5
    sh = torch.randn(Cin, device=dev) * 0.1
    dy = torch.randn(B, H, W, Cout, device=dev).to(TDT)
    yraw = torch.randn(B, H, W, Cin, device=dev).to(TDT)
    scale = torch.rand(Cin, device=dev) + 0.5
    shift = torch.randn(Cin, device=dev) * 0.1
    mean = torch.randn(Cin, device=dev) * 0.1
    invstd = torch.rand(Cin, device=dev) + 0.5
    # Fair protocol (round 5, after the first form of this tool was found to favour whichever kernel ran SECOND by up to 4 %: fresh output
    # buffers per variant): ONE set of output buffers, the two kernels alternate (old, v5, old, v5, old, v5), the reported time of each is the
    # median of its three measurements.
    tf = mode == "fwd_tf"
    if mode in ("fwd_tf", "fwd"):
        wp = pack(w, 0)
        y = torch.full((B, H, W, Cout), float("nan"), dtype=TDT, device=dev)
        stats = torch.full((ntiles * 2 * Cout,), float("nan"), device=dev)

        def run():
            rc = lib.cmu_conv3x3_fwd(vp(x.data_ptr()), i64(Cin), vp(sc.data_ptr() if tf else 0), vp(sh.data_ptr() if tf else 0), 0, vp(wp.data_ptr()),
                                     vp(y.data_ptr()), i64(Cout), vp(stats.data_ptr()), B, H, W, Cin, Cout, DT, vp(0))
            assert rc == 0, lib.cmu_last_error()
        outs = (y, stats)
    else:   # dgrad_bn: dY (B,H,W,K=Cout of the layer) -> dX (N = Cin of the layer) + BN-backward sums of the producer
        K, N = Cout, Cin
        wp = pack(w, 1)   # w: (Cout, Cin, 3, 3) -> flipped pack contracts over Cout
        dx = torch.full((B, H, W, N), float("nan"), dtype=TDT, device=dev)
        bst = torch.full((ntiles * 2 * N,), float("nan"), device=dev)

        def run():
            rc = lib.cmu_conv3x3_dgrad_bn(vp(dy.data_ptr()), i64(K), vp(wp.data_ptr()), vp(dx.data_ptr()), i64(N), vp(yraw.data_ptr()), i64(N),
                                          vp(scale.data_ptr()), vp(shift.data_ptr()), vp(mean.data_ptr()), vp(invstd.data_ptr()), vp(bst.data_ptr()),
                                          B, H, W, K, N, DT, vp(0))
            assert rc == 0, lib.cmu_last_error()
        outs = (dx, bst)
    times = {0: [], 1: []}
    for rnd in range(3):
        for v in (0, 1):
            override(v)
            run()
            kern = lib.cmu_last_kernel().decode()
            times[v].append(timeit(run))
            if rnd == 0:
                res[v] = [outs[0].clone(), outs[1].clone(), 0.0, kern]
    for v in (0, 1):
        res[v][2] = sorted(times[v])[1]
    override(-1)
    y0, s0, ms0, k0 = res[0]
    y1, s1, ms1, k1 = res[1]
    ry, rs = rel(y1, y0), rel(s1, s0)
    Kc, Nc = (Cin, Cout) if mode != "dgrad_bn" else (Cout, Cin)
    fl = 2.0 * B * H * W * Kc * Nc * 9
    print(f"{mode:8s} {Cin:4d}->{Cout:4d} @{H:3d} B={B}: old {ms0:.3f} ms {fl / ms0 / 1e9:5.0f} TF [{k0}] | new {ms1:.3f} ms {fl / ms1 / 1e9:5.0f} TF [{k1}]"
          f" | y relL2 {ry[0]:.2e} max {ry[1]:.2e} nan {int(torch.isnan(y1).sum())} | stats relL2 {rs[0]:.2e} nan {int(torch.isnan(s1).sum())}", flush=True)


shapes = [(256, 128, 128), (128, 256, 256), (64, 512, 512), (32, 1024, 1024), (256, 64, 128), (256, 256, 128), (128, 512, 256), (64, 1024, 512), (32, 512, 1024)]
if QUICK:
    shapes = [(64, 128, 128), (32, 64, 128)]
if os.environ.get("V5_SHAPES"):
    shapes = [tuple(int(v) for v in t.split(",")) for t in os.environ["V5_SHAPES"].split(";")]
for (H, ci, co) in shapes:
    for mode in MODES:
        if mode == "dgrad_bn" and ci % 128 != 0:
            continue
        case(H, ci, co, mode)
