"""Diagnostic: does partitioning the CUs between an MFMA-bound kernel and an HBM-bound pass hide the pass?

The MFMA kernels of the step are power-limited (csrc/probe.hip), the element-wise passes between them (BatchNorm-backward apply,
pool backward, ...: 8 of the 40 ms) are HBM-bound.  A power-limited kernel should lose little from running on fewer CUs (the clock
rises), so a pass running beside it on the CUs left over could be nearly free -- if the chip's power budget lets the two run
together.  This script measures it: the persistent 3x3 conv (cmu_conv3x3_fwd, 256 -> 256 @ 128 x 128, bs 32, f16) on a stream masked
to 256 - R CUs, cmu_bn_bwd_apply (32 x 512 x 512 x 64, f16: 3.2 GB per launch) on a stream masked to the other R CUs, alone and
together.  Streams come from hipExtStreamCreateWithCUMask (bit i of the mask = CU i in the runtime's enumeration; the reserved
CUs are taken evenly: every (256 / R)-th one).

    CMU_CONV_PERSIST_GRID=<256 - R> python tools/cu_partition.py R       (the conv kernel launches one workgroup per CU it may use)
"""
import ctypes
import glob
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cmunet_amd import _lib, ops  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 32
hip = ctypes.CDLL(glob.glob(torch.__path__[0] + "/lib/libamdhip64*")[0])
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
NCU = torch.cuda.get_device_properties(0).multi_processor_count


def masked_stream(bits):
    words = (NCU + 31) // 32
    arr = (ctypes.c_uint32 * words)()
    for i in bits:
        arr[i // 32] |= 1 << (i % 32)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(words), arr)
    assert rc == 0, f"hipExtStreamCreateWithCUMask -> {rc}"
    return torch.cuda.ExternalStream(st.value, device=dev)


step = max(1, NCU // max(R, 1))
reserved = [i for i in range(NCU) if R > 0 and i % step == step - 1][:R]
rest = [i for i in range(NCU) if i not in set(reserved)]
sM = masked_stream(rest) if R > 0 else torch.cuda.Stream()
sE = masked_stream(reserved) if R > 0 else torch.cuda.Stream()
sAll = torch.cuda.Stream()
print(f"CUs {NCU}: MFMA stream on {len(rest)}, element-wise stream on {len(reserved)}; CMU_CONV_PERSIST_GRID={os.environ.get('CMU_CONV_PERSIST_GRID')}")

g = torch.Generator(device=dev).manual_seed(0)
B, H, Cin, Cout = 32, 128, 256, 256
x = torch.randn(B, H, H, Cin, generator=g, device=dev).half()
w = torch.randn(Cout, Cin, 3, 3, generator=g, device=dev) * 0.05
sc, sh = torch.rand(Cin, generator=g, device=dev) + 0.5, torch.randn(Cin, generator=g, device=dev) * 0.1
wp = ops.pack_conv3x3(w, "f16")
xa = ops.Act(x, 0, Cin, sc, sh, 0)
y = ops.new_act(B, H, H, Cout, "f16", dev)
st = ops.new_stats(B, H, H, Cout, dev)
# element-wise pass: BatchNorm-backward apply on a 64-channel full-resolution tensor
C2, H2 = 64, 512
dA = ops.Act(torch.randn(B, H2, H2, C2, generator=g, device=dev).half())
y2 = ops.Act(torch.randn(B, H2, H2, C2, generator=g, device=dev).half(), 0, C2, torch.ones(C2, device=dev), torch.zeros(C2, device=dev), 0)
mean, invstd, coef = torch.zeros(C2, device=dev), torch.ones(C2, device=dev), torch.zeros(2, C2, device=dev)
dY = ops.new_act(B, H2, H2, C2, "f16", dev)
NCONV, NAPP = 40, 30


def conv_loop(stream):
    with torch.cuda.stream(stream):
        for _ in range(NCONV):
            ops.conv3x3_fwd(xa, wp, y, st)


def apply_loop(stream):
    with torch.cuda.stream(stream):
        for _ in range(NAPP):
            ops.bn_bwd_apply(dA, y2, mean, invstd, coef, dY)


def timed(fn):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(torch.cuda.current_stream())
    fn()
    torch.cuda.synchronize()
    e1.record(torch.cuda.current_stream())
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


def wall(fn):
    import time
    torch.cuda.synchronize()
    t = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) * 1e3


for _ in range(2):
    conv_loop(sM); apply_loop(sE)
torch.cuda.synchronize()
fl = 2.0 * B * H * H * Cin * Cout * 9
t_conv = wall(lambda: conv_loop(sM))
t_app = wall(lambda: apply_loop(sE))
t_app_all = wall(lambda: apply_loop(sAll))
t_both = wall(lambda: (conv_loop(sM), apply_loop(sE)))
print(f"conv alone on its CUs:        {t_conv:8.2f} ms for {NCONV} launches  ({fl * NCONV / t_conv / 1e9:.0f} TFLOP/s)")
print(f"apply alone on its CUs:       {t_app:8.2f} ms for {NAPP} launches  ({3 * dA.buf.numel() * 2 * NAPP / t_app / 1e9:.2f} TB/s)")
print(f"apply alone on all CUs:       {t_app_all:8.2f} ms  ({3 * dA.buf.numel() * 2 * NAPP / t_app_all / 1e9:.2f} TB/s)")
print(f"both together:                {t_both:8.2f} ms   (sum of the two alone {t_conv + t_app:.2f}; conv + apply-on-all-CUs back to back {t_conv + t_app_all:.2f})")
