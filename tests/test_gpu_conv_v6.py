"""Round 6: the 64-channel-item form of the 16x16x32 3x3 conv kernel (csrc/conv_igemm6.inc: four-wave workgroups, two per CU, halo by LDS-DMA)
-- the reference's ``nn.Conv2d(Cin, Cout, 3, padding=1)`` of ``DoubleConv`` (Finetuning/model.py:17-22) at the K = 64 / 64-output-channel shapes of
``down_conv1`` / ``up_conv1`` (model.py:96,107) -- against float64 arithmetic on the same 16-bit operands, against the 32x32x16 family
(``CMU_CONV_V6=0``) on the same tensors and, where both serve a shape, BIT FOR BIT against conv_igemm5: forward with a pending BatchNorm+ReLU
transform + batch statistics, plain forward / data gradient, channel slices of wider buffers (the concat-free decoder), ``relu_from`` of a
concat input, image borders (the DMA's out-of-range zeros), item lists of every length (``CMU_CONV_PERSIST_GRID``)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from cmunet_amd import ops as O
    return O


def rel_l2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def conv_ref64(xa, w):
    return torch.nn.functional.conv2d(xa.double().permute(0, 3, 1, 2), w.double(), padding=1).permute(0, 2, 3, 1)


def last_kernel():
    import ctypes
    from cmunet_amd import _lib
    lib = _lib.lib()
    lib.cmu_last_kernel.restype = ctypes.c_char_p
    return lib.cmu_last_kernel().decode()


# (B, H, W, Cin, Cout, extra input stride, extra output stride, relu_from): K in {64, 128}, N % 64 == 0, H % 16 == 0, W % 32 == 0
FWD_CASES = [
    (2, 32, 64, 64, 64, 0, 0, 0),           # down_conv1.conv2 / up_conv1.conv2 (model.py:20)
    (3, 16, 32, 128, 64, 0, 0, 64),         # up_conv1.conv1: concat input, channels [0, 64) carry no ReLU
    (2, 32, 32, 64, 128, 64, 128, 0),       # input = right half of a 128-channel buffer, output = left half of a 256-channel one
    (1, 48, 96, 128, 64, 0, 64, 32),        # four slices = two pairs per item, relu_from inside the view, output slice of a wider buffer
    (5, 16, 32, 64, 256, 0, 0, 0),          # four channel blocks per tile
    (1, 16, 32, 128, 128, 0, 0, 0),         # one tile: every halo pixel outside the image on two sides
]


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("case", FWD_CASES)
def test_v6_forward_transform_statistics_vs_float64(ops, dt, case):
    B, H, W, Cin, Cout, xs, ys, relu_from = case
    g = torch.Generator().manual_seed(29)
    tdt = ops.TORCH_DT[ops.dt_code(dt)]
    xbuf = torch.randn(B, H, W, Cin + xs, generator=g).to(tdt).cuda()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3.0 * Cin ** 0.5)).cuda()
    sc, sh = (torch.rand(Cin, generator=g) + 0.5).cuda(), (torch.randn(Cin, generator=g) * 0.3).cuda()
    x = ops.Act(xbuf, xs, Cin, sc, sh, relu_from)
    wq = w.to(tdt)
    z = xbuf[..., xs:].float() * sc + sh
    z = torch.cat([z[..., :relu_from], torch.relu(z[..., relu_from:])], -1)
    xa = z.to(tdt)                                             # what the kernel's staging rounds to
    ref = conv_ref64(xa, wq)
    outs = {}
    for v6 in (1, 0):
        with ops.dispatch_override("CMU_CONV_V6", v6), ops.dispatch_override("CMU_CONV_NARROW", 0):
            ybuf = torch.full((B, H, W, Cout + ys), float("nan"), dtype=tdt, device="cuda")
            y = ops.Act(ybuf, 0, Cout)
            st = torch.full_like(ops.new_stats(B, H, W, Cout, "cuda"), float("nan"))
            ops.conv3x3_fwd(x, ops.pack_conv3x3(w, dt), y, st)
            torch.cuda.synchronize()
            assert (last_kernel() == "conv_igemm6_kernel") == bool(v6)
            y2buf = torch.full((B, H, W, Cout), float("nan"), dtype=tdt, device="cuda")
            ops.conv3x3_fwd(ops.Act(xa.contiguous(), 0, Cin), ops.pack_conv3x3(w, dt), ops.Act(y2buf, 0, Cout), None)
            torch.cuda.synchronize()
            outs[v6] = (ybuf.clone(), st.clone(), y2buf.clone())
    ybuf, st, y2 = outs[1]
    if ys:
        assert bool(torch.isnan(ybuf[..., Cout:]).all()), "the kernel wrote outside its channel slice"
    tol = {"f16": 6e-4, "bf16": 5e-3}[dt]                      # one rounding of the stored type on fp32 sums
    for name, got in (("transform", ybuf[..., :Cout]), ("plain", y2)):
        assert bool(torch.isfinite(got).all()), name
        e = rel_l2(got, ref)
        assert e <= tol, (name, e)
    assert rel_l2(ybuf[..., :Cout], outs[0][0][..., :Cout]) <= 1.5 * tol
    t = ref.reshape(B, H // 16, 16, W // 16, 16, Cout)
    s1, s2 = t.sum((2, 4)).reshape(-1, Cout), (t * t).sum((2, 4)).reshape(-1, Cout)
    assert bool(torch.isfinite(st).all())
    assert rel_l2(st[:, 0], s1) <= 2e-5 + (0 if dt == "f16" else 1e-4) and rel_l2(st[:, 1], s2) <= 1e-4
    assert rel_l2(st, outs[0][1]) <= 1e-5


@pytest.mark.parametrize("dt", ["f16", "bf16"])
def test_v6_is_bit_identical_to_the_128_channel_kernel(ops, dt):
    """K = 128, N = 256: conv_igemm5 (128-channel items, 512 threads) and conv_igemm6 (64-channel items, 256 threads, LDS-DMA halo) run the same MFMA
    sequence per output and fold the statistics in the same order: same output bits, same slab bits."""
    B, H, W, Cin, Cout = 3, 32, 64, 128, 256
    g = torch.Generator().manual_seed(31)
    tdt = ops.TORCH_DT[ops.dt_code(dt)]
    x = torch.randn(B, H, W, Cin, generator=g).to(tdt).cuda()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3.0 * Cin ** 0.5)).cuda()
    sc, sh = (torch.rand(Cin, generator=g) + 0.5).cuda(), (torch.randn(Cin, generator=g) * 0.3).cuda()
    res = {}
    for v6 in (1, 0):
        with ops.dispatch_override("CMU_CONV_V6", v6), ops.dispatch_override("CMU_CONV_V5", 1), ops.dispatch_override("CMU_CONV_NARROW", 0):
            y = ops.new_act(B, H, W, Cout, dt, "cuda")
            st = ops.new_stats(B, H, W, Cout, "cuda")
            ops.conv3x3_fwd(ops.Act(x, 0, Cin, sc, sh, 0), ops.pack_conv3x3(w, dt), y, st)
            torch.cuda.synchronize()
            assert last_kernel() == ("conv_igemm6_kernel" if v6 else "conv_igemm5_kernel")
            res[v6] = (y.buf.clone(), st.clone())
    assert torch.equal(res[1][0].view(torch.int16), res[0][0].view(torch.int16))
    assert torch.equal(res[1][1], res[0][1])


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("case", [(2, 32, 64, 64, 128), (3, 16, 64, 128, 64), (2, 32, 32, 64, 64)])
def test_v6_plain_data_gradient_vs_float64(ops, dt, case):
    """The data gradient without BatchNorm-backward sums (up_conv1.conv1's: K = Cout = 64, N = Cin = 128; down_conv2.conv1's: K = 128, N = 64) =
    a plain launch on the flipped, transposed pack."""
    B, H, W, K, N = case
    g = torch.Generator().manual_seed(37)
    tdt = ops.TORCH_DT[ops.dt_code(dt)]
    dy = torch.randn(B, H, W, K, generator=g).to(tdt).cuda()
    wt = (torch.randn(K, N, 3, 3, generator=g) / (3.0 * K ** 0.5)).cuda()          # the layer's weight (Cout = K, Cin = N)
    ref = torch.nn.functional.conv_transpose2d(dy.double().permute(0, 3, 1, 2), wt.to(tdt).double(), padding=1).permute(0, 2, 3, 1)
    outs = {}
    for v6 in (1, 0):
        with ops.dispatch_override("CMU_CONV_V6", v6), ops.dispatch_override("CMU_CONV_NARROW", 0):
            dx = torch.full((B, H, W, N), float("nan"), dtype=tdt, device="cuda")
            ops.conv3x3_fwd(ops.Act(dy, 0, K), ops.pack_conv3x3(wt, dt, transpose_flip=True), ops.Act(dx, 0, N), None)
            torch.cuda.synchronize()
            assert (last_kernel() == "conv_igemm6_kernel") == bool(v6)
            outs[v6] = dx.clone()
    tol = {"f16": 6e-4, "bf16": 5e-3}[dt]
    assert bool(torch.isfinite(outs[1]).all()) and rel_l2(outs[1], ref) <= tol
    assert rel_l2(outs[1], outs[0]) <= 1.5 * tol


def test_v6_dispatch(ops):
    """The kernel is opt-in (measured no faster than the kernels its shapes run on: profiles/r06_conv_v6.txt): unforced, no shape takes it; forced, the
    shapes it can serve -- 16-bit, whole tiles, N % 64 == 0, K in {64, 128} -- do."""
    def kern(B, H, W, Cin, Cout, dt, v6=-1):
        tdt = ops.TORCH_DT[ops.dt_code(dt)]
        x = torch.randn(B, H, W, Cin).to(tdt).cuda()
        w = torch.randn(Cout, Cin, 3, 3).cuda() * 0.05
        with ops.dispatch_override("CMU_CONV_V6", v6):
            ops.conv3x3_fwd(ops.Act(x, 0, Cin), ops.pack_conv3x3(w, dt), ops.new_act(B, H, W, Cout, dt, "cuda"), None)
        torch.cuda.synchronize()
        return last_kernel()
    if os.environ.get("CMU_CONV_V6", "") != "1":
        assert kern(8, 128, 512, 64, 64, "f16") != "conv_igemm6_kernel"      # 1,024 items: the bench's kind of launch, not taken by default
    assert kern(2, 32, 64, 64, 64, "f16", v6=1) == "conv_igemm6_kernel"
    assert kern(2, 32, 64, 128, 64, "bf16", v6=1) == "conv_igemm6_kernel"
    assert kern(2, 32, 64, 64, 64, "f16", v6=0) != "conv_igemm6_kernel"
    assert kern(2, 32, 64, 64, 64, "f32", v6=1) != "conv_igemm6_kernel"
    assert kern(2, 28, 28, 64, 64, "f16", v6=1) != "conv_igemm6_kernel"     # partial tiles
    assert kern(2, 32, 64, 96, 64, "f16", v6=1) != "conv_igemm6_kernel"     # K = 96: not whole pairs of 64-byte slices
    assert kern(2, 32, 64, 256, 64, "f16", v6=1) != "conv_igemm6_kernel"    # K = 256: the table does not fit beside the halo


_GRID = r'''
import sys, torch
sys.path.insert(0, %r)
from cmunet_amd import ops
g = torch.Generator().manual_seed(3)
B, H, W, Cin, Cout = 7, 32, 64, 64, 128
x = torch.randn(B, H, W, Cin, generator=g).half().cuda()
w = (torch.randn(Cout, Cin, 3, 3, generator=g) / 24).cuda()
sc, sh = (torch.rand(Cin, generator=g) + 0.5).cuda(), (torch.randn(Cin, generator=g) * 0.3).cuda()
y = ops.new_act(B, H, W, Cout, "f16", "cuda")
st = ops.new_stats(B, H, W, Cout, "cuda")
with ops.dispatch_override("CMU_CONV_V6", 1):
    ops.conv3x3_fwd(ops.Act(x, 0, Cin, sc, sh, 0), ops.pack_conv3x3(w, "f16"), y, st)
torch.cuda.synchronize()
torch.save({"y": y.buf.cpu(), "st": st.cpu()}, sys.argv[1])
'''


def test_v6_item_lists_of_any_length_give_the_same_bits(ops, tmp_path):
    """56 items (7 images x 4 tiles x 2 channel blocks) over 512 / 5 / 3 / 1 workgroups (CMU_CONV_PERSIST_GRID): one item per workgroup, lists of
    11-12, of 18-19 and ONE list of 56 items with every item boundary in play -- same output bits, same statistics bits."""
    res, procs = [], []
    try:
        for grid in ("0", "5", "3", "1"):           # (the four children side by side: each is mostly interpreter start-up)
            o = str(tmp_path / f"g{grid}.pt")
            env = dict(os.environ, CMU_CONV_NARROW="0")
            if grid != "0":
                env["CMU_CONV_PERSIST_GRID"] = grid
            procs.append((o, subprocess.Popen([sys.executable, "-c", _GRID % ROOT, o], env=env)))
        for o, pr in procs:
            assert pr.wait(timeout=300) == 0
            res.append(torch.load(o))
    finally:
        for _, pr in procs:
            if pr.poll() is None:
                pr.kill()
            pr.wait()
    for r in res[1:]:
        assert torch.equal(r["y"].view(torch.uint8), res[0]["y"].view(torch.uint8))
        assert torch.equal(r["st"], res[0]["st"])
    assert bool(torch.isfinite(res[0]["y"].float()).all())
