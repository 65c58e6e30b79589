"""Round 5: the persistent 3x3 conv kernel on the 16x16x32 MFMA shape (csrc/conv_igemm5.inc) -- the reference's ``nn.Conv2d(Cin, Cout, 3, padding=1)``
of ``DoubleConv`` (Finetuning/model.py:17-22) at the whole-tile 128-channel shapes -- against float64 arithmetic on the same 16-bit operands and against
the 32x32x16 family (``CMU_CONV_V5=0``) on the same tensors: forward with a pending BatchNorm+ReLU transform + batch statistics, plain forward, data
gradient with the BatchNorm-backward sums; channel slices of wider buffers (the concat-free decoder), ``relu_from`` of a concat input, a last
workgroup with fewer items, several items per workgroup (``CMU_CONV_PERSIST_GRID``)."""
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
    """3x3 convolution, padding 1, in float64 on the GPU: xa (B,H,W,Cin) already activated, w (Cout,Cin,3,3)."""
    return torch.nn.functional.conv2d(xa.double().permute(0, 3, 1, 2), w.double(), padding=1).permute(0, 2, 3, 1)


# (B, H, W, Cin, Cout, extra input stride, extra output stride, relu_from): K >= 128, N % 128 == 0, H % 16 == 0, W % 32 == 0; enough items that the
# launch does not take the narrow blocks (2 * items > CUs is not needed: tests force CMU_CONV_NARROW off)
FWD_CASES = [
    (2, 32, 64, 128, 128, 0, 0, 0),
    (3, 16, 32, 256, 256, 0, 0, 0),
    (2, 32, 32, 128, 256, 128, 128, 0),     # input = right half of a 256-channel buffer, output = left half of a 384-channel one
    (1, 48, 96, 192, 128, 0, 64, 64),       # K = 192 = six 64-byte slices; relu_from inside the view; output slice of a wider buffer
    (2, 16, 64, 256, 128, 0, 0, 128),       # concat input: channels [0, 128) carry no ReLU
    (5, 16, 32, 512, 384, 0, 0, 0),
]


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("case", FWD_CASES)
def test_v5_forward_transform_statistics_vs_float64(ops, dt, case):
    B, H, W, Cin, Cout, xs, ys, relu_from = case
    g = torch.Generator().manual_seed(17)
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
    for v5 in (1, 0):
        with ops.dispatch_override("CMU_CONV_V5", v5), ops.dispatch_override("CMU_CONV_NARROW", 0):
            ybuf = torch.full((B, H, W, Cout + ys), float("nan"), dtype=tdt, device="cuda")
            y = ops.Act(ybuf, 0, Cout)
            st = torch.full_like(ops.new_stats(B, H, W, Cout, "cuda"), float("nan"))
            ops.conv3x3_fwd(x, ops.pack_conv3x3(w, dt), y, st)
            y2buf = torch.full((B, H, W, Cout), float("nan"), dtype=tdt, device="cuda")
            ops.conv3x3_fwd(ops.Act(xa.contiguous(), 0, Cin), ops.pack_conv3x3(w, dt), ops.Act(y2buf, 0, Cout), None)
            torch.cuda.synchronize()
            outs[v5] = (ybuf.clone(), st.clone(), y2buf.clone())
    ybuf, st, y2 = outs[1]
    if ys:
        assert bool(torch.isnan(ybuf[..., Cout:]).all()), "the kernel wrote outside its channel slice"
    tol = {"f16": 6e-4, "bf16": 5e-3}[dt]                      # one rounding of the stored type on fp32 sums
    for name, got in (("transform", ybuf[..., :Cout]), ("plain", y2)):
        assert bool(torch.isfinite(got).all()), name
        e = rel_l2(got, ref)
        assert e <= tol, (name, e)
    # the 32x32x16 family on the same tensors: both are one rounding away from the fp32 sums
    assert rel_l2(ybuf[..., :Cout], outs[0][0][..., :Cout]) <= 1.5 * tol
    # statistics: per 16 x 16 tile, per channel (sum, sum of squares) of the fp32 accumulators
    t = ref.reshape(B, H // 16, 16, W // 16, 16, Cout)
    s1, s2 = t.sum((2, 4)).reshape(-1, Cout), (t * t).sum((2, 4)).reshape(-1, Cout)
    assert bool(torch.isfinite(st).all())
    assert rel_l2(st[:, 0], s1) <= 2e-5 + (0 if dt == "f16" else 1e-4) and rel_l2(st[:, 1], s2) <= 1e-4
    assert rel_l2(st, outs[0][1]) <= 1e-5


DG_CASES = [
    (2, 32, 64, 256, 128, 0),      # K = Cout of the layer = 256, N = Cin = 128
    (3, 16, 32, 512, 256, 0),
    (2, 16, 64, 256, 256, 128),    # dX into the right half of a wider gradient buffer
]


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("case", DG_CASES)
def test_v5_data_gradient_with_bn_backward_sums_vs_float64(ops, dt, case):
    B, H, W, K, N, xs = case
    g = torch.Generator().manual_seed(23)
    tdt = ops.TORCH_DT[ops.dt_code(dt)]
    dy = torch.randn(B, H, W, K, generator=g).to(tdt).cuda()
    wt = (torch.randn(K, N, 3, 3, generator=g) / (3.0 * K ** 0.5)).cuda()          # the layer's weight (Cout = K, Cin = N)
    yraw = torch.randn(B, H, W, N, generator=g).to(tdt).cuda()                     # raw output of the layer the gradient flows into
    bsc, bsh = (torch.rand(N, generator=g) + 0.5).cuda(), (torch.randn(N, generator=g) * 0.3).cuda()
    mu, istd = (torch.randn(N, generator=g) * 0.1).cuda(), (torch.rand(N, generator=g) + 0.5).cuda()
    # dX = conv_transpose of dY with the layer's weight = conv of dY with the flipped, transposed weight
    wq = wt.to(tdt).double()
    ref = torch.nn.functional.conv_transpose2d(dy.double().permute(0, 3, 1, 2), wq, padding=1).permute(0, 2, 3, 1)
    outs = {}
    for v5 in (1, 0):
        with ops.dispatch_override("CMU_CONV_V5", v5), ops.dispatch_override("CMU_CONV_NARROW", 0):
            dxbuf = torch.full((B, H, W, N + xs), float("nan"), dtype=tdt, device="cuda")
            slab = torch.full_like(ops.new_stats(B, H, W, N, "cuda"), float("nan"))
            ops.conv3x3_dgrad_bn(ops.Act(dy, 0, K), ops.pack_conv3x3(wt, dt, transpose_flip=True), ops.Act(dxbuf, xs, N), ops.Act(yraw, 0, N, bsc, bsh, 0),
                                 mu, istd, slab)
            torch.cuda.synchronize()
            outs[v5] = (dxbuf.clone(), slab.clone())
    dxbuf, slab = outs[1]
    dx = dxbuf[..., xs:]
    if xs:
        assert bool(torch.isnan(dxbuf[..., :xs]).all())
    tol = {"f16": 6e-4, "bf16": 5e-3}[dt]
    assert bool(torch.isfinite(dx).all()) and rel_l2(dx, ref) <= tol
    assert rel_l2(dx, outs[0][0][..., xs:]) <= 1.5 * tol
    # BatchNorm+ReLU backward sums of the producer layer on dX AS STORED: per 16 x 16 tile sum(gate * dX) and sum(gate * dX * xhat)
    dxs = dx.double()
    gate = ((yraw.double() * bsc.double() + bsh.double()) > 0).double()
    xhat = (yraw.double() - mu.double()) * istd.double()
    dz = gate * dxs
    t1 = dz.reshape(B, H // 16, 16, W // 16, 16, N).sum((2, 4)).reshape(-1, N)
    t2 = (dz * xhat).reshape(B, H // 16, 16, W // 16, 16, N).sum((2, 4)).reshape(-1, N)
    assert bool(torch.isfinite(slab).all())
    assert rel_l2(slab[:, 0], t1) <= 2e-5 and rel_l2(slab[:, 1], t2) <= 2e-5
    assert rel_l2(slab, outs[0][1]) <= 2 * tol            # (the other family's sums are taken on ITS stored dX)


def test_v5_is_what_the_whole_tile_layers_run(ops):
    """The dispatch: 16-bit operands, N % 128 == 0, K % 64 == 0 and >= 128, whole 16 x 32 tiles -> conv_igemm5_kernel; CMU_CONV_V5=0, K = 64, partial
    tiles, fp32 -> the older kernels."""
    from cmunet_amd import _lib
    lib = _lib.lib()
    lib.cmu_last_kernel.restype = __import__("ctypes").c_char_p

    def kern(B, H, W, Cin, Cout, dt, v5=1):
        tdt = ops.TORCH_DT[ops.dt_code(dt)]
        x = torch.randn(B, H, W, Cin).to(tdt).cuda()
        w = torch.randn(Cout, Cin, 3, 3).cuda() * 0.05
        with ops.dispatch_override("CMU_CONV_V5", v5), ops.dispatch_override("CMU_CONV_NARROW", 0):
            ops.conv3x3_fwd(ops.Act(x, 0, Cin), ops.pack_conv3x3(w, dt), ops.new_act(B, H, W, Cout, dt, "cuda"), None)
        torch.cuda.synchronize()
        return lib.cmu_last_kernel().decode()
    assert kern(2, 32, 64, 128, 128, "f16") == "conv_igemm5_kernel"
    assert kern(2, 32, 64, 128, 128, "bf16") == "conv_igemm5_kernel"
    assert kern(2, 32, 64, 128, 128, "f16", v5=0) == "conv_igemm3p_kernel"
    assert kern(2, 32, 64, 64, 128, "f16") != "conv_igemm5_kernel"        # K = 64: two positions per item, measured slower
    assert kern(2, 32, 64, 160, 128, "f16") != "conv_igemm5_kernel"       # K = 160: an odd number of 64-byte slices
    assert kern(2, 32, 64, 192, 128, "f16") == "conv_igemm5_kernel"
    assert kern(2, 28, 28, 128, 128, "f16") != "conv_igemm5_kernel"       # partial tiles
    assert kern(2, 32, 64, 128, 128, "f32") != "conv_igemm5_kernel"
    assert kern(2, 32, 64, 128, 64, "f16") != "conv_igemm5_kernel"        # 64 output channels


_GRID = r'''
import sys, torch
sys.path.insert(0, %r)
from cmunet_amd import ops
g = torch.Generator().manual_seed(3)
B, H, W, Cin, Cout = 7, 32, 64, 256, 256
x = torch.randn(B, H, W, Cin, generator=g).half().cuda()
w = (torch.randn(Cout, Cin, 3, 3, generator=g) / 48).cuda()
sc, sh = (torch.rand(Cin, generator=g) + 0.5).cuda(), (torch.randn(Cin, generator=g) * 0.3).cuda()
y = ops.new_act(B, H, W, Cout, "f16", "cuda")
st = ops.new_stats(B, H, W, Cout, "cuda")
ops.conv3x3_fwd(ops.Act(x, 0, Cin, sc, sh, 0), ops.pack_conv3x3(w, "f16"), y, st)
torch.cuda.synchronize()
torch.save({"y": y.buf.cpu(), "st": st.cpu()}, sys.argv[1])
'''


def test_v5_item_lists_of_any_length_give_the_same_bits(ops, tmp_path):
    """56 items (7 images x 4 tiles x 2 channel blocks) over 256 / 5 / 3 / 1 workgroups (CMU_CONV_PERSIST_GRID): one item per workgroup, lists of
    11-12, of 18-19 and ONE list of 56 items with every item boundary, XCD range and parity of the statistics strips in play -- same output bits,
    same statistics bits (each item's arithmetic does not depend on which workgroup runs it)."""
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
