"""GPU parity of each forward C-ABI entry point against the CPU oracle's arithmetic (torch fp64 on the
same, dtype-quantised inputs).  Tolerances (stated per dtype below) cover accumulation order and the
final rounding to the storage dtype only, because the oracle is fed the already-quantised operands."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# relative-to-max tolerance per storage dtype: f32 exact-fp32 MFMA chain; f16 11-bit; bf16 8-bit mantissa
TOL = {"f32": 2e-5, "f16": 2e-3, "bf16": 1.6e-2}
DTS = ["f32", "f16", "bf16"]


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from cmunet_amd import ops as o
    return o


def q(t, dt, ops):
    """quantise an fp32 CPU tensor to the storage dtype and back (so the oracle sees what the GPU sees)."""
    return t.to(ops.TORCH_DT[ops.dt_code(dt)]).to(torch.float32)


def to_act(x_nchw, dt, ops, ld=None, coff=0):
    B, C, H, W = x_nchw.shape
    ld = C if ld is None else ld
    buf = torch.full((B, H, W, ld), 7.0, dtype=ops.TORCH_DT[ops.dt_code(dt)], device="cuda")
    buf[..., coff:coff + C] = x_nchw.permute(0, 2, 3, 1).to(buf.dtype).cuda()
    return ops.Act(buf, coff, C)


def from_act(a):
    return a.buf[..., a.coff:a.coff + a.C].float().cpu().permute(0, 3, 1, 2).contiguous()


def check(got, ref, tol, what):
    err = (got.double() - ref.double()).abs().max().item()
    scale = max(ref.abs().max().item(), 1e-6)
    assert err <= tol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e} (tol {tol})"


CONV_CASES = [
    # B, H, W, Cin, Cout, ldx_extra, ldy_extra, transform
    (2, 20, 24, 32, 64, 0, 0, False),
    (1, 16, 16, 16, 16, 0, 0, True),
    (2, 7, 9, 8, 8, 8, 16, True),
    (1, 33, 17, 64, 72, 0, 0, True),
    (1, 16, 32, 128, 64, 64, 0, False),
    # whole 64-byte K slices and 64/128-channel blocks -> wide-tile kernel (conv_igemm3.inc): partial tiles in both
    # directions, a tile whose right 16x16 half is outside the image, several n-blocks, channel-sliced buffers
    (1, 20, 33, 128, 128, 0, 0, True),
    (2, 16, 16, 256, 256, 0, 64, True),
    (1, 7, 9, 64, 256, 32, 0, False),
    (1, 32, 16, 512, 128, 0, 0, True),
    (2, 35, 70, 64, 64, 0, 0, True),
    (1, 17, 40, 96, 64, 0, 32, False),
    (1, 48, 64, 32, 384, 0, 0, True),
    # whole tiles (H % 16 == 0, W % 32 == 0): the persistent kernel -- several items per workgroup list, several channel
    # blocks, long K (table of 1024 channels), channel-sliced input and output buffers, no transform
    (2, 32, 64, 256, 256, 0, 64, True),
    (1, 48, 32, 512, 128, 32, 0, True),
    (1, 16, 64, 1024, 128, 0, 0, True),
    (3, 32, 32, 128, 64, 0, 0, False),
    (2, 16, 96, 64, 64, 64, 64, True),
    # not eligible (N = 192 is not a multiple of 128; K = 136 has a ragged slice): first kernel
    (2, 16, 16, 256, 192, 0, 64, True),
    (1, 7, 9, 136, 256, 8, 0, False),
]


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv3x3_fwd(ops, dt, case):
    B, H, W, Cin, Cout, ex, ey, tf = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = q(torch.randn(B, Cin, H, W, generator=g), dt, ops)
    w = q(torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5, dt, ops)
    xa = to_act(x, dt, ops, ld=Cin + ex, coff=ex)
    ref_in = x.double()
    if tf:
        sc = torch.rand(Cin, generator=g) + 0.5
        sh = torch.randn(Cin, generator=g) * 0.3
        rf = (Cin // 16) * 8          # a multiple of the 16-byte chunk (8 x 16-bit / 4 x f32)
        xa = xa.with_transform(sc.cuda(), sh.cuda(), rf)
        t = x.double() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)
        t[:, rf:] = t[:, rf:].clamp_min(0)
        # the kernel rounds the transformed value to the storage dtype before the MFMA
        ref_in = q(t.float(), dt, ops).double()
    wp = ops.pack_conv3x3(w.cuda(), dt)
    ybuf = torch.full((B, H, W, Cout + ey), -3.0, dtype=ops.TORCH_DT[ops.dt_code(dt)], device="cuda")
    ya = ops.Act(ybuf, ey, Cout)
    stats = ops.new_stats(B, H, W, Cout, "cuda")
    ops.conv3x3_fwd(xa, wp, ya, stats)
    torch.cuda.synchronize()
    ref = F.conv2d(ref_in, w.double(), padding=1)
    # a transformed 16-bit input is rounded once more inside the kernel: allow for it
    tol = TOL[dt] * (2.0 if tf else 1.0)
    check(from_act(ya), ref, tol, "conv3x3 y")
    if ey:
        assert (ybuf[..., :ey] == -3.0).all(), "conv wrote outside its channel slice"
    s = stats.double().sum(0).cpu()
    check(s[0], ref.sum((0, 2, 3)), 1e-3 if dt != "f32" else 1e-4, "stats sum")
    check(s[1], (ref * ref).sum((0, 2, 3)), 1e-3 if dt != "f32" else 1e-4, "stats sumsq")


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("seed", range(12))
def test_conv3x3_fwd_random_shapes_of_the_wide_kernels(ops, dt, seed):
    """Seeded random shapes with whole 64 / 128-channel blocks -- the shapes the small-shape dispatch of the wide kernels decides on
    (64-channel blocks for under-filled launches, 16 x 16 pixel tiles for an odd number of 16-pixel columns, partial tiles on the
    persistent kernel): batches of 1-6, 1-70 pixels per side, channel-sliced buffers, a pending transform whose ReLU starts at a
    random 16-byte chunk -- against float64, outputs and BatchNorm statistics."""
    import numpy as np
    rs = np.random.RandomState(2000 + seed)
    B, H, W = int(rs.randint(1, 7)), int(rs.randint(1, 71)), int(rs.randint(1, 71))
    Cin, Cout = 64 * int(rs.randint(1, 5)), 64 * int(rs.choice([1, 2, 4, 6]))
    tf = bool(rs.randint(0, 2))
    ex, ey = 16 * int(rs.randint(0, 3)), 16 * int(rs.randint(0, 3))
    g = torch.Generator().manual_seed(seed)
    x = q(torch.randn(B, Cin, H, W, generator=g), dt, ops)
    w = q(torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5, dt, ops)
    xa = to_act(x, dt, ops, ld=Cin + ex, coff=ex)
    ref_in = x.double()
    if tf:
        sc, sh = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
        rf = 8 * int(rs.randint(0, Cin // 8 + 1))
        neg = bool(seed % 2) and rf > 0      # a negative relu_from: the FIRST rf channels are the activated ones (round 4)
        xa = xa.with_transform(sc.cuda(), sh.cuda(), -rf if neg else rf)
        t = x.double() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)
        if neg:
            t[:, :rf] = t[:, :rf].clamp_min(0)
        else:
            t[:, rf:] = t[:, rf:].clamp_min(0)
        ref_in = q(t.float(), dt, ops).double()
    ybuf = torch.full((B, H, W, Cout + ey), -3.0, dtype=ops.TORCH_DT[ops.dt_code(dt)], device="cuda")
    ya = ops.Act(ybuf, ey, Cout)
    stats = ops.new_stats(B, H, W, Cout, "cuda")
    ops.conv3x3_fwd(xa, ops.pack_conv3x3(w.cuda(), dt), ya, stats)
    torch.cuda.synchronize()
    ref = F.conv2d(ref_in, w.double(), padding=1)
    what = f"{(B, H, W, Cin, Cout, tf, ex, ey)}"
    check(from_act(ya), ref, TOL[dt] * (2.0 if tf else 1.0), "conv3x3 y " + what)
    if ey:
        assert (ybuf[..., :ey] == -3.0).all(), "conv wrote outside its channel slice " + what
    sm = stats.double().sum(0).cpu()
    check(sm[0], ref.sum((0, 2, 3)), 1e-3 if dt != "f32" else 1e-4, "stats sum " + what)
    check(sm[1], (ref * ref).sum((0, 2, 3)), 1e-3 if dt != "f32" else 1e-4, "stats sumsq " + what)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("shape", [(2, 18, 21, 24, 40), (1, 18, 37, 128, 64), (2, 16, 32, 64, 128)])
def test_conv3x3_dgrad_via_flipped_pack(ops, dt, shape):
    B, H, W, Cin, Cout = shape
    g = torch.Generator().manual_seed(5)
    dy = q(torch.randn(B, Cout, H, W, generator=g), dt, ops)
    w = q(torch.randn(Cout, Cin, 3, 3, generator=g) / (Cout * 9) ** 0.5, dt, ops)
    wp = ops.pack_conv3x3(w.cuda(), dt, transpose_flip=True)
    dx = ops.new_act(B, H, W, Cin, dt, "cuda")
    ops.conv3x3_fwd(to_act(dy, dt, ops), wp, dx, None)
    ref = torch.nn.grad.conv2d_input((B, Cin, H, W), w.double(), dy.double(), padding=1)
    check(from_act(dx), ref, TOL[dt], "dgrad dx")


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("shape", [(2, 20, 24, 16), (1, 16, 16, 64), (2, 5, 37, 32)])
@pytest.mark.parametrize("masked", [0, 1, 2])
def test_conv3x3_c1_fwd(ops, dt, shape, masked):
    B, H, W, Cout = shape
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, H, W, generator=g)
    w = torch.randn(Cout, 1, 3, 3, generator=g) / 3
    mask = None
    xin = x
    if masked:
        m = (torch.rand(B, H, W, generator=g) > 0.5).to(torch.uint8)
        if masked == 1:   # reference behaviour: mask of sample 0 for the whole batch
            mask = m[:1].contiguous()
            xin = x * (1 - mask[0]).float()
        else:
            mask = m
            xin = x * (1 - m).float()
    ya = ops.new_act(B, H, W, Cout, dt, "cuda")
    stats = ops.new_stats(B, H, W, Cout, "cuda")
    ops.conv3x3_c1_fwd(x.cuda(), w.cuda(), ya, stats, None if mask is None else mask.cuda(), masked == 2)
    ref = F.conv2d(xin.double().unsqueeze(1), w.double(), padding=1)
    check(from_act(ya), ref, TOL[dt], "c1 conv")
    s = stats.double().sum(0).cpu()
    check(s[0], ref.sum((0, 2, 3)), 1e-4, "c1 stats sum")
    check(s[1], (ref * ref).sum((0, 2, 3)), 1e-4, "c1 stats sumsq")


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("case", [(2, 5, 7, 40, True), (3, 16, 16, 128, True), (2, 9, 4, 64, False), (3, 4, 4, 20, True), (1, 70, 3, 8, False)])
def test_gap_fwd(ops, dt, case):
    """Global average pool of relu(bn(y)) (MoCo encoder head, moco_data_module.py:62-64): 16-byte-chunk kernel where C allows, the
    element kernel otherwise; with and without a pending transform; more pixels than one pass of the pixel parts."""
    B, H, W, C, tf = case
    g = torch.Generator().manual_seed(sum(case[:4]))
    y = q(torch.randn(B, C, H, W, generator=g), dt, ops)
    ya = to_act(y, dt, ops, ld=C + 16, coff=8 if dt != "f32" else 4)
    ref = y.double()
    if tf:
        sc, sh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
        ya = ya.with_transform(sc.cuda(), sh.cuda(), 0)
        ref = F.relu(ref * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1))
    out = torch.full((B, C), 3.0, device="cuda")
    ops.gap_fwd(ya, out)
    check(out.cpu(), ref.mean((2, 3)), 1e-5, "gap")
    # backward: every pixel of a channel gets dout / (H W), written into a channel slice of a wider buffer
    dout = torch.randn(B, C, generator=g)
    dbuf = torch.full((B, H, W, C + 16), 5.0, dtype=ops.TORCH_DT[ops.dt_code(dt)], device="cuda")
    dA = ops.Act(dbuf, 8 if dt != "f32" else 4, C)
    ops.gap_bwd(dout.cuda(), dA)
    want = q((dout / (H * W)).float(), dt, ops).view(B, 1, 1, C).expand(B, H, W, C)
    got = dbuf[..., dA.coff:dA.coff + C].float().cpu()
    assert (got - want).abs().max().item() <= TOL[dt] * max(want.abs().max().item(), 1e-6)
    assert (dbuf[..., :dA.coff] == 5.0).all() and (dbuf[..., dA.coff + C:] == 5.0).all()


@pytest.mark.parametrize("training", [True, False])
def test_bn_finalize(ops, training):
    B, H, W, C = 3, 20, 24, 48
    g = torch.Generator().manual_seed(9)
    y = torch.randn(B, C, H, W, generator=g) * 2 + 0.5      # bias-free conv output
    bias = torch.randn(C, generator=g)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    rm, rv = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    # slab as the conv epilogue would write it: per-(image,row-block) partial sums
    parts = y.view(B, C, 4, H // 4, W).permute(0, 2, 1, 3, 4).reshape(B * 4, C, -1)
    stats = torch.stack([parts.sum(-1), (parts * parts).sum(-1)], 1).contiguous().cuda()
    rm_d, rv_d = rm.clone().cuda(), rv.clone().cuda()
    scale, shift = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    smean, sinv = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    from cmunet_amd import _lib
    ws = torch.empty(_lib.lib().cmu_bn_finalize_ws_bytes(C), dtype=torch.uint8, device="cuda")
    ops.bn_finalize(stats, B * H * W, bias.cuda(), gamma.cuda(), beta.cuda(), rm_d, rv_d, 0.1, 1e-5, training, scale,
                    shift, smean, sinv, ws)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    ref = F.batch_norm(y + bias.view(1, -1, 1, 1), rm_ref, rv_ref, gamma, beta, training, 0.1, 1e-5)
    got = y * scale.cpu().view(1, -1, 1, 1) + shift.cpu().view(1, -1, 1, 1)
    check(got, ref, 2e-5, "bn apply via scale/shift")
    check(rm_d.cpu(), rm_ref, 1e-5, "running_mean")
    check(rv_d.cpu(), rv_ref, 1e-5, "running_var")


@pytest.mark.parametrize("dt", DTS)
def test_bnrelu_maxpool(ops, dt):
    B, H, W, C = 2, 12, 20, 32
    g = torch.Generator().manual_seed(11)
    y = q(torch.randn(B, C, H, W, generator=g), dt, ops)
    sc, sh = torch.randn(C, generator=g), torch.randn(C, generator=g) * 0.2   # negative scales too
    ya = to_act(y, dt, ops, ld=C + 8, coff=8).with_transform(sc.cuda(), sh.cuda(), 0)
    out = ops.new_act(B, H // 2, W // 2, C, dt, "cuda")
    ops.bnrelu_maxpool_fwd(ya, out)
    ref = F.max_pool2d(F.relu(y.double() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)), 2)
    check(from_act(out), ref, TOL[dt], "pool")


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("case", [(2, 8, 8, 32, 16, True), (1, 5, 9, 64, 32, False), (2, 16, 16, 16, 8, True),
                                  # 4*Cout % 256 == 0 and Cin a whole number of 128-byte steps -> GEMM kernel (conv_gemm.inc):
                                  # partial tiles, several channel blocks, with / without pending transform
                                  (1, 5, 9, 64, 64, True), (2, 20, 17, 128, 128, False), (1, 16, 16, 192, 64, True),
                                  # (up to four 128-byte K steps: the two-workgroups-per-CU form; more: the ping-pong form)
                                  (1, 9, 7, 320, 64, True), (2, 20, 17, 384, 128, False)])
def test_convT2x2_fwd(ops, dt, case):
    B, H, W, Cin, Cout, tf = case
    if dt != "f32" and Cout % 8:
        pytest.skip("16-bit needs Cout % 8 == 0")
    g = torch.Generator().manual_seed(13)
    x = q(torch.randn(B, Cin, H, W, generator=g), dt, ops)
    w = q(torch.randn(Cin, Cout, 2, 2, generator=g) / (Cin) ** 0.5, dt, ops)
    bias = torch.randn(Cout, generator=g)
    xa = to_act(x, dt, ops)
    ref_in = x.double()
    if tf:
        sc, sh = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
        xa = xa.with_transform(sc.cuda(), sh.cuda(), 0)
        ref_in = q(F.relu(x.double() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)).float(), dt, ops).double()
    wp = ops.pack_convT2x2(w.cuda(), dt, 0)
    obuf = torch.full((B, 2 * H, 2 * W, 2 * Cout), 5.0, dtype=ops.TORCH_DT[ops.dt_code(dt)], device="cuda")
    oa = ops.Act(obuf, 0, Cout)        # left half of a concat buffer
    ops.convT2x2_fwd(xa, wp, bias.cuda(), oa)
    ref = F.conv_transpose2d(ref_in, w.double(), bias.double(), stride=2)
    check(from_act(oa), ref, TOL[dt] * (2.0 if tf else 1.0), "convT fwd")
    assert (obuf[..., Cout:] == 5.0).all()


@pytest.mark.parametrize("dt", DTS)
def test_convT2x2_dgrad(ops, dt):
    B, H, W, Cin, Cout = 2, 9, 6, 48, 24
    g = torch.Generator().manual_seed(15)
    dout = q(torch.randn(B, Cout, 2 * H, 2 * W, generator=g), dt, ops)
    w = q(torch.randn(Cin, Cout, 2, 2, generator=g) / (Cout * 4) ** 0.5, dt, ops)
    wp = ops.pack_convT2x2(w.cuda(), dt, 1)
    da = to_act(dout, dt, ops, ld=2 * Cout, coff=0)
    dx = ops.new_act(B, H, W, Cin, dt, "cuda")
    ops.convT2x2_dgrad(da, wp, dx)
    ref = F.conv2d(dout.double(), w.double(), stride=2)     # adjoint of conv_transpose2d(stride 2, k 2)
    check(from_act(dx), ref, TOL[dt], "convT dgrad")


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("C", [16, 64])
def test_conv1x1_head(ops, dt, C):
    B, H, W, K = 2, 10, 13, 2
    g = torch.Generator().manual_seed(17)
    x = q(torch.randn(B, C, H, W, generator=g), dt, ops)
    sc, sh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    w, b = torch.randn(K, C, generator=g) / C ** 0.5, torch.randn(K, generator=g)
    xa = to_act(x, dt, ops).with_transform(sc.cuda(), sh.cuda(), 0)
    logits = torch.empty(B, K, H, W, device="cuda")
    ops.conv1x1_head_fwd(xa, w.cuda(), b.cuda(), logits)
    a = F.relu(x.double() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1))
    ref = F.conv2d(a, w.double().view(K, C, 1, 1), b.double())
    check(logits.cpu(), ref, 2e-5, "head logits")


@pytest.mark.parametrize("dt", DTS)
def test_layout_roundtrip(ops, dt):
    B, C, H, W = 2, 24, 6, 10
    g = torch.Generator().manual_seed(19)
    x = q(torch.randn(B, C, H, W, generator=g), dt, ops)
    a = ops.new_act(B, H, W, C, dt, "cuda")
    ops.nchw_to_nhwc(x.cuda(), a)
    assert torch.equal(from_act(a), x)
    sc, sh = torch.randn(C, generator=g), torch.randn(C, generator=g)
    out = ops.apply_to_nchw(a.with_transform(sc.cuda(), sh.cuda(), 8))
    t = x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    t[:, 8:] = t[:, 8:].clamp_min(0)
    check(out.cpu(), t, 1e-6, "apply_to_nchw")


def test_bad_args_fail_loudly(ops):
    from cmunet_amd._lib import CmuError
    a = ops.new_act(1, 16, 16, 12, "bf16", "cuda")     # 12 channels: not a multiple of 8
    o = ops.new_act(1, 16, 16, 16, "bf16", "cuda")
    w = torch.zeros(16 * 9 * 64, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(CmuError):
        ops.conv3x3_fwd(a, w, o, None)


@pytest.mark.parametrize("chans,wide", [((256, 256), "1"), ((64, 64), "2"), ((128, 64), "2")])
def test_conv3x3_wide_tile_kernel_matches_first(chans, wide):
    """The wide-tile kernels (conv_igemm3.inc; CMU_CONV_WIDE=2 also sends 64-channel layers with few input channels to
    the 32-byte-slice variant) compute the same convolution as the first kernel, which CMU_CONV_WIDE=0 selects for every
    layer (different fp32 summation order only): run both in subprocesses and compare."""
    import os
    import subprocess
    import sys
    import tempfile
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, torch
sys.path.insert(0, %r)
from cmunet_amd import ops
g = torch.Generator().manual_seed(0)
B, H, W, Cin, Cout = 2, 20, 33, %d, %d
x = torch.randn(B, H, W, Cin, generator=g).to(torch.bfloat16).cuda()
w = (torch.randn(Cout, Cin, 3, 3, generator=g) / 48).cuda()
sc, sh = (torch.rand(Cin, generator=g) + 0.5).cuda(), (torch.randn(Cin, generator=g) * 0.3).cuda()
y = ops.new_act(B, H, W, Cout, "bf16", "cuda")
st = ops.new_stats(B, H, W, Cout, "cuda")
ops.conv3x3_fwd(ops.Act(x, 0, Cin, sc, sh, 0), ops.pack_conv3x3(w, "bf16"), y, st)
torch.save({"y": y.buf.float().cpu(), "s": st.sum(0).cpu()}, sys.argv[1])
''' % (root, chans[0], chans[1])
    outs = []
    for v in ("0", wide):
        with tempfile.NamedTemporaryFile(suffix=".pt", delete=False) as f:
            path = f.name
        subprocess.run([sys.executable, "-c", code, path], check=True, env=dict(os.environ, CMU_CONV_WIDE=v), timeout=300)
        outs.append(torch.load(path))
        os.unlink(path)
    check(outs[1]["y"], outs[0]["y"], 8e-3, "wide vs first y (bf16 rounding of different fp32 sums)")
    check(outs[1]["s"], outs[0]["s"], 1e-4, "wide vs first stats")


_PERSIST_SHAPES = [((64, 128), (48, 96)), ((128, 64), (48, 96)), ((64, 64), (48, 96)), ((128, 64), (64, 96)), ((64, 64), (96, 64)),
                   ((64, 128), (56, 56)), ((128, 64), (40, 80)), ((64, 64), (28, 28)), ((128, 128), (24, 48))]


def test_conv3x3_persistent_kernel_is_bit_identical():
    """The persistent wide kernel (conv_igemm3p.inc: one workgroup walks a list of tiles) against the one-tile-per-workgroup
    kernel (CMU_CONV_PERSIST=0) on whole-tile shapes and on shapes whose tiles overhang the image (the PART instantiation:
    predicated stores / statistics; 56, 40 x 80, 28, 24 x 48), with 13 workgroups forced so that every workgroup streams several
    tiles and the per-XCD item ranges are uneven: same MFMA order per accumulator, same statistics folding order ->
    identical bits, for the forward (pending transform + BN statistics), the plain data gradient and the data gradient
    with fused BN-backward sums.  (Round 6: ONE pair of child processes walks all nine shapes -- the switches are read once per process, so
    the two settings need two processes, not eighteen: 24 s of interpreter start-up less in the driver's GPU run.)"""
    import os
    import subprocess
    import sys
    import tempfile
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, torch
sys.path.insert(0, %r)
from cmunet_amd import ops
res = {}
for (Cin, Cout), (H, W) in %r:
  B = 3
  g = torch.Generator().manual_seed(0)
  x = torch.randn(B, H, W, Cin, generator=g).to(torch.bfloat16).cuda()
  w = (torch.randn(Cout, Cin, 3, 3, generator=g) / 48).cuda()
  sc, sh = (torch.rand(Cin, generator=g) + 0.5).cuda(), (torch.randn(Cin, generator=g) * 0.3).cuda()
  out = {}
  y = ops.new_act(B, H, W, Cout, "bf16", "cuda")
  st = ops.new_stats(B, H, W, Cout, "cuda")
  ops.conv3x3_fwd(ops.Act(x, 0, Cin, sc, sh, 0), ops.pack_conv3x3(w, "bf16"), y, st)
  out["y"], out["s"] = y.buf.clone(), st.clone()
  y2 = ops.new_act(B, H, W, Cout, "bf16", "cuda")
  ops.conv3x3_fwd(ops.Act(x, 0, Cin), ops.pack_conv3x3(w, "bf16"), y2, None)
  out["y2"] = y2.buf.clone()
  # data gradient into a conv+BN layer with raw output xr (Cout channels after the transposed-flipped pack)
  xr = torch.randn(B, H, W, Cout, generator=g).to(torch.bfloat16).cuda()
  bsc, bsh = (torch.rand(Cout, generator=g) + 0.5).cuda(), (torch.randn(Cout, generator=g) * 0.3).cuda()
  mu, istd = (torch.randn(Cout, generator=g) * 0.1).cuda(), (torch.rand(Cout, generator=g) + 0.5).cuda()
  wt = (torch.randn(Cin, Cout, 3, 3, generator=g) / 48).cuda()       # the layer dY belongs to: Cout -> Cin
  dX = ops.new_act(B, H, W, Cout, "bf16", "cuda")
  slab = ops.new_stats(B, H, W, Cout, "cuda")
  ops.conv3x3_dgrad_bn(ops.Act(x, 0, Cin), ops.pack_conv3x3(wt, "bf16", transpose_flip=True), dX, ops.Act(xr, 0, Cout, bsc, bsh, 0), mu, istd, slab)
  out["dx"], out["slab"] = dX.buf.clone(), slab.clone()
  res[(Cin, Cout, H, W)] = {k: v.cpu() for k, v in out.items()}
torch.save(res, sys.argv[1])
''' % (root, _PERSIST_SHAPES)
    outs = []
    for env in ({"CMU_CONV_PERSIST": "0"}, {"CMU_CONV_PERSIST": "1", "CMU_CONV_PERSIST_GRID": "13"}):
        with tempfile.NamedTemporaryFile(suffix=".pt", delete=False) as f:
            path = f.name
        subprocess.run([sys.executable, "-c", code, path], check=True, env=dict(os.environ, CMU_CONV_WIDE="2", **env),
                       timeout=300)
        outs.append(torch.load(path))
        os.unlink(path)
    assert len(outs[0]) == len(_PERSIST_SHAPES)
    for shape in outs[0]:
        for k in outs[0][shape]:
            assert torch.equal(outs[0][shape][k].view(torch.uint8), outs[1][shape][k].view(torch.uint8)), (shape, k)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("hw", [(32, 64), (16, 16), (28, 28)])
def test_conv3x3_narrow_channel_blocks_equal_the_wide_ones(ops, dt, hw):
    """Launches with fewer 128-channel items than half the chip's CUs take 64-channel blocks (conv_igemm3.inc::igemm3_narrow_blocks:
    small batches, deep levels).  Same MFMA order per accumulator -> the outputs are bit-identical to the 128-channel blocks
    (CMU_CONV_NARROW forced off through cmu_set_dispatch_override); the per-tile statistics fold the same pixels in another lane order.  Whole-tile shape
    (persistent kernel), a 16 x 16 image (half-empty tiles, one-tile kernel) and 28 x 28 (partial tiles); forward with pending
    transform + statistics, plain data gradient, data gradient with fused BN-backward sums."""
    import os
    g = torch.Generator().manual_seed(5)
    B, (H, W), Cin, Cout = 2, hw, 128, 256
    tdt = ops.TORCH_DT[ops.dt_code(dt)]
    x = torch.randn(B, H, W, Cin, generator=g).to(tdt).cuda()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / 32).cuda()
    sc, sh = (torch.rand(Cin, generator=g) + 0.5).cuda(), (torch.randn(Cin, generator=g) * 0.3).cuda()
    xr = torch.randn(B, H, W, Cout, generator=g).to(tdt).cuda()
    bsc, bsh = (torch.rand(Cout, generator=g) + 0.5).cuda(), (torch.randn(Cout, generator=g) * 0.3).cuda()
    mu, istd = (torch.randn(Cout, generator=g) * 0.1).cuda(), (torch.rand(Cout, generator=g) + 0.5).cuda()
    wt = (torch.randn(Cin, Cout, 3, 3, generator=g) / 32).cuda()
    outs = []
    for v in (0, 1):
        # (the bit identity is a property of the 32x32x16 family: the 16x16x32 kernel of round 5, which takes the whole-tile 128-channel
        # launches by default, sums 32 channels per MFMA -- tests/test_gpu_conv_v5.py holds it to rounding against both)
        with ops.dispatch_override("CMU_CONV_NARROW", v), ops.dispatch_override("CMU_CONV_V5", 0):
            y = ops.new_act(B, H, W, Cout, dt, "cuda")
            st = ops.new_stats(B, H, W, Cout, "cuda")
            ops.conv3x3_fwd(ops.Act(x, 0, Cin, sc, sh, 0), ops.pack_conv3x3(w, dt), y, st)
            y2 = ops.new_act(B, H, W, Cout, dt, "cuda")
            ops.conv3x3_fwd(ops.Act(x, 0, Cin), ops.pack_conv3x3(w, dt), y2, None)
            dX = ops.new_act(B, H, W, Cout, dt, "cuda")
            slab = ops.new_stats(B, H, W, Cout, "cuda")
            ops.conv3x3_dgrad_bn(ops.Act(x, 0, Cin), ops.pack_conv3x3(wt, dt, transpose_flip=True), dX, ops.Act(xr, 0, Cout, bsc, bsh, 0), mu, istd, slab)
            torch.cuda.synchronize()
            outs.append({"y": y.buf.clone(), "y2": y2.buf.clone(), "dx": dX.buf.clone(), "s": st.clone(), "slab": slab.clone()})
    for k in ("y", "y2", "dx"):
        assert torch.equal(outs[0][k].view(torch.uint8), outs[1][k].view(torch.uint8)), k
    for k in ("s", "slab"):
        check(outs[1][k].cpu(), outs[0][k].cpu(), 2e-5, f"{k}: narrow vs wide blocks")
    # and against torch on the same operands (the narrow path is what small shapes now run)
    xa = torch.relu(x.float() * sc + sh).to(tdt).float()
    ref = torch.nn.functional.conv2d(xa.permute(0, 3, 1, 2), w.to(tdt).float(), padding=1).permute(0, 2, 3, 1)
    check(outs[1]["y"].float().cpu(), ref.cpu(), {"f32": 2e-5, "f16": 4e-3, "bf16": 2e-2}[dt], "narrow blocks vs torch")


@pytest.mark.parametrize("dt", DTS)
def test_conv3x3_short_last_batch_takes_another_kernel_same_outputs(ops, dt):
    """Advisor (round 3): the kernel form depends on the batch size (64-channel blocks while 2 x items <= CUs) -- the short last
    batch of an epoch runs another form than the full ones.  Its outputs for the images both batches hold are the same bits (same
    MFMA order per accumulator); only the statistics' fold order differs.  128 -> 256 @ 32 x 64 on a 256-CU chip:
    16 images = 128 items (narrow blocks), 17 images = 136 items (128-channel blocks)."""
    g = torch.Generator().manual_seed(6)
    Cin, Cout, H, W = 128, 256, 32, 64
    tdt = ops.TORCH_DT[ops.dt_code(dt)]
    x = torch.randn(17, H, W, Cin, generator=g).to(tdt).cuda()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / 32).cuda()
    sc, sh = (torch.rand(Cin, generator=g) + 0.5).cuda(), (torch.randn(Cin, generator=g) * 0.3).cuda()
    for v5 in (0, 1):
        outs, stats = [], []
        with ops.dispatch_override("CMU_CONV_V5", v5):
            for B in (17, 16):
                y, st = ops.new_act(B, H, W, Cout, dt, "cuda"), ops.new_stats(B, H, W, Cout, "cuda")
                ops.conv3x3_fwd(ops.Act(x[:B].contiguous(), 0, Cin, sc, sh, 0), ops.pack_conv3x3(w, dt), y, st)
                outs.append(y.buf)
                stats.append(st)
        if v5 == 0 or dt == "f32":
            assert torch.equal(outs[0][:16].view(torch.uint8), outs[1].view(torch.uint8))
        else:
            # round 5: at 16 bits the 17-image batch runs the 16x16x32 kernel (32 channels per MFMA), the 16-image batch the 64-channel
            # blocks of the 32x32x16 family: the same fp32 sums in another order -- equal to rounding of the stored type, not bit for bit
            check(outs[0][:16].float().cpu(), outs[1].float().cpu(), {"f16": 1e-3, "bf16": 8e-3}[dt], "shared images, two kernel families")
        rows = stats[1].shape[0]
        check(stats[1].sum(0).cpu(), stats[0][:rows].sum(0).cpu(), 2e-5, "statistics of the 16 shared images, two fold orders")


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("bhw", [(2, 16, 16), (3, 14, 14), (2, 9, 7), (22, 48, 48)])
def test_conv3x3_slim_tiles_equal_the_32_pixel_ones(ops, dt, bhw):
    """Images with an odd number of 16-pixel columns run on 16 x 16 pixel tiles (conv_igemm3.inc::igemm3_slim_tiles: one-column
    images at every dtype; f32 up to 7 columns once the 32-pixel tiles fill the chip) instead of leaving half of a 16 x 32 tile
    empty.  Same MFMA order per accumulator -> bit-identical outputs against CMU_CONV_SLIM forced off (cmu_set_dispatch_override); statistics in
    another fold order.  Forward with pending transform + statistics, data gradient with fused BN-backward sums."""
    import os
    B, H, W = bhw
    if B > 8 and dt != "f32":
        pytest.skip("the wide-image form is f32-only")
    g = torch.Generator().manual_seed(9)
    Cin, Cout = 64, 256
    tdt = ops.TORCH_DT[ops.dt_code(dt)]
    x = torch.randn(B, H, W, Cin, generator=g).to(tdt).cuda()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / 24).cuda()
    sc, sh = (torch.rand(Cin, generator=g) + 0.5).cuda(), (torch.randn(Cin, generator=g) * 0.3).cuda()
    xr = torch.randn(B, H, W, Cout, generator=g).to(tdt).cuda()
    bsc, bsh = (torch.rand(Cout, generator=g) + 0.5).cuda(), (torch.randn(Cout, generator=g) * 0.3).cuda()
    mu, istd = (torch.randn(Cout, generator=g) * 0.1).cuda(), (torch.rand(Cout, generator=g) + 0.5).cuda()
    wt = (torch.randn(Cin, Cout, 3, 3, generator=g) / 24).cuda()
    outs = []
    for v in (0, 1):
        with ops.dispatch_override("CMU_CONV_SLIM", v):
            y = ops.new_act(B, H, W, Cout, dt, "cuda")
            st = ops.new_stats(B, H, W, Cout, "cuda")
            ops.conv3x3_fwd(ops.Act(x, 0, Cin, sc, sh, 0), ops.pack_conv3x3(w, dt), y, st)
            dX = ops.new_act(B, H, W, Cout, dt, "cuda")
            slab = ops.new_stats(B, H, W, Cout, "cuda")
            ops.conv3x3_dgrad_bn(ops.Act(x, 0, Cin), ops.pack_conv3x3(wt, dt, transpose_flip=True), dX, ops.Act(xr, 0, Cout, bsc, bsh, 0), mu, istd, slab)
            torch.cuda.synchronize()
            outs.append({"y": y.buf.clone(), "dx": dX.buf.clone(), "s": st.clone(), "slab": slab.clone()})
    for k in ("y", "dx"):
        assert torch.equal(outs[0][k].view(torch.uint8), outs[1][k].view(torch.uint8)), k
    for k in ("s", "slab"):
        check(outs[1][k].sum(0).cpu(), outs[0][k].sum(0).cpu(), 2e-5, f"{k}: 16-pixel vs 32-pixel tiles")
    xa = torch.relu(x.float() * sc + sh).to(tdt).float()
    ref = torch.nn.functional.conv2d(xa.permute(0, 3, 1, 2), w.to(tdt).float(), padding=1).permute(0, 2, 3, 1)
    check(outs[1]["y"].float().cpu(), ref.cpu(), {"f32": 2e-5, "f16": 4e-3, "bf16": 2e-2}[dt], "16-pixel tiles vs torch")
    ssum = outs[1]["s"].sum(0).cpu()          # per-channel (sum, sum of squares) over all tiles == the stored output's
    yf = outs[1]["y"].float().cpu().reshape(-1, Cout)
    check(ssum[0], ref.cpu().reshape(-1, Cout).sum(0), {"f32": 2e-4, "f16": 5e-3, "bf16": 2e-2}[dt], "statistics: sum")


@pytest.mark.parametrize("dt", DTS)
def test_pack_batch_matches_single_packs(ops, dt):
    """cmu_pack_batch (all packs of a step in one launch) writes exactly what the per-weight entries write."""
    g = torch.Generator().manual_seed(23)
    ws = [torch.randn(40, 24, 3, 3, generator=g).cuda(), torch.randn(128, 64, 3, 3, generator=g).cuda(),
          torch.randn(64, 32, 2, 2, generator=g).cuda(), torch.randn(8, 8, 3, 3, generator=g).cuda()]
    items = [(ws[0], 0, 0), (ws[0], 0, 1), (ws[1], 0, 0), (ws[1], 0, 1), (ws[2], 1, 0), (ws[2], 1, 1), (ws[3], 0, 1)]
    plan = ops.PackPlan(items, dt)
    plan.run()
    torch.cuda.synchronize()
    for (w, kind, mode), out in zip(items, plan.outs):
        ref = ops.pack_conv3x3(w, dt, transpose_flip=bool(mode)) if kind == 0 else ops.pack_convT2x2(w, dt, mode)
        assert torch.equal(out.view(torch.uint8), ref.view(torch.uint8)), (kind, mode, tuple(w.shape))
