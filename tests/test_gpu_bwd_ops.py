"""GPU parity of the backward / loss / optimiser C-ABI entry points against the oracle's arithmetic
(torch CPU fp64 autograd on the same dtype-quantised operands)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from test_gpu_fwd_ops import DTS, TOL, check, from_act, q, to_act  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from cmunet_amd import ops as o
    return o


def ws_bytes(n):
    return torch.empty(max(int(n), 16), dtype=torch.uint8, device="cuda")


def bn_consts(y, gamma, beta, eps=1e-5):
    mean = y.double().mean((0, 2, 3))
    var = y.double().var((0, 2, 3), unbiased=False)
    invstd = 1.0 / torch.sqrt(var + eps)
    scale = gamma.double() * invstd
    shift = beta.double() - mean * scale
    return mean.float(), invstd.float(), scale.float(), shift.float()


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("shape", [(2, 32, 12, 20), (1, 72, 9, 7), (2, 1024 + 64, 4, 4)])
def test_bn_relu_backward(ops, dt, shape):
    from cmunet_amd import _lib
    B, C, H, W = shape
    g = torch.Generator().manual_seed(C)
    y = q(torch.randn(B, C, H, W, generator=g) * 1.5 + 0.3, dt, ops)
    dA = q(torch.randn(B, C, H, W, generator=g), dt, ops)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    mean, invstd, scale, shift = bn_consts(y, gamma, beta)
    # oracle: autograd through batch_norm(train) + relu in fp64
    yd = y.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    a = F.relu(F.batch_norm(yd, None, None, gd, bd, True, 0.1, 1e-5))
    (a * dA.double()).sum().backward()
    ya = to_act(y, dt, ops, ld=C + 8, coff=8).with_transform(scale.cuda(), shift.cuda(), 0)
    da = to_act(dA, dt, ops)
    dgamma, dbeta = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    coef = torch.empty(2, C, device="cuda")
    ws = ws_bytes(_lib.lib().cmu_bn_bwd_ws_bytes(C))
    ops.bn_bwd_reduce(da, ya, mean.cuda(), invstd.cuda(), dgamma, dbeta, coef, ws)
    dy = ops.new_act(B, H, W, C, dt, "cuda")
    ops.bn_bwd_apply(da, ya, mean.cuda(), invstd.cuda(), coef, dy)
    # the ReLU gate is evaluated on fp32 scale/shift: elements within rounding of 0 may flip -> tolerance on few
    check(dgamma.cpu(), gd.grad, 2e-3, "dgamma")
    check(dbeta.cpu(), bd.grad, 2e-3, "dbeta")
    check(from_act(dy), yd.grad, max(TOL[dt], 2e-3), "dy")


WG_CASES = [(2, 20, 24, 32, 64, True), (1, 16, 16, 16, 16, False), (2, 7, 9, 8, 8, True), (1, 33, 17, 72, 40, True),
            (3, 16, 32, 128, 64, False),
            # Cout % 128 == 0 and Cin % 64 == 0 -> wide-tile kernel (conv_wgrad2.inc) for the 16-bit dtypes: partial tiles in
            # both directions, several blocks per operand, a single tile per split, more splits than tiles
            (2, 20, 24, 64, 128, True), (1, 7, 37, 128, 256, False), (3, 33, 16, 192, 128, True), (1, 8, 16, 64, 128, True),
            (1, 5, 3, 64, 128, True),
            # Cout == 64 and Cin % 128 == 0 -> the same kernel with the operands' roles swapped (halo on dY, transform on the
            # plain images, taps flipped at the store); (3, 16, 32, 128, 64) above is one more
            (2, 20, 24, 128, 64, True), (1, 7, 37, 256, 64, True), (1, 5, 3, 128, 64, False), (2, 33, 16, 128, 64, True),
            # whole 64-blocks on both sides where neither form applies -> 64 n x 64 c blocks with the wave halves as k-parts
            # (conv_wgrad2s.inc, 16-bit); at f32 all of the above with Cout % 64 == Cin % 64 == 0 run on conv_wgrad2f.inc
            (2, 20, 24, 64, 64, True), (1, 7, 37, 64, 64, False), (3, 33, 16, 64, 192, True), (1, 5, 3, 64, 64, True),
            (2, 40, 24, 192, 64, True), (4, 8, 16, 64, 64, True)]


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("case", WG_CASES)
def test_conv3x3_wgrad(ops, dt, case):
    from cmunet_amd import _lib
    B, H, W, Cin, Cout, tf = case
    g = torch.Generator().manual_seed(sum(case))
    x = q(torch.randn(B, Cin, H, W, generator=g), dt, ops)
    dy = q(torch.randn(B, Cout, H, W, generator=g), dt, ops)
    xa = to_act(x, dt, ops, ld=Cin + 16, coff=16)
    ref_in = x.double()
    if tf:
        sc, sh = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
        rf = (Cin // 16) * 8          # a multiple of the 16-byte chunk (8 x 16-bit / 4 x f32)
        xa = xa.with_transform(sc.cuda(), sh.cuda(), rf)
        t = x.double() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)
        t[:, rf:] = t[:, rf:].clamp_min(0)
        ref_in = q(t.float(), dt, ops).double()
    dW = torch.full((Cout, Cin, 3, 3), 9.0, device="cuda")
    ws = ws_bytes(_lib.lib().cmu_conv3x3_wgrad_ws_bytes(B, H, W, Cin, Cout, ops.dt_code(dt)))
    ops.conv3x3_wgrad(xa, to_act(dy, dt, ops), dW, ws)
    ref = torch.nn.grad.conv2d_weight(ref_in, (Cout, Cin, 3, 3), dy.double(), padding=1)
    check(dW.cpu(), ref, 1e-4 if dt == "f32" else 2e-3, "dW 3x3")


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("seed", range(10))
def test_conv3x3_wgrad_random_shapes_of_the_wide_kernels(ops, dt, seed):
    """Seeded random shapes with whole 64-channel blocks on both sides -- what the wide weight-gradient kernels serve (16-bit:
    conv_wgrad2.inc, its swapped form and conv_wgrad2s.inc; fp32: conv_wgrad2f.inc) -- against float64: heights and widths that do
    not divide into K tiles (down to one pixel), batches of 1-4, channel-sliced buffers, pending transform with the ReLU starting
    at a random 16-byte chunk."""
    from cmunet_amd import _lib
    rs = np.random.RandomState(1000 + seed)
    B, H, W = int(rs.randint(1, 5)), int(rs.randint(1, 41)), int(rs.randint(1, 41))
    Cin, Cout = 64 * int(rs.randint(1, 4)), 64 * int(rs.randint(1, 4))
    tf = bool(rs.randint(0, 2))
    g = torch.Generator().manual_seed(seed)
    x = q(torch.randn(B, Cin, H, W, generator=g), dt, ops)
    dy = q(torch.randn(B, Cout, H, W, generator=g), dt, ops)
    ex, ed = 16 * int(rs.randint(0, 3)), 16 * int(rs.randint(0, 3))
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
    dW = torch.full((Cout, Cin, 3, 3), 9.0, device="cuda")
    ws = ws_bytes(_lib.lib().cmu_conv3x3_wgrad_ws_bytes(B, H, W, Cin, Cout, ops.dt_code(dt)))
    ops.conv3x3_wgrad(xa, to_act(dy, dt, ops, ld=Cout + ed, coff=ed), dW, ws)
    ref = torch.nn.grad.conv2d_weight(ref_in, (Cout, Cin, 3, 3), dy.double(), padding=1)
    check(dW.cpu(), ref, 1e-4 if dt == "f32" else 2e-3, f"dW 3x3 {(B, H, W, Cin, Cout, tf, ex, ed)}")


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("masked", [0, 1, 2])
def test_conv3x3_c1_wgrad(ops, dt, masked):
    from cmunet_amd import _lib
    B, H, W, Cout = 2, 20, 37, 32
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, H, W, generator=g)
    dy = q(torch.randn(B, Cout, H, W, generator=g), dt, ops)
    mask, xin = None, x
    if masked:
        m = (torch.rand(B, H, W, generator=g) > 0.5).to(torch.uint8)
        mask = m[:1].contiguous() if masked == 1 else m
        xin = x * (1 - (m[0] if masked == 1 else m)).float()
    dW = torch.empty(Cout, 1, 3, 3, device="cuda")
    ws = ws_bytes(_lib.lib().cmu_conv3x3_c1_wgrad_ws_bytes(B, H, W, Cout))
    ops.conv3x3_c1_wgrad(x.cuda(), to_act(dy, dt, ops), dW, ws, None if mask is None else mask.cuda(), masked == 2)
    ref = torch.nn.grad.conv2d_weight(xin.double().unsqueeze(1), (Cout, 1, 3, 3), dy.double(), padding=1)
    check(dW.cpu(), ref, 1e-4, "dW c1")


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("masked", [0, 2])
def test_conv3x3_c1_wgrad_bn(ops, dt, masked):
    """First-layer wgrad with the BN+ReLU backward applied on the fly == unfused (apply in f64, then wgrad)."""
    from cmunet_amd import _lib
    B, H, W, Cout = 2, 20, 37, 32
    g = torch.Generator().manual_seed(4)
    x = torch.randn(B, H, W, generator=g)
    dA = q(torch.randn(B, Cout, H, W, generator=g), dt, ops)
    yraw = q(torch.randn(B, Cout, H, W, generator=g), dt, ops)
    sc, sh = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.3
    mu, istd = torch.randn(Cout, generator=g) * 0.1, torch.rand(Cout, generator=g) + 0.5
    coef = torch.randn(2, Cout, generator=g) * 0.1
    mask, xin = None, x
    if masked:
        mask = (torch.rand(B, H, W, generator=g) > 0.5).to(torch.uint8)
        xin = x * (1 - mask).float()
    v = lambda t: t.double().view(1, -1, 1, 1)
    dz = torch.where(yraw.double() * v(sc) + v(sh) > 0, dA.double(), torch.zeros((), dtype=torch.float64))
    dy = v(sc) * (dz - v(coef[0]) - (yraw.double() - v(mu)) * v(istd) * v(coef[1]))
    dW = torch.empty(Cout, 1, 3, 3, device="cuda")
    ws = ws_bytes(_lib.lib().cmu_conv3x3_c1_wgrad_ws_bytes(B, H, W, Cout))
    ops.conv3x3_c1_wgrad_bn(x.cuda(), to_act(dA, dt, ops), to_act(yraw, dt, ops), sc.cuda(), sh.cuda(), mu.cuda(), istd.cuda(),
                            coef.cuda(), dW, ws, None if mask is None else mask.cuda(), masked == 2)
    ref = torch.nn.grad.conv2d_weight(xin.double().unsqueeze(1), (Cout, 1, 3, 3), dy, padding=1)
    check(dW.cpu(), ref, 1e-4, "dW c1 (fused BN backward)")


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("case", [(2, 20, 37, 32, 0), (1, 48, 64, 64, 1), (3, 33, 16, 64, 2), (2, 16, 16, 8, 0)])
def test_conv3x3_c1_wgrad_bn_recomputed_output_is_bit_identical(ops, dt, case):
    """cmu_conv3x3_c1_wgrad_bn_w (round 4): the layer's raw output recomputed from the image and the forward weights instead of
    read -- the bits cmu_conv3x3_c1_fwd stored, so the weight gradient equals the reading form's bit for bit (partial tiles, several
    tiles per workgroup, batch mask / per-sample mask)."""
    from cmunet_amd import _lib
    B, H, W, Cout, masked = case
    g = torch.Generator().manual_seed(11 + Cout)
    x = torch.randn(B, H, W, generator=g).cuda()
    w = (torch.randn(Cout, 1, 3, 3, generator=g) / 3).cuda()
    dA = to_act(q(torch.randn(B, Cout, H, W, generator=g), dt, ops), dt, ops)
    sc, sh = (torch.rand(Cout, generator=g) + 0.5).cuda(), (torch.randn(Cout, generator=g) * 0.3).cuda()
    mu, istd = (torch.randn(Cout, generator=g) * 0.1).cuda(), (torch.rand(Cout, generator=g) + 0.5).cuda()
    coef = (torch.randn(2, Cout, generator=g) * 0.1).cuda()
    mask = None
    if masked:
        m = (torch.rand(B, H, W, generator=g) > 0.5).to(torch.uint8)
        mask = (m[:1] if masked == 1 else m).contiguous().cuda()
    y = ops.new_act(B, H, W, Cout, dt, "cuda")
    ops.conv3x3_c1_fwd(x, w, y, None, mask, masked == 2)
    ws = ws_bytes(_lib.lib().cmu_conv3x3_c1_wgrad_ws_bytes(B, H, W, Cout))
    dW_read, dW_rec = torch.empty(Cout, 1, 3, 3, device="cuda"), torch.empty(Cout, 1, 3, 3, device="cuda")
    ops.conv3x3_c1_wgrad_bn(x, dA, y, sc, sh, mu, istd, coef, dW_read, ws, mask, masked == 2)
    ops.conv3x3_c1_wgrad_bn(x, dA, y, sc, sh, mu, istd, coef, dW_rec, ws, mask, masked == 2, w=w)
    assert torch.equal(dW_read, dW_rec)
    assert float(dW_read.abs().max()) > 0


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("with_skip", [True, False])
def test_maxpool_bwd(ops, dt, with_skip):
    B, C, H, W = 2, 32, 8, 12
    g = torch.Generator().manual_seed(5)
    y = q(torch.randn(B, C, H, W, generator=g), dt, ops)
    sc, sh = torch.randn(C, generator=g), torch.randn(C, generator=g) * 0.2
    dP = q(torch.randn(B, C, H // 2, W // 2, generator=g), dt, ops)
    dS = q(torch.randn(B, C, H, W, generator=g), dt, ops)
    a = F.relu(y.double() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)).requires_grad_(True)
    (F.max_pool2d(a, 2) * dP.double()).sum().backward()
    ref = a.grad + (dS.double() if with_skip else 0)
    ya = to_act(y, dt, ops).with_transform(sc.cuda(), sh.cuda(), 0)
    dA = ops.new_act(B, H, W, C, dt, "cuda")
    sk = to_act(dS, dt, ops, ld=2 * C, coff=C) if with_skip else None
    # fused BatchNorm-backward statistics (phase 1) of the same layer, checked against the separate reduce kernel
    from cmunet_amd import _lib
    mean, invstd = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    ws = ws_bytes(_lib.lib().cmu_bn_bwd_ws_bytes(C))
    ops.maxpool_bwd(to_act(dP, dt, ops), sk, ya, dA, mean.cuda(), invstd.cuda(), ws)
    coef_f, coef_r = torch.empty(2, C, device="cuda"), torch.empty(2, C, device="cuda")
    dg_f, db_f, dg_r, db_r = (torch.empty(C, device="cuda") for _ in range(4))
    ops.bn_bwd_finalize(ws, B * H * W, dg_f, db_f, coef_f)
    ops.bn_bwd_reduce(dA, ya, mean.cuda(), invstd.cuda(), dg_r, db_r, coef_r, ws_bytes(_lib.lib().cmu_bn_bwd_ws_bytes(C)))
    check(dg_f.cpu(), dg_r.cpu(), 1e-5, "fused dgamma"); check(db_f.cpu(), db_r.cpu(), 1e-5, "fused dbeta")
    check(coef_f.cpu(), coef_r.cpu(), 1e-5, "fused coef")
    # positions whose activation is 0 (all-negative windows) may differ in which zero got the gradient;
    # that gradient is killed by the ReLU gate downstream, so compare after gating with a > 0
    gate = (a.detach() > 0).double()
    check(from_act(dA).double() * gate, ref * gate, TOL[dt], "maxpool bwd (gated)")


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("skips", [0, 1, 2])
def test_maxpool_bwd_apply_never_stores_the_pooled_gradient(ops, dt, skips):
    """Round 3: ``maxpool_bwd(dA=None)`` (BatchNorm-backward sums only) + ``bn_bwd_finalize`` + ``maxpool_bwd_apply`` against the stored
    form ``maxpool_bwd`` + ``bn_bwd_finalize`` + ``bn_bwd_apply``: the same sums bit for bit (they are taken on the gradient rounded as it
    would have been stored) and the same dY bit for bit, with zero, one and two skip gradients, strided operands."""
    from cmunet_amd import _lib
    B, C, H, W = 2, 32, 8, 12
    g = torch.Generator().manual_seed(25 + skips)
    y = q(torch.randn(B, C, H, W, generator=g), dt, ops)
    sc, sh = torch.randn(C, generator=g), torch.randn(C, generator=g) * 0.2
    dP = to_act(q(torch.randn(B, C, H // 2, W // 2, generator=g), dt, ops), dt, ops)
    s1 = to_act(q(torch.randn(B, C, H, W, generator=g), dt, ops), dt, ops, ld=2 * C, coff=C) if skips >= 1 else None
    s2 = to_act(q(torch.randn(B, C, H, W, generator=g), dt, ops), dt, ops, ld=3 * C, coff=2 * C) if skips >= 2 else None
    ya = to_act(y, dt, ops).with_transform(sc.cuda(), sh.cuda(), 0)
    mean, invstd = (torch.randn(C, generator=g) * 0.1).cuda(), (torch.rand(C, generator=g) + 0.5).cuda()
    n = _lib.lib().cmu_bn_bwd_ws_bytes(C)
    ws_a, ws_b = ws_bytes(n), ws_bytes(n)
    # stored form
    dA = ops.new_act(B, H, W, C, dt, "cuda")
    ops.maxpool_bwd(dP, s1, ya, dA, mean, invstd, ws_a, dSkip2=s2)
    coef_a, dg_a, db_a = torch.empty(2, C, device="cuda"), torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    ops.bn_bwd_finalize(ws_a, B * H * W, dg_a, db_a, coef_a)
    dY_a = ops.new_act(B, H, W, C, dt, "cuda")
    ops.bn_bwd_apply(dA, ya, mean, invstd, coef_a, dY_a)
    # never-stored form
    ops.maxpool_bwd(dP, s1, ya, None, mean, invstd, ws_b, dSkip2=s2)
    coef_b, dg_b, db_b = torch.empty(2, C, device="cuda"), torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    ops.bn_bwd_finalize(ws_b, B * H * W, dg_b, db_b, coef_b)
    assert torch.equal(coef_a, coef_b) and torch.equal(dg_a, dg_b) and torch.equal(db_a, db_b)
    dY_b = ops.Act(torch.full((B, H, W, 2 * C), 7.0, device="cuda").to(ops.TORCH_DT[ops.dt_code(dt)]), C, C)     # strided destination
    ops.maxpool_bwd_apply(dP, s1, ya, mean, invstd, coef_b, dY_b, dSkip2=s2)
    assert torch.equal(dY_b.buf[..., C:].contiguous().view(torch.uint8), dY_a.buf.view(torch.uint8))
    assert bool((dY_b.buf[..., :C] == 7.0).all())                                   # the other half of the buffer is untouched
    with pytest.raises(_lib.CmuError):
        ops.maxpool_bwd(dP, s1, ya, None)                                           # dA may only be missing in the sums-only form


@pytest.mark.parametrize("dt", DTS)
def test_maxpool_bwd_two_skip_gradients(ops, dt):
    """cmu_maxpool_bwd2: two gradients of the same skip tensor (CM_UNet's pixel and feature decoders hang on one encoder,
    cmunet.py:121-124 of the reference; autograd sums them) added in fp32 inside the pass, with different strides / channel offsets;
    the fused BatchNorm-backward sums see the summed gradient."""
    from cmunet_amd import _lib
    B, C, H, W = 2, 32, 8, 12
    g = torch.Generator().manual_seed(15)
    y = q(torch.randn(B, C, H, W, generator=g), dt, ops)
    sc, sh = torch.randn(C, generator=g), torch.randn(C, generator=g) * 0.2
    dP = q(torch.randn(B, C, H // 2, W // 2, generator=g), dt, ops)
    dS1 = q(torch.randn(B, C, H, W, generator=g), dt, ops)
    dS2 = q(torch.randn(B, C, H, W, generator=g), dt, ops)
    a = F.relu(y.double() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)).requires_grad_(True)
    (F.max_pool2d(a, 2) * dP.double()).sum().backward()
    ref = a.grad + dS1.double() + dS2.double()
    ya = to_act(y, dt, ops).with_transform(sc.cuda(), sh.cuda(), 0)
    dA = ops.new_act(B, H, W, C, dt, "cuda")
    mean, invstd = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    ws = ws_bytes(_lib.lib().cmu_bn_bwd_ws_bytes(C))
    ops.maxpool_bwd(to_act(dP, dt, ops), to_act(dS1, dt, ops, ld=2 * C, coff=C), ya, dA, mean.cuda(), invstd.cuda(), ws,
                    dSkip2=to_act(dS2, dt, ops, ld=3 * C, coff=2 * C))
    gate = (a.detach() > 0).double()
    check(from_act(dA).double() * gate, ref * gate, TOL[dt], "maxpool bwd, two skip gradients (gated)")
    coef_f, coef_r = torch.empty(2, C, device="cuda"), torch.empty(2, C, device="cuda")
    dg_f, db_f, dg_r, db_r = (torch.empty(C, device="cuda") for _ in range(4))
    ops.bn_bwd_finalize(ws, B * H * W, dg_f, db_f, coef_f)
    ops.bn_bwd_reduce(dA, ya, mean.cuda(), invstd.cuda(), dg_r, db_r, coef_r, ws_bytes(_lib.lib().cmu_bn_bwd_ws_bytes(C)))
    check(dg_f.cpu(), dg_r.cpu(), 1e-5, "fused dgamma"); check(db_f.cpu(), db_r.cpu(), 1e-5, "fused dbeta")


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("case", [(2, 8, 8, 32, 16, True), (1, 5, 9, 64, 32, False), (2, 16, 16, 128, 64, True),
                                  # Cin % 128 == 0 and Cout % 64 == 0 -> wide kernel for the 16-bit dtypes (conv_wgrad2.inc)
                                  (1, 5, 9, 128, 64, True), (2, 7, 33, 256, 128, False), (3, 4, 16, 128, 192, True),
                                  (1, 1, 2, 256, 64, True)])
def test_convT2x2_wgrad(ops, dt, case):
    from cmunet_amd import _lib
    B, H, W, Cin, Cout, tf = case
    g = torch.Generator().manual_seed(7)
    x = q(torch.randn(B, Cin, H, W, generator=g), dt, ops)
    dout = q(torch.randn(B, Cout, 2 * H, 2 * W, generator=g), dt, ops)
    xa = to_act(x, dt, ops)
    ref_in = x.double()
    if tf:
        sc, sh = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
        xa = xa.with_transform(sc.cuda(), sh.cuda(), 0)
        ref_in = q(F.relu(x.double() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)).float(), dt, ops).double()
    w = torch.zeros(Cin, Cout, 2, 2, dtype=torch.float64, requires_grad=True)
    bias = torch.zeros(Cout, dtype=torch.float64, requires_grad=True)
    (F.conv_transpose2d(ref_in, w, bias, stride=2) * dout.double()).sum().backward()
    dW, db = torch.empty(Cin, Cout, 2, 2, device="cuda"), torch.empty(Cout, device="cuda")
    ws = ws_bytes(_lib.lib().cmu_convT2x2_wgrad_ws_bytes(B, H, W, Cin, Cout, ops.dt_code(dt)))
    ops.convT2x2_wgrad(xa, to_act(dout, dt, ops, ld=2 * Cout, coff=0), dW, db, ws)
    check(dW.cpu(), w.grad, 1e-4 if dt == "f32" else 2e-3, "dW convT")
    check(db.cpu(), bias.grad, 1e-4, "dbias convT")


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("seed", range(12))
def test_convT2x2_wgrad_random_shapes_of_the_wide_kernels(ops, dt, seed):
    """Seeded random shapes with Cin % 128 == 0 and Cout % 64 == 0 (16-bit: conv_wgradT2_kernel; fp32: conv_wgradT2f_kernel): 1-3 images
    of 1-23 low-res pixels per side, channel-sliced dOut, with and without the pending transform -- weight and bias gradient vs float64."""
    from cmunet_amd import _lib
    rs = np.random.RandomState(3000 + seed)
    B, H, W = int(rs.randint(1, 4)), int(rs.randint(1, 24)), int(rs.randint(1, 24))
    Cin, Cout = 128 * int(rs.randint(1, 3)), 64 * int(rs.randint(1, 4))
    if seed >= 4:
        Cin = 256 * int(rs.randint(1, 3))       # (round 4: Cin % 256 == 0 takes the 256-channel X block of conv_wgradT2_kernel)
    tf = bool(rs.randint(0, 2))
    ed = 16 * int(rs.randint(0, 3))
    g = torch.Generator().manual_seed(seed)
    x = q(torch.randn(B, Cin, H, W, generator=g), dt, ops)
    dout = q(torch.randn(B, Cout, 2 * H, 2 * W, generator=g), dt, ops)
    xa = to_act(x, dt, ops)
    ref_in = x.double()
    if tf:
        sc, sh = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
        xa = xa.with_transform(sc.cuda(), sh.cuda(), 0)
        ref_in = q(F.relu(x.double() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)).float(), dt, ops).double()
    w = torch.zeros(Cin, Cout, 2, 2, dtype=torch.float64, requires_grad=True)
    bias = torch.zeros(Cout, dtype=torch.float64, requires_grad=True)
    (F.conv_transpose2d(ref_in, w, bias, stride=2) * dout.double()).sum().backward()
    dW, db = torch.empty(Cin, Cout, 2, 2, device="cuda"), torch.empty(Cout, device="cuda")
    ws = ws_bytes(_lib.lib().cmu_convT2x2_wgrad_ws_bytes(B, H, W, Cin, Cout, ops.dt_code(dt)))
    ops.convT2x2_wgrad(xa, to_act(dout, dt, ops, ld=Cout + ed, coff=ed), dW, db, ws)
    what = f"{(B, H, W, Cin, Cout, tf, ed)}"
    check(dW.cpu(), w.grad, 1e-4 if dt == "f32" else 2e-3, "dW convT " + what)
    check(db.cpu(), bias.grad, 1e-4, "dbias convT " + what)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("C", [16, 64])
def test_conv1x1_head_bwd(ops, dt, C):
    from cmunet_amd import _lib
    B, H, W, K = 2, 10, 13, 2
    g = torch.Generator().manual_seed(9)
    x = q(torch.randn(B, C, H, W, generator=g), dt, ops)
    sc, sh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    w, b = torch.randn(K, C, generator=g) / C ** 0.5, torch.randn(K, generator=g)
    dl = torch.randn(B, K, H, W, generator=g)
    a = F.relu(x.double() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)).requires_grad_(True)
    wd, bd = w.double().requires_grad_(True), b.double().requires_grad_(True)
    (F.conv2d(a, wd.view(K, C, 1, 1), bd) * dl.double()).sum().backward()
    xa = to_act(x, dt, ops).with_transform(sc.cuda(), sh.cuda(), 0)
    dX = ops.new_act(B, H, W, C, dt, "cuda")
    dW, db = torch.empty(K, C, device="cuda"), torch.empty(K, device="cuda")
    ws = ws_bytes(_lib.lib().cmu_conv1x1_head_bwd_ws_bytes(B, H, W, C, K))
    mean, invstd = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    bws = ws_bytes(_lib.lib().cmu_bn_bwd_ws_bytes(C))
    ops.conv1x1_head_bwd(dl.cuda(), xa, w.cuda(), dX, dW, db, ws, mean.cuda(), invstd.cuda(), bws)
    coef_f, coef_r = torch.empty(2, C, device="cuda"), torch.empty(2, C, device="cuda")
    dg_f, db_f, dg_r, db_r = (torch.empty(C, device="cuda") for _ in range(4))
    ops.bn_bwd_finalize(bws, B * H * W, dg_f, db_f, coef_f)
    ops.bn_bwd_reduce(dX, xa, mean.cuda(), invstd.cuda(), dg_r, db_r, coef_r, ws_bytes(_lib.lib().cmu_bn_bwd_ws_bytes(C)))
    check(dg_f.cpu(), dg_r.cpu(), 1e-5, "fused dgamma (head)"); check(coef_f.cpu(), coef_r.cpu(), 1e-5, "fused coef (head)")
    check(from_act(dX), a.grad, TOL[dt], "head dX")
    check(dW.cpu(), wd.grad, 1e-4, "head dW")
    check(db.cpu(), bd.grad, 1e-4, "head dbias")
    # the fused form: the rank-K input gradient is never stored -- head backward leaves sums + parameter gradients (dX None),
    # cmu_conv1x1_head_bn_apply writes the producing layer's dY straight from dlogits: the same bits as stored dX + bn_bwd_apply
    dY_ref = ops.new_act(B, H, W, C, dt, "cuda")
    ops.bn_bwd_apply(dX, xa, mean.cuda(), invstd.cuda(), coef_f, dY_ref)
    dW2, db2 = torch.empty(K, C, device="cuda"), torch.empty(K, device="cuda")
    bws2 = ws_bytes(_lib.lib().cmu_bn_bwd_ws_bytes(C))
    ops.conv1x1_head_bwd(dl.cuda(), xa, w.cuda(), None, dW2, db2, ws, mean.cuda(), invstd.cuda(), bws2)
    coef2, dg2, db2b = torch.empty(2, C, device="cuda"), torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    ops.bn_bwd_finalize(bws2, B * H * W, dg2, db2b, coef2)
    assert torch.equal(coef2, coef_f) and torch.equal(dg2, dg_f) and torch.equal(dW2, dW) and torch.equal(db2, db)
    dY = ops.new_act(B, H, W, C, dt, "cuda")
    ops.conv1x1_head_bn_apply(dl.cuda(), xa, w.cuda(), mean.cuda(), invstd.cuda(), coef2, dY)
    assert torch.equal(dY.buf.view(torch.uint8), dY_ref.buf.view(torch.uint8)), "fused head + BN apply differs from the two passes"


@pytest.mark.parametrize("dt", ["f16", "bf16", "f32"])
@pytest.mark.parametrize("K", [1, 3])
def test_conv1x1_head_bn_apply_other_class_counts(ops, dt, K):
    """The fused head backward + BatchNorm apply for class counts other than two (the generic kernel): bit-identical to the stored form."""
    from cmunet_amd import _lib
    B, H, W, C = 3, 12, 20, 64
    g = torch.Generator().manual_seed(19 + K)
    x = q(torch.randn(B, C, H, W, generator=g), dt, ops)
    sc, sh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    w = torch.randn(K, C, generator=g) / C ** 0.5
    dl = torch.randn(B, K, H, W, generator=g)
    mean, invstd = (torch.randn(C, generator=g) * 0.1).cuda(), (torch.rand(C, generator=g) + 0.5).cuda()
    xa = to_act(x, dt, ops).with_transform(sc.cuda(), sh.cuda(), 0)
    outs = []
    for fused in (False, True):
        dX = None if fused else ops.new_act(B, H, W, C, dt, "cuda")
        dW, db = torch.empty(K, C, device="cuda"), torch.empty(K, device="cuda")
        ws, bws = ws_bytes(_lib.lib().cmu_conv1x1_head_bwd_ws_bytes(B, H, W, C, K)), ws_bytes(_lib.lib().cmu_bn_bwd_ws_bytes(C))
        ops.conv1x1_head_bwd(dl.cuda(), xa, w.cuda(), dX, dW, db, ws, mean, invstd, bws)
        coef, dg, dbt = torch.empty(2, C, device="cuda"), torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
        ops.bn_bwd_finalize(bws, B * H * W, dg, dbt, coef)
        dY = ops.new_act(B, H, W, C, dt, "cuda")
        if fused:
            ops.conv1x1_head_bn_apply(dl.cuda(), xa, w.cuda(), mean, invstd, coef, dY)
        else:
            ops.bn_bwd_apply(dX, xa, mean, invstd, coef, dY)
        outs.append((dY.buf.clone(), dW, db, coef))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a.view(torch.uint8), b.view(torch.uint8))


def test_masked_mse(ops):
    from cmunet_amd import _lib
    from oracle import cmunet as OC
    B, K, H, W = 3, 2, 20, 33
    g = torch.Generator().manual_seed(11)
    logits = torch.randn(B, K, H, W, generator=g)
    img = torch.randn(B, H, W, generator=g) * 2 + 1
    mask = (torch.rand(B, H, W, generator=g) > 0.4).to(torch.uint8)
    lo = logits.clone().requires_grad_(True)
    ref = OC.masked_mse(lo[:, 1], img, mask)
    (ref * 3.0).backward()
    loss = torch.empty(1, device="cuda")
    dl = torch.empty(B, K, H, W, device="cuda")
    ws = ws_bytes(_lib.lib().cmu_masked_mse_ws_bytes(B, H))
    ops.masked_mse_fwd_bwd(logits.cuda(), 1, img.cuda(), mask.cuda(), loss, dl, 3.0, ws)
    check(loss.cpu(), ref.detach().view(1), 1e-5, "masked mse loss")
    check(dl.cpu(), lo.grad, 1e-5, "masked mse grad")


@pytest.mark.parametrize("case", ["all_masked", "none_masked", "constant_rows", "one_pixel"])
def test_masked_mse_edge_cases(ops, case):
    """Edges of cmunet_head.py:62-70: every pixel masked; no pixel masked (0/0: the reference's loss is NaN -- so is ours, a
    silent 0 would hide a broken mask); constant image rows (variance 0: the target is (x - mean) / sqrt(0 + 1e-6) = 0); one
    masked pixel."""
    from cmunet_amd import _lib
    from oracle import cmunet as OC
    B, K, H, W = 2, 2, 16, 32
    g = torch.Generator().manual_seed(12)
    logits = torch.randn(B, K, H, W, generator=g)
    img = torch.randn(B, H, W, generator=g)
    mask = (torch.rand(B, H, W, generator=g) > 0.5).to(torch.uint8)
    if case == "all_masked":
        mask[:] = 1
    elif case == "none_masked":
        mask[:] = 0
    elif case == "constant_rows":
        img[:, ::2, :] = 3.25
    else:
        mask[:] = 0
        mask[1, 5, 7] = 1
    lo = logits.clone().requires_grad_(True)
    ref = OC.masked_mse(lo[:, 1], img, mask)
    ref.backward()
    loss = torch.empty(1, device="cuda")
    dl = torch.empty(B, K, H, W, device="cuda")
    ws = ws_bytes(_lib.lib().cmu_masked_mse_ws_bytes(B, H))
    ops.masked_mse_fwd_bwd(logits.cuda(), 1, img.cuda(), mask.cuda(), loss, dl, 1.0, ws)
    if case == "none_masked":
        assert torch.isnan(ref) and torch.isnan(loss.cpu()).all()
        return
    check(loss.cpu(), ref.detach().view(1), 1e-5, "masked mse loss")
    check(dl.cpu(), lo.grad, 1e-5, "masked mse grad")


def test_softmax_ce_dice(ops, golden_dir):
    import numpy as np
    from cmunet_amd import _lib
    d = np.load(golden_dir + "/losses.npz")
    logits, y1h = torch.from_numpy(d["logits"]), torch.from_numpy(d["y1h"])
    B, K, H, W = logits.shape
    out = torch.empty(6, device="cuda")
    dl = torch.empty(B, K, H, W, device="cuda")
    ws = ws_bytes(_lib.lib().cmu_softmax_ce_dice_ws_bytes(B, H, W))
    ops.softmax_ce_dice_fwd_bwd(logits.cuda(), y1h.cuda(), out, dl, 1.0, ws)
    o = out.cpu()
    assert abs(o[0].item() - float(d["ce"])) < 1e-5
    assert abs(o[1].item() - float(d["dice"])) < 1e-6          # thresholded counters: exact up to fp32 of the ratio
    assert abs(o[2].item() - float(d["iou"])) < 1e-6
    check(dl.cpu(), torch.from_numpy(d["dlogits"]), 1e-5, "dlogits (CE only: Dice has no gradient, A-4)")


@pytest.mark.parametrize("case", ["no_foreground", "all_foreground", "prediction_empty"])
def test_softmax_ce_dice_edge_cases(ops, case):
    """Degenerate masks of metrics.py:135-220: no foreground anywhere (Dice loss 1 - eps/eps = 0, IoU eps/eps = 1), everything
    foreground, and a foreground target the prediction misses entirely (Dice loss ~1, IoU ~0)."""
    from cmunet_amd import _lib
    from oracle import losses as OL
    B, K, H, W = 2, 2, 12, 20
    g = torch.Generator().manual_seed(19)
    logits = torch.randn(B, K, H, W, generator=g) * 0.3
    fg = torch.rand(B, H, W, generator=g) > 0.7
    if case == "no_foreground":
        fg[:] = False
        logits[:, 0] += 5.0                       # predicts background everywhere
    elif case == "all_foreground":
        fg[:] = True
        logits[:, 1] += 5.0
    else:
        logits[:, 0] += 5.0                       # target has foreground, prediction has none
    y1h = torch.stack([~fg, fg], 1).double()
    lo = logits.clone().requires_grad_(True)
    ce = OL.cross_entropy_prob(lo, y1h)
    ce.backward()
    out = torch.empty(6, device="cuda")
    dl = torch.empty(B, K, H, W, device="cuda")
    ws = ws_bytes(_lib.lib().cmu_softmax_ce_dice_ws_bytes(B, H, W))
    ops.softmax_ce_dice_fwd_bwd(logits.cuda(), y1h.cuda(), out, dl, 1.0, ws)
    o = out.cpu()
    assert abs(o[0].item() - float(ce.detach())) < 1e-5 * max(1.0, float(ce.detach()))
    assert abs(o[1].item() - float(OL.dice_loss(logits, y1h))) < 1e-6
    assert abs(o[2].item() - (1.0 - float(OL.iou_loss(logits, y1h)))) < 1e-6 or abs(o[2].item() - float(OL.iou_loss(logits, y1h))) < 1e-6
    check(dl.cpu(), lo.grad.float(), 1e-5, "dlogits")


@pytest.mark.parametrize("rank,world,B", [(0, 1, 8), (1, 4, 8), (0, 1, 1), (2, 3, 5)])
def test_infonce_inbatch(ops, rank, world, B):
    from oracle import cmunet as OC
    D = 256
    g = torch.Generator().manual_seed(13)
    pred = torch.randn(B, D, generator=g)
    keys = F.normalize(torch.randn(B * world, D, generator=g), dim=1)
    p = pred.clone().requires_grad_(True)
    ref = OC.infonce_inbatch(p, keys, 0.07, rank, 1.0)
    ref.backward()
    loss = torch.empty(1 + B, device="cuda")
    dp = torch.empty(B, D, device="cuda")
    ops.infonce_inbatch_fwd_bwd(pred.cuda(), keys.cuda(), loss, dp, rank, 0.07, 1.0)
    check(loss[:1].cpu(), ref.detach().view(1), 2e-5, "infonce loss")
    check(dp.cpu(), p.grad, 1e-4, "infonce dpred")


@pytest.mark.parametrize("shape", [(8, 128, 512), (32, 1024, 4096), (5, 72, 260), (40, 64, 1024)])
@pytest.mark.parametrize("gathered", [False, True])
def test_moco_infonce_enqueue(ops, gathered, shape):
    """(32, 1024, 4096) = BASELINE config 3 per GPU; (5, 72, 260): ragged tiles; (40, 64, 1024): more than one 32-row tile."""
    from cmunet_amd import _lib
    from oracle import moco as OM
    B, D, K = shape
    T = 0.2
    if K % (2 * B if gathered else B) != 0:
        K = (K // (2 * B)) * 2 * B                          # moco2_module.py:169: K % gathered batch == 0
    if K % 4 != 0:
        pytest.skip("K % 4")
    g = torch.Generator().manual_seed(17)
    q_raw, k_raw = torch.randn(B, D, generator=g), torch.randn(B, D, generator=g)
    queue = OM.init_queue(D, K, seed=3)
    ptr = torch.tensor([K - (2 * B if gathered else B)], dtype=torch.long)    # wraps to 0 after the enqueue
    qr = q_raw.clone().requires_grad_(True)
    logits, labels, k, _ = OM.logits_from_embeddings(qr, k_raw, queue, T)
    keys_all = torch.cat([k, F.normalize(torch.randn(B, D, generator=g), dim=1)]) if gathered else k
    ref_queue, ref_ptr = queue.clone(), ptr.clone()
    OM.dequeue_and_enqueue(keys_all, ref_queue, ref_ptr, K)
    ref = F.cross_entropy(logits, labels)
    ref.backward()
    qd, pd = queue.clone().cuda(), ptr.clone().cuda()
    loss, dq, kn = torch.empty(1, device="cuda"), torch.empty(B, D, device="cuda"), torch.empty(B, D, device="cuda")
    ws = ws_bytes(_lib.lib().cmu_moco_ws_bytes(B, D, K))
    ops.moco_infonce_enqueue(q_raw.cuda(), k_raw.cuda(), keys_all.cuda() if gathered else None, qd, pd, loss, dq, kn, T, ws)
    check(loss.cpu(), ref.detach().view(1), 2e-5, "moco loss")
    check(dq.cpu(), qr.grad, 1e-4, "moco dq")
    check(kn.cpu(), k, 1e-6, "normalised keys")
    check(qd.cpu(), ref_queue, 1e-6, "queue after enqueue")
    assert int(pd.item()) == int(ref_ptr.item()) == 0
    # bitwise reproducible (fixed-order sums across workgroups), and a second call works on the moved pointer
    qd2, pd2 = queue.clone().cuda(), ptr.clone().cuda()
    loss2, dq2 = torch.empty(1, device="cuda"), torch.empty(B, D, device="cuda")
    ops.moco_infonce_enqueue(q_raw.cuda(), k_raw.cuda(), keys_all.cuda() if gathered else None, qd2, pd2, loss2, dq2, None, T, ws)
    assert torch.equal(loss2, loss) and torch.equal(dq2, dq) and torch.equal(qd2, qd)
    ops.moco_infonce_enqueue(q_raw.cuda(), k_raw.cuda(), keys_all.cuda() if gathered else None, qd2, pd2, loss2, None, None, T, ws)
    assert int(pd2.item()) == (2 * B if gathered else B) % K and bool(torch.isfinite(loss2).all())


def test_l2norm_ema_adam(ops):
    g = torch.Generator().manual_seed(19)
    x = torch.randn(5, 300, generator=g)
    out = torch.empty(5, 300, device="cuda")
    ops.l2_normalize_rows(x.cuda(), out)
    check(out.cpu(), F.normalize(x, dim=1), 1e-6, "l2norm")
    n = 1003
    t, o = torch.randn(n, generator=g), torch.randn(n, generator=g)
    td = t.clone().cuda()
    ops.ema_update(td, o.cuda(), 0.996)
    check(td.cpu(), t * 0.996 + o * (1 - 0.996), 1e-6, "ema")
    for decoupled, wd in ((0, 0.0), (1, 0.05), (0, 1e-4)):
        p0 = torch.randn(n, generator=g)
        p = torch.nn.Parameter(p0.clone())
        opt = (torch.optim.AdamW if decoupled else torch.optim.Adam)([p], lr=1e-3, betas=(0.9, 0.95), weight_decay=wd)
        pd, m, v = p0.clone().cuda(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
        for step in range(1, 4):
            gr = torch.randn(n, generator=g)
            p.grad = gr.clone()
            opt.step()
            ops.adam_step(pd, gr.cuda(), m, v, None, 1e-3, 0.9, 0.95, 1e-8, wd, decoupled, step)
        check(pd.cpu(), p.detach(), 1e-5, f"adam decoupled={decoupled} wd={wd}")


def _bn_bwd_sums(dx_stored, y, scale, shift, mean, invstd):
    """sum(gate*dX), sum(gate*dX*xhat) per channel, fp64, gate evaluated as the kernels do (fp32 scale/shift)."""
    gate = (y.float() * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)) > 0
    dz = dx_stored.double() * gate
    xhat = (y.double() - mean.double().view(1, -1, 1, 1)) * invstd.double().view(1, -1, 1, 1)
    return dz.sum((0, 2, 3)), (dz * xhat).sum((0, 2, 3))


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("shape", [(2, 18, 21, 24, 40), (1, 18, 37, 128, 64), (2, 16, 32, 64, 128), (1, 33, 40, 64, 256),
                                   # whole-tile shapes: the persistent kernel (several channel blocks, long K, 64-channel form)
                                   (2, 32, 64, 256, 256), (1, 48, 32, 128, 512), (3, 16, 96, 64, 64)])
def test_conv3x3_dgrad_bn(ops, dt, shape):
    """Data gradient + fused BatchNorm-backward partial sums of the producer layer (first and wide-tile kernels)."""
    from cmunet_amd import _lib
    B, H, W, Cin, Cout = shape
    g = torch.Generator().manual_seed(31)
    dy = q(torch.randn(B, Cout, H, W, generator=g), dt, ops)
    w = q(torch.randn(Cout, Cin, 3, 3, generator=g) / (Cout * 9) ** 0.5, dt, ops)
    y = q(torch.randn(B, Cin, H, W, generator=g) * 1.5 + 0.2, dt, ops)          # raw output of the producer layer
    gamma, beta = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
    mean, invstd, scale, shift = bn_consts(y, gamma, beta)
    ya = to_act(y, dt, ops, ld=Cin + 16, coff=16).with_transform(scale.cuda(), shift.cuda(), 0)
    dx = ops.new_act(B, H, W, Cin, dt, "cuda")
    slab = ops.new_stats(B, H, W, Cin, "cuda")
    ops.conv3x3_dgrad_bn(to_act(dy, dt, ops), ops.pack_conv3x3(w.cuda(), dt, transpose_flip=True), dx, ya, mean.cuda(), invstd.cuda(), slab)
    ref = torch.nn.grad.conv2d_input((B, Cin, H, W), w.double(), dy.double(), padding=1)
    check(from_act(dx), ref, TOL[dt], "dgrad dx")
    s1, s2 = _bn_bwd_sums(from_act(dx), y, scale, shift, mean, invstd)
    got = slab.double().sum(0).cpu()
    check(got[0], s1, 2e-4, "sum gate*dX")
    check(got[1], s2, 2e-4, "sum gate*dX*xhat")
    # and the two-level finalisation of that slab
    dgamma, dbeta, coef = torch.empty(Cin, device="cuda"), torch.empty(Cin, device="cuda"), torch.empty(2, Cin, device="cuda")
    ops.bn_bwd_finalize_tiles(slab, B * H * W, dgamma, dbeta, coef, ws_bytes(_lib.lib().cmu_bn_finalize_ws_bytes(Cin)))
    check(dbeta.cpu(), s1, 2e-4, "dbeta")
    check(dgamma.cpu(), s2, 2e-4, "dgamma")
    check(coef.cpu()[0], s1 / (B * H * W), 2e-4, "coef mean(dz)")
    check(coef.cpu()[1], s2 / (B * H * W), 2e-4, "coef mean(dz*xhat)")


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("shape", [(2, 8, 8, 32, 16), (1, 5, 9, 64, 32), (2, 16, 16, 128, 64),
                                   # Cin % 256 == 0 and Cout a whole number of 128-byte steps -> GEMM kernel (conv_gemm.inc)
                                   (1, 5, 9, 256, 64), (2, 20, 17, 512, 128)])
def test_convT2x2_dgrad_bn(ops, dt, shape):
    B, H, W, Cin, Cout = shape
    g = torch.Generator().manual_seed(33)
    dout = q(torch.randn(B, Cout, 2 * H, 2 * W, generator=g), dt, ops)
    w = q(torch.randn(Cin, Cout, 2, 2, generator=g) / (Cout * 4) ** 0.5, dt, ops)
    y = q(torch.randn(B, Cin, H, W, generator=g) * 1.5 + 0.2, dt, ops)
    gamma, beta = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
    mean, invstd, scale, shift = bn_consts(y, gamma, beta)
    ya = to_act(y, dt, ops).with_transform(scale.cuda(), shift.cuda(), 0)
    dx = ops.new_act(B, H, W, Cin, dt, "cuda")
    slab = ops.new_stats(B, H, W, Cin, "cuda")
    ops.convT2x2_dgrad_bn(to_act(dout, dt, ops, ld=2 * Cout, coff=0), ops.pack_convT2x2(w.cuda(), dt, 1), dx, ya, mean.cuda(),
                          invstd.cuda(), slab)
    ref = F.conv2d(dout.double(), w.double(), stride=2)
    check(from_act(dx), ref, TOL[dt], "convT dgrad")
    s1, s2 = _bn_bwd_sums(from_act(dx), y, scale, shift, mean, invstd)
    got = slab.double().sum(0).cpu()
    check(got[0], s1, 2e-4, "sum gate*dX")
    check(got[1], s2, 2e-4, "sum gate*dX*xhat")


def test_fused_sgd_and_lamb_vs_reference_traces(golden_dir):
    """FusedSGD / FusedLAMB on the flat arena against three-step traces of torch.optim.SGD and of the reference's LAMB class
    (tests/golden/optim_traces.npz), including the clipped first step, excluded tensors, trust clipping."""
    import numpy as np
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from cmunet_amd.optim import FlatParams, FusedLAMB, FusedSGD
    d = np.load(f"{golden_dir}/optim_traces.npz")
    wds = [float(w) for w in d["wds"]]
    n = len(wds)
    shapes = [tuple(int(v) for v in row if v) for row in d["shapes"]]

    class Holder(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.ps = torch.nn.ParameterList([torch.nn.Parameter(torch.from_numpy(d[f"p0.{i}"]).clone().view(shapes[i])) for i in range(n)])

    def load_grads(flat, k):
        for i, name in enumerate(flat.names):
            flat.grad_views[name].copy_(torch.from_numpy(d[f"g{k}.{i}"]).view(shapes[i]))

    decay = lambda name, prm: wds[int(name.split(".")[-1])] != 0.0
    for tag, kw in (("a", dict(trust_clip=False, always_adapt=False)), ("b", dict(trust_clip=True, always_adapt=True))):
        flat = FlatParams(Holder().cuda())
        opt = FusedLAMB(flat, lr=2e-2, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.05, max_grad_norm=2.0, decay_filter=decay, **kw)
        for k in range(3):
            load_grads(flat, k)
            opt.step()
        for i, name in enumerate(flat.names):
            check(flat.views[name].cpu().flatten(), torch.from_numpy(d[f"lamb_{tag}.{i}"]).flatten(), 2e-5, f"lamb {tag} tensor {i}")
        assert opt.global_grad_norm > 0
    # SparK's annealed run: learning rate and weight decay set per iteration (FusedLAMB.set_lr_wd) to what the reference's own
    # lr_wd_annealing handed its LAMB class; tensors 1 and 3 are the no-decay group of the reference's get_param_groups
    scales = [float(v) for v in d["sched_wd_scale"]]
    flat = FlatParams(Holder().cuda())
    opt = FusedLAMB(flat, lr=1.0, betas=(0.9, 0.95), eps=1e-6, weight_decay=1.0, max_grad_norm=5.0,
                    decay_filter=lambda name, prm: scales[int(name.split(".")[-1])] != 0.0)
    from cmunet_amd.pretrain import spark_lr_wd
    pk, wd0, wde, wp_it, max_it = (float(v) for v in d["sched_args"])
    for k, (it, lr, cur_wd) in enumerate(d["sched"]):
        got = spark_lr_wd(pk, wd0, wde, int(it), wp_it, int(max_it))
        assert abs(got[0] - lr) < 1e-15 and abs(got[1] - cur_wd) < 1e-15
        opt.set_lr_wd(*got)
        load_grads(flat, k % 3)
        opt.step()
    for i, name in enumerate(flat.names):
        check(flat.views[name].cpu().flatten(), torch.from_numpy(d[f"lamb_sched.{i}"]).flatten(), 2e-5, f"annealed lamb tensor {i}")
    for tag, kw in (("a", dict(momentum=0.9, weight_decay=1e-4)), ("b", dict(momentum=0.9, weight_decay=1e-2, nesterov=True)),
                    ("c", dict(momentum=0.0, weight_decay=0.0))):
        flat = FlatParams(Holder().cuda())
        opt = FusedSGD(flat, lr=0.03, **kw)
        for k in range(3):
            load_grads(flat, k)
            opt.step()
        for i, name in enumerate(flat.names):
            check(flat.views[name].cpu().flatten(), torch.from_numpy(d[f"sgd_{tag}.{i}"]).flatten(), 2e-6, f"sgd {tag} tensor {i}")
