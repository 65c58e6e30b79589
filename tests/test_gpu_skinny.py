"""Weight-streaming GEMMs of the projector / predictor necks (csrc/skinny.hip) against torch float64:
y = x.w^T + b, dx = dy.w, dw = dy^T.x, db = sum dy for <= 256 rows (groups of 32).  fp32 MFMA products with fp32 accumulation: tolerance
1e-5 relative to the result's scale times sqrt(K) growth (written below)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from cmunet_amd import ops as o
    return o


def _close(got, ref, K, what):
    err = (got.double().cpu() - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= 2e-6 * scale * max(1.0, K ** 0.5 / 8), f"{what}: {err:.3e} vs scale {scale:.3e}"


@pytest.mark.parametrize("M,K,N,bias", [(32, 50176, 1536, True), (32, 1536, 256, False), (5, 104, 70, True), (1, 8, 1, False),
                                        (17, 4096, 33, True), (32, 40, 128, False),
                                        # round 4: more than one 32-row group (the reference's batch sizes: 64 and 256 per GPU)
                                        (33, 3136, 96, True), (64, 50176, 1536, True), (256, 12544, 1536, False), (256, 1536, 256, True),
                                        (100, 104, 70, True), (256, 4096, 1024, False)])
def test_skinny_gemm_three_products(ops, M, K, N, bias):
    g = torch.Generator().manual_seed(M * 7 + N)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g) if bias else None
    dy = torch.randn(M, N, generator=g)
    y = ops.skinny_gemm_fwd(x.cuda(), w.cuda(), None if b is None else b.cuda())
    _close(y, x.double() @ w.double().t() + (0 if b is None else b.double()), K, "y")
    dx = ops.skinny_gemm_dgrad(dy.cuda(), w.cuda())
    _close(dx, dy.double() @ w.double(), N, "dx")
    dw, db = ops.skinny_gemm_wgrad(dy.cuda(), x.cuda(), with_bias=bias)
    _close(dw, dy.double().t() @ x.double(), M, "dw")
    if bias:
        _close(db, dy.double().sum(0), M, "db")
    y2 = ops.skinny_gemm_fwd(x.cuda(), w.cuda(), None if b is None else b.cuda())
    assert torch.equal(y, y2)                                   # fixed-order split-K: bitwise reproducible


def test_neck_linear_autograd_vs_float64(ops):
    """The necks' Linear through the skinny kernels as an autograd node, one to eight 32-row groups: value and all three
    gradients against float64 products on the CPU (round 4: there is no library GEMM in the product to compare with or to fall
    back to -- more than ops.SKINNY_MAX_ROWS rows, or a K that is not a multiple of 8, raises)."""
    from cmunet_amd.cmunet import neck_linear
    torch.manual_seed(3)
    fc = torch.nn.Linear(3136, 96, bias=True).cuda()
    w64, b64 = fc.weight.detach().double().cpu(), fc.bias.detach().double().cpu()
    for rows in (8, 32, 33, 48, 64, 256):
        x = torch.randn(rows, 3136, device="cuda", requires_grad=True)
        go = torch.randn(rows, 96, device="cuda")
        fc.zero_grad()
        y = neck_linear(fc, x)
        y.backward(go)
        xd, gd = x.detach().double().cpu(), go.double().cpu()
        refs = (xd @ w64.t() + b64, gd @ w64, gd.t() @ xd, gd.sum(0))
        for a, b_, what in zip((y.detach(), x.grad, fc.weight.grad, fc.bias.grad), refs, ("y", "dx", "dw", "db")):
            assert (a.double().cpu() - b_).abs().max().item() <= 2e-5 * max(1.0, b_.abs().max().item()), (rows, what)
    with pytest.raises(RuntimeError, match="no library"):
        neck_linear(fc, torch.randn(ops.SKINNY_MAX_ROWS + 1, 3136, device="cuda"))
    with pytest.raises(RuntimeError, match="no library"):
        neck_linear(torch.nn.Linear(20, 8).cuda(), torch.randn(4, 20, device="cuda"))


def test_queue_logits_rows_beyond_one_group_vs_float64(ops):
    """Moco_v2.forward's l_neg = q @ queue (moco2_module.py:262) at the reference's batch size (256 rows per GPU) on the skinny
    kernels, value and dq against float64; more rows raise (no einsum behind it)."""
    from cmunet_amd.moco import queue_logits
    g = torch.Generator().manual_seed(9)
    queue = torch.nn.functional.normalize(torch.randn(128, 4096, generator=g), dim=0).cuda()
    for rows in (32, 40, 256):
        q = torch.nn.functional.normalize(torch.randn(rows, 128, generator=g), dim=1).cuda().requires_grad_(True)
        go = torch.randn(rows, 4096, generator=g).cuda()
        l = queue_logits(q, queue)
        l.backward(go)
        ref = q.detach().double().cpu() @ queue.double().cpu()
        refd = go.double().cpu() @ queue.double().cpu().t()
        assert (l.double().cpu() - ref).abs().max().item() <= 2e-6 and (q.grad.double().cpu() - refd).abs().max().item() <= 2e-5 * refd.abs().max().item()
    with pytest.raises(RuntimeError, match="no library"):
        queue_logits(torch.randn(257, 128, device="cuda"), queue)


def test_skinny_rejects_what_it_cannot_do(ops):
    from cmunet_amd._lib import CmuError
    with pytest.raises(CmuError):
        ops.skinny_gemm_fwd(torch.randn(257, 64, device="cuda"), torch.randn(8, 64, device="cuda"))
    with pytest.raises(CmuError):
        ops.skinny_gemm_fwd(torch.randn(4, 20, device="cuda"), torch.randn(8, 20, device="cuda"))


@pytest.mark.parametrize("cdt", ["f16", "bf16"])
@pytest.mark.parametrize("M,K,N,bias", [(32, 50176, 1536, True), (32, 1536, 256, False), (5, 112, 70, True), (1, 16, 1, False),
                                        (17, 4096, 33, True), (32, 48, 128, False),
                                        (33, 3136, 96, True), (64, 50176, 1536, True), (256, 12544, 1536, False), (100, 112, 70, True)])
def test_skinny16_gemm_three_products(ops, M, K, N, bias, cdt):
    """16-bit-operand variants (csrc/necks.hip; the AMP arithmetic of cmunet_config.py:76-78): the operands are rounded to
    f16 / bf16 in registers and multiplied into fp32 -- against float64 products of the SAME rounded operands the error is
    accumulation order only."""
    tdt = ops.TORCH_DT[ops.dt_code(cdt)]
    g = torch.Generator().manual_seed(M * 7 + N)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g) if bias else None
    dy = torch.randn(M, N, generator=g)
    xq, wq, dq = x.to(tdt).double(), w.to(tdt).double(), dy.to(tdt).double()
    y = ops.skinny_gemm_fwd(x.cuda(), w.cuda(), None if b is None else b.cuda(), compute_dt=cdt)
    _close(y, xq @ wq.t() + (0 if b is None else b.double()), K, "y")
    dx = ops.skinny_gemm_dgrad(dy.cuda(), w.cuda(), compute_dt=cdt)
    _close(dx, dq @ wq, N, "dx")
    dw, db = ops.skinny_gemm_wgrad(dy.cuda(), x.cuda(), with_bias=bias, compute_dt=cdt)
    _close(dw, dq.t() @ xq, M, "dw")
    if bias:
        _close(db, dy.double().sum(0), M, "db")
    assert torch.equal(y, ops.skinny_gemm_fwd(x.cuda(), w.cuda(), None if b is None else b.cuda(), compute_dt=cdt))
    # and within the storage type's rounding of the exact product
    exact = x.double() @ w.double().t() + (0 if b is None else b.double())
    tol = {"f16": 2e-3, "bf16": 1.6e-2}[cdt]
    assert (y.double().cpu() - exact).abs().max().item() <= tol * max(1.0, exact.abs().max().item())


@pytest.mark.parametrize("M,N,relu,affine", [(32, 1536, True, True), (4, 96, True, True), (7, 33, False, True), (8, 64, False, False),
                                             (64, 1536, True, True)])
def test_bn1d_relu_kernels_vs_torch_float64(ops, M, N, relu, affine):
    """BatchNorm1d (+ ReLU) of the necks (nonlinear_neck.py:58-60, 95-98; eps 1e-6) forward, running statistics and backward
    against torch.nn.functional.batch_norm in float64; the exchanged-sums form (SyncBN) against the local form."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, N, generator=g) * 2 + 0.5
    gamma = (1 + 0.3 * torch.randn(N, generator=g)) if affine else None
    beta = 0.2 * torch.randn(N, generator=g) if affine else None
    dy = torch.randn(M, N, generator=g)
    rm, rv = torch.zeros(N), torch.ones(N)
    xd = x.double().requires_grad_(True)
    gd = gamma.double().requires_grad_(True) if affine else None
    bd = beta.double().requires_grad_(True) if affine else None
    rmd, rvd = rm.double().clone(), rv.double().clone()
    ref = F.batch_norm(xd, rmd, rvd, gd, bd, True, 0.1, 1e-6)
    if relu:
        ref = F.relu(ref)
    ref.backward(dy.double())
    c = lambda t: None if t is None else t.cuda()
    rmc, rvc = rm.cuda(), rv.cuda()
    y, mean, invstd = ops.bn1d_relu_fwd(x.cuda(), c(gamma), c(beta), rmc, rvc, 0.1, 1e-6, True, relu)
    close = lambda a, b, tol=2e-5: (a.double().cpu() - b.detach()).abs().max().item() <= tol * max(1.0, b.detach().abs().max().item())
    assert close(y, ref) and close(rmc, rmd) and close(rvc, rvd)
    dx, dg, db = ops.bn1d_relu_bwd(dy.cuda(), x.cuda(), y, mean, invstd, c(gamma), relu, affine=affine)
    assert close(dx, xd.grad, 1e-4)
    if affine:
        assert close(dg, gd.grad, 1e-4) and close(db, bd.grad, 1e-4)
    # SyncBN form: the column sums of two half batches added up == the statistics of the whole batch
    if M % 2 == 0:
        h = M // 2
        s = ops.bn1d_colsums(x[:h].cuda()) + ops.bn1d_colsums(x[h:].cuda())
        rm2, rv2 = rm.cuda(), rv.cuda()
        y0, m0, i0 = ops.bn1d_relu_fwd(x[:h].cuda(), c(gamma), c(beta), rm2, rv2, 0.1, 1e-6, True, relu, s, M)
        assert close(y0, ref[:h], 1e-4) and close(rm2, rmd) and close(rv2, rvd, 1e-4)
        y1, _, _ = ops.bn1d_relu_fwd(x[h:].cuda(), c(gamma), c(beta), None, None, 0.1, 1e-6, True, relu, s, M)
        bs = ops.bn1d_bwd_colsums(dy[:h].cuda(), x[:h].cuda(), y0, m0, i0, relu) + ops.bn1d_bwd_colsums(dy[h:].cuda(), x[h:].cuda(), y1, m0, i0, relu)
        dx0, dg0, _ = ops.bn1d_relu_bwd(dy[:h].cuda(), x[:h].cuda(), y0, m0, i0, c(gamma), relu, bs, M, affine=affine)
        assert close(dx0, xd.grad[:h], 2e-4)
        if affine:
            assert close(bs[1], gd.grad, 2e-4)              # the rank-local dgamma halves add up to the whole
    # eval mode: running statistics
    ye, _, _ = ops.bn1d_relu_fwd(x.cuda(), c(gamma), c(beta), rmc, rvc, 0.1, 1e-6, False, relu)
    refe = F.batch_norm(x.double(), rmd, rvd, None if gamma is None else gamma.double(), None if beta is None else beta.double(), False, 0.1, 1e-6)
    assert close(ye, F.relu(refe) if relu else refe)


@pytest.mark.parametrize("dt", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("shape", [(2, 8, 8, 64, 16, True), (3, 4, 12, 1024, 256, True), (1, 2, 2, 32, 40, False), (2, 16, 16, 128, 96, True)])
def test_conv1x1_nchw_vs_conv2d(ops, dt, shape):
    """The target latent's Conv2d(C, C/4, 1) (cmunet.py:128-131) from the raw NHWC latent with its pending BatchNorm+ReLU into
    an NCHW fp32 tensor, against F.conv2d in float64 on the same quantised operands."""
    import torch.nn.functional as F
    B, H, W, K, N, tf = shape
    tdt = ops.TORCH_DT[ops.dt_code(dt)]
    g = torch.Generator().manual_seed(K + N)
    x = torch.randn(B, H, W, K, generator=g).to(tdt)
    w = torch.randn(N, K, 1, 1, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    sc = (1 + 0.2 * torch.randn(K, generator=g)) if tf else None
    sh = 0.3 * torch.randn(K, generator=g) if tf else None
    act = ops.Act(x.cuda().contiguous(), 0, K, None if sc is None else sc.cuda(), None if sh is None else sh.cuda(), 0)
    out = ops.conv1x1_nchw_fwd(act, w.cuda(), b.cuda())
    xin = x.double()
    if tf:
        xin = torch.relu(xin * sc.double() + sh.double()).to(tdt).double()      # the kernel rounds the activated operand to dt
    ref = F.conv2d(xin.permute(0, 3, 1, 2), w.to(tdt).double(), b.double())
    tol = {"f32": 2e-5, "f16": 2e-3, "bf16": 1.6e-2}[dt]
    err = (out.double().cpu() - ref).abs().max().item()
    assert out.shape == (B, N, H, W) and err <= tol * max(1.0, ref.abs().max().item()), err
