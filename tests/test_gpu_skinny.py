"""Weight-streaming GEMMs of the projector / predictor necks (csrc/skinny.hip) against torch float64:
y = x.w^T + b, dx = dy.w, dw = dy^T.x, db = sum dy for <= 32 rows.  fp32 MFMA products with fp32 accumulation: tolerance
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
                                        (17, 4096, 33, True), (32, 40, 128, False)])
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


def test_neck_linear_autograd_matches_library_gemm(ops):
    """The necks' Linear through the skinny kernels (<= 32 rows) and through the library GEMM (more rows) agree with
    torch.nn.functional.linear in value and in all three gradients."""
    from cmunet_amd.cmunet import neck_linear
    torch.manual_seed(3)
    fc = torch.nn.Linear(3136, 96, bias=True).cuda()
    for rows in (8, 32, 48):
        x = torch.randn(rows, 3136, device="cuda", requires_grad=True)
        go = torch.randn(rows, 96, device="cuda")
        fc.zero_grad()
        y = neck_linear(fc, x)
        y.backward(go)
        got = (y.detach().clone(), x.grad.clone(), fc.weight.grad.clone(), fc.bias.grad.clone())
        x2 = x.detach().clone().requires_grad_(True)
        fc.zero_grad()
        y2 = torch.nn.functional.linear(x2, fc.weight, fc.bias)
        y2.backward(go)
        for a, b_, what in zip(got, (y2.detach(), x2.grad, fc.weight.grad, fc.bias.grad), ("y", "dx", "dw", "db")):
            assert (a - b_).abs().max().item() <= 2e-5 * max(1.0, b_.abs().max().item()), (rows, what)


def test_skinny_rejects_what_it_cannot_do(ops):
    from cmunet_amd._lib import CmuError
    with pytest.raises(CmuError):
        ops.skinny_gemm_fwd(torch.randn(33, 64, device="cuda"), torch.randn(8, 64, device="cuda"))
    with pytest.raises(CmuError):
        ops.skinny_gemm_fwd(torch.randn(4, 20, device="cuda"), torch.randn(8, 20, device="cuda"))
