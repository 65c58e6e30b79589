"""Full-size checks (BASELINE configs[1]: bs 32, 512 x 512, bf16) through size-independent properties -- the oracle cannot
run these shapes in seconds, the properties can:
  * exact homogeneity: scaling an operand by 2 scales conv / data-gradient / weight-gradient outputs by exactly 2 (a power
    of two commutes with every fp32 accumulation and bf16 rounding in the kernels);
  * batch equivariance: permuting the images permutes the outputs bit for bit (tiles never mix images);
  * the statistics slab is the checksum of the output: its column sums equal the sums over the stored tensor;
  * bitwise reproducibility of a whole training step (no float atomics anywhere), the loss equal to a torch evaluation of
    the same logits, and a falling loss on a repeated batch.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

B, H, W = 32, 512, 512


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from cmunet_amd import ops as o
    return o


def _act(ops, t):
    return ops.Act(t, 0, t.shape[3])


# (Cin, Cout, H): first kernel (64->64), 64-channel wide variant (128->64), wide kernel (128->128 at the second level)
@pytest.mark.parametrize("shape", [(64, 64, 512), (128, 64, 512), (128, 128, 256)])
def test_conv3x3_homogeneity_permutation_checksum(ops, shape):
    Cin, Cout, S = shape
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(B, S, S, Cin, generator=g, device="cuda").to(torch.bfloat16)
    w = torch.randn(Cout, Cin, 3, 3, generator=g, device="cuda") / (3 * Cin ** 0.5)
    wp = ops.pack_conv3x3(w, "bf16")

    def conv(inp):
        y = ops.new_act(B, S, S, Cout, "bf16", "cuda")
        st = ops.new_stats(B, S, S, Cout, "cuda")
        ops.conv3x3_fwd(_act(ops, inp), wp, y, st)
        return y.buf, st

    y1, st1 = conv(x)
    y2, st2 = conv(x * 2)
    assert torch.equal(y2, y1 * 2), "conv(2x) != 2 conv(x)"
    assert torch.equal(st2[:, 0], st1[:, 0] * 2) and torch.equal(st2[:, 1], st1[:, 1] * 4), "statistics are not homogeneous"
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(2)).cuda()
    y3, _ = conv(x[perm].contiguous())
    assert torch.equal(y3, y1[perm]), "outputs depend on the position of an image in the batch"
    # checksum: slab sums (fp32 accumulators before rounding) against the stored bf16 tensor
    s = st1.double().sum(0)
    yd = y1.double()
    ref1, ref2 = yd.sum((0, 1, 2)), (yd * yd).sum((0, 1, 2))
    assert ((s[0] - ref1).abs() <= 2e-3 * yd.abs().sum((0, 1, 2))).all()
    assert ((s[1] - ref2).abs() <= 4e-3 * ref2).all()


@pytest.mark.parametrize("shape", [(64, 64, 512), (64, 128, 256), (128, 64, 512)])
def test_conv3x3_backward_homogeneity(ops, shape):
    from cmunet_amd import _lib
    Cin, Cout, S = shape
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(B, S, S, Cin, generator=g, device="cuda").to(torch.bfloat16)
    dy = torch.randn(B, S, S, Cout, generator=g, device="cuda").to(torch.bfloat16)
    w = torch.randn(Cout, Cin, 3, 3, generator=g, device="cuda") / (3 * Cin ** 0.5)
    wpf = ops.pack_conv3x3(w, "bf16", transpose_flip=True)
    ws = torch.empty(_lib.lib().cmu_conv3x3_wgrad_ws_bytes(B, S, S, Cin, Cout, ops.dt_code("bf16")), dtype=torch.uint8, device="cuda")

    def bwd(d):
        dx = ops.new_act(B, S, S, Cin, "bf16", "cuda")
        ops.conv3x3_fwd(_act(ops, d), wpf, dx, None)
        dW = torch.empty(Cout, Cin, 3, 3, device="cuda")
        ops.conv3x3_wgrad(_act(ops, x), _act(ops, d), dW, ws)
        return dx.buf, dW

    dx1, dW1 = bwd(dy)
    dx2, dW2 = bwd(dy * 2)
    assert torch.equal(dx2, dx1 * 2), "dgrad(2 dY) != 2 dgrad(dY)"
    assert torch.equal(dW2, dW1 * 2), "wgrad(2 dY) != 2 wgrad(dY)"
    dx3, dW3 = bwd(dy)
    assert torch.equal(dx3, dx1) and torch.equal(dW3, dW1), "backward kernels are not bitwise reproducible"


def test_training_step_reproducible_and_consistent():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from cmunet_amd import model as M
    from cmunet_amd.pretrain import MaskedReconPretrainer, random_patch_mask_device
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(5)
    img = torch.randn(B, H, W, generator=g, device=dev)
    mask = random_patch_mask_device(B, H, W, 16, 0.6, g, dev)

    def run(steps):
        torch.manual_seed(0)
        net = M.UNet(out_classes=2, dtype="bf16").to(dev)
        tr = MaskedReconPretrainer(net, lr=1.5e-4 * B / 256.0, betas=(0.9, 0.95), weight_decay=0.05)
        losses = [float(tr.step(img, mask)) for _ in range(steps)]
        return losses, tr.flat.grad.clone(), tr.flat.arena.clone(), tr

    l1, g1, p1, tr = run(3)
    l2, g2, p2, _ = run(3)
    assert l1 == l2 and torch.equal(g1, g2) and torch.equal(p1, p2), "two identical runs differ"
    assert l1[2] < l1[0], l1
    # forward/backward is a pure function of (parameters, batch), and its loss equals the oracle's masked MSE
    # (cmunet_head.py:62-70, oracle/cmunet.py) evaluated by torch on the same logits
    from oracle import cmunet as OC
    tr.forward_backward(img, mask)
    loss_again = float(tr.loss)
    tr.forward_backward(img, mask)
    assert float(tr.loss) == loss_again
    logits, _ = tr.engine.unet_forward(tr.sd, img, True, mask, mask_per_sample=False)
    ref = float(OC.masked_mse(logits[:, 1].float().cpu(), img.cpu(), mask.cpu()))
    # (the forward above moved the BatchNorm running statistics, not the batch-statistics output: same logits)
    assert abs(loss_again - ref) <= 2e-4 * max(1.0, abs(ref)), (loss_again, ref)
    assert torch.isfinite(g1).all() and float(g1.abs().max()) > 0
