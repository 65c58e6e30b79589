"""GPU parity of the drop-in modules (cmunet_amd.model) against the golden fixtures produced by the
reference itself (tests/golden/*.npz, oracle/gen_golden.py) and against the CPU oracle on fresh inputs.

Tolerances:
  f32  : max |got - ref| <= 1e-3 * max |ref|  -- exact-fp32 MFMA chains, different summation order than ATen's
         CPU kernels, amplified by BatchNorm's 1/std and the depth of the network;
  f16 / bf16 : ||got - ref||_2 <= 2e-2 / 8e-2 * ||ref||_2  -- storage rounding of every activation / gradient
         (2^-11 / 2^-8 per tensor).  The element-wise max norm is NOT used for 16-bit storage: a pre-activation
         within rounding of zero flips its ReLU gate, which moves that one gradient element by O(1) while
         the tensor as a whole stays within rounding (a property of low-precision ReLU networks, also of
         the reference's own AMP run, cmunet_config.py:76-78).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
TOLS = {"f32": 1e-3, "f16": 2e-2, "bf16": 8e-2}
DTS = ["f32", "f16", "bf16"]


@pytest.fixture(scope="module")
def M():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from cmunet_amd import model
    return model


def load_fx(golden_dir, name):
    d = np.load(f"{golden_dir}/{name}.npz", allow_pickle=False)
    return {k: torch.from_numpy(np.asarray(d[k])) if d[k].dtype.kind in "fiu" else d[k] for k in d.files}


def load_sd(module, fx, prefix="sd."):
    sd = {k[len(prefix):]: v for k, v in fx.items() if k.startswith(prefix)}
    module.load_state_dict(sd, strict=True)


_MODE = {"norm": "max"}


def rel_err(got, ref):
    d = got.detach().double().cpu() - ref.double()
    if _MODE["norm"] == "l2":
        return d.norm().item() / max(ref.double().norm().item(), 1e-9)
    return d.abs().max().item() / max(ref.abs().max().item(), 1e-6)


@pytest.fixture(autouse=True)
def _norm_for_dtype(request):
    dt = request.node.callspec.params.get("dt", "f32") if hasattr(request.node, "callspec") else "f32"
    _MODE["norm"] = "max" if dt == "f32" else "l2"
    yield
    _MODE["norm"] = "max"


def assert_close(got, ref, tol, what, report):
    e = rel_err(got, ref)
    report.append((what, e))
    assert e <= tol, f"{what}: rel err {e:.3e} > {tol}"


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("tag,cin,cout", [("a", 1, 16), ("b", 16, 32), ("c", 8, 8)])
def test_double_conv_fixture(M, golden_dir, dt, tag, cin, cout):
    fx = load_fx(golden_dir, f"double_conv_{tag}")
    m = M.DoubleConv(cin, cout, dtype=dt).cuda().train()
    load_sd(m, fx)
    x = fx["x"].cuda().requires_grad_(True)
    y = m(x)
    (y * fx["go"].cuda()).sum().backward()
    rep, tol = [], TOLS[dt]
    assert_close(y, fx["y"], tol, "y", rep)
    if cin > 1:
        assert_close(x.grad, fx["dx"], tol, "dx", rep)
    for k, p in m.named_parameters():
        ref = fx["grad." + k]
        if k.endswith("0.bias") or k.endswith("3.bias"):        # conv bias before train-mode BN: exactly 0 here,
            assert p.grad.abs().max().item() == 0.0             # rounding noise (<1e-5 of the weights' grads) in the reference
            continue
        assert_close(p.grad, ref, tol * 3, "d" + k, rep)
    for k, v in m.state_dict().items():
        if "running" in k:
            assert_close(v, fx["after." + k], 1e-4 if dt == "f32" else tol, k, rep)
        if "num_batches" in k:
            assert int(v) == int(fx["after." + k])


@pytest.mark.parametrize("dt", DTS)
def test_down_block_fixture(M, golden_dir, dt):
    fx = load_fx(golden_dir, "down_block")
    m = M.DownBlock(16, 32, dtype=dt).cuda().train()
    load_sd(m, fx)
    x = fx["x"].cuda().requires_grad_(True)
    down, skip = m(x)
    ((down * fx["gd"].cuda()).sum() + (skip * fx["gs"].cuda()).sum()).backward()
    rep, tol = [], TOLS[dt]
    assert_close(down, fx["down"], tol, "down", rep)
    assert_close(skip, fx["skip"], tol, "skip", rep)
    assert_close(x.grad, fx["dx"], tol * 2, "dx", rep)
    for k, p in m.named_parameters():
        if ".0.bias" in k or ".3.bias" in k:
            continue
        assert_close(p.grad, fx["grad." + k], tol * 3, "d" + k, rep)


@pytest.mark.parametrize("dt", DTS)
def test_up_block_fixture(M, golden_dir, dt):
    fx = load_fx(golden_dir, "up_block")
    m = M.UpBlock(32, 16, "conv_transpose", dtype=dt).cuda().train()
    load_sd(m, fx)
    xd, xs = fx["xd"].cuda().requires_grad_(True), fx["xs"].cuda().requires_grad_(True)
    y = m(xd, xs)
    (y * fx["go"].cuda()).sum().backward()
    rep, tol = [], TOLS[dt]
    assert_close(y, fx["y"], tol, "y", rep)
    assert_close(xd.grad, fx["dxd"], tol * 2, "dxd", rep)
    assert_close(xs.grad, fx["dxs"], tol * 2, "dxs", rep)
    for k, p in m.named_parameters():
        if ".0.bias" in k or ".3.bias" in k:
            continue
        assert_close(p.grad, fx["grad." + k], tol * 3, "d" + k, rep)


def test_bad_up_sample_mode(M, golden_dir):
    fx = load_fx(golden_dir, "losses")
    with pytest.raises(ValueError) as e:
        M.UpBlock(4, 2, "nearest")
    assert str(e.value) == str(fx["bad_mode_msg"])


@pytest.mark.parametrize("dt", DTS)
def test_unet_small_fixture(M, golden_dir, dt):
    """BASELINE config 1 shape family: base 16, depth 3, Dice+CE criterion (train.py:455)."""
    from oracle import losses as OL
    fx = load_fx(golden_dir, "unet_small")
    m = M.UNet(base_ch=16, depth=3, dtype=dt).cuda().train()
    load_sd(m, fx)
    x, y1h = fx["x"].cuda(), fx["y1h"].cuda()
    logits = m(x)
    loss = F.cross_entropy(logits, y1h)          # Dice term has no gradient (A-4); its value is checked below
    loss.backward()
    rep, tol = [], TOLS[dt]
    assert_close(logits, fx["logits"], tol, "logits", rep)
    total = OL.dice_ce_loss(logits.detach().cpu(), fx["y1h"])
    assert abs(float(total) - float(fx["loss"])) <= tol * max(1.0, abs(float(fx["loss"]))), (float(total), float(fx["loss"]))
    worst = 0.0
    for k, p in m.named_parameters():
        if ".0.bias" in k or ".3.bias" in k:
            continue
        e = rel_err(p.grad, fx["grad." + k])
        worst = max(worst, e)
        assert e <= tol * 5, f"d{k}: {e:.3e}"
    for k, v in m.state_dict().items():
        if "running" in k:
            assert_close(v, fx["after." + k], tol, k, rep)
    # argmax agreement (segmentation mask): exact wherever the reference's logit margin exceeds the tolerance
    ref = fx["logits"]
    margin = (ref[:, 1] - ref[:, 0]).abs()
    sure = margin > 2 * tol * ref.abs().max()
    got_mask = logits.detach().cpu().argmax(1)
    assert torch.equal(got_mask[sure], ref.argmax(1)[sure])
    if dt == "f32":
        assert (got_mask != ref.argmax(1)).float().mean().item() < 1e-3
    m.eval()
    with torch.no_grad():
        le = m(x)
    assert_close(le, fx["logits_eval"], tol * 2, "eval logits", rep)
    print(f"[unet_small {dt}] logits err {rep[0][1]:.2e}, worst param-grad err {worst:.2e}")


@pytest.mark.parametrize("dt", DTS)
def test_unet_full_fixture(M, golden_dir, dt):
    """The reference UNet() itself (31 042 434 parameters), weights regenerated from the fixture's seed."""
    from oracle import unet as OU
    fx = load_fx(golden_dir, "unet_full")
    m = M.UNet(dtype=dt)
    assert sum(p.numel() for p in m.parameters()) == 31042434
    keys = [str(k) for k in fx["state_keys"]]
    assert list(m.state_dict().keys()) == keys
    assert [str(tuple(v.shape)) for v in m.state_dict().values()] == [str(s) for s in fx["state_shapes"]]
    m.load_state_dict(OU.make_state_dict(base_ch=64, depth=5, seed=int(fx["seed"])))
    m = m.cuda().train()
    logits = m(fx["x"].cuda())
    (logits * fx["go"].cuda()).sum().backward()
    tol = TOLS[dt]
    e = rel_err(logits, fx["logits"])
    assert e <= tol * 2, f"logits {e:.3e}"
    # Gradients: the bottleneck of a 32x32 input is 2x2x2 = 8 values per channel -- BatchNorm there is ill-conditioned and a ReLU network's
    # gradient is a discontinuous function of its forward, so rounds 1-4 held the fixture's gradient NORMS to tol * 20.
    # Since round 5 every parameter gradient is held to the kernels' own arithmetic instead: the oracle's float64 backward pass on the
    # HIP path's OWN forward (every raw conv output, ReLU gate and activated value taken from the engine's saved state), flat bars
    # 1e-4 (f32) / 1e-2 (f16) / 1e-1 (bf16) on the relative L2 error of each tensor -- the technique of
    # tests/test_gpu_pretrain.py::test_masked_recon_step_gate_forced_backward on the reference network and the fixture's input.
    from cmunet_amd import ops
    from cmunet_amd.optim import FlatParams
    sd0 = OU.make_state_dict(base_ch=64, depth=5, seed=int(fx["seed"]))
    net = M.UNet(dtype=dt)
    net.load_state_dict(sd0)
    net = net.cuda().train()
    flat = FlatParams(net)
    sdd = dict(net.named_parameters())
    sdd.update(dict(net.named_buffers()))
    eng = net._engine(torch.device("cuda"))
    eng.prepack(sdd)
    x = fx["x"]
    x3 = x if x.dim() == 3 else x[:, 0]
    lg, ctx = eng.unet_forward(sdd, x3.contiguous().float().cuda(), True, None)
    go = fx["go"]
    dlogits = go.contiguous().float().cuda()            # the engine's logits are (B, K, H, W) fp32
    assert dlogits.shape == lg.shape
    layers = {}

    def add(st):
        y = st["y"]
        raw = y.buf[..., y.coff:y.coff + y.C].float()
        z = (raw.double() * y.scale.double() + y.shift.double()).float()
        layers[st["pconv"]] = (raw.permute(0, 3, 1, 2).contiguous().cpu(), torch.clamp_min(z, 0).permute(0, 3, 1, 2).contiguous().cpu(),
                               (z > 0).permute(0, 3, 1, 2).contiguous().cpu())
    for lv in ctx["enc"]["levels"]:
        add(lv["s1"]); add(lv["s2"])
    add(ctx["enc"]["bott"]["s1"]); add(ctx["enc"]["bott"]["s2"])
    for lv in ctx["dec"]["levels"]:
        add(lv["s1"]); add(lv["s2"])
    assert len(layers) == 18
    eng.grad_target, eng.grad_prefix = flat.grad_views, ""
    try:
        eng.unet_backward(sdd, ctx, dlogits)
    finally:
        eng.grad_target = None

    class Taps:
        seen = set()

        def conv(self, key, y):
            self.seen.add(key)
            return y + (layers[key][0].to(y.dtype) - y).detach()

        def act(self, key, z):
            a = z * layers[key][2].to(z.dtype)
            return a + (layers[key][1].to(z.dtype) - a).detach()
    taps = Taps()
    tdt = ops.TORCH_DT[ops.dt_code(dt)]

    def wq(k, v):        # float64 on the weights the kernels multiply with (16-bit storage: the packed copies hold the rounded weights)
        if not v.is_floating_point():
            return v.clone()
        v = v.to(tdt).double() if (v.dim() == 4 and dt != "f32" and "conv_last" not in k) else v.double()
        return v.requires_grad_(True) if "running" not in k else v
    osd = {k: wq(k, v) for k, v in sd0.items()}
    OU.TAP = taps
    try:
        out = OU.unet_forward(x3.double(), osd, training=True)
        (out * go.double()).sum().backward()
    finally:
        OU.TAP = None
    assert taps.seen == set(layers)
    bar = {"f32": 1e-4, "f16": 1e-2, "bf16": 1e-1}[dt]
    worst = ("", 0.0)
    for k, v in osd.items():
        if not (torch.is_tensor(v) and v.requires_grad) or k.endswith(("0.bias", "3.bias")):     # conv biases in front of BN: identically zero
            continue
        got = flat.grad_views[k].detach().double().cpu()
        e2 = float((got - v.grad).norm() / v.grad.norm().clamp_min(1e-30))
        if k.endswith("up_sample.bias"):       # the border remainder of sums that cancel: held to the bar on its layer's scale
            wg = osd[k[:-len("bias")] + "weight"].grad
            e2 = (got - v.grad).norm().item() / max(v.grad.norm().item(), 1e-2 * wg.norm().item())
        if e2 > worst[1]:
            worst = (k, e2)
        assert e2 <= bar, f"d{k}: relative L2 error {e2:.2e} with the gates forced (bar {bar:.0e}, {dt})"
    print(f"[unet_full {dt}] logits err {e:.2e}; gate-forced float64 backward on the 32 x 32 fixture input: worst {worst[0]} {worst[1]:.2e} (bar {bar:.0e})")


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_unet_vs_oracle_masked(M, dt):
    """Fresh seeded input at a non-square size with the fused patch mask (UNet_encoder.py:156) vs the CPU oracle."""
    from oracle import unet as OU
    sd = OU.make_state_dict(base_ch=16, depth=4, seed=5)
    m = M.UNet(base_ch=16, depth=4, dtype=dt)
    m.load_state_dict(sd)
    m = m.cuda().train()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(3, 48, 80, generator=g)
    mask = (torch.rand(1, 3, 5, generator=g) > 0.4).to(torch.uint8).repeat_interleave(16, 1).repeat_interleave(16, 2)
    logits = m(x.cuda(), mask=mask.cuda())
    osd = OU.clone_sd(sd)
    ref = OU.unet_forward(x * (1 - mask[0]).float(), osd, training=True)
    e = rel_err(logits, ref)
    assert e <= TOLS[dt] * 2, f"{e:.3e}"


@pytest.mark.parametrize("shape", [(2, 16, 16), (1, 32, 16), (1, 16, 48)])
def test_unet_smallest_inputs_vs_oracle(M, shape):
    """The smallest inputs a depth-5 UNet accepts (H, W multiples of 16): the bottleneck is 1 x 1 (or 2 x 1 / 1 x 3) pixels, so
    its BatchNorm averages over as few as two values; batch 1 at 32 x 16.  f32 storage vs the CPU oracle, logits and gradients."""
    from oracle import unet as OU
    sd = OU.make_state_dict(base_ch=8, depth=5, seed=9)
    m = M.UNet(base_ch=8, depth=5, dtype="f32")
    m.load_state_dict(sd)
    m = m.cuda().train()
    g = torch.Generator().manual_seed(2)
    x = torch.randn(*shape, generator=g)
    go = torch.randn(shape[0], 2, shape[1], shape[2], generator=g)
    logits = m(x.cuda())
    (logits * go.cuda()).sum().backward()
    osd = OU.clone_sd(sd, requires_grad=True)
    ref = OU.unet_forward(x, osd, training=True)
    (ref * go).sum().backward()
    assert rel_err(logits, ref) <= 2e-3
    for k in ("conv_last.weight", "down_conv1.double_conv.double_conv.0.weight", "double_conv.double_conv.3.weight", "up_conv4.up_sample.weight"):
        e = rel_err(m.get_parameter(k).grad, osd[k].grad)
        assert e <= 2e-2, f"d{k}: {e:.3e}"
    with pytest.raises(AssertionError):
        m(torch.randn(1, 24, 16).cuda())                     # not a multiple of 2^4


_REF_SIZES = {}


def _reference_size_case(size):
    """CPU oracle of the full-depth base-64 UNet on one size x size image, in float32 (what the reference computes) and in float64
    (the truth both float32 runs scatter around); computed once per size."""
    if size not in _REF_SIZES:
        from oracle import unet as OU
        sd = OU.make_state_dict(base_ch=64, depth=5, seed=21)
        g = torch.Generator().manual_seed(size)
        x = torch.randn(1, size, size, generator=g)
        go = torch.randn(1, 2, size, size, generator=g)
        osd = OU.clone_sd(sd, requires_grad=True)
        ref = OU.unet_forward(x, osd, training=True)
        (ref * go).sum().backward()
        osd64 = {k: (v.double().requires_grad_(True) if v.is_floating_point() and "running" not in k else (v.double() if v.is_floating_point() else v.clone()))
                 for k, v in sd.items()}
        ref64 = OU.unet_forward(x.double(), osd64, training=True)
        (ref64 * go.double()).sum().backward()
        _REF_SIZES[size] = (sd, x, go, ref.detach(), {k: v.grad for k, v in osd.items() if v.requires_grad}, ref64.detach(),
                            {k: v.grad for k, v in osd64.items() if torch.is_tensor(v) and v.requires_grad})
    return _REF_SIZES[size]


@pytest.mark.parametrize("dt", ["f32", "f16"])
@pytest.mark.parametrize("size", [224] + ([256] if __import__("os").environ.get("CMU_TEST_SLOW") == "1" else []))   # (256: whole tiles, covered by the finetuning tests; CMU_TEST_SLOW=1)
def test_unet_reference_image_sizes_vs_oracle(M, size, dt):
    """The sizes the reference actually trains at: 224 x 224 (CM-UNet / MoCo crops, cmunet_dataset.py:60-88) and
    256 x 256 (finetuning, dataset.py:46-47) through the reference UNet (base 64, depth 5): at 224 the levels are 224 / 112 / 56 /
    28 / 14 pixels wide -- partial 16 x 32 tiles at every level but the first, the one-tile kernels instead of the persistent one.
    With random weights and a random output gradient the float32 gradients of the oracle itself sit 3e-3 ... 2e-2 (max norm) from
    its float64 run, so the bar is stated against the float64 truth: logits within the dtype's tolerance, argmax equal where the
    margin allows, and for f32 storage every checked gradient within 8x the oracle's own float32 error in relative L2 (floor 2e-3)."""
    sd, x, go, ref32, g32, ref64, g64 = _reference_size_case(size)
    m = M.UNet(dtype=dt)
    m.load_state_dict(sd)
    m = m.cuda().train()
    logits = m(x.cuda())
    (logits * go.cuda()).sum().backward()
    tol = TOLS[dt]
    e, e_cpu = rel_err(logits, ref64), rel_err(ref32, ref64)
    assert e <= max(tol * 2, 5 * e_cpu), f"logits {e:.3e} (oracle f32: {e_cpu:.3e})"
    margin = (ref64[:, 1] - ref64[:, 0]).abs()
    sure = margin > 4 * tol * ref64.abs().max()
    assert torch.equal(logits.detach().cpu().argmax(1)[sure], ref64.argmax(1)[sure])
    worst = 0.0
    for k in ("conv_last.weight", "down_conv1.double_conv.double_conv.0.weight", "down_conv3.double_conv.double_conv.3.weight",
              "double_conv.double_conv.3.weight", "up_conv4.up_sample.weight", "up_conv4.double_conv.double_conv.3.weight",
              "up_conv1.double_conv.double_conv.0.weight"):
        # relative L2 error of the whole tensor (the max norm of a 2-million-element gradient picks its single worst outlier)
        l2 = lambda a, b: (a.detach().double().cpu() - b).norm().item() / max(b.norm().item(), 1e-12)
        e2, e2_cpu = l2(m.get_parameter(k).grad, g64[k]), l2(g32[k].double(), g64[k])
        worst = max(worst, e2 / max(e2_cpu, 1e-9))
        bar = max(8 * e2_cpu, 2e-3) if dt == "f32" else tol * 20
        assert e2 <= bar, f"d{k}: {e2:.3e} (oracle f32: {e2_cpu:.3e})"
    print(f"[unet {size}x{size} {dt}] logits err vs f64 {e:.2e} (oracle f32 {e_cpu:.2e}); worst gradient error ratio to the oracle's f32 {worst:.1f}")


def test_parameter_gradients_land_in_the_arena_and_shared_weights_still_accumulate():
    """optim.claim_grad_sink: with the parameters held by a FlatParams arena the UNet's autograd node writes every parameter gradient
    into its arena slot and autograd adopts the alias as ``.grad`` (no copy at gather time); the same network applied twice in one
    graph -- its weights used by two backward nodes -- still gets the SUM of both gradients (the second node must not overwrite
    the slot), and gradient accumulation over two backward calls adds up.  Checked against the same network without an arena."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from cmunet_amd import model as M
    from cmunet_amd.optim import FlatParams
    torch.manual_seed(3)
    ref = M.UNet(base_ch=16, depth=3, dtype="f32").cuda().train()
    net = M.UNet(base_ch=16, depth=3, dtype="f32").cuda().train()
    net.load_state_dict(ref.state_dict())
    flat = FlatParams(net)
    g = torch.Generator(device="cuda").manual_seed(1)
    x1 = torch.randn(2, 32, 32, generator=g, device="cuda")
    x2 = torch.randn(2, 32, 32, generator=g, device="cuda")

    def loss_of(m, a, b=None):
        out = m(a).square().mean()
        return out if b is None else out + 0.5 * m(b).square().mean()

    # one use: adopted in place
    loss_of(ref, x1).backward()
    loss_of(net, x1).backward()
    named = dict(net.named_parameters())
    w = "down_conv1.double_conv.double_conv.0.weight"
    assert named[w].grad.data_ptr() == flat.grad_views[w].data_ptr()
    flat.gather_autograd_grads()
    for n, p in ref.named_parameters():
        assert torch.allclose(flat.grad_views[n], p.grad, rtol=1e-5, atol=1e-7), n
    # gradient accumulation: a second backward adds into the adopted .grad
    loss_of(ref, x2).backward()
    loss_of(net, x2).backward()
    flat.gather_autograd_grads()
    for n, p in ref.named_parameters():
        assert torch.allclose(flat.grad_views[n], p.grad, rtol=1e-5, atol=1e-7), n
    # two uses of the same weights in one graph
    for m in (ref, net):
        for p in m.parameters():
            p.grad = None
    loss_of(ref, x1, x2).backward()
    loss_of(net, x1, x2).backward()
    flat.gather_autograd_grads()
    for n, p in ref.named_parameters():
        assert torch.allclose(flat.grad_views[n], p.grad, rtol=1e-5, atol=1e-7), n
