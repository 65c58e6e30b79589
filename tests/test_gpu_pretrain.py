"""GPU parity of the pretraining models (CM-UNet joint step, MoCo-v2 step, fused masked-recon trainer)
against the CPU oracle restatement (oracle/cmunet.py, oracle/moco.py) on small seeded configurations.
fp32 storage; tolerance 2e-3 relative (max norm) on losses and gradients -- the chain is
conv blocks -> mean -> Linear(H*W, 1536) -> BN1d over 4 rows -> ... so rounding is amplified."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
# Round 6 (the driver's GPU run took 702 s of its 900 s limit): representatives run by default, their twins -- the second input mask of the joint
# step's geometry / gate-forced tests (float64 oracle on the CPU: 12-13 s each), the 256-pixel SparK gate-forced case -- with CMU_TEST_SLOW=1
# (run once per round by the builder; numbers in profiles/r06_parity.txt).
SLOW = __import__("os").environ.get("CMU_TEST_SLOW") == "1"
J224_BASE = __import__("os").environ.get("CMU_JOINT224_BASE", "32")       # (tests/joint224_case.py: 64 = the reference's width, run once per round)


def _parity_record(line):
    """Achieved parity numbers of this run -> gpurun_out/parity_record.txt (copied to profiles/r03_parity.txt): what the bars
    passed BY, not only that they passed."""
    import os
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "parity_record.txt"), "a") as f:
            f.write(line + "\n")
    except OSError:
        pass


@pytest.fixture(scope="module")
def cuda():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda")


def rel(got, ref):
    return (got.detach().double().cpu() - ref.detach().double()).abs().max().item() / max(ref.detach().abs().max().item(), 1e-9)


def test_cmunet_joint_step_vs_oracle(cuda):
    from cmunet_amd import cmunet as C
    from cmunet_amd.pretrain import create_random_patch_mask
    from oracle import cmunet as OC
    torch.manual_seed(0)
    B, S = 4, 32
    model = C.build_model(C.cmunet_config(img_size=S, dtype="f32", base_ch=16, depth=3)).to(cuda).train()
    with torch.no_grad():                                   # non-trivial BN affine parameters everywhere
        for n, p in model.named_parameters():
            if p.dim() == 1 and ("bn" in n or ".1." in n or ".4." in n):
                p.add_(0.2 * torch.randn_like(p))
        for pb, pm in zip(model.backbone.parameters(), model.target_backbone.parameters()):
            pm.copy_(pb * 0.9)
        for pb, pm in zip(model.projector.parameters(), model.target_projector.parameters()):
            pm.copy_(pb * 1.1)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(1)
    img, img_t = torch.randn(B, S, S, generator=g), torch.randn(B, S, S, generator=g)
    mask = torch.from_numpy(create_random_patch_mask(B, S, 16, 0.65, np.random.RandomState(2)))
    Cr = model.reduced_channels()
    rw, rb = torch.randn(Cr, 64, 1, 1, generator=g) * 0.1, torch.randn(Cr, generator=g) * 0.1

    losses = model(img.to(cuda), mode='loss', img_t=img_t.to(cuda), mask=mask.to(cuda), reduce_w=rw.to(cuda), reduce_b=rb.to(cuda))
    (losses['loss_ct'] + losses['loss_rc']).backward()

    osd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k and not k.startswith("target_") else v.clone())
           for k, v in sd.items()}
    ref = OC.forward_train(img, img_t, mask.numpy(), rw, rb, osd, temperature=0.07, ct_weight=1.0, rc_weight=1.0)
    (ref['loss_ct'] + ref['loss_rc']).backward()

    assert abs(float(losses['loss_rc']) - float(ref['loss_rc'])) <= 2e-4 * max(1, abs(float(ref['loss_rc'])))
    assert abs(float(losses['loss_ct']) - float(ref['loss_ct'])) <= 2e-3 * max(1, abs(float(ref['loss_ct'])))
    params = dict(model.named_parameters())
    checked = 0
    for k, v in osd.items():
        if not (torch.is_tensor(v) and v.requires_grad) or v.grad is None:
            continue
        if ".0.bias" in k or ".3.bias" in k:                 # conv bias under train-mode BN: identically zero here
            continue
        if v.grad.abs().max() < 1e-6:                         # analytically zero (e.g. a constant shift in front of a
            assert params[k].grad.abs().max() < 1e-5, k        # batch-normalised Linear): only rounding noise on both sides
            continue
        e = rel(params[k].grad, v.grad)
        assert e <= 5e-3, f"d{k}: {e:.2e}"
        checked += 1
    assert checked > 60
    for k in sd:                                             # target networks got no gradient, BN buffers advanced
        if k.startswith("target_") and k in params:
            assert params[k].grad is None
    assert int(model.state_dict()["target_backbone.double_conv.double_conv.1.num_batches_tracked"]) == 1

    # EMA (cmunet.py:78-92)
    before = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    model.momentum = 0.9
    model.momentum_update()
    OC.momentum_update(before, 0.9)
    for k, v in model.state_dict().items():
        if k.startswith("target_") and v.is_floating_point() and "running" not in k:
            assert rel(v, before[k]) <= 1e-6, k


def test_cmunet_joint_step_vs_reference_fixture(cuda, golden_dir):
    """The HIP path against what the REFERENCE's own CM_UNet produced (tests/golden/cmunet_ref.npz, written by
    oracle/gen_golden.py::gen_cmunet from the reference's cmae modules; shipped cmunet_config.py at 224 x 224, bs 4): the same
    seeded state loaded under the reference's key names (strict), the same images, the reference's patch mask and per-call
    reduce_channels conv -> forward_train's two losses, the gradient norm of every trainable parameter, sampled gradients in
    full, BatchNorm buffers, the EMA of momentum_update, the hook's schedule, and CMUNetPretrainHead.forward alone."""
    from cmunet_amd import cmunet as C
    from cmunet_amd.pretrain import create_random_patch_mask
    from oracle import cmunet as OC
    f = np.load(f"{golden_dir}/cmunet_ref.npz")
    seed, B, S = int(f["seed"]), int(f["B"]), int(f["S"])
    img, img_t, mask, rw, rb = OC.cmunet_fixture_inputs(seed, B, S)
    # the product's own host-side mask generator draws the reference's mask from the same numpy stream (UNet_encoder.py:106-139)
    pm = create_random_patch_mask(B, S, 16, 0.65, np.random.RandomState(seed + 11))
    assert np.array_equal(pm, mask) and np.array_equal(pm[:, ::16, ::16], f["mask_patches"])
    model = C.build_model(C.cmunet_config(img_size=S, dtype="f32")).to(cuda).train()
    sd = OC.make_cmunet_sd(seed, S)
    model.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)        # the reference's state_dict key names
    trainable = [str(k) for k in f["trainable"]]
    named = dict(model.named_parameters())
    assert sorted(k for k, p in named.items() if p.requires_grad) == trainable
    losses = model(img.to(cuda), mode='loss', img_t=img_t.to(cuda), mask=torch.from_numpy(mask).to(cuda), reduce_w=rw.to(cuda), reduce_b=rb.to(cuda))
    (losses['loss_ct'] + losses['loss_rc']).backward()
    d_rc, d_ct = abs(float(losses['loss_rc']) - float(f["loss_rc"])), abs(float(losses['loss_ct']) - float(f["loss_ct"]))
    print(f"CM_UNet vs reference: loss_rc {float(losses['loss_rc']):.6f} (ref {float(f['loss_rc']):.6f}), loss_ct {float(losses['loss_ct']):.6f} "
          f"(ref {float(f['loss_ct']):.6f})")
    assert d_rc <= 2e-4 * max(1.0, abs(float(f["loss_rc"]))) and d_ct <= 2e-3 * max(1.0, abs(float(f["loss_ct"])))
    ref = torch.from_numpy(f["grad_norms"]).double()
    got = torch.stack([named[k].grad.double().norm().cpu() for k in trainable])
    noise = torch.tensor([k.endswith((".0.bias", ".3.bias", "fc0.bias")) or k == "feature_decoder.conv_last.bias" for k in trainable])
    relerr = ((got - ref).abs() / ref.clamp_min(1e-30))[~noise]
    worst = int(torch.argmax(relerr))
    print(f"  gradient norms of {int((~noise).sum())} parameters: worst relative difference {float(relerr[worst]):.2e} "
          f"({[k for k, n in zip(trainable, noise) if not n][worst]})")
    assert float(relerr.max()) <= 5e-3
    assert float(got[noise].max()) <= 1e-2                  # analytically zero (bias in front of a training-mode BatchNorm)
    for k in f.files:
        if k.startswith("grad."):
            e = rel(named[k[5:]].grad, torch.from_numpy(f[k]))
            assert e <= 5e-3, f"d{k[5:]}: {e:.2e}"
        if k.startswith("after."):
            assert rel(model.state_dict()[k[6:]].float(), torch.from_numpy(f[k]).float()) <= 1e-4, k
    for k in named:
        if k.startswith("target_"):
            assert named[k].grad is None
    # EMA (cmunet.py:78-92) at the fixture's momentum
    model.momentum = 0.9
    model.momentum_update()
    tk = [str(k) for k in f["target_keys"]]
    en = torch.stack([named[k].detach().double().norm().cpu() for k in tk])
    assert float(((en - torch.from_numpy(f["ema_norms"])).abs() / torch.from_numpy(f["ema_norms"]).clamp_min(1e-12)).max()) <= 1e-6
    assert rel(named["target_backbone.down_conv1.double_conv.double_conv.0.weight"].detach(), torch.from_numpy(f["ema_sample"])) <= 1e-6
    for (it, mx, base, end), m in zip(f["hook_cases"], f["hook_momentum"]):
        assert abs(C.momentum_schedule(int(it), int(mx), float(base), float(end)) - float(m)) < 1e-12
    # the head alone (cmunet_head.py:47-91): the reference takes pred_pixel[:, 1]; channel 1 of a 2-channel map here
    model.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)        # (predictor BatchNorm buffers back to the fixture's start)
    model.zero_grad()
    x, mk = torch.from_numpy(f["head.x"]).to(cuda), torch.from_numpy(f["head.mask"]).to(cuda)
    pred = torch.from_numpy(f["head.pred"]).to(cuda)
    logits = torch.stack([torch.zeros_like(pred), pred], 1).contiguous().requires_grad_(True)
    ps = torch.from_numpy(f["head.proj_s"]).to(cuda).requires_grad_(True)
    hl = model.head(x, logits, mk, ps, torch.from_numpy(f["head.proj_t"]).to(cuda))
    (hl["loss_ct"] + hl["loss_rc"]).backward()
    assert abs(float(hl["loss_rc"]) - float(f["head.loss_rc"])) <= 2e-5 * max(1.0, abs(float(f["head.loss_rc"])))
    assert abs(float(hl["loss_ct"]) - float(f["head.loss_ct"])) <= 1e-4 * max(1.0, abs(float(f["head.loss_ct"])))
    assert rel(logits.grad[:, 1], torch.from_numpy(f["head.dpred"])) <= 1e-4 and float(logits.grad[:, 0].abs().max()) == 0.0
    assert rel(ps.grad, torch.from_numpy(f["head.dproj_s"])) <= 1e-3
    # ... and with other hyper-parameters (temperature 0.2, ct_weight 0.5, rc_weight 2), as the reference's head computed them
    t2, cw2, rw2 = (float(v) for v in f["head2.hyper"])
    cfg2 = C.cmunet_config(img_size=S, dtype="f32", temperature=t2, ct_weight=cw2, rc_weight=rw2)["head"]
    head2 = C.build_model(cfg2).to(cuda).train()
    head2.load_state_dict({k[len("head."):]: v.clone() for k, v in sd.items() if k.startswith("head.")}, strict=True)
    lg2 = torch.stack([torch.zeros_like(pred), pred], 1).contiguous().requires_grad_(True)
    ps2 = torch.from_numpy(f["head.proj_s"]).to(cuda).requires_grad_(True)
    h2 = head2(x, lg2, mk, ps2, torch.from_numpy(f["head.proj_t"]).to(cuda))
    (h2["loss_ct"] + h2["loss_rc"]).backward()
    assert abs(float(h2["loss_rc"]) - float(f["head2.loss_rc"])) <= 2e-5 * max(1.0, abs(float(f["head2.loss_rc"])))
    assert abs(float(h2["loss_ct"]) - float(f["head2.loss_ct"])) <= 1e-4 * max(1.0, abs(float(f["head2.loss_ct"])))
    assert rel(lg2.grad[:, 1], torch.from_numpy(f["head2.dpred"])) <= 1e-4 and rel(ps2.grad, torch.from_numpy(f["head2.dproj_s"])) <= 1e-3


def test_nonlinear_neck_eval_mode_gradients(cuda):
    """NonLinearNeck with its BatchNorm in eval mode (running statistics: a fixed affine map): output, input gradient and the
    gradients of fc / norm parameters against torch autograd of the same functions (nonlinear_neck.py:88-102).  (The eval-mode
    backward once read an unwritten invstd -- found by the reference's two-rank head fixture.)"""
    from cmunet_amd import cmunet as C
    g = torch.Generator().manual_seed(12)
    neck = C.NonLinearNeck(in_channels=96, hid_channels=128, out_channels=32, num_layers=2, with_bias=True, with_last_bn=False,
                           with_avg_pool=False).to(cuda)
    with torch.no_grad():
        neck.bn0.running_mean.copy_(0.2 * torch.randn(128, generator=g)); neck.bn0.running_var.copy_(0.5 + torch.rand(128, generator=g))
        neck.bn0.weight.copy_(1 + 0.2 * torch.randn(128, generator=g)); neck.bn0.bias.copy_(0.1 * torch.randn(128, generator=g))
    neck.train()
    neck.bn0.eval()
    x = torch.randn(6, 1, 96, generator=g)
    go = torch.randn(6, 1, 32, generator=g)
    xg = x.to(cuda).requires_grad_(True)
    y = neck(xg)
    (y * go.to(cuda)).sum().backward()
    # torch reference on the CPU
    import torch.nn.functional as F
    p = {k: v.detach().cpu().double().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in neck.state_dict().items()}
    xr = x.double().requires_grad_(True)
    h = F.linear(xr[:, 0, :], p["fc0.weight"], p["fc0.bias"])
    h = F.batch_norm(h, p["bn0.running_mean"], p["bn0.running_var"], p["bn0.weight"], p["bn0.bias"], False, 0.1, 1e-6)
    yr = F.linear(F.relu(h), p["fc1.weight"]).unsqueeze(1)
    (yr * go.double()).sum().backward()
    assert rel(y, yr) <= 1e-5 and rel(xg.grad, xr.grad) <= 1e-4
    named = dict(neck.named_parameters())
    for k in ("fc0.weight", "fc0.bias", "bn0.weight", "bn0.bias", "fc1.weight"):
        assert rel(named[k].grad, p[k].grad) <= 2e-4, k
    assert int(neck.bn0.num_batches_tracked) == 0                       # eval mode: buffers untouched


def test_joint_trainer_dynamic_loss_scale(cuda):
    """JointPretrainer with the AmpOptimWrapper protocol (cmunet_config.py:76-78) at f16: the loss is scaled by the device-side
    scale, clean steps update and count, an overflowing step (scale 2^120) is skipped on the device -- parameters, Adam moments and
    the step count untouched -- and the scale backs off."""
    from cmunet_amd import cmunet as C, ops
    from cmunet_amd.pretrain import JointPretrainer, create_random_patch_mask
    torch.manual_seed(0)
    B, S = 4, 32
    model = C.build_model(C.cmunet_config(img_size=S, dtype="f16", base_ch=16, depth=3)).to(cuda).train()
    tr = JointPretrainer(model, lr=1e-3, amp=True)
    g = torch.Generator().manual_seed(3)
    img, img_t = torch.randn(B, S, S, generator=g).to(cuda), torch.randn(B, S, S, generator=g).to(cuda)
    mask = torch.from_numpy(create_random_patch_mask(B, S, 16, 0.65, np.random.RandomState(4))).to(cuda)
    for _ in range(3):
        l = tr.step(img, img_t, mask)
        assert torch.isfinite(l["loss_ct"]) and torch.isfinite(l["loss_rc"])
    scale, found, growth, good, skipped = tr.amp.read()
    assert (scale, found, good, skipped) == (65536.0, 0.0, 3, 0) and growth == 3
    # an overflowing step
    tr.amp = ops.AmpScaler(cuda, init_scale=2.0 ** 120)
    before, m_before = tr.flat.arena.clone(), tr.opt.m.clone()
    tr.step(img, img_t, mask)
    scale, found, growth, good, skipped = tr.amp.read()
    assert skipped == 1 and good == 0 and scale == 2.0 ** 119
    assert torch.equal(tr.flat.arena, before) and torch.equal(tr.opt.m, m_before)
    # without the scaler the f16 step still runs (static scale 1): the trainer's amp is optional
    tr.amp = None
    tr.step(img, img_t, mask)
    assert not torch.equal(tr.flat.arena, before)


def test_default_constructed_trainers_scale_the_loss_at_f16(cuda):
    """Advisor (round 3): ``cmunet_config()`` / ``CM_UNet`` default to f16 storage, so ``JointPretrainer(model)`` and
    ``MaskedReconPretrainer(model)`` with NO amp argument must come up with the reference's dynamic loss scaler
    (cmunet_config.py:76-78) -- their first step equals the ``amp=True`` one bit for bit and moves the encoder --, f32 models come up
    without one, and ``amp=False`` switches it off."""
    from cmunet_amd import cmunet as C, model as M
    from cmunet_amd.pretrain import JointPretrainer, MaskedReconPretrainer, create_random_patch_mask
    B, S = 4, 32
    g = torch.Generator().manual_seed(3)
    img, img_t = torch.randn(B, S, S, generator=g).to(cuda), torch.randn(B, S, S, generator=g).to(cuda)
    mask = torch.from_numpy(create_random_patch_mask(B, S, 16, 0.65, np.random.RandomState(4))).to(cuda)

    def joint(**kw):
        torch.manual_seed(0)
        cfg = C.cmunet_config(img_size=S, base_ch=16, depth=3)
        model = C.build_model(cfg).to(cuda).train()
        assert model.dtype == "f16"
        tr = JointPretrainer(model, lr=1e-3, **kw)
        w0 = tr.flat.arena.clone()
        tr.step(img, img_t, mask)
        enc = tr.flat.prefix_range("backbone.")
        return tr, w0, enc
    tr_def, w0, enc = joint()
    tr_amp, _, _ = joint(amp=True)
    assert tr_def.amp is not None and tr_def.amp.read()[3] == 1                       # one good (unskipped) update
    assert torch.equal(tr_def.flat.arena, tr_amp.flat.arena) and torch.equal(tr_def.flat.grad, tr_amp.flat.grad)
    assert not torch.equal(tr_def.flat.arena[enc[0]:enc[1]], w0[enc[0]:enc[1]]) and float(tr_def.flat.grad[enc[0]:enc[1]].abs().max()) > 0
    tr_off, _, _ = joint(amp=False)
    assert tr_off.amp is None

    def recon(dt, **kw):
        torch.manual_seed(0)
        net = M.UNet(out_classes=2, dtype=dt, base_ch=16, depth=3).to(cuda)
        tr = MaskedReconPretrainer(net, lr=1e-3, **kw)
        tr.step(img, mask)
        return tr
    r_def, r_amp = recon("f16"), recon("f16", amp=True)
    assert r_def.amp is not None and torch.equal(r_def.flat.arena, r_amp.flat.arena)
    assert recon("f32").amp is None and recon("bf16").amp is None and recon("f16", amp=False).amp is None


def test_cmunet_modules_standalone(cuda):
    """UNet_encoder / MUNetPretrainDecoder used on their own keep the reference's tensor contract."""
    from cmunet_amd import cmunet as C
    from oracle import unet as OU, cmunet as OC
    enc = C.UNet_encoder(base_ch=16, depth=3, dtype="f32", mask_ratio=0.5).to(cuda).train()
    dec = C.MUNetPretrainDecoder(base_ch=16, depth=3, dtype="f32").to(cuda).train()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 32, 48, generator=g)
    latent, mask, skips = enc(x.to(cuda))
    assert latent.shape == (2, 64, 8, 12) and mask.shape == (2, 32, 48) and mask.dtype == torch.uint8
    assert [tuple(s.shape) for s in skips] == [(2, 16, 32, 48), (2, 32, 16, 24)]
    assert int(mask[0].sum()) == int(0.5 * 32 * 48) // 256 * 256
    esd = {k: v.detach().cpu().clone() for k, v in enc.state_dict().items()}
    lat_ref, _, sk_ref = OC.encoder(x, mask.cpu().numpy(), esd, "", True, True)
    assert rel(latent, lat_ref) <= 1e-3 and rel(skips[0], sk_ref[0]) <= 1e-3
    out = dec(latent, skips)
    dsd = {k: v.detach().cpu().clone() for k, v in dec.state_dict().items()}
    ref = OC.decoder(lat_ref, sk_ref, dsd, "", True)
    assert out.shape == (2, 2, 32, 48) and rel(out, ref) <= 2e-3
    out.sum().backward()
    assert enc.down_conv1.double_conv.double_conv[0].weight.grad is not None


def test_moco_step_vs_oracle(cuda):
    from cmunet_amd import moco as M
    from oracle import moco as OM
    torch.manual_seed(0)
    B, S, K, T = 4, 32, 64, 0.2
    m = M.Moco_v2(emb_dim=64, num_negatives=K, softmax_temperature=T, encoder_momentum=0.99, dtype="f32", base_ch=16, depth=3).to(cuda).train()
    with torch.no_grad():
        for pk in m.encoder_k.parameters():
            pk.mul_(0.95)
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(5)
    xq, xk = torch.randn(B, 1, S, S, generator=g), torch.randn(B, 1, S, S, generator=g)
    loss = m.training_step(((xq.to(cuda), xk.to(cuda)), 0))
    loss.backward()
    osd = {k: (v.clone().requires_grad_(True) if k.startswith("encoder_q.") and v.is_floating_point() and "running" not in k else v.clone())
           for k, v in sd.items()}
    queue, ptr = sd["queue"].clone(), sd["queue_ptr"].clone()
    ref, logits, k = OM.training_step(xq, xk, osd, queue, ptr, T, 0.99)
    ref.backward()
    assert abs(float(loss) - float(ref)) <= 1e-3 * max(1.0, abs(float(ref)))
    assert rel(m.queue, queue) <= 1e-4 and int(m.queue_ptr) == int(ptr) == B
    pq = dict(m.named_parameters())
    for kname in ("encoder_q.down_conv1.double_conv.double_conv.0.weight", "encoder_q.double_conv.double_conv.3.weight",
                  "encoder_q.down_conv2.double_conv.double_conv.4.bias"):
        assert rel(pq[kname].grad, osd[kname].grad) <= 5e-3, kname
    # key encoder after the EMA (before the forward, A-8)
    assert rel(m.encoder_k.double_conv.double_conv[0].weight, osd["encoder_k.double_conv.double_conv.0.weight"]) <= 1e-6
    # API-faithful forward(): logits (N, 1+K), labels 0
    m.zero_grad()
    lg, lb, kk, qq = m(xq.to(cuda), xk.to(cuda), m.queue)
    assert lg.shape == (B, 1 + K) and int(lb.sum()) == 0 and kk.shape == (B, 64)
    # ... and its logits / gradient (through the skinny-kernel q @ queue) equal the oracle's on the same, already updated state
    F.cross_entropy(lg, lb).backward()
    sd2 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    osd2 = {k: (v.clone().requires_grad_(True) if k.startswith("encoder_q.") and v.is_floating_point() and "running" not in k else v.clone())
            for k, v in sd2.items()}
    for k in list(osd2):                       # the forward above already advanced the BatchNorm buffers: rewind them for the oracle
        if "running" in k or "num_batches" in k:
            osd2[k] = osd[k].clone() if k in osd else osd2[k]
    q2 = OM.encoder_gap(xq, osd2, "encoder_q.", True)
    with torch.no_grad():
        k2 = OM.encoder_gap(xk, osd2, "encoder_k.", True)
    lg2, lb2, _, _ = OM.logits_from_embeddings(q2, k2, sd2["queue"], T)
    assert rel(lg, lg2) <= 1e-3
    F.cross_entropy(lg2, lb2).backward()
    kname = "encoder_q.double_conv.double_conv.3.weight"
    assert rel(pq[kname].grad, osd2[kname].grad) <= 5e-3


@pytest.mark.parametrize("dt", ["f16", "bf16"])
def test_cmunet_joint_step_16bit_vs_reference_fixture(cuda, golden_dir, dt):
    """The AMP arithmetic (16-bit storage / MFMA operands incl. the 16-bit-operand projector GEMMs, fp32 accumulation) of the joint
    step at the reference's shipped 224 x 224 geometry against the reference's own fp32 CM_UNet (tests/golden/cmunet_ref.npz): the two
    losses and the gradient norms of all trainable tensors, with the bars of 16-bit storage."""
    from cmunet_amd import cmunet as C
    from oracle import cmunet as OC
    f = np.load(f"{golden_dir}/cmunet_ref.npz")
    seed, B, S = int(f["seed"]), int(f["B"]), int(f["S"])
    img, img_t, mask, rw, rb = OC.cmunet_fixture_inputs(seed, B, S)
    model = C.build_model(C.cmunet_config(img_size=S, dtype=dt)).to(cuda).train()
    model.load_state_dict({k: v.clone() for k, v in OC.make_cmunet_sd(seed, S).items()}, strict=True)
    losses = model(img.to(cuda), mode='loss', img_t=img_t.to(cuda), mask=torch.from_numpy(mask).to(cuda), reduce_w=rw.to(cuda), reduce_b=rb.to(cuda))
    # (f16: a static loss scale, as AMP training uses one -- the masked-MSE gradient per pixel is ~1e-5 here, below f16's normals)
    scale = 4096.0 if dt == "f16" else 1.0
    ((losses['loss_ct'] + losses['loss_rc']) * scale).backward()
    e_rc = abs(float(losses['loss_rc'].detach()) - float(f["loss_rc"])) / abs(float(f["loss_rc"]))
    e_ct = abs(float(losses['loss_ct'].detach()) - float(f["loss_ct"])) / abs(float(f["loss_ct"]))
    trainable = [str(k) for k in f["trainable"]]
    named = dict(model.named_parameters())
    ref = torch.from_numpy(f["grad_norms"]).double()
    got = torch.stack([named[k].grad.double().norm().cpu() for k in trainable]) / scale
    noise = torch.tensor([k.endswith((".0.bias", ".3.bias", "fc0.bias")) or k == "feature_decoder.conv_last.bias" for k in trainable])
    # (a ConvTranspose bias feeds a conv + training-mode BatchNorm: its gradient is the border remainder of sums that cancel, ~1e-3
    # of its layer's weight gradient -- measured on that scale, as in test_spark_step_vs_reference_fixture)
    denom = ref.clone()
    for i, k in enumerate(trainable):
        if k.endswith("up_sample.bias"):
            denom[i] = ref[trainable.index(k[:-len("bias")] + "weight")]
    relerr = ((got - ref).abs() / denom.clamp_min(1e-30))[~noise]
    worst = sorted(zip(relerr.tolist(), [k for k, n in zip(trainable, noise) if not n]))[-3:]
    print("  worst:", [(f"{e:.2e}", k) for e, k in worst])
    _parity_record(f"CM_UNet joint step {dt} vs the reference's fp32 fixture (224x224, bs 4): loss_rc {e_rc:.2e}, loss_ct {e_ct:.2e} relative; "
                   f"per-tensor gradient-norm error worst {float(relerr.max()):.2e}, median {float(relerr.median()):.2e}; worst three "
                   + "; ".join(f"{k}: {e:.2e}" for e, k in worst))
    print(f"CM_UNet {dt} vs the reference's fp32 run: loss_rc {e_rc:.2e}, loss_ct {e_ct:.2e} relative; gradient norms worst {float(relerr.max()):.2e}, "
          f"median {float(relerr.median()):.2e}")
    # measured: f16 loss_rc 1e-5, loss_ct 5e-3, gradient norms median 7e-3; bf16 (8 significand bits against 11) 1e-4, 2.5e-2, 9e-2 --
    # the contrastive gradient goes through a BatchNorm over bs 4 rows and a softmax at temperature 0.07, which amplify operand rounding
    # round 3: with the ConvTranspose biases on their layer's scale the f16 worst per-tensor error is 2.8e-2 (median 4.9e-3), bf16 0.26
    # (median 8.9e-2) -- profiles/r03_parity.txt; the f16 bars are halved accordingly (worst 10 % -> 5 %, median 2 % -> 1 %)
    bar_l, bar_g, bar_m = (5e-3, 5e-2, 1e-2) if dt == "f16" else (3e-2, 3.5e-1, 1.5e-1)
    assert e_rc <= bar_l and e_ct <= 10 * bar_l, (e_rc, e_ct)
    assert float(relerr.max()) <= bar_g and float(relerr.median()) <= bar_m


def test_moco_step_vs_reference_fixture(cuda, golden_dir):
    """The HIP path against what the REFERENCE's own Moco_v2 produced (tests/golden/moco_ref.npz, oracle/gen_golden.py::gen_moco):
    the same seeded state under the reference's key names, two training steps (EMA before the forward, logits from the
    pre-enqueue queue, enqueue before the loss: A-8) -- losses, enqueued keys, pointer, gradient norms of the query encoder, the
    key encoder after its EMA -- and forward()'s (logits, labels, k, q) contract in between."""
    from cmunet_amd import moco as M
    from oracle import moco as OM
    f = np.load(f"{golden_dir}/moco_ref.npz")
    seed, B, S, K, T, EM = int(f["seed"]), int(f["B"]), int(f["S"]), int(f["K"]), float(f["T"]), float(f["EM"])
    m = M.Moco_v2(emb_dim=1024, num_negatives=K, softmax_temperature=T, encoder_momentum=EM, dtype="f32").to(cuda).train()
    sd = OM.make_moco_sd(seed, K)
    have = m.state_dict()
    assert set(sd) <= set(have) and set(have) - set(sd) <= {"val_queue", "val_queue_ptr"}, sorted(set(have) ^ set(sd))[:6]
    m.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=False)
    xq, xk, xq2, xk2 = (t.to(cuda) for t in OM.moco_fixture_inputs(seed, B, S))
    named = dict(m.named_parameters())
    qkeys = [str(k) for k in f["qkeys"]]
    assert sorted(k for k, p in named.items() if p.requires_grad) == qkeys
    loss = m.training_step(((xq, xk), 0))
    loss.backward()
    print(f"Moco_v2 vs reference: loss {float(loss):.6f} (ref {float(f['loss']):.6f})")
    assert abs(float(loss) - float(f["loss"])) <= 1e-3 * max(1.0, abs(float(f["loss"])))
    assert rel(m.queue[:, :B].t(), torch.from_numpy(f["keys"])) <= 1e-4 and int(m.queue_ptr) == int(f["queue_ptr"]) == B
    assert torch.equal(m.queue[:, B:].cpu(), sd["queue"][:, B:])
    got = torch.stack([named[k].grad.double().norm().cpu() for k in qkeys])
    ref = torch.from_numpy(f["grad_norms"]).double()
    live = torch.tensor([not k.endswith((".0.bias", ".3.bias")) for k in qkeys])
    relerr = ((got - ref).abs() / ref.clamp_min(1e-30))[live]
    print(f"  gradient norms of {int(live.sum())} query-encoder parameters: worst relative difference {float(relerr.max()):.2e}")
    assert float(relerr.max()) <= 5e-3
    for name in f.files:
        if name.startswith("grad."):
            assert rel(named[name[5:]].grad, torch.from_numpy(f[name])) <= 5e-3, name
    kk = [str(k) for k in f["kkeys"]]
    en = torch.stack([named[k].detach().double().norm().cpu() for k in kk])
    assert float(((en - torch.from_numpy(f["ema_norms"])).abs() / torch.from_numpy(f["ema_norms"])).max()) <= 1e-6
    assert rel(named["encoder_k.double_conv.double_conv.0.weight"].detach()[:8], torch.from_numpy(f["ema_sample"])) <= 1e-6
    with torch.no_grad():
        lg, lb, k2, q2 = m(xq2, xk2, m.queue)
    assert lg.shape == (B, 1 + K) and k2.shape == (B, 1024) and q2.shape == (B, 1024) and int(lb.sum()) == 0
    assert rel(lg, torch.from_numpy(f["fwd.logits"])) <= 1e-3
    m.zero_grad()
    loss2 = m.training_step(((xq2, xk2), 0))
    assert abs(float(loss2) - float(f["loss2"])) <= 1e-3 * max(1.0, abs(float(f["loss2"])))
    assert rel(m.queue[:, B:2 * B].t(), torch.from_numpy(f["keys2"])) <= 1e-4 and int(m.queue_ptr) == int(f["queue_ptr2"]) == 2 * B


def test_moco_validation_step_vs_reference_fixture(cuda, golden_dir):
    """Moco_v2.validation_step against the REFERENCE's own (tests/golden/moco_val_ref.npz, gen_golden.py::gen_moco_val: module in eval
    mode, two batches against val_queue): loss, top-1 / top-5 precision, the keys enqueued into val_queue, its pointer; the training
    queue stays untouched."""
    from cmunet_amd import moco as M
    from oracle import moco as OM
    f = np.load(f"{golden_dir}/moco_val_ref.npz")
    seed, B, S, K, T = int(f["seed"]), int(f["B"]), int(f["S"]), int(f["K"]), float(f["T"])
    m = M.Moco_v2(emb_dim=1024, num_negatives=K, softmax_temperature=T, dtype="f32").to(cuda)
    sd = OM.make_moco_sd(seed, K)
    sd["val_queue"], sd["val_queue_ptr"] = OM.init_queue(1024, K, seed + 1), torch.zeros(1, dtype=torch.long)
    m.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
    m.eval()
    outs = []
    for i, s_ in enumerate((seed, seed + 1)):
        x1, x2 = (t.to(cuda) for t in OM.moco_fixture_inputs(s_, B, S)[:2])
        r = m.validation_step(((x1, x2), torch.zeros(B)), i)
        outs.append(r)
        print(f"Moco_v2.validation_step {i}: loss {float(r['val_loss']):.6f} (ref {float(f[f'val_loss{i}']):.6f}) acc1 {float(r['val_acc1']):.1f} acc5 {float(r['val_acc5']):.1f}")
        assert abs(float(r["val_loss"]) - float(f[f"val_loss{i}"])) <= 1e-3 * max(1.0, abs(float(f[f"val_loss{i}"])))
        assert r["val_acc1"].shape == (1,) and float(r["val_acc1"]) == float(f[f"val_acc1_{i}"].reshape(-1)[0]) and float(r["val_acc5"]) == float(f[f"val_acc5_{i}"].reshape(-1)[0])
        assert rel(m.val_queue[:, i * B:(i + 1) * B].t(), torch.from_numpy(f[f"keys{i}"])) <= 1e-4
    assert int(m.val_queue_ptr) == int(f["val_queue_ptr"].reshape(-1)[0]) == 2 * B
    assert torch.equal(m.queue.cpu(), sd["queue"]) and int(m.queue_ptr) == 0
    mean = m.validation_epoch_end(outs)
    assert abs(float(mean["val_acc5"]) - 0.5 * (float(f["val_acc5_0"].reshape(-1)[0]) + float(f["val_acc5_1"].reshape(-1)[0]))) <= 1e-4


def test_moco_forward_enqueue_backward_order_and_batch_size_change(cuda):
    """Advisor (round 2).  (i) The reference's order is forward -> _dequeue_and_enqueue(keys) -> loss.backward() (moco2_module.py:
    287-309): ``Moco_v2.forward`` must therefore hand autograd a copy of the queue (moco2_module.py:262 ``queue.clone().detach()``),
    or the enqueue trips 'modified by an inplace operation' -- and the gradient must be the one of the PRE-enqueue queue.
    (ii) A batch size that changes between fused steps leaves the ring pointer off a multiple of the new batch: the reference fails
    with a shape error at moco2_module.py:172; here the host-side check raises and the kernel's own enqueue wraps inside the queue."""
    import torch.nn.functional as F
    from cmunet_amd import moco as MO, ops
    torch.manual_seed(0)
    m = MO.Moco_v2(emb_dim=64, num_negatives=48, softmax_temperature=0.2, dtype="f32", base_ch=16, depth=3).to(cuda).train()
    g = torch.Generator().manual_seed(1)
    xq, xk = torch.randn(8, 1, 32, 32, generator=g).to(cuda), torch.randn(8, 1, 32, 32, generator=g).to(cuda)
    queue0 = m.queue.clone()
    logits, labels, k, q = m(xq, xk, m.queue)
    q.retain_grad()
    m._dequeue_and_enqueue(k, queue_ptr=m.queue_ptr, queue=m.queue)
    assert not torch.equal(m.queue, queue0)
    F.cross_entropy(logits.float(), labels).backward()              # must not raise
    # d loss / d q through l_neg uses the queue as it was at forward time
    p = torch.softmax(logits.detach().float(), 1)
    p[:, 0] -= 1.0
    dq_ref = (p[:, :1] * k + p[:, 1:] @ queue0.t()) / (0.2 * 8)
    assert (q.grad - dq_ref).abs().max().item() <= 1e-5 * max(1.0, dq_ref.abs().max().item())
    # validation path (no_grad): no copy is made
    with torch.no_grad():
        m.eval()
        m(xq, xk, m.val_queue)
        m.train()
    # (ii) fused steps: 8 keys, then 16 keys with the pointer at 8 -> 16 | 48 but 8 % 16 != 0
    m.zero_grad()
    m.queue_ptr.zero_()
    m.training_step((xq, xk)).backward()
    assert int(m.queue_ptr) == 8
    xq2, xk2 = torch.randn(16, 1, 32, 32, generator=g).to(cuda), torch.randn(16, 1, 32, 32, generator=g).to(cuda)
    with pytest.raises(RuntimeError, match="not a multiple"):
        m.training_step((xq2, xk2))
    with pytest.raises(AssertionError):
        m.training_step((xq2[:5], xk2[:5]))                         # 48 % 5 != 0: the reference's assert (moco2_module.py:169)
    # the kernel itself stays inside the queue for any pointer: ptr 40, 16 keys -> columns 40..47 and 0..7
    D, K = 64, 48
    qd, pd = F.normalize(torch.randn(D, K, generator=g), dim=0).to(cuda), torch.tensor([40], dtype=torch.int64, device=cuda)
    guard = torch.full((D * K + 64,), 7.0, device=cuda)
    guard[:D * K] = qd.reshape(-1)
    qv = guard[:D * K].view(D, K)
    qr, kr = torch.randn(16, D, generator=g).to(cuda), torch.randn(16, D, generator=g).to(cuda)
    loss, dq = torch.empty(1, device=cuda), torch.empty(16, D, device=cuda)
    ws = torch.empty(ops._lib.lib().cmu_moco_ws_bytes(16, D, K), dtype=torch.uint8, device=cuda)
    ops.moco_infonce_enqueue(qr, kr, None, qv, pd, loss, dq, None, 0.2, ws)
    kn = F.normalize(kr, dim=1)
    assert torch.equal(guard[D * K:], torch.full((64,), 7.0, device=cuda))       # nothing past the buffer
    assert (qv[:, 40:48] - kn[:8].t()).abs().max().item() <= 1e-6 and (qv[:, 0:8] - kn[8:].t()).abs().max().item() <= 1e-6
    assert torch.equal(qv[:, 8:40], qd[:, 8:40]) and int(pd) == (40 + 16) % 48 and bool(torch.isfinite(loss).all())


def test_moco_compute_l_s_and_configure_optimizers(cuda):
    """The reference's non-fused call sequence (moco2_module.py:287-309: forward -> _compute_l_s -> backward -> optimizer.step with
    the objects of configure_optimizers): same loss and queue as the fused training_step, and the fused SGD + cosine schedule
    returned by ``configure_optimizers`` follow torch.optim.SGD + CosineAnnealingLR on a copy of the model."""
    from cmunet_amd import moco as MO

    def make():
        torch.manual_seed(0)
        return MO.Moco_v2(emb_dim=64, num_negatives=64, softmax_temperature=0.2, encoder_momentum=0.99, learning_rate=0.05, momentum=0.9,
                          weight_decay=1e-4, dtype="f32", base_ch=16, depth=3).to(cuda).train()
    g = torch.Generator().manual_seed(8)
    xq, xk = torch.randn(8, 1, 32, 32, generator=g).to(cuda), torch.randn(8, 1, 32, 32, generator=g).to(cuda)
    a, b = make(), make()
    # (a) the reference's sequence on the API-faithful methods; (b) the fused step
    a._momentum_update_key_encoder()
    output, target, keys, _ = a(img_q=xq, img_k=xk, queue=a.queue)
    loss_a = a._compute_l_s(output, target, keys, a.queue)
    loss_b = b.training_step((xq, xk))
    assert abs(float(loss_a) - float(loss_b)) <= 1e-5 * max(1.0, abs(float(loss_b)))
    assert int(a.queue_ptr) == int(b.queue_ptr) == 8 and (a.queue - b.queue).abs().max().item() <= 1e-6
    (opt,), (sched,) = a.configure_optimizers(max_epochs=10)
    ref_params = [p for p in b.parameters() if p.requires_grad]
    ref_opt = torch.optim.SGD(ref_params, 0.05, momentum=0.9, weight_decay=1e-4)
    ref_sched = torch.optim.lr_scheduler.CosineAnnealingLR(ref_opt, 10)
    for step in range(2):
        if step:
            output, target, keys, _ = a(img_q=xq, img_k=xk, queue=a.queue)
            loss_a = a._compute_l_s(output, target, keys, a.queue)
            loss_b = b.training_step((xq, xk))
        opt.zero_grad()
        ref_opt.zero_grad()
        loss_a.backward()
        loss_b.backward()
        opt.step()
        ref_opt.step()
        sched.step()
        ref_sched.step()
        assert abs(sched.get_last_lr()[0] - ref_sched.get_last_lr()[0]) <= 1e-12
    pa, pb = dict(a.named_parameters()), dict(b.named_parameters())
    for n, p in pb.items():
        if p.requires_grad:
            assert (pa[n].detach() - p.detach()).abs().max().item() <= 2e-4 * max(1.0, float(p.detach().abs().max())), n
    with pytest.raises(ValueError):
        a.configure_optimizers()                     # no Lightning trainer, no max_epochs


def test_moco_trainer_static_loss_scale_is_transparent(cuda):
    """MocoPretrainer.step(loss_scale=s): the scale multiplies the loss before backward and is divided out by the SGD kernel -- at f32
    the update equals the unscaled one (a power of two: to rounding of the momentum buffer only)."""
    from cmunet_amd import moco as M
    from cmunet_amd.pretrain import MocoPretrainer
    outs = []
    for ls in (1.0, 256.0):
        torch.manual_seed(0)
        m = M.Moco_v2(emb_dim=64, num_negatives=64, softmax_temperature=0.2, encoder_momentum=0.99, learning_rate=0.05, dtype="f32",
                      base_ch=16, depth=3).to(cuda).train()
        tr = MocoPretrainer(m)
        g = torch.Generator().manual_seed(8)
        xq, xk = torch.randn(4, 1, 32, 32, generator=g).to(cuda), torch.randn(4, 1, 32, 32, generator=g).to(cuda)
        losses = [float(tr.step(xq, xk, loss_scale=ls)) for _ in range(2)]
        outs.append((losses, tr.flat.arena.clone()))
    assert np.allclose(outs[0][0], outs[1][0], rtol=1e-6)
    assert rel(outs[1][1], outs[0][1].cpu()) <= 1e-6


def test_masked_recon_trainer_matches_autograd_path(cuda):
    """The fused trainer (arena gradients, fused AdamW) and the drop-in autograd path agree after 3 steps."""
    from cmunet_amd import model as M
    from cmunet_amd.cmunet import masked_mse_loss
    from cmunet_amd.pretrain import MaskedReconPretrainer, random_patch_mask_device
    from oracle import unet as OU
    sd = OU.make_state_dict(base_ch=16, depth=3, seed=9)
    a, b = M.UNet(base_ch=16, depth=3, dtype="f32"), M.UNet(base_ch=16, depth=3, dtype="f32")
    a.load_state_dict(sd); b.load_state_dict(sd)
    a, b = a.to(cuda).train(), b.to(cuda).train()
    tr = MaskedReconPretrainer(a, lr=1e-3, betas=(0.9, 0.95), weight_decay=0.05)
    keys = ("ln", "bias", "pos_embed", "mask_token", "cls_token")       # cmunet_config.py:84-91 as mmengine applies it: substring match, BatchNorm weights decay
    decay = [p for n, p in b.named_parameters() if not any(k in n for k in keys)]
    nodecay = [p for n, p in b.named_parameters() if any(k in n for k in keys)]
    opt = torch.optim.AdamW([{"params": decay, "weight_decay": 0.05}, {"params": nodecay, "weight_decay": 0.0}], lr=1e-3, betas=(0.9, 0.95))
    g = torch.Generator(device=cuda).manual_seed(0)
    for it in range(3):
        x = torch.randn(2, 32, 32, generator=g, device=cuda)
        mask = random_patch_mask_device(2, 32, 32, 16, 0.5, g, cuda)
        la = tr.step(x, mask)
        opt.zero_grad()
        lb = masked_mse_loss(b(x, mask=mask), 1, x, mask)
        lb.backward()
        opt.step()
        assert abs(float(la) - float(lb)) <= 1e-5 * max(1.0, abs(float(lb)))
    for (n, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        assert rel(pa, pb.detach().cpu()) <= 1e-4, n


def _spark_from_fixture(d, dt, cuda):
    from cmunet_amd import spark as S
    from oracle import unet as OU
    x = torch.from_numpy(d["x"])
    size, ratio = x.shape[-1], (float(d["mask_ratio"]) if "mask_ratio" in d.files else 0.6)
    enc = S.build_sparse_encoder("unet_sparse", input_size=size, dtype=dt)
    model = S.SparK(enc, S.UnetDecoder(dtype=dt), mask_ratio=ratio, densify_norm='', dtype=dt)
    sd = OU.make_state_dict(base_ch=64, depth=5, seed=int(d["seed"]))
    msd = model.state_dict()
    for k, v in sd.items():
        if "up_conv" in k or "conv_last" in k:
            kk = "dense_decoder." + k
            if "conv_last" in k:
                v = v[:1].clone()
        else:
            kk = "sparse_encoder.sp_cnn." + k
        assert kk in msd and msd[kk].shape == v.shape, kk
        msd[kk] = v.clone()
    off = 0
    tok = torch.from_numpy(d["tokens_flat"])
    for i, p in enumerate(model.mask_tokens):
        msd[f"mask_tokens.{i}"] = tok[off:off + p.numel()].view_as(p).clone()
        off += p.numel()
    model.load_state_dict(msd)
    return model.to(cuda).train()


@pytest.mark.parametrize("fixture", ["spark_unet", "spark_unet_m75", "spark_unet_m75_b8"])
@pytest.mark.parametrize("dt", ["f32", "f16", "bf16"])
def test_spark_step_vs_reference_fixture(cuda, golden_dir, dt, fixture):
    """SparK sparse masked-conv step against fixtures produced by the reference's own SparK code (oracle/gen_golden.py:
    Pretraining/Spark imported behind stubs): ``spark_unet`` = 64 px at mask ratio 0.6, ``spark_unet_m75`` = 128 px at BASELINE
    config 5's ratio 0.75 (16 of 64 patches kept per image; the tile-skipping conv path has whole tiles to skip there).
    ``spark_unet_m75_b8`` = the same configuration at batch 8 (round-2 review): 128 active positions per channel in the
    bottleneck's sparse BatchNorm instead of 32.  On all three, every tensor's gradient norm is held to the stated bar itself
    (f32 2e-3, f16 10 %, bf16 15 %; measured worst 9e-4 / 4.2e-2 / 8.2e-2, profiles/r03_parity.txt) -- the factor of 5 round 2
    allowed was only ever needed by the ConvTranspose biases, whose gradient is a cancellation remainder (see below).
    f32: max-norm 2e-3; f16 / bf16: loss 1.5e-2 / 3e-2 (storage rounding through 18 sparse BatchNorms)."""
    d = np.load(f"{golden_dir}/{fixture}.npz")
    x, active = torch.from_numpy(d["x"]), torch.from_numpy(d["active"]).bool()
    model = _spark_from_fixture(d, dt, cuda)
    if fixture == "spark_unet":
        assert model.fmap_h == 4 and model.len_keep == 6 and model.hierarchy == 5
    else:
        assert model.fmap_h == 8 and model.len_keep == 16 and model.mask_ratio == 0.75
        assert x.shape[0] == (8 if fixture.endswith("_b8") else 2)
    slack = 1.0        # (round 2 allowed 5x the per-tensor bar on the two small fixtures; with the ConvTranspose biases measured on
    #                    their layer's scale -- see below -- every fixture meets the stated bar itself: profiles/r03_parity.txt)
    loss = model(x.to(cuda), active_b1ff=active.to(cuda))
    loss.backward()
    ltol, gtol = {"f32": (2e-4, 2e-3), "f16": (1.5e-2, 0.10), "bf16": (3e-2, 0.15)}[dt]
    assert abs(float(loss) - float(d["loss"])) <= ltol * max(1.0, abs(float(d["loss"]))), (float(loss), float(d["loss"]))
    named = dict(model.named_parameters())
    worst, errs = 0.0, []
    ref_norm = {str(k): float(n) for k, n in zip(d["grad_norm_keys"], d["grad_norms"])}
    for k, n in ref_norm.items():
        if ".0.bias" in k or ".3.bias" in k or n < 1e-7:
            continue
        got = named[k].grad.norm().item()
        if k.endswith("up_sample.bias"):
            # A ConvTranspose bias feeds a conv + training-mode BatchNorm, which is invariant to a constant shift of its input
            # except through the zero padding at the image border: the true gradient is a small remainder of sums that cancel
            # (norm ~1e-3 against ~1 for the layer's weight), so its RELATIVE error only measures the cancellation.  It is held
            # to the bar relative to the gradient norm of the same layer's weight instead.
            scale = ref_norm[k[:-len("bias")] + "weight"]
            e = abs(got - n) / scale
        else:
            e = abs(got - n) / n
        errs.append((e, k, got, n))
        worst = max(worst, e)
    errs.sort(reverse=True)
    top = "; ".join(f"{k}: {e:.2e}" for e, k, _, _ in errs[:4])
    assert errs[0][0] <= gtol * slack, f"worst per-tensor gradient-norm errors: {top}"
    tg = torch.cat([p.grad.flatten() for p in model.mask_tokens]).cpu()
    # Sparse BatchNorm here normalises as few as 6 active positions per channel, so the reference's OWN f32 gradients sit
    # ~3e-3 from its float64 run (fixture keys *64).  The f32 bar is therefore stated against the float64 truth:
    # our error may not exceed 5x the reference's own f32 error (floor 2e-3) -- both are
    # single draws of rounding noise amplified by a condition number of ~5e4.
    tg64, tg32 = torch.from_numpy(d["token_grads64_flat"]), torch.from_numpy(d["token_grads_flat"])
    if dt == "f32":
        assert rel(tg, tg64) <= max(5 * rel(tg32, tg64), 2e-3), (rel(tg, tg64), rel(tg32, tg64))
        for k in ("dense_decoder.conv_last.weight", "sparse_encoder.sp_cnn.down_conv1.double_conv.double_conv.0.weight",
                  "sparse_encoder.sp_cnn.down_conv1.double_conv.double_conv.1.bias"):
            g64, g32 = torch.from_numpy(d["grad64." + k]), torch.from_numpy(d["grad." + k])
            e, e_ref = rel(named[k].grad, g64), rel(g32, g64)
            print(f"[spark f32 {fixture}] {k}: err vs f64 {e:.2e} (reference f32: {e_ref:.2e})")
            assert e <= max(5 * e_ref, 2e-3), (k, e, e_ref)
        assert rel(model.state_dict()["sparse_encoder.sp_cnn.double_conv.double_conv.4.running_var"],
                   torch.from_numpy(d["bott_running_var"])) <= 1e-3
    else:
        assert rel(tg, tg64) <= 0.3
    print(f"[spark {dt} {fixture}] loss {float(loss):.5f} vs {float(d['loss']):.5f}, worst grad-norm err {worst:.2e}")
    _parity_record(f"spark step vs reference fixture {fixture} {dt}: loss {float(loss):.6f} vs {float(d['loss']):.6f}, "
                   f"worst per-tensor gradient-norm errors {top} (bar {gtol * slack:.2g})")


def test_spark_sync_batchnorm_two_ranks(cuda):
    """SparseSyncBatchNorm2d (sbn=True: the bottleneck's two BatchNorms, SURVEY A-10) on two ranks (gloo, both on cuda:0).
    (i) Both ranks fed the SAME batch: the all-reduced sums and counts are twice the local ones, so loss, gradients and
    running statistics must equal the single-process sbn=False step (to the float32 rounding of the exchanged totals).  (ii) Different batches per rank: the bottleneck's running
    statistics come out identical on both ranks (they are the same all-reduced numbers) and differ from the unsynchronised
    run, while a down block's stay per-rank."""
    import os
    import socket
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import os, sys, torch
sys.path.insert(0, %r)
import torch.distributed as dist
from cmunet_amd import spark as S
world = int(os.environ.get("WORLD_SIZE", "1"))
rank = int(os.environ.get("RANK", "0"))
if world > 1:
    dist.init_process_group("gloo", rank=rank, world_size=world)
mode, sbn, out = sys.argv[1], sys.argv[2] == "1", sys.argv[3]
torch.manual_seed(5)
enc = S.build_sparse_encoder("unet_sparse", input_size=64, sbn=sbn, base_ch=16, depth=3, dtype="f32")
model = S.SparK(enc, S.UnetDecoder(base_ch=16, depth=3, dtype="f32"), mask_ratio=0.6, densify_norm="", dtype="f32").cuda().train()
g = torch.Generator().manual_seed(11 if mode == "same" else 11 + rank)
x = torch.randn(4, 1, 64, 64, generator=g)
f = model.fmap_h
active = torch.zeros(4, 1, f, f, dtype=torch.bool)
for b in range(4):
    idx = torch.randperm(f * f, generator=g)[:model.len_keep]
    active[b, 0].view(-1)[idx] = True
loss = model(x.cuda(), active_b1ff=active.cuda())
loss.backward()
sd = model.state_dict()
bk = [k for k in sd if "sp_cnn.double_conv" in k and k.endswith("running_mean")]
dk = [k for k in sd if "sp_cnn.down_conv1" in k and k.endswith("running_mean")]
gp = dict(model.named_parameters())
gk = [k for k in gp if "sp_cnn.double_conv.double_conv.0.weight" in k or "sp_cnn.down_conv1.double_conv.double_conv.3.weight" in k]
torch.save({"loss": loss.detach().cpu(), "bott": [sd[k].cpu() for k in bk], "down": [sd[k].cpu() for k in dk],
            "grads": [gp[k].grad.cpu() for k in gk]}, out + f".{rank}")
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
''' % root

    def run(mode, sbn, world):
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "r")
            procs = []
            with socket.socket() as so:
                so.bind(("127.0.0.1", 0))
                port = so.getsockname()[1]
            for rk in range(world):
                env = dict(os.environ)
                if world > 1:
                    env.update(WORLD_SIZE=str(world), RANK=str(rk), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
                else:
                    for k in ("WORLD_SIZE", "RANK", "MASTER_ADDR", "MASTER_PORT"):
                        env.pop(k, None)
                procs.append(subprocess.Popen([sys.executable, "-c", code, mode, "1" if sbn else "0", out], env=env))
            try:
                for p in procs:
                    assert p.wait(timeout=300) == 0
            finally:                                  # a failed / hung rank must not leave its peer holding the GPU
                for p in procs:
                    if p.poll() is None:
                        p.kill()
                    p.wait()
            return [torch.load(out + f".{rk}") for rk in range(world)]

    single = run("same", False, 1)[0]
    same = run("same", True, 2)
    def near(a, b, tol):
        return (a.double() - b.double()).abs().max().item() <= tol * max(b.double().abs().max().item(), 1e-6)
    for r in same:
        assert near(r["loss"], single["loss"], 1e-6)
        for a, b in zip(r["bott"] + r["down"], single["bott"] + single["down"]):
            assert near(a, b, 1e-5)
        for a, b in zip(r["grads"], single["grads"]):
            assert near(a, b, 2e-4)
    diff = run("diff", True, 2)
    nosync = run("diff", False, 2)
    for a, b in zip(diff[0]["bott"], diff[1]["bott"]):
        assert torch.equal(a, b)                                  # synchronised statistics: the same numbers on both ranks
    assert not torch.equal(diff[0]["down"][0], diff[1]["down"][0])   # the down blocks' SparseBatchNorm2d stay per-rank
    assert not torch.equal(diff[0]["bott"][0], nosync[0]["bott"][0])


def _joint224_hip_vs_oracle(cuda, mode):
    """One joint step of tests/joint224_case.py on the HIP path and on the oracle -> (errors per gradient, losses, oracle losses)."""
    import joint224_case as J
    model, sd, (img, img_t, mask, rw, rb) = J.build(mode)
    model = model.to(cuda)
    losses = model(img.to(cuda), mode='loss', img_t=img_t.to(cuda), mask=mask.to(cuda), reduce_w=rw.to(cuda), reduce_b=rb.to(cuda))
    (losses['loss_ct'] + losses['loss_rc']).backward()
    ref_losses, ref = J.oracle_step(sd, (img, img_t, mask, rw, rb))
    params = dict(model.named_parameters())
    errs = {k: J.rel_l2(params[k].grad.detach().cpu(), ref[k]) for k in J.KEYS}
    return J, sd, errs, {k: float(v.detach()) for k, v in losses.items()}, ref_losses


def _joint224_spread(J, sd, mode):
    """Per-tensor MAX over six perturbation seeds of how far the oracle's own gradient moves under four-ulp weight noise
    (tests/golden/joint224_spread.npz, written by tests/gen_joint224_spread.py for exactly these weights)."""
    import os
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "joint224_spread.npz"))
    assert list(d["keys"]) == list(J.KEYS)
    assert abs(float(d[f"{mode}_checksum"]) - J.checksum(sd)) <= 1e-9 * abs(float(d[f"{mode}_checksum"])), "the spread fixture was made for other weights"
    return dict(zip(J.KEYS, d[f"{mode}_spread"].max(0))), len(d["seeds"])


class _HipForwardTaps:
    """oracle.unet.TAP object that forces the HIP path's forward onto the oracle: every 3x3 conv's raw output takes the VALUE the HIP
    kernel stored (the derivative still flows through the oracle's own convolution), and every ReLU takes the HIP path's gate
    (a = z * gate, derivative = gate) and the value its consumers compute, max(fmaf(y, scale, shift), 0) in fp32."""

    def __init__(self, layers):
        self.layers, self.seen = layers, set()

    def conv(self, key, y):
        if key not in self.layers:           # the target encoder: no gradient flows through it, its forward stays the oracle's own
            return y
        yh = self.layers[key][0].to(y.dtype)
        self.seen.add(key)
        return y + (yh - y).detach()

    def act(self, key, z):
        if key not in self.layers:
            return F.relu(z)
        ah = self.layers[key][1].to(z.dtype)
        a = z * (ah > 0).to(z.dtype)
        return a + (ah - a).detach()


def _collect_hip_layers(last_ctx):
    """{conv prefix: (raw output (B,C,H,W) fp32 on the CPU, activated value max(fmaf(y, scale, shift), 0) as the consumers compute it)}
    of every conv + BatchNorm layer of the online encoder and the two decoders, from the engine's saved forward state."""
    out = {}

    def add(s):
        y = s["y"]
        raw = y.buf[..., y.coff:y.coff + y.C].float()
        z = (raw.double() * y.scale.double() + y.shift.double()).float()            # = fmaf(y, scale, shift) (exact product, one rounding)
        out[s["pconv"]] = (raw.permute(0, 3, 1, 2).contiguous().cpu(), torch.clamp_min(z, 0).permute(0, 3, 1, 2).contiguous().cpu())
    ectx, pctx, fctx = last_ctx
    for lv in ectx["levels"]:
        add(lv["s1"]); add(lv["s2"])
    add(ectx["bott"]["s1"]); add(ectx["bott"]["s2"])
    for dctx in (pctx, fctx):
        for lv in dctx["levels"]:
            add(lv["s1"]); add(lv["s2"])
    return out


@pytest.mark.parametrize("case", [("f32", 224, 4, 32), ("f16", 224, 4, 32), ("f16", 512, 2, 64), ("bf16", 224, 4, 32)])
def test_masked_recon_step_gate_forced_backward(cuda, case):
    """The HEADLINE step (BASELINE config 2: masked reconstruction through ``MaskedReconPretrainer``'s engine path -- mask fused into the
    first conv, skips written into the concat buffers, head / pool fusions, first-layer weight gradient with the recomputed raw
    output) at 224 x 224 (partial tiles: 112 / 56 / 28 / 14-pixel levels), base 32, depth 5, bs 4, with the oracle's float64 backward
    pass on the HIP path's OWN forward (every conv output, ReLU gate and activated value taken from the engine's saved state, see
    test_cmunet_joint_step_gate_forced_backward).  EVERY parameter gradient of the network is compared: f32 storage within 1e-4
    relative L2 (what is left is the backward kernels' summation order); f16 storage -- weights rounded to f16 on both sides, the
    HIP path storing every activation gradient in f16 -- within 1e-2 (measured 2.8e-3).  Third case: the REFERENCE network (base 64,
    depth 5, 31 M parameters) at the bench's 512 x 512 geometry and arithmetic (f16; whole tiles: the weight-resident 64-channel
    tile and the NB = 128 persistent forms the bench runs), bs 2 so that the float64 oracle finishes in about a minute."""
    import joint224_case as J
    dt, S, B, base = case
    from cmunet_amd import model as M, ops
    from cmunet_amd.pretrain import MaskedReconPretrainer, create_random_patch_mask
    from oracle import cmunet as OC, unet as OU
    torch.manual_seed(0)
    net = M.UNet(out_classes=2, dtype=dt, base_ch=base, depth=5).train()
    gw = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for n, p in net.named_parameters():
            if p.dim() == 1 and (".1." in n or ".4." in n):
                p.add_(0.2 * torch.randn(p.shape, generator=gw))
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    g = torch.Generator().manual_seed(1)
    img = torch.randn(B, S, S, generator=g)
    mask = torch.from_numpy(create_random_patch_mask(B, S, 16, 0.6, np.random.RandomState(2)))
    net = net.to(cuda)
    scale = 1.0 if dt == "f32" else 1024.0
    tr = MaskedReconPretrainer(net, lr=1e-3, amp=False, loss_scale=scale)
    eng = tr.engine
    eng.prepack(tr.sd)
    x, m = img.to(cuda), mask.to(cuda)
    logits, ctx = eng.unet_forward(tr.sd, x, True, m, mask_per_sample=False)
    loss, dlogits = torch.zeros(1, device=cuda), torch.empty_like(logits)
    ws = torch.empty(ops._lib.lib().cmu_masked_mse_ws_bytes(B, S), dtype=torch.uint8, device=cuda)
    ops.masked_mse_fwd_bwd(logits, 1, x, m, loss, dlogits, scale, ws, None)
    tdt = ops.TORCH_DT[ops.dt_code(dt)]
    layers = {}

    def add(st):
        y = st["y"]
        raw = y.buf[..., y.coff:y.coff + y.C].float()
        z = (raw.double() * y.scale.double() + y.shift.double()).float()            # = fmaf(y, scale, shift)
        # (the consumers round this to the storage type while staging; the oracle keeps the unrounded value so that its max-pools pick
        # the element the HIP pool picks -- it compares the fp32 values -- instead of tying on equal 16-bit roundings)
        act = torch.clamp_min(z, 0)
        layers[st["pconv"]] = (raw.permute(0, 3, 1, 2).contiguous().cpu(), act.permute(0, 3, 1, 2).contiguous().cpu(),
                               (z > 0).permute(0, 3, 1, 2).contiguous().cpu())
    for lv in ctx["enc"]["levels"]:
        add(lv["s1"]); add(lv["s2"])
    add(ctx["enc"]["bott"]["s1"]); add(ctx["enc"]["bott"]["s2"])
    for lv in ctx["dec"]["levels"]:
        add(lv["s1"]); add(lv["s2"])
    assert len(layers) == 18
    eng.grad_target, eng.grad_prefix = tr.flat.grad_views, ""
    try:
        eng.unet_backward(tr.sd, ctx, dlogits)
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
    # the oracle in float64 on the weights the kernels multiply with (16-bit storage: the packed copies hold the rounded weights)
    # (round 6: float64 where the bar needs it -- f32 storage, 1e-4 -- and float32 for the 16-bit cases, whose bars are 1e-2 / 1e-1: the fp32
    # oracle sits ~1e-5 from the fp64 one with the gates forced, DESIGN section 2, and runs the CPU convs several times faster: 47 s -> 12 s of the
    # driver's GPU test run)
    odt = torch.float64 if dt == "f32" else torch.float32

    def wq(k, v):
        if not v.is_floating_point():
            return v.clone()
        v = v.to(tdt).to(odt) if (v.dim() == 4 and dt != "f32" and "conv_last" not in k) else v.to(odt)
        return v.requires_grad_(True) if "running" not in k else v
    osd = {k: wq(k, v) for k, v in sd.items()}
    OU.TAP = taps
    try:
        out = OU.unet_forward(img.to(odt) * (1 - mask[0]).to(odt), osd, training=True)
        ref_loss = OC.masked_mse(out[:, 1], img.to(odt), mask)
        ref_loss.backward()
    finally:
        OU.TAP = None
    assert taps.seen == set(layers)
    assert abs(float(loss) - float(ref_loss.detach())) <= {"f32": 2e-5, "f16": 2e-3, "bf16": 2e-2}[dt] * max(1.0, abs(float(ref_loss.detach())))
    # (bf16, the opt-in storage type: eight times f16's rounding step -- with the gates forced this is a bar on the kernels' arithmetic
    # alone, where the unforced whole-step comparison has to allow 15-35 % for gate flips)
    bar = {"f32": 1e-4, "f16": 1e-2, "bf16": 1e-1}[dt]          # measured worst: 1.2e-5 / 2.8e-3 / 5.9e-2 (a ConvTranspose bias each time)
    worst, n = ("", 0.0), 0
    for k, v in osd.items():
        if not (torch.is_tensor(v) and v.requires_grad) or k.endswith(("0.bias", "3.bias")):     # (conv biases in front of BN: identically zero)
            continue
        got = tr.flat.grad_views[k].detach().double().cpu() / scale
        e = J.rel_l2(got, v.grad)
        if k.endswith("up_sample.bias"):
            # a bias in front of conv + training-mode BatchNorm: its gradient is the border remainder of sums that cancel (norm ~1e-3
            # of its layer's weight gradient) -- held to the bar on the layer's scale
            wg = osd[k[:-len("bias")] + "weight"].grad
            e = (got - v.grad).norm().item() / max(v.grad.norm().item(), 1e-2 * wg.norm().item())
        n += 1
        if e > worst[1]:
            worst = (k, e)
        assert e <= bar, f"d{k}: relative L2 error {e:.2e} with the gates forced (bar {bar:.0e}, {dt})"
    print(f"[masked-recon step @ {S} {dt} bs {B} base {base}, oracle backward on the HIP forward] {n} parameter gradients, worst {worst[0]}: {worst[1]:.2e} (bar {bar:.0e})")
    _parity_record(f"masked-reconstruction step (engine path of MaskedReconPretrainer) {dt} at {S}x{S}, bs {B}, base {base}, depth 5, {'float64' if dt == 'f32' else 'float32'} oracle backward on the HIP "
                   f"path's own forward values and ReLU gates: {n} parameter gradients, worst relative L2 error {worst[1]:.2e} ({worst[0]}), bar {bar:.0e}")


@pytest.mark.parametrize("mode", ["random65"] + (["tie_free"] if SLOW else []))
def test_cmunet_joint_step_gate_forced_backward(cuda, mode):
    """The twin of test_cmunet_joint_step_reference_geometry that CAN fail on the conv chain.  Two fp32 implementations of this step
    cannot agree to better than ~3e-3 on the encoder's gradients because a rounding-sized change of the forward flips ReLU gates
    (that test's docstring) -- so here the oracle's float64 backward pass runs on the HIP path's OWN forward: every conv output,
    every gate and every activated value of the online encoder and both decoders are taken from the engine's saved state
    (oracle.unet.TAP), the max-pools then pick the same elements, and what is left between the two gradients is the arithmetic of the
    HIP backward kernels alone (data / weight gradients of the partial-tile, narrow and slim dispatch of this geometry, BatchNorm
    backward, pool backward with two skip gradients, ConvTranspose, the necks' skinny GEMMs).  ONE flat bar at f32 for every tensor
    -- head, projector, both decoders, the deepest ConvTranspose, the first encoder layer: 1e-4 (measured: 1e-8 ... 1.9e-5; the
    review asked for 1e-4 on head / projector and 1e-3 on the conv chain)."""
    import joint224_case as J
    from oracle import unet as OU
    model, sd, (img, img_t, mask, rw, rb) = J.build(mode)
    model = model.to(cuda)
    model.keep_ctx = True
    losses = model(img.to(cuda), mode='loss', img_t=img_t.to(cuda), mask=mask.to(cuda), reduce_w=rw.to(cuda), reduce_b=rb.to(cuda))
    layers = _collect_hip_layers(model.last_ctx)
    (losses['loss_ct'] + losses['loss_rc']).backward()
    model.last_ctx = None
    assert len(layers) == 10 + 8 + 8
    taps = _HipForwardTaps(layers)
    OU.TAP = taps
    try:
        # float64 oracle: its own rounding is out of the picture (sd64 holds the same fp32 weights)
        sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
        ref_losses, ref = J.oracle_step(sd64, (img.double(), img_t.double(), mask, rw.double(), rb.double()))
    finally:
        OU.TAP = None
    assert taps.seen == set(layers), "a layer of the oracle ran without the HIP path's forward"
    assert abs(float(losses['loss_rc'].detach()) - ref_losses['loss_rc']) <= 2e-5 * max(1, abs(ref_losses['loss_rc']))
    assert abs(float(losses['loss_ct'].detach()) - ref_losses['loss_ct']) <= 2e-4 * max(1, abs(ref_losses['loss_ct']))
    params = dict(model.named_parameters())
    errs = {k: J.rel_l2(params[k].grad.detach().cpu(), ref[k]) for k in J.KEYS}
    line = ", ".join(f"{k}: {e:.2e}" for k, e in errs.items())
    print(f"[joint step @ 224 {mode}, f32, oracle backward on the HIP forward] relative L2 error of the gradients: " + line)
    _parity_record(f"CM_UNet joint step f32 at the reference geometry (224x224, bs 4, base {J224_BASE}; input mask '{mode}'), float64 oracle backward on the HIP "
                   "path's own forward values and ReLU gates (one flat bar, 1e-4, on every tensor): " + line)
    for k, e in errs.items():
        assert e <= 1e-4, f"d{k}: relative L2 error {e:.2e} with the gates forced (bar 1e-4)"


@pytest.mark.parametrize("mode", ["random65"] + (["tie_free"] if SLOW else []))
def test_cmunet_joint_step_reference_geometry(cuda, mode):
    """The joint step at the reference's own geometry (224 x 224 crops, depth 5, projector in_channels = 224*224 =
    50,176 -> 1,536 -> 256, cmunet_config.py:18-26; mask ratio 0.65 -> 127 of 196 patches) with base 32 channels, f32 storage,
    against the oracle: both losses and gradients at every stage of the chain (head, projector, decoders, encoder).

    How far can two correct fp32 implementations be apart here?  Round 4 measured the mechanism (tests/gen_joint224_spread.py and
    the probes recorded in DESIGN.md section 2): the oracle's own encoder / deep-decoder gradients move by 2 ... 7e-3 (relative L2) when
    its weights are perturbed by four ulps -- in float64 arithmetic just as much as in float32, with the max-pools replaced by
    average pools just as much, and with mask[0] = 0 (mode 'tie_free': nothing zeroed in the input, no tied pool windows) just as
    much; with the ReLUs replaced by softplus the same perturbation moves them by 1e-5.  So it is not rounding that accumulates and
    not tied max-pool windows (round 3's reading): the gradient of a ReLU network is a DISCONTINUOUS function of its weights -- a
    perturbation of relative size eps flips a share ~eps * depth of the gates, each flip changes the backward signal by O(1), and the
    gradient moves by ~sqrt(eps * depth): ~3e-3 for eps = 2^-22 and ~2e-3 for fp32 rounding, whatever the implementation.  Tensors
    with no gate between them and the loss (the heads' last layers) move by 1e-6 ... 1e-4 instead.
    The bars: every tensor is held to max(floor, 2 x the MAX over six perturbation seeds of the oracle's own spread), the spread
    taken from a committed fixture made for exactly these weights (floor 1e-4; 1e-3 inside the conv chain).  What tells a kernel
    error from a flip on the conv chain is the gate-forced twin below (test_cmunet_joint_step_gate_forced_backward)."""
    J, sd, errs, losses, ref_losses = _joint224_hip_vs_oracle(cuda, mode)
    assert abs(losses['loss_rc'] - ref_losses['loss_rc']) <= 2e-4 * max(1, abs(ref_losses['loss_rc']))
    assert abs(losses['loss_ct'] - ref_losses['loss_ct']) <= 2e-3 * max(1, abs(ref_losses['loss_ct']))
    sens, nseeds = _joint224_spread(J, sd, mode)
    line = ", ".join(f"{k}: {errs[k]:.2e} (oracle spread: {sens[k]:.2e})" for k in errs)
    print(f"[joint step @ 224 {mode}, f32 vs oracle] relative L2 error of the gradients: " + line)
    _parity_record(f"CM_UNet joint step f32 at the reference geometry (224x224, bs 4, base 32; input mask '{mode}') vs the fp32 oracle, relative L2 "
                   f"error per gradient (in brackets: the max over {nseeds} seeds of how far the oracle's own gradient moves under four-ulp weight "
                   "noise, tests/golden/joint224_spread.npz): " + line)
    for k, e in errs.items():
        floor = 1e-3 if J.in_conv_chain(k) else 1e-4
        assert e <= max(floor, 2.0 * sens[k]), f"d{k}: relative L2 error {e:.2e} against a spread of {sens[k]:.2e} (max over {nseeds} seeds)"




def test_adamw_decay_groups_one_step_batchnorm_weight_moves_by_lr_wd_gamma(cuda):
    """the CM-UNet trainers' AdamW follows cmunet_config.py:84-91 as mmengine applies it -- only names containing
    'bias' (/'ln'/...) are exempt, so a BatchNorm WEIGHT with a zero gradient moves by exactly lr * wd * gamma in one step (decoupled
    decay; Adam's own update of a zero gradient is 0), a bias does not move at all."""
    from cmunet_amd import model as M
    from cmunet_amd.pretrain import JointPretrainer, MaskedReconPretrainer
    from cmunet_amd import cmunet as C
    lr, wd = 1e-2, 0.05

    def check(tr, tag):
        g = torch.Generator().manual_seed(5)
        with torch.no_grad():
            tr.flat.arena.copy_((torch.rand(tr.flat.arena.shape, generator=g) + 0.5).to(cuda))   # gammas away from 1, nothing zero
        w0 = tr.flat.arena.clone()
        tr.flat.grad.zero_()
        tr.opt.step()
        torch.cuda.synchronize()
        n_bn = n_bias = 0
        for n in tr.flat.names:
            off, cnt = tr.flat.offsets[n]
            a, b = w0[off:off + cnt], tr.flat.arena[off:off + cnt]
            if "bias" in n:
                assert torch.equal(a, b), (tag, n)
                n_bias += 1
            else:
                want = a * (1.0 - lr * wd)
                assert float((b - want).abs().max()) <= 1e-7, (tag, n, float((b - want).abs().max()))
                assert float((a - b).abs().min()) > 0.4 * lr * wd, (tag, n)     # it did move (gamma >= 0.5)
                n_bn += tr.flat.params[n].dim() == 1
        return n_bn, n_bias
    net = M.UNet(out_classes=2, dtype="f32", base_ch=16, depth=3).to(cuda)
    n_bn, n_bias = check(MaskedReconPretrainer(net, lr=lr, weight_decay=wd, amp=False), "recon")
    assert n_bn == 2 * (2 * 3 - 1) and n_bias > n_bn          # five DoubleConvs of a depth-3 UNet: ten BatchNorm weights, all decayed
    torch.manual_seed(0)
    model = C.build_model(C.cmunet_config(img_size=32, base_ch=16, depth=3, dtype="f32")).to(cuda).train()
    n_bn, _ = check(JointPretrainer(model, lr=lr, weight_decay=wd, amp=False), "joint")
    assert n_bn > 10


def test_moco_nonfused_api_pieces_vs_torch(cuda):
    """Round 5: ``Moco_v2.forward`` / ``_compute_l_s`` / ``validation_step`` no longer call ATen compute ops -- the
    row normalisation, the [q.k | q @ queue] / T logits and the mean cross entropy (+ precision@k from the target's rank) run on the library's
    kernels.  Each piece, forward and backward, against the torch expression of moco2_module.py:256-285 in float64."""
    import torch.nn.functional as F
    from cmunet_amd import moco as MO, ops
    g = torch.Generator().manual_seed(3)
    B, D, K, T = 48, 128, 512, 0.2
    x = torch.randn(B, D, generator=g)
    x[5] = 0.0                                                     # a zero row: F.normalize's eps clamp
    xq = x.clone().to(cuda).requires_grad_(True)
    y = MO._L2NormRowsFn.apply(xq)
    go = torch.randn(B, D, generator=g)
    y.backward(go.to(cuda))
    xr = x.double().requires_grad_(True)
    yr = F.normalize(xr, dim=1)
    yr.backward(go.double())
    assert rel(y, yr.float()) <= 1e-6 and rel(xq.grad[torch.arange(B) != 5], xr.grad[torch.arange(B) != 5].float()) <= 1e-5
    # logits
    q = F.normalize(torch.randn(B, D, generator=g), dim=1)
    k = F.normalize(torch.randn(B, D, generator=g), dim=1)
    queue = F.normalize(torch.randn(D, K, generator=g), dim=0)
    qc = q.clone().to(cuda).requires_grad_(True)
    logits = MO._MocoLogitsFn2.apply(qc, k.to(cuda), queue.to(cuda), T)
    gl = torch.randn(B, K + 1, generator=g)
    logits.backward(gl.to(cuda))
    qd = q.double().requires_grad_(True)
    lr_ = torch.cat([(qd * k.double()).sum(1, keepdim=True), qd @ queue.double()], 1) / T
    lr_.backward(gl.double())
    assert logits.shape == (B, K + 1) and rel(logits, lr_.float()) <= 1e-6 and rel(qc.grad, qd.grad.float()) <= 1e-5
    # cross entropy (mean) with an incoming gradient that is not 1, and the target's rank
    lo = (torch.randn(B, K + 1, generator=g) * 3).to(cuda).requires_grad_(True)
    tgt = torch.randint(0, K + 1, (B,), generator=g).to(cuda)
    loss = MO.row_cross_entropy(lo, tgt)
    (loss * 2.5).backward()
    lod = lo.detach().double().cpu().requires_grad_(True)
    lref = F.cross_entropy(lod, tgt.cpu())
    (lref * 2.5).backward()
    assert loss.shape == () and abs(float(loss) - float(lref)) <= 1e-5 * abs(float(lref)) and rel(lo.grad, lod.grad.float()) <= 1e-5
    _, _, rank = ops.row_cross_entropy(lo.detach(), tgt, want_grad=False, want_rank=True)
    a1, a5 = MO.precision_from_rank(rank, (1, 5))
    r1, r5 = MO.precision_at_k(lo.detach(), tgt, (1, 5))
    assert float(a1) == float(r1) and float(a5) == float(r5)
    assert torch.equal(rank.cpu().long(), (lo.detach().cpu() > lo.detach().cpu().gather(1, tgt.cpu().view(-1, 1))).sum(1))


def test_spark_patchify_unpatchify_are_the_reference_einsums(cuda):
    """Spark/spark.py:133-148 on cmu_patchify: bit for bit the reference's einsum + reshape, and inverse of each other."""
    from cmunet_amd import spark as S
    enc = S.build_sparse_encoder("unet_sparse", input_size=64, base_ch=16, depth=5, dtype="f32")
    model = S.SparK(enc, S.UnetDecoder(base_ch=16, depth=5, dtype="f32"), mask_ratio=0.75, densify_norm="", dtype="f32").to(cuda)
    p, h, w = model.downsample_raito, model.fmap_h, model.fmap_w
    g = torch.Generator().manual_seed(2)
    for C in (1, 3):
        x = torch.randn(3, C, h * p, w * p, generator=g).to(cuda)
        ref = torch.einsum('bchpwq->bhwpqc', x.reshape(3, C, h, p, w, p)).reshape(3, h * w, C * p * p)
        got = model.patchify(x)
        assert got.shape == ref.shape and torch.equal(got, ref)
        assert torch.equal(model.unpatchify(got), x)
        assert torch.equal(model.unpatchify(ref), torch.einsum('bhwpqc->bchpwq', ref.reshape(3, h, w, p, p, C)).reshape(3, C, h * p, w * p))


@pytest.mark.parametrize("case", [("f32", 128, 8), ("f16", 128, 8)] + ([("f16", 256, 2)] if SLOW else []))
def test_spark_step_gate_forced_backward(cuda, case):
    """SparK's list-driven backward pinned the way the dense paths are (round-5 review, item 2): BASELINE config 5's model (base 64, depth 5,
    ``unet_sparse`` + ``UnetDecoder``, mask ratio 0.75) at 128 x 128, bs 8 (and 256 x 256, bs 2: 16-pixel patches span whole 16 x 32 tiles at level 1,
    so the tile-list kernels, the gather levels and the patch-organised passes all run), every list-driven switch at its default (on), with the
    oracle's FLOAT64 backward (``oracle/spark.py``: Spark/encoder.py:20-52, spark.py:88-131) run on the HIP path's OWN forward: the masked raw
    output, the ReLU gates and the activated values of all ten sparse-encoder convs and eight decoder convs come from the step's saved state
    (``SparK.keep_ctx``) through ``oracle.unet.TAP``.  A gradient of a ReLU network is discontinuous in its weights, and a sparse
    BatchNorm over a few hundred active positions amplifies every flip -- so the unforced comparisons of this path need 10 % bars at f16 and cannot
    tell a kernel that is off by a few per cent from a flipped gate; this one can.  One flat bar on EVERY parameter gradient (87 tensors incl. the
    mask tokens): 1e-4 (f32) / 1e-2 (f16) relative L2.  Measured (profiles/r06_parity.txt): f32 worst 5.0e-6, f16 1.3e-3; the 256-pixel case
    (CMU_TEST_SLOW=1; 14 s of float64 on the CPU) 1.25e-3."""
    dt, S, B = case
    from cmunet_amd import spark as SP, ops
    from oracle import spark as OS, unet as OU
    torch.manual_seed(0)
    enc = SP.build_sparse_encoder("unet_sparse", input_size=S, dtype=dt)
    model = SP.SparK(enc, SP.UnetDecoder(dtype=dt), mask_ratio=0.75, densify_norm="", dtype=dt).train()
    gw = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.dim() == 1 and (".1." in n or ".4." in n):
                p.add_(0.2 * torch.randn(p.shape, generator=gw))          # BatchNorm weights / biases off their init values
    sd = {k: v.detach().clone() for k, v in model.state_dict().items() if k != "config"}
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, 1, S, S, generator=g)
    active = model.mask(B, "cpu", torch.Generator().manual_seed(5))
    model = model.to(cuda)
    scale = 1.0 if dt == "f32" else 1024.0
    model.grad_scale, model.keep_ctx = scale, True
    loss = model(x.to(cuda), active_b1ff=active.to(cuda))
    loss.backward()
    torch.cuda.synchronize()
    ctx = model.last_ctx
    tdt = ops.TORCH_DT[ops.dt_code(dt)]
    layers = {}

    def add(st, act_mask):
        y = st["y"]
        raw = y.buf[..., y.coff:y.coff + y.C].float()
        z = (raw.double() * y.scale.double() + y.shift.double()).float()            # = fmaf(y, scale, shift) as the consumers compute it
        if act_mask is not None:                                                   # sparse layers: nothing lives at a masked position
            m = act_mask.repeat_interleave(y.H // act_mask.shape[1], 1).repeat_interleave(y.W // act_mask.shape[2], 2).unsqueeze(-1).bool()
            raw, z = torch.where(m, raw, torch.zeros_like(raw)), torch.where(m, z, torch.full_like(z, -1.0))
        layers[st["pconv"]] = (raw.permute(0, 3, 1, 2).contiguous().cpu(), torch.clamp_min(z, 0).permute(0, 3, 1, 2).contiguous().cpu(),
                               (z > 0).permute(0, 3, 1, 2).contiguous().cpu())
    for lv in ctx["levels"]:
        add(lv["s1"], ctx["active"]); add(lv["s2"], ctx["active"])
    add(ctx["b1"], ctx["active"]); add(ctx["b2"], ctx["active"])
    for lv in ctx["dctx"]["levels"]:
        add(lv["s1"], None); add(lv["s2"], None)
    assert len(layers) == 18

    class Taps:
        seen = set()

        def conv(self, key, y):
            self.seen.add(key)
            return y + (layers[key][0].to(y.dtype) - y).detach()

        def act(self, key, z):
            a = z * layers[key][2].to(z.dtype)
            return a + (layers[key][1].to(z.dtype) - a).detach()
    taps = Taps()

    odt = torch.float64 if dt == "f32" else torch.float32     # (the f16 bar is 1e-2: a float32 oracle is exact enough and several times faster)

    def wq(k, v):                     # the oracle on the weights the kernels multiply with (16-bit storage: the packs hold the rounded weights)
        if not v.is_floating_point():
            return v.clone()
        v = v.to(tdt).to(odt) if (v.dim() == 4 and dt != "f32" and "conv_last" not in k and "mask_tokens" not in k) else v.to(odt)
        return v.requires_grad_(True) if "running" not in k else v
    osd = {k: wq(k, v) for k, v in sd.items()}
    toks = [osd[f"mask_tokens.{i}"] for i in range(len(model.mask_tokens))]
    OU.TAP = taps
    try:
        ref_loss, _ = OS.forward(x.to(odt), active, osd, toks)
        ref_loss.backward()
    finally:
        OU.TAP = None
    assert taps.seen == set(layers)
    lbar = {"f32": 2e-5, "f16": 2e-3}[dt]
    assert abs(float(loss) - float(ref_loss.detach())) <= lbar * max(1.0, abs(float(ref_loss.detach()))), (float(loss), float(ref_loss))
    bar = {"f32": 1e-4, "f16": 1e-2}[dt]
    errs = []
    for k, p in model.named_parameters():
        if k.startswith("densify_projs") or not osd[k].requires_grad:
            continue
        gref = osd[k].grad
        assert gref is not None and p.grad is not None, k
        got = p.grad.detach().double().cpu() / scale
        if k.endswith(".0.bias") or k.endswith(".3.bias"):
            # a conv bias in front of a training-mode BatchNorm: the true gradient is zero, both sides hold rounding noise -- measured on the
            # scale of the same layer's weight gradient
            den = osd[k[:-len("bias")] + "weight"].grad.norm().item()
        elif k.endswith("up_sample.bias"):
            den = osd[k[:-len("bias")] + "weight"].grad.norm().item()      # cancellation remainder (see test_spark_step_vs_reference_fixture)
        else:
            den = gref.norm().item()
        errs.append(((got - gref).norm().item() / max(den, 1e-30), k))
    errs.sort(reverse=True)
    top = "; ".join(f"{k}: {e:.2e}" for e, k in errs[:4])
    print(f"[spark gate-forced {dt} {S}x{S} bs {B}] {len(errs)} gradients, worst {top}")
    _parity_record(f"SparK step {dt} at {S}x{S}, bs {B}, base 64, mask 0.75, list-driven kernels, {'float64' if dt == 'f32' else 'float32'} oracle backward on the HIP path's own forward values and "
                   f"gates: {len(errs)} parameter gradients, worst relative L2 {top} (bar {bar:.0e})")
    assert len(errs) >= 85 and errs[0][0] <= bar, top
