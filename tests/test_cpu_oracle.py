"""CPU suite (-m "not gpu"): the oracle restatement against the golden vectors produced by the reference
(tests/golden/*.npz: dense UNet, losses, SparK, LAMB, soft-clDice, and -- cmunet_ref.npz / moco_ref.npz -- the reference's own CM_UNet and
Moco_v2 modules), plus closed-form checks of the CM-UNet head and MoCo pieces."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import cmunet as OC, losses as OL, moco as OM, unet as OU


def fx(golden_dir, name):
    d = np.load(f"{golden_dir}/{name}.npz", allow_pickle=False)
    return {k: torch.from_numpy(np.asarray(d[k])) if d[k].dtype.kind in "fiu" else d[k] for k in d.files}


def sd_of(f, prefix="sd."):
    return {k[len(prefix):]: v.clone() for k, v in f.items() if k.startswith(prefix)}


def close(a, b, tol=2e-5):
    return (a.double() - b.double()).abs().max().item() <= tol * max(1.0, b.abs().max().item())


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_double_conv_golden(golden_dir, tag):
    f = fx(golden_dir, f"double_conv_{tag}")
    sd = OU.clone_sd(sd_of(f), requires_grad=True)
    x = f["x"].clone().requires_grad_(True)
    y = OU.double_conv(x, sd, "double_conv.", True)
    (y * f["go"]).sum().backward()
    assert close(y, f["y"]) and close(x.grad, f["dx"])
    for k in sd:
        if ("grad." + k) in f:
            assert close(sd[k].grad, f["grad." + k], 1e-4), k
        if ("after." + k) in f:
            assert close(sd[k].detach().float(), f["after." + k].float()), k


def test_down_up_block_golden(golden_dir):
    f = fx(golden_dir, "down_block")
    sd = OU.clone_sd(sd_of(f))
    d, s = OU.down_block(f["x"], sd, "", True)
    assert close(d, f["down"]) and close(s, f["skip"])
    f = fx(golden_dir, "up_block")
    sd = OU.clone_sd(sd_of(f))
    y = OU.up_block(f["xd"], f["xs"], sd, "", "conv_transpose", True)
    assert close(y, f["y"])
    with pytest.raises(ValueError) as e:
        OU.up_block(f["xd"], f["xs"], sd, "", "nearest", True)
    assert str(e.value) == str(fx(golden_dir, "losses")["bad_mode_msg"])


def test_unet_small_golden_and_adam_trace(golden_dir):
    f = fx(golden_dir, "unet_small")
    sd = OU.clone_sd(sd_of(f), requires_grad=True)
    logits = OU.unet_forward(f["x"], sd, training=True)
    loss = OL.dice_ce_loss(logits, f["y1h"])
    assert close(logits, f["logits"]) and abs(float(loss) - float(f["loss"])) < 1e-5
    assert loss.dtype == torch.float64                   # A-5: float64 targets -> float64 loss
    assert str(f["crit_name"]) == "dice_loss + cross_entropy_loss"
    # two Adam steps (train.py:163-169 restated) reproduce the reference trace
    t = fx(golden_dir, "unet_small_adam_trace")
    sd = OU.clone_sd(sd_of(f), requires_grad=True)
    params = [v for v in sd.values() if v.requires_grad]
    opt = torch.optim.Adam(params, lr=1e-3)
    for step in range(2):
        opt.zero_grad()
        l = OL.dice_ce_loss(OU.unet_forward(t["x"], sd, training=True), t["y1h"])
        l.backward()
        opt.step()
        assert abs(float(l) - float(t["losses"][step])) < 1e-5
    assert close(sd["conv_last.weight"].detach(), t["conv_last_weight"], 1e-5)
    assert close(sd["double_conv.double_conv.4.running_var"], t["bott_bn_running_var"], 1e-5)


def test_unet_full_golden(golden_dir):
    f = fx(golden_dir, "unet_full")
    sd = OU.make_state_dict(base_ch=64, depth=5, seed=int(f["seed"]))
    assert [str(k) for k in f["state_keys"]] == list(sd.keys())
    assert sum(v.numel() for k, v in sd.items() if "running" not in k and "num_batches" not in k) == 31042434
    logits = OU.unet_forward(f["x"], OU.clone_sd(sd), training=True)
    assert close(logits, f["logits"], 1e-4)


def test_losses_golden(golden_dir):
    f = fx(golden_dir, "losses")
    lo = f["logits"].clone().requires_grad_(True)
    assert abs(float(OL.dice_loss(lo, f["y1h"])) - float(f["dice"])) < 1e-6
    assert abs(float(OL.iou_loss(lo, f["y1h"])) - float(f["iou"])) < 1e-6
    assert abs(float(OL.cross_entropy_prob(lo, f["y1h"])) - float(f["ce"])) < 1e-6
    OL.dice_ce_loss(lo, f["y1h"]).backward()
    assert close(lo.grad, f["dlogits"], 1e-5)            # Dice contributes no gradient (A-4)


def test_patch_mask_counts_and_reference_loop():
    """UNet_encoder.py:106-139: floor(ratio*H*W/256) patches per sample; 224@.65 -> 127, 512@.6 -> 614, 512@.75 -> 768."""
    for size, ratio, n in ((224, 0.65, 127), (512, 0.6, 614), (512, 0.75, 768), (256, 0.6, 153)):
        m = OC.create_random_patch_mask(2, size, 16, ratio, np.random.RandomState(0))
        assert m.shape == (2, size, size) and m.dtype == np.uint8
        assert (m.reshape(2, size // 16, 16, size // 16, 16).sum((2, 4)) // 256).sum(axis=(1, 2)).tolist() == [n, n]
    # the product's vectorised host generator consumes the RNG identically
    from cmunet_amd.pretrain import create_random_patch_mask
    a = OC.create_random_patch_mask(3, 96, 16, 0.6, np.random.RandomState(7))
    b = create_random_patch_mask(3, 96, 16, 0.6, np.random.RandomState(7))
    assert np.array_equal(a, b)


def test_cmunet_head_closed_forms():
    g = torch.Generator().manual_seed(0)
    img = torch.randn(2, 8, 12, generator=g)
    t = OC.recon_target(img)
    assert torch.allclose(t.mean(-1), torch.zeros(2, 8), atol=1e-6)
    assert torch.allclose(t.var(-1), torch.ones(2, 8), atol=1e-4)         # unbiased variance, per ROW (A-3)
    pred, mask = torch.randn(2, 8, 12, generator=g), (torch.rand(2, 8, 12, generator=g) > 0.5)
    ref = (((pred - t) ** 2) * mask).sum() / mask.sum()
    assert torch.allclose(OC.masked_mse(pred, img, mask.to(torch.uint8)), ref)
    # InfoNCE: labels offset by B*rank, scaled by 2*t (cmunet_head.py:84-88)
    p, k = torch.randn(4, 16, generator=g), F.normalize(torch.randn(12, 16, generator=g), dim=1)
    l = OC.infonce_inbatch(p, k, 0.07, rank=2)
    s = F.normalize(p, dim=1) @ k.t() / 0.07
    assert torch.allclose(l, 2 * 0.07 * F.cross_entropy(s, torch.arange(4) + 8))
    assert abs(OC.momentum_schedule(10, 100) - 0.996) < 1e-12              # A-9: constant when end == base


def test_moco_queue_semantics():
    """moco2_module.py:160-175, 256-285: logits use the PRE-enqueue queue; pointer wraps; K % batch == 0."""
    g = torch.Generator().manual_seed(1)
    queue, ptr = OM.init_queue(8, 32, 0), torch.zeros(1, dtype=torch.long)
    assert torch.allclose(queue.norm(dim=0), torch.ones(32), atol=1e-6)
    q, k = torch.randn(4, 8, generator=g), torch.randn(4, 8, generator=g)
    logits, labels, kn, _ = OM.logits_from_embeddings(q, k, queue, 0.2)
    assert logits.shape == (4, 33) and labels.sum() == 0
    before = queue.clone()
    for it in range(9):
        OM.dequeue_and_enqueue(kn, queue, ptr, 32)
    assert int(ptr) == (9 * 4) % 32
    assert torch.equal(queue[:, 4:32], torch.cat([kn.T] * 7, 1)) and not torch.equal(queue, before)
    with pytest.raises(AssertionError):
        OM.dequeue_and_enqueue(torch.randn(5, 8), queue, ptr, 32)


@pytest.mark.skipif(not os.path.exists("/root/reference/Finetuning/model.py"), reason="reference not present (GPU box)")
def test_oracle_vs_imported_reference():
    """Build container only: import the reference model and compare the oracle on a fresh seeded case."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_model", "/root/reference/Finetuning/model.py")
    ref = importlib.util.module_from_spec(spec)
    import sys
    sys.dont_write_bytecode = True
    spec.loader.exec_module(ref)
    sd = OU.make_state_dict(base_ch=64, depth=5, seed=123)
    m = ref.UNet()
    m.load_state_dict(sd)
    m.train()
    x = torch.randn(1, 16, 32, generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        assert close(OU.unet_forward(x, OU.clone_sd(sd), training=True), m(x), 1e-4)


@pytest.mark.parametrize("fixture", ["spark_unet", "spark_unet_m75", "spark_unet_m75_b8"])
def test_spark_golden(golden_dir, fixture):
    """oracle/spark.py reproduces the loss the reference's SparK produced (fixtures written by gen_golden.py: 64 px at mask
    ratio 0.6, 128 px at BASELINE config 5's ratio 0.75)."""
    from oracle import spark as OS
    d = np.load(f"{golden_dir}/{fixture}.npz")
    sd = OU.make_state_dict(base_ch=64, depth=5, seed=int(d["seed"]))
    osd = {}
    for k, v in sd.items():
        if "up_conv" in k or "conv_last" in k:
            osd["dense_decoder." + k] = v[:1].clone() if "conv_last" in k else v.clone()
        else:
            osd["sparse_encoder.sp_cnn." + k] = v.clone()
    tok, off, toks = torch.from_numpy(d["tokens_flat"]), 0, []
    for c in (1024, 512, 256, 128, 64):
        toks.append(tok[off:off + c].view(1, c, 1, 1)); off += c
    loss, rec = OS.forward(torch.from_numpy(d["x"]), torch.from_numpy(d["active"]).bool(), osd, toks)
    assert abs(float(loss) - float(d["loss"])) < 1e-5 and rec.shape == tuple(d["x"].shape)
    if fixture.startswith("spark_unet_m75"):
        assert float(d["mask_ratio"]) == 0.75 and int(d["active"].sum()) == d["x"].shape[0] * 16
    a = OS.make_active(3, 16, 0.6, torch.Generator().manual_seed(0))
    assert a.shape == (3, 1, 16, 16) and a.view(3, -1).sum(1).tolist() == [round(256 * 0.4)] * 3     # spark.py:29,82-86


def test_neck_syncbn_two_rank_golden(golden_dir):
    """oracle/cmunet.py::nonlinear_neck on the concatenated rows of both ranks reproduces what the reference's NonLinearNeck gave
    there (tests/golden/neck_syncbn_2rank.npz): the SyncBN semantics the two-rank GPU test is held to."""
    from oracle import cmunet as OC
    d = np.load(f"{golden_dir}/neck_syncbn_2rank.npz")
    S, B = int(d["S"]), int(d["B"])
    sd = OC.make_neck_sd("", S * S, 1536, 256, int(d["seed"]))
    osd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone()) for k, v in sd.items()}
    x = torch.from_numpy(d["x"]).reshape(2 * B, 1, S, S).clone().requires_grad_(True)
    y = OC.nonlinear_neck(x, osd, "", training=True)
    (y * torch.from_numpy(d["go"]).reshape(2 * B, 1, 256)).sum().backward()
    assert close(y.detach().view(2, B, 1, 256), torch.from_numpy(d["y"]), 1e-5)
    assert close(x.grad.view(2, B, 1, S, S), torch.from_numpy(d["dx"]), 2e-4)
    assert close(osd["bn0.weight"].grad, torch.from_numpy(d["dbn0_weight"]), 2e-4)
    assert close(osd["bn0.running_var"], torch.from_numpy(d["running_var"]), 1e-5)


def test_optimizer_traces_golden(golden_dir):
    """oracle/optim.py (SGD as MoCo configures it, LAMB as SparK does) against traces of torch.optim.SGD and of the
    reference's own LAMB class (oracle/gen_golden.py)."""
    import numpy as np
    import torch
    from oracle import optim as OO
    d = np.load(f"{golden_dir}/optim_traces.npz")
    wds = [float(w) for w in d["wds"]]
    n = len(wds)
    p0 = [torch.from_numpy(d[f"p0.{i}"]) for i in range(n)]
    grads = [[torch.from_numpy(d[f"g{k}.{i}"]) for i in range(n)] for k in range(3)]
    for tag, kw in (("a", dict(trust_clip=False, always_adapt=False)), ("b", dict(trust_clip=True, always_adapt=True))):
        ps, ms, vs = [t.clone() for t in p0], [torch.zeros_like(t) for t in p0], [torch.zeros_like(t) for t in p0]
        for k in range(3):
            OO.lamb_step(ps, grads[k], ms, vs, 2e-2, wds, betas=(0.9, 0.98), eps=1e-6, max_grad_norm=2.0, step=k + 1, **kw)
        for i in range(n):
            assert torch.allclose(ps[i], torch.from_numpy(d[f"lamb_{tag}.{i}"]), atol=2e-6), (tag, i)
    for tag, kw in (("a", dict(momentum=0.9, weight_decay=1e-4)), ("b", dict(momentum=0.9, weight_decay=1e-2, nesterov=True)),
                    ("c", dict(momentum=0.0, weight_decay=0.0))):
        ps, bufs = [t.clone() for t in p0], [torch.zeros_like(t) for t in p0]
        for k in range(3):
            OO.sgd_step(ps, grads[k], bufs, 0.03, step=k + 1, **kw)
        for i in range(n):
            assert torch.allclose(ps[i], torch.from_numpy(d[f"sgd_{tag}.{i}"]), atol=2e-6), (tag, i)


def test_soft_cldice_golden(golden_dir):
    """oracle/losses.py soft skeleton / soft clDice against the reference's own classes (fixture from gen_golden.py)."""
    import numpy as np
    import torch
    from oracle import losses as OL
    d = np.load(f"{golden_dir}/cldice.npz")
    logits, y1h = torch.from_numpy(d["logits"]), torch.from_numpy(d["y1h"])
    assert abs(float(OL.soft_cldice(logits, y1h)) - float(d["cldice"])) < 1e-6
    yp = (torch.softmax(logits, 1) > 0.5).float()[:, 1:2]
    assert torch.allclose(OL.soft_skel(yp, 10), torch.from_numpy(d["skel_pred"]), atol=1e-6)
    assert str(d["metric_name"]) == "soft_clDice"


def test_augment_oracle_against_pillow_fixture_and_live(golden_dir):
    """oracle/augment.py: the bicubic resize restatement reproduces Pillow's own outputs bit for bit (committed fixture made
    by oracle/gen_golden.py --only-augment; and Pillow itself when it is importable here), the ShiftPixel / GaussNoise
    arithmetic matches the reference's lines, and the in-kernel generator is Philox4x32-10 (Random123 known answers)."""
    from oracle import augment as OA
    z = np.load(os.path.join(golden_dir, "augment.npz"))
    i = 0
    while f"resize{i}.in" in z:
        a, ref = z[f"resize{i}.in"], z[f"resize{i}.out"]
        assert np.array_equal(OA.resize_bicubic(a, *ref.shape).view(np.uint32), ref.view(np.uint32)), i
        i += 1
    assert i == 5
    i = 0
    while f"resize_u8_{i}.in" in z:             # 8-bit images (PIL mode 'L') and the NEAREST resize of the masks
        assert np.array_equal(OA.resize_bicubic_u8(z[f"resize_u8_{i}.in"], *z[f"resize_u8_{i}.out"].shape), z[f"resize_u8_{i}.out"]), i
        assert np.array_equal(OA.resize_nearest(z[f"nearest{i}.in"], *z[f"nearest{i}.out"].shape), z[f"nearest{i}.out"]), i
        i += 1
    assert i == 6
    _, t = OA.two_view(z["tv_in"], z["tv_shifts"], z["tv_noise"], 32)
    assert np.array_equal(t.view(np.uint32), z["tv_img_t"].view(np.uint32))
    try:
        from PIL import Image
    except ImportError:
        Image = None
    if Image is not None:
        rng = np.random.RandomState(9)
        for (h, w, oh, ow) in [(300, 200, 256, 256), (51, 77, 256, 256), (512, 512, 256, 256)]:
            a = rng.standard_normal((h, w)).astype(np.float32)
            ref = np.asarray(Image.fromarray(a).resize((ow, oh), resample=Image.BICUBIC))
            assert np.array_equal(OA.resize_bicubic(a, oh, ow).view(np.uint32), ref.view(np.uint32))
            a8 = rng.randint(0, 256, (h, w)).astype(np.uint8)
            assert np.array_equal(OA.resize_bicubic_u8(a8, oh, ow), np.asarray(Image.fromarray(a8).resize((ow, oh), resample=Image.BICUBIC)))
            assert np.array_equal(OA.resize_nearest(a8, oh, ow), np.asarray(Image.fromarray(a8).resize((ow, oh), resample=Image.NEAREST)))
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for c, k, exp in kat:
        assert tuple(int(v) for v in OA.philox4x32_10(np.array([c], dtype=np.uint64), k)[0]) == exp


def test_random_resized_crop_params_contract():
    from cmunet_amd.dataset import random_resized_crop_params
    rng = np.random.RandomState(0)
    for _ in range(500):
        x0, y0, cw, ch = random_resized_crop_params(256, 256, rng)
        assert 0 <= x0 and 0 <= y0 and cw >= 1 and ch >= 1 and x0 + cw <= 256 and y0 + ch <= 256
        assert 0.18 <= cw * ch / 65536 <= 1.0 and 0.70 <= cw / ch <= 1.40


def test_cmunet_oracle_vs_reference_fixture(golden_dir):
    """tests/golden/cmunet_ref.npz holds what the REFERENCE's own CM_UNet produced in the build container (cmae modules behind the
    mmengine / mmcv plumbing stand-in of oracle/gen_golden.py::import_cmae; shipped cmunet_config.py, 224 x 224, bs 4): the patch
    mask under a numpy seed, forward_train's two losses, the gradient norm of every trainable parameter, a few gradients in full, the
    EMA of momentum_update, MomentumUpdateHook's schedule and CMUNetPretrainHead.forward alone.  The oracle restatement, fed with the
    regenerated weights and inputs, must reproduce them: rows a7-a12 of SURVEY section 8 are pinned by the reference."""
    f = fx(golden_dir, "cmunet_ref")
    seed, B, S = int(f["seed"]), int(f["B"]), int(f["S"])
    img, img_t, mask, rw, rb = OC.cmunet_fixture_inputs(seed, B, S)
    assert np.array_equal(mask[:, ::16, ::16], f["mask_patches"].numpy())          # create_random_patch_mask (UNet_encoder.py:106-139)
    assert np.array_equal(np.repeat(np.repeat(mask[:, ::16, ::16], 16, 1), 16, 2), mask)
    sd = OC.make_cmunet_sd(seed, S)
    trainable = [str(k) for k in f["trainable"]]
    osd = {k: (v.clone().requires_grad_(True) if k in set(trainable) else v.clone()) for k, v in sd.items()}
    assert sorted(k for k, v in osd.items() if v.is_floating_point() and "running" not in k and not k.startswith("target_")) == trainable
    out = OC.forward_train(img, img_t, mask, rw, rb, osd, temperature=0.07, ct_weight=1.0, rc_weight=1.0)
    (out["loss_ct"] + out["loss_rc"]).backward()
    assert abs(float(out["loss_rc"]) - float(f["loss_rc"])) <= 2e-5 * max(1.0, abs(float(f["loss_rc"])))
    assert abs(float(out["loss_ct"]) - float(f["loss_ct"])) <= 1e-4 * max(1.0, abs(float(f["loss_ct"])))
    norms = torch.stack([osd[k].grad.double().norm() for k in trainable])
    ref = f["grad_norms"].double()
    # (a bias in front of a training-mode BatchNorm -- the conv biases, fc0.bias of the necks -- has an analytically zero gradient:
    # what both sides hold there is rounding noise of the summation order, compared by magnitude only)
    # (so has feature_decoder.conv_last.bias: a constant added to the projector's input shifts fc0's output by a constant)
    noise = torch.tensor([k.endswith((".0.bias", ".3.bias", "fc0.bias")) or k == "feature_decoder.conv_last.bias" for k in trainable])
    big = ~noise
    assert ((norms[big] - ref[big]).abs() / ref[big]).max().item() <= 2e-3
    assert norms[noise].max().item() <= 1e-2 and ref[noise].max().item() <= 1e-2
    for k in f:
        if k.startswith("grad."):
            g, r = osd[k[5:]].grad, f[k]
            assert (g - r).norm().item() <= 2e-3 * r.norm().item() + 1e-7, k
        if k.startswith("after."):
            assert close(osd[k[6:]].float(), f[k].float(), 1e-4), k
    # EMA (cmunet.py:78-92) at momentum 0.9 from the post-step state
    st = {k: v.detach().clone() for k, v in osd.items()}
    OC.momentum_update(st, 0.9)
    tk = [str(k) for k in f["target_keys"]]
    en = torch.stack([st[k].double().norm() for k in tk])
    assert ((en - f["ema_norms"].double()).abs() / f["ema_norms"].double().clamp_min(1e-12)).max().item() <= 1e-6
    assert close(st["target_backbone.down_conv1.double_conv.double_conv.0.weight"], f["ema_sample"], 1e-6)
    # MomentumUpdateHook's schedule (momentum_update_hook.py:29-40)
    for (it, mx, base, end), m in zip(f["hook_cases"].numpy(), f["hook_momentum"].numpy()):
        assert abs(OC.momentum_schedule(int(it), int(mx), float(base), float(end)) - float(m)) < 1e-12
    # the patch mask over more geometries and ratios (incl. 0 and 1), oracle and the product's vectorised host generator
    from cmunet_amd.pretrain import create_random_patch_mask as product_mask
    for ci, (mb, ms_, mr) in enumerate(f["mask_cases"].numpy()):
        mb, ms_ = int(mb), int(ms_)
        want = np.repeat(np.repeat(f[f"mask{ci}"].numpy(), 16, 1), 16, 2)
        assert np.array_equal(OC.create_random_patch_mask(mb, ms_, 16, float(mr), np.random.RandomState(seed + 300 + ci)), want), ci
        assert np.array_equal(product_mask(mb, ms_, 16, float(mr), np.random.RandomState(seed + 300 + ci)), want), ci
    # the head with other hyper-parameters (temperature 0.2, ct_weight 0.5, rc_weight 2)
    t2, cw2, rw2 = (float(v) for v in f["head2.hyper"])
    hsd2 = {k: (v.clone().requires_grad_(v.is_floating_point() and "running" not in k)) for k, v in sd.items() if k.startswith("head.")}
    p2, s2 = f["head.pred"].clone().requires_grad_(True), f["head.proj_s"].clone().requires_grad_(True)
    h2 = OC.head(f["head.x"], p2, f["head.mask"], s2, f["head.proj_t"], hsd2, "head.", t2, cw2, rw2)
    (h2["loss_ct"] + h2["loss_rc"]).backward()
    assert abs(float(h2["loss_rc"]) - float(f["head2.loss_rc"])) <= 2e-5 * max(1.0, abs(float(f["head2.loss_rc"])))
    assert abs(float(h2["loss_ct"]) - float(f["head2.loss_ct"])) <= 1e-4 * max(1.0, abs(float(f["head2.loss_ct"])))
    assert close(p2.grad, f["head2.dpred"], 1e-4) and close(s2.grad, f["head2.dproj_s"], 2e-4)
    # the head alone (cmunet_head.py:47-91)
    hsd = {k: (v.clone().requires_grad_(v.is_floating_point() and "running" not in k)) for k, v in sd.items() if k.startswith("head.")}
    pred, ps = f["head.pred"].clone().requires_grad_(True), f["head.proj_s"].clone().requires_grad_(True)
    hl = OC.head(f["head.x"], pred, f["head.mask"], ps, f["head.proj_t"], hsd, "head.", 0.07, 1.0, 1.0)
    (hl["loss_ct"] + hl["loss_rc"]).backward()
    assert abs(float(hl["loss_rc"]) - float(f["head.loss_rc"])) <= 2e-5 * max(1.0, abs(float(f["head.loss_rc"])))
    assert abs(float(hl["loss_ct"]) - float(f["head.loss_ct"])) <= 1e-4 * max(1.0, abs(float(f["head.loss_ct"])))
    assert close(pred.grad, f["head.dpred"], 1e-4) and close(ps.grad, f["head.dproj_s"], 2e-4)


def test_moco_oracle_vs_reference_fixture(golden_dir):
    """tests/golden/moco_ref.npz holds what the REFERENCE's own Moco_v2 produced in the build container (moco2_module.py behind the
    lightning / torchvision plumbing stand-in of oracle/gen_golden.py::import_moco): two training steps at bs 4, 64 x 64, K = 64,
    tau 0.2, m 0.99.  The oracle restatement on the regenerated state and inputs must reproduce them: row a13 is pinned by the
    reference."""
    f = fx(golden_dir, "moco_ref")
    seed, B, S, K, T, EM = int(f["seed"]), int(f["B"]), int(f["S"]), int(f["K"]), float(f["T"]), float(f["EM"])
    sd = OM.make_moco_sd(seed, K)
    xq, xk, xq2, xk2 = OM.moco_fixture_inputs(seed, B, S)
    qkeys = [str(k) for k in f["qkeys"]]
    osd = {k: (v.clone().requires_grad_(True) if k in set(qkeys) else v.clone()) for k, v in sd.items()}
    queue, ptr = sd["queue"].clone(), sd["queue_ptr"].clone()
    loss, logits, k = OM.training_step(xq, xk, osd, queue, ptr, T, EM)
    loss.backward()
    assert abs(float(loss) - float(f["loss"])) <= 2e-5 * max(1.0, abs(float(f["loss"])))
    assert close(queue[:, :B].t(), f["keys"], 1e-5) and int(ptr) == int(f["queue_ptr"]) == B
    assert close(queue[:, B:], sd["queue"][:, B:], 0.0)                       # the rest of the queue is untouched
    norms = torch.stack([osd[q].grad.double().norm() for q in qkeys])
    live = torch.tensor([not q.endswith((".0.bias", ".3.bias")) for q in qkeys])          # (conv bias under training-mode BN: noise)
    assert ((norms - f["grad_norms"].double()).abs() / f["grad_norms"].double().clamp_min(1e-30))[live].max().item() <= 2e-3
    for name in f:
        if name.startswith("grad."):
            assert (osd[name[5:]].grad - f[name]).norm().item() <= 2e-3 * f[name].norm().item(), name
    kk = [str(q) for q in f["kkeys"]]
    en = torch.stack([osd[q].double().norm() for q in kk])
    assert ((en - f["ema_norms"].double()).abs() / f["ema_norms"].double()).max().item() <= 1e-6
    assert close(osd["encoder_k.double_conv.double_conv.0.weight"][:8], f["ema_sample"], 1e-6)
    # forward() on the updated state (no enqueue), then the second step
    with torch.no_grad():
        q2 = OM.encoder_gap(xq2, osd, "encoder_q.", True)
        k2 = OM.encoder_gap(xk2, osd, "encoder_k.", True)
    lg, lb, _, _ = OM.logits_from_embeddings(q2, k2, queue, T)
    assert close(lg, f["fwd.logits"], 1e-4) and int(lb.sum()) == int(f["fwd.labels"].sum()) == 0
    loss2, _, _ = OM.training_step(xq2, xk2, osd, queue, ptr, T, EM)
    assert abs(float(loss2) - float(f["loss2"])) <= 2e-5 * max(1.0, abs(float(f["loss2"])))
    assert close(queue[:, B:2 * B].t(), f["keys2"], 1e-5) and int(ptr) == int(f["queue_ptr2"]) == 2 * B


def test_moco_validation_oracle_vs_reference_fixture(golden_dir):
    """tests/golden/moco_val_ref.npz: the REFERENCE's own Moco_v2.validation_step (eval mode, two batches against val_queue); the
    oracle restatement on the regenerated state reproduces loss, top-1 / top-5 precision, enqueued keys and pointer."""
    f = fx(golden_dir, "moco_val_ref")
    seed, B, S, K, T = int(f["seed"]), int(f["B"]), int(f["S"]), int(f["K"]), float(f["T"])
    sd = OM.make_moco_sd(seed, K)
    vq, vp = OM.init_queue(1024, K, seed + 1), torch.zeros(1, dtype=torch.long)
    for i, s_ in enumerate((seed, seed + 1)):
        x1, x2 = OM.moco_fixture_inputs(s_, B, S)[:2]
        loss, a1, a5 = OM.validation_step(x1, x2, sd, vq, vp, T)
        assert abs(float(loss) - float(f[f"val_loss{i}"])) <= 2e-5 * max(1.0, abs(float(f[f"val_loss{i}"])))
        assert float(a1) == float(f[f"val_acc1_{i}"].reshape(-1)[0]) and float(a5) == float(f[f"val_acc5_{i}"].reshape(-1)[0])
        assert close(vq[:, i * B:(i + 1) * B].t(), f[f"keys{i}"], 1e-5)
    assert int(vp) == int(f["val_queue_ptr"].reshape(-1)[0]) == 2 * B


def test_moco_two_rank_oracle_vs_reference_fixture(golden_dir):
    """tests/golden/moco_ref_2rank.npz: the REFERENCE's own Moco_v2 run on two gloo ranks in the build container (DDP strategy:
    shuffle-BN around the key encoder with rank 0's broadcast permutation, gathered keys, 2B rows enqueued).  The one-process oracle
    emulation of the two ranks reproduces every rank's loss, local gradient norms, key-encoder BatchNorm buffer and the queue."""
    f = fx(golden_dir, "moco_ref_2rank")
    seed, B, S, K, T, EM = int(f["seed"]), int(f["B"]), int(f["S"]), int(f["K"]), float(f["T"]), float(f["EM"])
    torch.manual_seed(seed + 7)
    perm = torch.randperm(2 * B)
    assert torch.equal(perm, f["perm"])                                   # rank 0's torch.randperm under the same seed
    sd0 = OM.make_moco_sd(seed, K)
    ins = [OM.moco_fixture_inputs(seed + 50 * (r + 1), B, S)[:2] for r in range(2)]
    out, queue = OM.two_rank_step(sd0, ins, perm, T, EM)
    qkeys = [str(k) for k in f["qkeys"]]
    live = torch.tensor([not k.endswith((".0.bias", ".3.bias")) for k in qkeys])
    for r, (loss, sd) in enumerate(out):
        assert abs(float(loss) - float(f["loss"][r])) <= 2e-5 * max(1.0, abs(float(f["loss"][r])))
        gn = torch.stack([sd[k].grad.double().norm() for k in qkeys])
        assert ((gn - f["grad_norms"][r].double()).abs() / f["grad_norms"][r].double().clamp_min(1e-30))[live].max().item() <= 2e-3
        assert (sd["encoder_q.down_conv1.double_conv.double_conv.0.weight"].grad - f["grad0"][r]).norm().item() <= 2e-3 * f["grad0"][r].norm().item()
        assert close(sd["encoder_k.down_conv1.double_conv.double_conv.1.running_mean"], f["bn_k"][r], 1e-5)
    assert close(queue[:, :2 * B].t(), f["keys"], 1e-5) and int(f["queue_ptr"]) == 2 * B
    assert float((f["loss"][0] - f["loss"][1]).abs()) > 1e-4              # the ranks had different images


def test_finetune_sensitivity_fixture_is_consistent(golden_dir):
    """tests/golden/finetune_sensitivity.npz (the reference's loop re-run from initial weights perturbed by four-ulp noise): the stored
    deviations are those of the stored runs against finetune_ref.npz, epoch 0 barely moves, and the epoch-1 validation Dice -- a
    thresholded metric after six Adam steps -- moves by more than 1e-3: the spread the GPU trajectory test takes its bar from."""
    d = np.load(f"{golden_dir}/finetune_ref.npz")
    s = np.load(f"{golden_dir}/finetune_sensitivity.npz")
    keys = [str(k) for k in s["log_keys"]]
    assert keys == [str(k) for k in d["log_keys"]] and int(s["seed"]) == int(d["seed"]) and float(s["eps"]) == 2.0 ** -22
    assert s["train_logs"].shape == (int(s["runs"]), 2, len(keys))
    assert np.array_equal(s["train_dev"], np.abs(s["train_logs"] - d["train_logs"][None]).max(0))
    assert np.array_equal(s["valid_dev"], np.abs(s["valid_logs"] - d["valid_logs"][None]).max(0))
    i = keys.index("dice_loss")
    assert s["train_dev"][0, i] < 5e-4 and s["valid_dev"][0, i] < 5e-4
    assert 1e-3 < s["valid_dev"][1, i] < 1e-2


def test_finetune_loop_oracle_vs_reference_fixture(golden_dir):
    """tests/golden/finetune_ref.npz: per-epoch logs of the REFERENCE's own TrainEpoch / ValidEpoch / train() (Finetuning/train.py)
    over two epochs on a synthetic split, reference UNet + DiceLoss + CrossEntropyLoss + Adam.  The oracle loop reproduces the
    logs (the reference's log keys included) and the trained parameters: row a5 is pinned by the reference's own loop."""
    d = np.load(f"{golden_dir}/finetune_ref.npz")
    seed = int(d["seed"])
    keys = [str(k) for k in d["log_keys"]]
    assert keys == sorted(["dice_loss + cross_entropy_loss", "dice_loss", "cross_entropy_loss", "iou_loss", "soft_clDice"])
    sd = OU.make_state_dict(base_ch=64, depth=5, seed=seed)
    train_loader, valid_loader = OU.finetune_fixture_data(seed + 1)
    osd = OU.clone_sd(sd, requires_grad=True)
    opt = torch.optim.Adam([v for v in osd.values() if v.requires_grad], lr=1e-3)

    def run(loader, training):
        acc = {k: [] for k in keys}
        for x, y in loader:
            if training:
                opt.zero_grad()
                lo = OU.unet_forward(x, osd, training=True)
                l = OL.dice_ce_loss(lo, y)
                l.backward()
                opt.step()
            else:
                with torch.no_grad():
                    lo = OU.unet_forward(x, osd, training=False)
                    l = OL.dice_ce_loss(lo, y)
            lo = lo.detach()
            acc["dice_loss + cross_entropy_loss"].append(float(l)); acc["dice_loss"].append(float(OL.dice_loss(lo, y)))
            acc["cross_entropy_loss"].append(float(OL.cross_entropy_prob(lo, y))); acc["iou_loss"].append(float(OL.iou_loss(lo, y)))
            acc["soft_clDice"].append(float(OL.soft_cldice(lo, y)))
        return [float(np.mean(acc[k])) for k in keys]
    for ep in range(2):
        got_t, got_v = run(train_loader, True), run(valid_loader, False)
        for got, ref in ((got_t, d["train_logs"][ep]), (got_v, d["valid_logs"][ep])):
            for k, a, b in zip(keys, got, ref):
                # (a training trajectory: the summation order of ATen's CPU kernels depends on the thread count, and six Adam steps
                # at lr 1e-3 amplify that to a few 1e-4 by the second epoch; the generator's own run agreed to 2e-4)
                # (the thresholded metrics -- dice, iou, soft-clDice -- move by pixel flips on top of that)
                assert abs(a - float(b)) <= (1e-3 if k == "cross_entropy_loss" else 3e-3) * max(1.0, abs(float(b))), (ep, k, a, float(b))
    pk = [str(k) for k in d["param_keys"]]
    norms = torch.stack([osd[k].detach().double().norm() for k in pk])
    ref = torch.from_numpy(d["param_norms"])
    # (a conv bias in front of a training-mode BatchNorm has a zero gradient up to rounding noise, and Adam normalises noise to full
    # steps of lr: those parameters random-walk on both sides)
    live = torch.tensor([not k.endswith((".0.bias", ".3.bias")) for k in pk])
    assert ((norms - ref).abs() / ref.clamp_min(1e-9))[live].max().item() <= 2e-3


def test_cmunet_head_two_rank_oracle_vs_reference_fixture(golden_dir):
    """tests/golden/cmunet_head_2rank.npz: the REFERENCE's own CMUNetPretrainHead.forward on two gloo ranks (concat_all_gather of the
    target projections, labels arange(B) + B * rank; predictor BatchNorm in eval mode).  The oracle's head with the gather emulated
    reproduces every rank's losses and gradients."""
    f = fx(golden_dir, "cmunet_head_2rank")
    seed = int(f["seed"])
    hsd0 = OC.make_neck_sd("head.predictor.", 256, 1536, 256, seed)
    g0 = torch.Generator().manual_seed(seed + 1)
    hsd0["head.predictor.bn0.running_mean"] = 0.1 * torch.randn(1536, generator=g0)
    hsd0["head.predictor.bn0.running_var"] = 0.5 + torch.rand(1536, generator=g0)
    ins = [OC.head_fixture_inputs(seed + 10 * (r + 1)) for r in range(2)]
    keys_all = torch.cat([F.normalize(ins[r][4].squeeze(1), dim=1, p=2) for r in range(2)])
    for r in range(2):
        x, pred, mk, ps, pt = ins[r]
        hsd = {k: (v.clone().requires_grad_(v.is_floating_point() and "running" not in k)) for k, v in hsd0.items()}
        po, pso = pred.clone().requires_grad_(True), ps.clone().requires_grad_(True)
        loss_rc = OC.masked_mse(po, x, mk)
        loss_ct = OC.infonce_inbatch(OC.nonlinear_neck(pso, hsd, "head.predictor.", training=False).squeeze(1), keys_all, 0.07, rank=r)
        (loss_ct + loss_rc).backward()
        assert abs(float(loss_rc) - float(f["loss_rc"][r])) <= 2e-5 * max(1.0, abs(float(f["loss_rc"][r])))
        assert abs(float(loss_ct) - float(f["loss_ct"][r])) <= 1e-4 * max(1.0, abs(float(f["loss_ct"][r])))
        assert close(po.grad, f["dpred"][r], 1e-4) and close(pso.grad, f["dproj_s"][r], 2e-4)
        assert abs(float(hsd["head.predictor.fc1.weight"].grad.double().norm()) - float(f["dfc1_norm"][r])) <= 2e-4 * float(f["dfc1_norm"][r])
    # the label offset matters: with rank 0's labels on rank 1 the loss is another one
    x, pred, mk, ps, pt = ins[1]
    hsd = {k: v.clone() for k, v in hsd0.items()}
    wrong = OC.infonce_inbatch(OC.nonlinear_neck(ps, hsd, "head.predictor.", training=False).squeeze(1), keys_all, 0.07, rank=0)
    assert abs(float(wrong) - float(f["loss_ct"][1])) > 1e-3
