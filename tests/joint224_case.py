"""The CM-UNet joint step at the reference's own geometry (224 x 224 crops, depth 5, projector in_channels = 224*224 =
50,176 -> 1,536 -> 256, cmunet_config.py:18-26) with base 32 channels, f32 storage -- the case shared by
tests/test_gpu_pretrain.py::test_cmunet_joint_step_reference_geometry{,_tie_free} and by tests/gen_joint224_spread.py (which writes
tests/golden/joint224_spread.npz).  Everything is drawn from CPU generators, so the weights and inputs are the same numbers in the
build container (where the spread fixture is made) and on the GPU box (where the HIP path is compared).

Two mask variants:
  'random65'  mask ratio 0.65 (127 of 196 patches) drawn as the reference draws it.  The reference multiplies every image of the
              batch by (1 - mask[0]) (quirk A-1, UNet_encoder.py:155-156): the online encoder sees 65 % zeros, max-pool windows
              inside masked patches TIE, and the last bit of a sum decides where a pooled gradient goes.
  'tie_free'  the same masks for the loss (cmunet_head.py:70 uses each sample's own mask), but mask[0] = 0: nothing is zeroed in
              the input, no window ties, and two correct fp32 implementations agree to rounding on EVERY tensor -- the twin that
              can tell a kernel error from a tie.
"""
import numpy as np
import torch

B, S = 4, 224
KEYS = ("head.predictor.fc1.weight", "head.predictor.bn0.weight", "projector.fc1.weight", "projector.fc0.weight", "projector.bn0.bias",
        "feature_decoder.conv_last.weight", "pixel_decoder.conv_last.weight", "pixel_decoder.up_conv4.up_sample.weight",
        "feature_decoder.up_conv1.double_conv.double_conv.3.weight", "backbone.double_conv.double_conv.3.weight",
        "backbone.down_conv1.double_conv.double_conv.0.weight", "backbone.down_conv3.double_conv.double_conv.1.weight",
        "backbone.down_conv2.double_conv.double_conv.3.weight", "backbone.down_conv4.double_conv.double_conv.0.weight",
        "pixel_decoder.up_conv1.double_conv.double_conv.0.weight", "feature_decoder.up_conv3.up_sample.weight")


def in_conv_chain(key):
    return "backbone." in key or "up_conv" in key


def build(mask_mode="random65"):
    """(model on the CPU, state dict, (img, img_t, mask, reduce_w, reduce_b))."""
    from cmunet_amd import cmunet as C
    from cmunet_amd.pretrain import create_random_patch_mask
    torch.manual_seed(0)
    # (CMU_JOINT224_BASE=64: the reference's own width, run once per round by the builder for the gate-forced test -- its float64 oracle takes
    # four times as long; the committed spread fixture is for base 32)
    base = int(__import__("os").environ.get("CMU_JOINT224_BASE", "32"))
    model = C.build_model(C.cmunet_config(img_size=S, dtype="f32", base_ch=base, depth=5)).train()
    model.init_weights()
    gw = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.dim() == 1 and ("bn" in n or ".1." in n or ".4." in n):
                p.add_(0.2 * torch.randn(p.shape, generator=gw))
    assert model.projector.fc0.weight.shape == (1536, 50176) and model.reduced_channels() == 256
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(1)
    img, img_t = torch.randn(B, S, S, generator=g), torch.randn(B, S, S, generator=g)
    mask = torch.from_numpy(create_random_patch_mask(B, S, 16, 0.65, np.random.RandomState(2)))
    assert int(mask[0].sum()) == 127 * 256
    if mask_mode == "tie_free":
        mask = mask.clone()
        mask[0] = 0
    else:
        assert mask_mode == "random65"
    rw, rb = torch.randn(256, 16 * base, 1, 1, generator=g) * 0.05, torch.randn(256, generator=g) * 0.1
    return model, sd, (img, img_t, mask, rw, rb)


def checksum(sd):
    """A number that pins the weights a spread fixture was made for."""
    return float(sum(v.double().abs().sum() for k, v in sorted(sd.items()) if v.is_floating_point() and k in KEYS))


def oracle_step(sd, inputs, perturb_seed=None, eps=2.0 ** -22):
    """The oracle's losses and gradients (dict over KEYS); ``perturb_seed``: the weights times (1 + eps * u), u uniform in [-1, 1)
    -- four ulps of noise, the sensitivity probe."""
    from oracle import cmunet as OC
    img, img_t, mask, rw, rb = inputs
    trainable = lambda k, v: v.is_floating_point() and "running" not in k and not k.startswith("target_")
    if perturb_seed is None:
        osd = {k: (v.clone().requires_grad_(True) if trainable(k, v) else v.clone()) for k, v in sd.items()}
    else:
        gp = torch.Generator().manual_seed(perturb_seed)
        osd = {k: ((v * (1 + eps * (2 * torch.rand(v.shape, generator=gp) - 1))).requires_grad_(True) if trainable(k, v) else v.clone())
               for k, v in sd.items()}
    ref = OC.forward_train(img, img_t, mask.numpy(), rw, rb, osd, temperature=0.07, ct_weight=1.0, rc_weight=1.0)
    (ref['loss_ct'] + ref['loss_rc']).backward()
    return {"loss_ct": float(ref['loss_ct'].detach()), "loss_rc": float(ref['loss_rc'].detach())}, {k: osd[k].grad.detach().double() for k in KEYS}


def rel_l2(a, b):
    return (a.double() - b.double()).norm().item() / max(b.double().norm().item(), 1e-12)
