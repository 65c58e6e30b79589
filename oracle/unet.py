"""Oracle: dense UNet blocks, restated functionally on a reference-named state dict.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows /root/reference/Finetuning/model.py:
  double_conv   model.py:16-26   [Conv3x3 p1 bias -> BatchNorm2d(eps 1e-5, mom .1) -> ReLU] x2
  down_block    model.py:42-45   DoubleConv -> MaxPool2d(2); returns (down, skip)
  up_block      model.py:67-81   ConvTranspose2d(k2,s2) | bilinear(align_corners) -> cat([up, skip],1) -> DoubleConv
  unet_forward  model.py:110-131 4 down, bottleneck, 4 up, 1x1 head; input (B,H,W) -> unsqueeze(1)
The same block code is what UNet_encoder.py:9-49 / munet_neck.py:12-49 /
moco_data_module.py:18-66 re-declare, so these functions are the
oracle for those copies too.

All functions take ``sd`` (a dict name -> tensor with the reference's
``state_dict`` key names) and a key ``prefix``; BatchNorm running statistics in
``sd`` are updated in place in training mode exactly like nn.BatchNorm2d
(momentum 0.1, unbiased running variance, num_batches_tracked += 1).
"""
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1

# Optional forward taps (tests only; None = the plain restatement).  An object with
#   conv(key, y) -> y     called on every 3x3 conv's raw output (key = the conv's state-dict prefix, e.g. 'backbone.down_conv1.
#                         double_conv.double_conv.0.'),
#   act(key, z) -> a      called INSTEAD of relu(z) on that layer's BatchNorm output.
# tests/test_gpu_pretrain.py::test_cmunet_joint_step_gate_forced_backward uses it to run the oracle's backward pass on the HIP
# path's own forward values and ReLU gates (the gradient of a ReLU network is discontinuous in its weights: see that test).
TAP = None


def _bn(x, sd, p, training):
    rm, rv = sd[p + "running_mean"], sd[p + "running_var"]
    y = F.batch_norm(x, rm, rv, sd[p + "weight"], sd[p + "bias"], training, BN_MOMENTUM, BN_EPS)
    if training and (p + "num_batches_tracked") in sd:
        sd[p + "num_batches_tracked"] += 1
    return y


def double_conv(x, sd, prefix, training=True):
    """model.py:16-26. ``prefix`` ends with 'double_conv.' (the nn.Sequential)."""
    for conv, bn in ((0, 1), (3, 4)):
        x = F.conv2d(x, sd[f"{prefix}{conv}.weight"], sd[f"{prefix}{conv}.bias"], padding=1)
        if TAP is not None:
            x = TAP.conv(f"{prefix}{conv}.", x)
        x = _bn(x, sd, f"{prefix}{bn}.", training)
        x = F.relu(x) if TAP is None else TAP.act(f"{prefix}{conv}.", x)
    return x


def down_block(x, sd, prefix, training=True):
    """model.py:42-45. prefix e.g. 'down_conv1.'"""
    skip = double_conv(x, sd, prefix + "double_conv.double_conv.", training)
    return F.max_pool2d(skip, 2), skip


def up_block(down_input, skip_input, sd, prefix, up_sample_mode="conv_transpose", training=True):
    """model.py:67-81. prefix e.g. 'up_conv4.'"""
    if up_sample_mode == "conv_transpose":
        x = F.conv_transpose2d(down_input, sd[prefix + "up_sample.weight"], sd[prefix + "up_sample.bias"], stride=2)
    elif up_sample_mode == "bilinear":
        x = F.interpolate(down_input, scale_factor=2, mode="bilinear", align_corners=True)
    else:
        raise ValueError("Unsupported `up_sample_mode` (can take one of `conv_transpose` or `bilinear`)")
    x = torch.cat([x, skip_input], dim=1)
    return double_conv(x, sd, prefix + "double_conv.double_conv.", training)


def depth_of(sd, prefix=""):
    """Number of resolution levels (reference: 5 = 4 down blocks + bottleneck)."""
    n = 0
    while f"{prefix}down_conv{n + 1}.double_conv.double_conv.0.weight" in sd:
        n += 1
    return n + 1


def encoder_forward(x_b1hw, sd, prefix="", training=True):
    """Down path + bottleneck (model.py:121-125; UNet_encoder.py:79-83). Returns (latent, [skip1..])."""
    n_down = depth_of(sd, prefix) - 1
    skips = []
    x = x_b1hw
    for i in range(1, n_down + 1):
        x, s = down_block(x, sd, f"{prefix}down_conv{i}.", training)
        skips.append(s)
    x = double_conv(x, sd, prefix + "double_conv.double_conv.", training)
    return x, skips


def decoder_forward(latent, skips, sd, prefix="", up_sample_mode="conv_transpose", training=True):
    """Up path + 1x1 head (model.py:126-130; munet_neck.py:75-82)."""
    x = latent
    for i in range(len(skips), 0, -1):
        x = up_block(x, skips[i - 1], sd, f"{prefix}up_conv{i}.", up_sample_mode, training)
    return F.conv2d(x, sd[prefix + "conv_last.weight"], sd[prefix + "conv_last.bias"])


def unet_forward(x_bhw, sd, up_sample_mode="conv_transpose", training=True):
    """model.py:110-131: (B,H,W) -> logits (B,out_classes,H,W)."""
    latent, skips = encoder_forward(x_bhw.unsqueeze(1), sd, "", training)
    return decoder_forward(latent, skips, sd, "", up_sample_mode, training)


def make_state_dict(base_ch=64, depth=5, out_classes=2, in_ch=1, up_sample_mode="conv_transpose",
                    seed=0, dtype=torch.float32, encoder=True, decoder=True):
    """Seeded random state dict with the reference's key names/shapes.

    base_ch=64, depth=5 is the reference structure (model.py:96-108); other values are the build's
    extension.  Initialisation mimics nn.Conv2d / nn.ConvTranspose2d defaults only loosely
    (uniform +-1/sqrt(fan_in)); tests always pass explicit tensors so the exact init law is irrelevant.
    BN affine parameters are randomised (not 1/0) so that tests exercise them.
    """
    g = torch.Generator().manual_seed(seed)

    def u(shape, fan_in):
        b = 1.0 / fan_in ** 0.5
        return ((torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1) * b).to(dtype)

    sd = {}

    def dconv(p, cin, cout):
        for conv, bn, ci in ((0, 1, cin), (3, 4, cout)):
            sd[f"{p}{conv}.weight"] = u((cout, ci, 3, 3), ci * 9)
            sd[f"{p}{conv}.bias"] = u((cout,), ci * 9)
            sd[f"{p}{bn}.weight"] = (1.0 + 0.25 * (torch.rand(cout, generator=g, dtype=torch.float64) * 2 - 1)).to(dtype)
            sd[f"{p}{bn}.bias"] = (0.2 * (torch.rand(cout, generator=g, dtype=torch.float64) * 2 - 1)).to(dtype)
            sd[f"{p}{bn}.running_mean"] = torch.zeros(cout, dtype=dtype)
            sd[f"{p}{bn}.running_var"] = torch.ones(cout, dtype=dtype)
            sd[f"{p}{bn}.num_batches_tracked"] = torch.zeros((), dtype=torch.int64)

    chans = [base_ch * 2 ** i for i in range(depth)]
    if encoder:
        cin = in_ch
        for i in range(depth - 1):
            dconv(f"down_conv{i + 1}.double_conv.double_conv.", cin, chans[i])
            cin = chans[i]
        dconv("double_conv.double_conv.", cin, chans[-1])
    if decoder:
        for i in range(depth - 1, 0, -1):
            cin, cout = chans[i], chans[i - 1]
            if up_sample_mode == "conv_transpose":
                sd[f"up_conv{i}.up_sample.weight"] = u((cin, cout, 2, 2), cout * 4)
                sd[f"up_conv{i}.up_sample.bias"] = u((cout,), cout * 4)
            dconv(f"up_conv{i}.double_conv.double_conv.", cin, cout)
        sd["conv_last.weight"] = u((out_classes, chans[0], 1, 1), chans[0])
        sd["conv_last.bias"] = u((out_classes,), chans[0])
    return sd


def clone_sd(sd, requires_grad=False):
    out = {}
    for k, v in sd.items():
        t = v.detach().clone()
        if requires_grad and t.is_floating_point() and "running_" not in k:
            t.requires_grad_(True)
        out[k] = t
    return out


def finetune_fixture_data(seed, n_train=6, n_valid=2, S=64, bs=2):
    """Synthetic finetuning split behind tests/golden/finetune_ref.npz: z-scored random images and blob-like vessel masks
    (one-hot (2,H,W) float64, as dataset.py:48 hands them over), batched as lists of (x (B,S,S) f32, y (B,2,S,S) f64)."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(seed)
    n = n_train + n_valid
    x = torch.randn(n, S, S, generator=g)
    blob = F.avg_pool2d(torch.randn(n, 1, S, S, generator=g), 7, 1, 3)[:, 0]
    y1 = (blob > blob.flatten(1).quantile(0.8, dim=1).view(-1, 1, 1)).double()
    y = torch.stack([1 - y1, y1], 1)
    x = x + 0.8 * y1.float()                                   # (something to learn)
    mk = lambda lo, hi: [(x[i:i + bs].clone(), y[i:i + bs].clone()) for i in range(lo, hi, bs)]
    return mk(0, n_train), mk(n_train, n)


def checkpoint_layout_cases(sd):
    """Synthetic checkpoints in the five foreign layouts load_model accepts (train.py:240-308), built from a full UNet state dict:
    file name -> object for torch.save.  Used by gen_golden.py::gen_load_model (the reference's own load_model) and by the tests."""
    enc = {k: v for k, v in sd.items() if "down_conv" in k or k.startswith("double_conv")}
    dec = {k: v for k, v in sd.items() if "up_conv" in k or "conv_last" in k}
    return {
        "spark.pth": {"module": {**{"sparse_encoder.sp_cnn." + k: v for k, v in enc.items()}, **{"dense_decoder." + k: v for k, v in dec.items()},
                                 "mask_tokens.0": torch.zeros(1)}, "epoch": 3},
        "cmunet.pth": {"meta": {"mmengine_version": "0.10.5"},
                       "state_dict": {**{"backbone." + k: v for k, v in enc.items()}, **{"pixel_decoder." + k: v for k, v in dec.items()},
                                      **{"target_backbone." + k: v * 0 for k, v in enc.items()}}},
        # (a raw encoder dict reaches the reference's "encoder only" branch only if it ALSO has a "meta" entry without
        # mmengine_version: train.py:262 indexes checkpoint["meta"] before it gets there -- a plain dict raises KeyError('meta'))
        "encoder.pth": {"meta": {}, **{"module." + k: v for k, v in enc.items()}},
        "moco.ckpt": {"state_dict": {**{"encoder_q." + k: v for k, v in enc.items()}, "queue": torch.zeros(4, 4)}},
        "genesis.pt": {"epoch": 1, "state_dict": {"module." + k: v for k, v in sd.items()}},
    }
