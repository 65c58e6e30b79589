"""Oracle: CM-UNet pretraining step, restated on CPU.  TEST INFRASTRUCTURE ONLY.

PINNED BY THE REFERENCE: oracle/gen_golden.py::gen_cmunet runs the reference's own cmae modules (CM_UNet.forward_train /
backward / momentum_update, CMUNetPretrainHead, NonLinearNeck, UNet_encoder incl. its numpy patch mask, MomentumUpdateHook)
in the build container behind an mmengine / mmcv plumbing stand-in, asserts this file equal on every number and writes
tests/golden/cmunet_ref.npz (shipped cmunet_config.py, 224 x 224, bs 4; one rank).

Follows, under /root/reference/Pretraining/CM-UNet/:
  create_random_patch_mask  cmae/models/backbones/UNet_encoder.py:106-139
  random_masking            UNet_encoder.py:141-158   (mask of sample 0 applied to the whole batch: A-1)
  encoder                   UNet_encoder.py:76-84
  decoder                   cmae/models/necks/munet_neck.py:75-82
  nonlinear_neck            cmae/models/necks/nonlinear_neck.py:88-102 with cfg cmunet_config.py:18-38
                            (with_avg_pool=False, num_layers=2, with_bias=True, with_last_bn=False;
                             SyncBN eps 1e-6 == BatchNorm1d on one rank)
  head                      cmae/models/heads/cmunet_head.py:47-91
  forward_train             cmae/models/algorithms/cmunet.py:108-135
  momentum_update           cmunet.py:78-92
  momentum_schedule         cmae/core/hooks/momentum_update_hook.py:29-40
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import unet as U

NECK_BN_EPS = 1e-6


def create_random_patch_mask(batch_size, img_size, patch_size=16, mask_ratio=0.65, rng=None):
    """UNet_encoder.py:106-139.  ``rng``: a numpy RandomState-like with .shuffle (the reference uses the
    global numpy RNG).  Returns uint8 (B, img, img) with 1 = masked."""
    rng = np.random if rng is None else rng
    per_side = img_size // patch_size
    num_patches = per_side ** 2
    target = int(mask_ratio * img_size * img_size)
    area = patch_size * patch_size
    mask = np.zeros((batch_size, img_size, img_size), dtype=np.uint8)
    for i in range(batch_size):
        cur = 0
        idx = np.arange(num_patches)
        rng.shuffle(idx)
        for k in idx:
            r = (k // per_side) * patch_size
            c = (k % per_side) * patch_size
            if cur + area <= target:
                mask[i, r:r + patch_size, c:c + patch_size] = 1
                cur += area
            if cur >= target:
                break
    return mask


def encoder(x_bhw, mask_bhw_u8, sd, prefix, training=True, ref_compat=True):
    """UNet_encoder.py:76-84, 141-158.  ``mask_bhw_u8`` is an explicit input (RNG lives outside).
    ref_compat=True multiplies the whole batch by (1 - mask[0]) (UNet_encoder.py:156)."""
    x = x_bhw.unsqueeze(1)
    m = torch.as_tensor(mask_bhw_u8)
    if ref_compat:
        x = x * (1 - m[0]).to(x.dtype)
    else:
        x = x * (1 - m).to(x.dtype).unsqueeze(1)
    latent, skips = U.encoder_forward(x, sd, prefix, training)
    return latent, m, skips


def decoder(latent, skips, sd, prefix, training=True):
    """munet_neck.py:75-82."""
    return U.decoder_forward(latent, skips, sd, prefix, "conv_transpose", training)


def nonlinear_neck(x_b1n, sd, prefix, training=True):
    """nonlinear_neck.py:88-102 in the cmunet_config.py configuration:
    x[:,0,:] -> flatten -> fc0 (bias) -> BN(eps 1e-6) -> ReLU -> fc1 (no bias) -> unsqueeze(1)."""
    x = x_b1n[:, 0, :]
    x = x.reshape(x.size(0), -1)
    x = F.linear(x, sd[prefix + "fc0.weight"], sd.get(prefix + "fc0.bias"))
    x = F.batch_norm(x, sd[prefix + "bn0.running_mean"], sd[prefix + "bn0.running_var"],
                     sd[prefix + "bn0.weight"], sd[prefix + "bn0.bias"], training, 0.1, NECK_BN_EPS)
    if training and (prefix + "bn0.num_batches_tracked") in sd:
        sd[prefix + "bn0.num_batches_tracked"] += 1
    x = F.relu(x)
    x = F.linear(x, sd[prefix + "fc1.weight"], sd.get(prefix + "fc1.bias"))
    return x.unsqueeze(1)


def recon_target(img_bhw):
    """cmunet_head.py:63-67 -- per-ROW normalisation (dim=-1 of (B,H,W)), unbiased variance (A-3)."""
    mean = img_bhw.mean(dim=-1, keepdim=True)
    var = img_bhw.var(dim=-1, keepdim=True)
    return (img_bhw - mean) / (var + 1.e-6) ** .5


def masked_mse(pred_bhw, img_bhw, mask_bhw):
    """cmunet_head.py:62-70."""
    target = recon_target(img_bhw)
    rec = (pred_bhw - target) ** 2
    m = mask_bhw.to(rec.dtype) if not mask_bhw.is_floating_point() else mask_bhw
    return (rec * m).sum() / m.sum()


def infonce_inbatch(pred_s_bd, proj_t_all_nd, temperature, rank=0, ct_weight=1.0):
    """cmunet_head.py:72-88.  ``proj_t_all_nd`` is the all-gathered (world*B, D) target projection
    (already L2-normalised and detached); labels i + B*rank; loss scaled by ct_weight*2*t."""
    pred = F.normalize(pred_s_bd, dim=1, p=2)
    score = pred @ proj_t_all_nd.t().detach() / temperature
    bs = score.size(0)
    label = torch.arange(bs, dtype=torch.long) + bs * rank
    return ct_weight * 2 * temperature * F.cross_entropy(score, label)


def head(img, pred_pixel, mask_s, proj_s, proj_t, sd, prefix="head.", temperature=0.07,
         ct_weight=1.0, rc_weight=1.0, training=True, gather=None, rank=0):
    """cmunet_head.py:47-91.  ``gather``: callable emulating concat_all_gather (identity on one rank)."""
    loss_rc = masked_mse(pred_pixel, img, mask_s)
    pred_s = nonlinear_neck(proj_s, sd, prefix + "predictor.", training)
    proj_t = F.normalize(proj_t.squeeze(1), dim=1, p=2)
    proj_t_all = proj_t if gather is None else gather(proj_t)
    loss_ct = infonce_inbatch(pred_s.squeeze(1), proj_t_all, temperature, rank, ct_weight)
    return {"loss_ct": loss_ct, "loss_rc": rc_weight * loss_rc}


def forward_train(img, img_t, mask_s, reduce_w, reduce_b, sd, temperature=0.07, ct_weight=1.0,
                  rc_weight=1.0, training=True, ref_compat=True, gather=None, rank=0):
    """cmunet.py:108-135.  Keys of ``sd`` use the CM_UNet module names: backbone., target_backbone.,
    pixel_decoder., feature_decoder., projector., target_projector., head.predictor.
    ``reduce_w/reduce_b``: the per-call random Conv2d(1024,256,1) of cmunet.py:128 as explicit inputs (A-2).
    The target branch carries no gradient into trainable parameters (target params frozen)."""
    B, H, W = img.shape
    latent_s, mask_s, skip_s = encoder(img, mask_s, sd, "backbone.", training, ref_compat)
    with torch.no_grad():
        zero_mask = torch.zeros_like(torch.as_tensor(mask_s))
        latent_t, _, _ = encoder(img_t, zero_mask, sd, "target_backbone.", training, ref_compat)
    pred_pixel = decoder(latent_s, skip_s, sd, "pixel_decoder.", training)
    pred_feature = decoder(latent_s, skip_s, sd, "feature_decoder.", training)
    # cmunet.py:126 -- mean over the decoder's 2 output channels, kept as a 1-channel image
    proj_s = nonlinear_neck(torch.mean(pred_feature, dim=1, keepdim=True), sd, "projector.", training)
    with torch.no_grad():
        # cmunet.py:128-131 -- 256*(H/16)*(W/16) == H*W, so the reduced latent is re-viewed as an image
        lt = F.conv2d(latent_t, reduce_w, reduce_b)
        lt = lt.reshape(B, -1).reshape(B, 1, H, W)
        proj_t = nonlinear_neck(torch.mean(lt, dim=1, keepdim=True), sd, "target_projector.", training)
    return head(img, pred_pixel[:, 1], mask_s, proj_s, proj_t, sd, "head.", temperature, ct_weight,
                rc_weight, training, gather, rank)


def momentum_schedule(cur_iter, max_iter, base_momentum=0.996, end_momentum=0.996):
    """momentum_update_hook.py:29-40."""
    return end_momentum - (end_momentum - base_momentum) * (math.cos(math.pi * cur_iter / float(max_iter)) + 1) / 2


def momentum_update(sd, momentum, pairs=(("backbone.", "target_backbone."), ("projector.", "target_projector."))):
    """cmunet.py:78-92: p_t = p_t*m + p_o*(1-m) over *parameters* (not buffers) of backbone and projector."""
    for src, dst in pairs:
        for k in list(sd.keys()):
            if k.startswith(src) and "running_" not in k and "num_batches" not in k:
                kt = dst + k[len(src):]
                sd[kt] = sd[kt] * momentum + sd[k].detach() * (1. - momentum)


def make_neck_sd(prefix, in_ch, hid, out, seed, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    sd[prefix + "fc0.weight"] = (torch.randn(hid, in_ch, generator=g, dtype=torch.float64) / in_ch ** 0.5).to(dtype)
    sd[prefix + "fc0.bias"] = (0.1 * torch.randn(hid, generator=g, dtype=torch.float64)).to(dtype)
    sd[prefix + "bn0.weight"] = (1 + 0.2 * torch.randn(hid, generator=g, dtype=torch.float64)).to(dtype)
    sd[prefix + "bn0.bias"] = (0.1 * torch.randn(hid, generator=g, dtype=torch.float64)).to(dtype)
    sd[prefix + "bn0.running_mean"] = torch.zeros(hid, dtype=dtype)
    sd[prefix + "bn0.running_var"] = torch.ones(hid, dtype=dtype)
    sd[prefix + "bn0.num_batches_tracked"] = torch.zeros((), dtype=torch.int64)
    sd[prefix + "fc1.weight"] = (torch.randn(out, hid, generator=g, dtype=torch.float64) / hid ** 0.5).to(dtype)
    return sd


def make_cmunet_sd(seed, img_size=224):
    """Seeded state dict of the whole CM_UNet at the shipped configuration (cmunet_config.py: projector in = img_size^2 =
    50,176), in the reference's key names.  tests/golden/cmunet_ref.npz was produced by loading exactly this into the
    reference's own CM_UNet (oracle/gen_golden.py::gen_cmunet); the tests regenerate it from the seed (217.8 M parameters do
    not fit a fixture)."""
    sd = {}
    for prefix, s, enc in (("backbone.", seed, True), ("target_backbone.", seed + 1, True), ("pixel_decoder.", seed + 2, False),
                           ("feature_decoder.", seed + 3, False)):
        part = U.make_state_dict(base_ch=64, depth=5, seed=s, encoder=enc, decoder=not enc)
        sd.update({prefix + k: v for k, v in part.items()})
    sd.update(make_neck_sd("projector.", img_size * img_size, 1536, 256, seed + 4))
    sd.update(make_neck_sd("target_projector.", img_size * img_size, 1536, 256, seed + 5))
    sd.update(make_neck_sd("head.predictor.", 256, 1536, 256, seed + 6))
    return sd


def cmunet_fixture_inputs(seed, B=4, S=224):
    """The inputs of tests/golden/cmunet_ref.npz, regenerated: images from a seeded generator, the patch mask as the reference
    draws it (numpy's global stream seeded with seed + 11: UNet_encoder.py:106-139), the per-call Conv2d(1024, 256, 1) of
    cmunet.py:128 as torch creates it right after torch.manual_seed(seed + 12), and the head-only case's tensors."""
    g = torch.Generator().manual_seed(seed + 10)
    img, img_t = torch.randn(B, S, S, generator=g), torch.randn(B, S, S, generator=g)
    mask = create_random_patch_mask(B, S, 16, 0.65, np.random.RandomState(seed + 11))
    state = torch.get_rng_state()
    torch.manual_seed(seed + 12)
    rc = torch.nn.Conv2d(1024, 256, kernel_size=1)
    torch.set_rng_state(state)
    return img, img_t, mask, rc.weight.detach().clone(), rc.bias.detach().clone()


def head_fixture_inputs(seed, B=4, H=32, W=48):
    """(x, pred_pixel, mask, proj_s, proj_t) of one rank for tests/golden/cmunet_head_2rank.npz."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, H, W, generator=g) * 2 + 0.5
    pred = torch.randn(B, H, W, generator=g)
    mk = (torch.rand(B, H, W, generator=g) > 0.4).to(torch.uint8)
    ps = torch.randn(B, 1, 256, generator=g)
    pt = torch.randn(B, 1, 256, generator=g)
    return x, pred, mk, ps, pt
