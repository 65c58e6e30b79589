"""Oracle: SparK sparse masked-conv pretraining on the UNet, restated on CPU.  TEST INFRASTRUCTURE ONLY.

Pinned against the reference itself: oracle/gen_golden.py imports Pretraining/Spark (behind the timm /
tensorboard / UNET stubs of SURVEY Appendix C-3) and checks this restatement before writing
tests/golden/spark_unet.npz.

Follows /root/reference/Pretraining/Spark/:
  sparse conv / pool      encoder.py:20-23   dense op, then  x *= active (up-sampled to the op's resolution)
  sparse BatchNorm        encoder.py:26-36   BN1d over the ACTIVE positions only, scattered into zeros
  sparse UNet encoder     models/custom.py:15-40, 113-182  (DoubleConv / DownBlock converted by
                          SparseEncoder.dense_model_to_sparse, encoder.py:158-208); feature maps = 4 skips + bottleneck
  mask                    spark.py:82-86     rand -> argsort -> first len_keep = round(f*f*(1-ratio)) patches ACTIVE
  densify                 spark.py:98-111    where(active, feat, mask_token) per scale (no norm / proj for the full UNet)
  decoder                 decoder.py:39-58   UpBlocks of Finetuning/model.py + Conv2d(64, 1, 1)
  loss                    spark.py:112-123   per-patch normalised L2, averaged over NON-active patches
"""
import torch
import torch.nn.functional as F

from . import unet as U

BN_EPS = 1e-5


def _active_ex(active_b1ff, H, W):
    return active_b1ff.repeat_interleave(H // active_b1ff.shape[-2], 2).repeat_interleave(W // active_b1ff.shape[-1], 3)


def sp_conv(x, w, b, active_b1ff):
    y = F.conv2d(x, w, b, padding=1)
    return y * _active_ex(active_b1ff, y.shape[2], y.shape[3]).to(y.dtype)


def sp_bn(x, sd, p, active_b1ff, training=True):
    ii = _active_ex(active_b1ff, x.shape[2], x.shape[3]).squeeze(1).nonzero(as_tuple=True)
    bhwc = x.permute(0, 2, 3, 1)
    nc = bhwc[ii]
    nc = F.batch_norm(nc, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"], sd[p + "bias"], training, 0.1, BN_EPS)
    if training and (p + "num_batches_tracked") in sd:
        sd[p + "num_batches_tracked"] += 1
    out = torch.zeros_like(bhwc)
    out[ii] = nc
    return out.permute(0, 3, 1, 2)


def sp_double_conv(x, sd, prefix, active, training=True):
    """(``oracle.unet.TAP``, test hook: see oracle/unet.py -- the masked conv output and the ReLU of a layer can be replaced by values taken
    from the HIP path's own forward, so that the float64 backward runs on the gates the kernels saw.)"""
    for conv, bn in ((0, 1), (3, 4)):
        x = sp_conv(x, sd[f"{prefix}{conv}.weight"], sd[f"{prefix}{conv}.bias"], active)
        if U.TAP is not None:
            x = U.TAP.conv(f"{prefix}{conv}.", x)
        x = sp_bn(x, sd, f"{prefix}{bn}.", active, training)
        x = F.relu(x) if U.TAP is None else U.TAP.act(f"{prefix}{conv}.", x)
    return x


def sparse_encoder(x_b1hw, sd, prefix, active, training=True):
    """custom.py:152-182 with hierarchical=True: [skip1, skip2, skip3, skip4, bottleneck]."""
    n_down = U.depth_of(sd, prefix) - 1
    feats = []
    x = x_b1hw
    for i in range(1, n_down + 1):
        skip = sp_double_conv(x, sd, f"{prefix}down_conv{i}.double_conv.double_conv.", active, training)
        pooled = F.max_pool2d(skip, 2)
        x = pooled * _active_ex(active, pooled.shape[2], pooled.shape[3]).to(pooled.dtype)
        feats.append(skip)
    feats.append(sp_double_conv(x, sd, prefix + "double_conv.double_conv.", active, training))
    return feats


def make_active(B, f, mask_ratio, generator=None):
    """spark.py:82-86 -> bool (B,1,f,f), True = kept (active)."""
    len_keep = round(f * f * (1 - mask_ratio))
    idx = torch.rand(B, f * f, generator=generator).argsort(dim=1)[:, :len_keep]
    return torch.zeros(B, f * f, dtype=torch.bool).scatter_(1, idx, True).view(B, 1, f, f)


def patchify(bchw, p):
    B, C, H, W = bchw.shape
    h, w = H // p, W // p
    x = bchw.reshape(B, C, h, p, w, p)
    x = torch.einsum('bchpwq->bhwpqc', x)
    return x.reshape(B, h * w, C * p * p)


def forward(inp_b1hw, active_b1ff, sd, mask_tokens, enc_prefix="sparse_encoder.sp_cnn.", dec_prefix="dense_decoder.", training=True):
    """spark.py:88-131 for the full-UNet configuration (densify_norm '', no projections).
    ``mask_tokens``: list ordered from the smallest feature map (bottleneck) to the largest, each (1,C,1,1)."""
    ratio = inp_b1hw.shape[-1] // active_b1ff.shape[-1]
    active_b1hw = active_b1ff.repeat_interleave(ratio, 2).repeat_interleave(ratio, 3)
    feats = sparse_encoder(inp_b1hw * active_b1hw.to(inp_b1hw.dtype), sd, enc_prefix, active_b1ff, training)
    feats = feats[::-1]
    cur = active_b1ff
    to_dec = []
    for i, f in enumerate(feats):
        to_dec.append(torch.where(cur.expand_as(f), f, mask_tokens[i].expand_as(f)))
        cur = cur.repeat_interleave(2, 2).repeat_interleave(2, 3)
    # decoder.py:49-55
    x = to_dec[0]
    n_up = len(to_dec) - 1
    for k in range(n_up):
        x = U.up_block(x, to_dec[k + 1], sd, f"{dec_prefix}up_conv{n_up - k}.", "conv_transpose", training)
    rec = F.conv2d(x, sd[dec_prefix + "conv_last.weight"], sd[dec_prefix + "conv_last.bias"])
    return recon_loss(inp_b1hw, rec, active_b1ff), rec


def recon_loss(inp_b1hw, rec_b1hw, active_b1ff):
    """spark.py:112-123: per-patch normalised L2 between the reconstruction and the input, averaged over the NON-active patches."""
    ratio = inp_b1hw.shape[-1] // active_b1ff.shape[-1]
    inp, recp = patchify(inp_b1hw, ratio), patchify(rec_b1hw, ratio)
    mean = inp.mean(dim=-1, keepdim=True)
    var = (inp.var(dim=-1, keepdim=True) + 1e-6) ** .5
    inp = (inp - mean) / var
    l2 = ((recp - inp) ** 2).mean(dim=2)
    non_active = active_b1ff.logical_not().int().view(active_b1ff.shape[0], -1)
    return (l2 * non_active).sum() / (non_active.sum() + 1e-8)
