"""Drop-in for the CM-UNet pretraining model (reference: Pretraining/CM-UNet/cmae/models/...), on the HIP engine.

  UNet_encoder(out_classes, up_sample_mode, patch_size=16, mask_ratio=0.65)   backbones/UNet_encoder.py:51-158
  MUNetPretrainDecoder(out_classes=2, up_sample_mode='conv_transpose')         necks/munet_neck.py:52-82
  NonLinearNeck(in_channels, hid_channels, out_channels, num_layers=2, ...)    necks/nonlinear_neck.py:35-102
  CMUNetPretrainHead(predictor, temperature, ct_weight, rc_weight)             heads/cmunet_head.py:25-91
  CM_UNet(backbone, neck, head, base_momentum=0.996)                           algorithms/cmunet.py:7-135
  MomentumUpdateHook.momentum(cur_iter, max_iter)                              core/hooks/momentum_update_hook.py:29-40

The mmengine registry / Runner are not rebuilt (out of scope); the classes take the same config
dicts (``cmunet_config.py:5-42``) and build their children directly.  Conv blocks, losses, EMA, the optimiser, the MLP
necks (fc -> BN1d -> ReLU -> fc on <= 32 rows per GPU: weight-streaming skinny GEMMs + BatchNorm1d kernels, csrc/skinny.hip,
necks.hip) and the target latent's 1x1 reduction all run on the HIP kernels; only a Linear on more than 32 rows would be a
plain library GEMM.

Reference quirks are replicated behind ``ref_compat=True``: the mask of sample 0 masks
the whole batch (A-1); a fresh randomly initialised Conv2d(1024,256,1) reduces the target latent at every
call (A-2, weights injectable for tests); per-row target normalisation (A-3).  Hard-coded ``.cuda()`` /
``"cuda:0"`` strings of the reference become the module's device.
"""
import math
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn

from . import _lib, ops
from .model import DoubleConv, DownBlock, UpBlock, _EngineOwner, _named_state, _param_args, _require_cuda
from .ops import Act
from .optim import claim_grad_sink, dp_exchanges
from .pretrain import create_random_patch_mask, random_patch_mask_device


def _init_weights_ref(m):
    """UNet_encoder.py:90-104 / munet_neck.py:84-116: kaiming-normal(fan_out, relu) convs, BN (1, 0),
    xavier-normal linears, zero biases."""
    if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
        nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
        if m.bias is not None:
            nn.init.zeros_(m.bias)
    elif isinstance(m, nn.BatchNorm2d):
        nn.init.constant_(m.weight, 1)
        nn.init.zeros_(m.bias)
    elif isinstance(m, nn.Linear):
        nn.init.xavier_normal_(m.weight)
        if m.bias is not None:
            nn.init.zeros_(m.bias)


# ---------------------------------------------------------------------------------------------------
# autograd-wrapped fused losses
# ---------------------------------------------------------------------------------------------------
class _MaskedMSEFn(torch.autograd.Function):
    """cmunet_head.py:62-70 on logits[:, channel] (kernel cmu_masked_mse_fwd_bwd)."""

    @staticmethod
    def forward(ctx, logits, channel, img, mask):
        B, K, H, W = logits.shape
        loss = torch.empty(1, dtype=torch.float32, device=logits.device)
        dl = torch.empty_like(logits)
        ws = torch.empty(_lib.lib().cmu_masked_mse_ws_bytes(B, H), dtype=torch.uint8, device=logits.device)
        ops.masked_mse_fwd_bwd(logits.detach().contiguous(), channel, img.contiguous(), mask.contiguous(), loss, dl, 1.0, ws)
        ctx.save_for_backward(dl)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl * g, None, None, None


def masked_mse_loss(logits, channel, img, mask):
    return _MaskedMSEFn.apply(logits, channel, img, mask)


class _InfoNCEFn(torch.autograd.Function):
    """cmunet_head.py:72-88 (kernel cmu_infonce_inbatch_fwd_bwd); keys are detached by construction."""

    @staticmethod
    def forward(ctx, pred, keys, rank, temperature, ct_weight):
        B, D = pred.shape
        loss = torch.empty(1 + B, dtype=torch.float32, device=pred.device)
        dp = torch.empty_like(pred)
        ops.infonce_inbatch_fwd_bwd(pred.detach().contiguous(), keys.detach().contiguous(), loss, dp, rank, temperature, ct_weight)
        ctx.save_for_backward(dp)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dp,) = ctx.saved_tensors
        return dp * g, None, None, None, None


@torch.no_grad()
def concat_all_gather(tensor):
    """cmunet_head.py:9-22 (mmengine all_gather -> cat) as one RCCL all-gather into a single tensor."""
    if not dp_exchanges():
        return tensor
    out = torch.empty((dist.get_world_size() * tensor.shape[0],) + tuple(tensor.shape[1:]), dtype=tensor.dtype, device=tensor.device)
    dist.all_gather_into_tensor(out, tensor.contiguous())
    return out


# ---------------------------------------------------------------------------------------------------
# encoder / decoder modules (module-boundary tensors are NCHW fp32 like the reference's)
# ---------------------------------------------------------------------------------------------------
class _EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, x, mask, mask_per_sample, names, *params):
        eng = module._engine(x.device)
        sd = _named_state(module)
        ectx = eng.encoder_forward(sd, x.detach().float().contiguous(), module.training, "", mask, mask_per_sample)
        ctx.module, ctx.ectx, ctx.names, ctx.eng = module, ectx, names, eng
        outs = [ops.apply_to_nchw(ectx["latent"])] + [ops.apply_to_nchw(s) for s in ectx["skips"]]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        eng, ectx = ctx.eng, ctx.ectx
        sd = _named_state(ctx.module)

        def to_act(g, like):
            a = eng._new(like.B, like.H, like.W, like.C)
            ops.nchw_to_nhwc(g.contiguous().float(), a)
            return a
        d_latent = to_act(gouts[0], ectx["latent"])
        d_skips = [to_act(g, s) for g, s in zip(gouts[1:], ectx["skips"])]
        grads = {}
        eng.encoder_backward(sd, ectx, d_latent, d_skips, grads)
        ctx.ectx = None
        return (None, None, None, None, None, *[grads.get(n) for n in ctx.names])


class UNet_encoder(_EngineOwner, nn.Module):
    """UNet down path + bottleneck with random patch masking (UNet_encoder.py:51-158).

    forward(x (B,H,W)) -> (latent (B,C,H/2^d,W/2^d), mask (B,H,W) uint8 on the device, [skip1..skip4]).
    """

    def __init__(self, out_classes=2, up_sample_mode='conv_transpose', patch_size=16, mask_ratio=0.65,
                 base_ch=64, depth=5, dtype="f16", ref_compat=True):
        super().__init__()
        self.up_sample_mode = up_sample_mode
        self.dtype = dtype
        self.ref_compat = ref_compat
        chans = [base_ch * 2 ** i for i in range(depth)]
        cin = 1
        for i in range(depth - 1):
            setattr(self, f"down_conv{i + 1}", DownBlock(cin, chans[i], dtype))
            cin = chans[i]
        self.double_conv = DoubleConv(cin, chans[-1], dtype)
        self.patch_size = patch_size
        self.mask_ratio = mask_ratio

    def init_weights(self):
        self.apply(_init_weights_ref)

    def create_random_patch_mask(self, batch_size, img_size=256):
        """numpy (B,img,img) uint8, the reference's RNG consumption (UNet_encoder.py:106-139)."""
        return create_random_patch_mask(batch_size, img_size, self.patch_size, self.mask_ratio)

    def make_mask(self, x, generator=None, host_rng=None):
        B, H, W = x.shape[0], x.shape[-2], x.shape[-1]
        if host_rng is not None:          # reference-exact host generation
            return torch.from_numpy(create_random_patch_mask(B, H, self.patch_size, self.mask_ratio, host_rng)).to(x.device)
        return random_patch_mask_device(B, H, W, self.patch_size, self.mask_ratio, generator, x.device)

    def forward(self, x, mask=None):
        _require_cuda(x, "UNet_encoder")
        if mask is None:
            mask = self.make_mask(x)
        names, params = _param_args(self)
        outs = _EncoderFn.apply(self, x, mask, not self.ref_compat, names, *params)
        return outs[0], mask, list(outs[1:])


class _DecoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, nskip, names, latent, *rest):
        skips, params = rest[:nskip], rest[nskip:]
        eng = module._engine(latent.device)
        sd = _named_state(module)

        def to_act(t):
            B, C, H, W = t.shape
            a = eng._new(B, H, W, C)
            ops.nchw_to_nhwc(t.detach().float().contiguous(), a)
            return a
        dctx = eng.decoder_forward(sd, to_act(latent), [to_act(s) for s in skips], module.training, "", None, True)
        ctx.module, ctx.dctx, ctx.names, ctx.eng, ctx.nskip = module, dctx, names, eng, nskip
        return dctx["logits"]

    @staticmethod
    def backward(ctx, dlogits):
        sd = _named_state(ctx.module)
        grads = {}
        d_latent, d_skips = ctx.eng.decoder_backward(sd, ctx.dctx, dlogits.contiguous().float(), grads, True)
        gl = ops.apply_to_nchw(d_latent)
        gs = [ops.apply_to_nchw(d) for d in d_skips]
        ctx.dctx = None
        return (None, None, None, gl, *gs, *[grads.get(n) for n in ctx.names])


class MUNetPretrainDecoder(_EngineOwner, nn.Module):
    """UNet up path + 1x1 head (munet_neck.py:52-82). forward(x, skip) with skip = [skip1..skip4]."""

    def __init__(self, out_classes=2, up_sample_mode='conv_transpose', base_ch=64, depth=5, dtype="f16"):
        super().__init__()
        self.up_sample_mode = up_sample_mode
        self.dtype = dtype
        chans = [base_ch * 2 ** i for i in range(depth)]
        for i in range(depth - 1, 0, -1):
            setattr(self, f"up_conv{i}", UpBlock(chans[i], chans[i - 1], up_sample_mode, dtype))
        self.conv_last = nn.Conv2d(chans[0], out_classes, kernel_size=1)

    def init_weights(self):
        self.apply(_init_weights_ref)

    def forward(self, x, skip):
        _require_cuda(x, "MUNetPretrainDecoder")
        names, params = _param_args(self)
        return _DecoderFn.apply(self, len(skip), names, x, *skip, *params)


class _SkinnyLinearFn(torch.autograd.Function):
    """nn.Linear on <= 256 rows through the weight-streaming kernels (csrc/skinny.hip, necks.hip): per group of 32 rows one pass
    over the weights for the forward and one for the input gradient; one write of the weight gradient.  ``compute_dt`` None: exact fp32 products
    everywhere; 'f16' / 'bf16' (the AMP configuration): the 16-bit-operand kernel where it is the faster one -- measured at
    K = 262,144, N = 1,536 (tools/skinny_bench.py, profiles/r02_workloads.txt): weight gradient 0.37 ms against 0.55 ms exact;
    the exact forward stages its operands through LDS as contiguous runs (0.41 ms; the 16-bit-operand form still reads 32 rows
    per wave a megabyte apart: 0.72 ms) and the input gradient is faster exact (0.38 against 0.48 ms), so those two stay on the
    exact fp32 kernels (at least the precision autocast would give)."""

    @staticmethod
    def forward(ctx, x, weight, bias, compute_dt):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.compute_dt = compute_dt
        ctx.weight_param = weight if isinstance(weight, torch.nn.Parameter) else None
        return ops.skinny_gemm_fwd(x.detach(), weight.detach(), None if bias is None else bias.detach())

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        dx = ops.skinny_gemm_dgrad(dy, weight.detach()) if ctx.needs_input_grad[0] else None
        # a trainer's gradient arena as the destination (optim.claim_grad_sink): no 1.6 GB copy of the projector's gradient per step
        sink = claim_grad_sink(ctx.weight_param) if ctx.needs_input_grad[1] else None
        dw, db = (ops.skinny_gemm_wgrad(dy, x.detach(), ctx.has_bias, ctx.compute_dt, out=sink) if ctx.needs_input_grad[1] else (None, None))
        if ctx.has_bias and db is None and ctx.needs_input_grad[2]:
            db = dy.sum(0)
        return dx, dw, db, None


def neck_linear(fc, x, compute_dt=None):
    """``fc(x)`` for the necks' Linear layers (nonlinear_neck.py:63-66, 95-101) on the weight-streaming kernels: up to
    ``ops.SKINNY_MAX_ROWS`` = 256 fp32 rows per GPU (the reference's own batch size, cmunet_config.py:55), in_features a multiple
    of 8.  There is no library GEMM behind this (round 4): anything else raises."""
    if ops.skinny_eligible(x, fc.weight):
        return _SkinnyLinearFn.apply(x, fc.weight, fc.bias, compute_dt)
    raise RuntimeError(f"neck_linear: needs a CUDA fp32 (rows <= {ops.SKINNY_MAX_ROWS}, in_features % 8 == 0) input, got "
                       f"{tuple(x.shape)} {x.dtype} on {x.device} (the HIP path has no library / CPU fallback)")


class _BN1dFn(torch.autograd.Function):
    """(Sync)BatchNorm1d (+ the ReLU that follows it) of the necks on the kernels of csrc/necks.hip.  More than one rank and
    ``sync``: the column sums are exchanged (one all-reduce forward, one backward), as nn.SyncBatchNorm does."""

    @staticmethod
    def forward(ctx, x, weight, bias, bn, relu, sync):
        x = x.detach().float().contiguous()
        training = bn.training or bn.running_mean is None
        world = dist.get_world_size() if (sync and training and dist.is_available() and dist.is_initialized()) else 1
        sums, count = None, x.shape[0]
        if world > 1:
            sums = ops.bn1d_colsums(x)
            dist.all_reduce(sums)
            count = x.shape[0] * world
        y, mean, invstd = ops.bn1d_relu_fwd(x, None if weight is None else weight.detach(), None if bias is None else bias.detach(),
                                            bn.running_mean, bn.running_var, bn.momentum if bn.momentum is not None else 0.1, bn.eps,
                                            training, relu, sums, count)
        if training and bn.num_batches_tracked is not None:
            bn.num_batches_tracked += 1
        ctx.save_for_backward(x, y, mean, invstd, weight)
        ctx.relu, ctx.world, ctx.count, ctx.training = relu, world, count, training
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, mean, invstd, weight = ctx.saved_tensors
        dy = dy.contiguous().float()
        sums = None
        if not ctx.training:
            # eval mode: y = gamma * (x - running_mean) * invstd + beta is a fixed affine map -- the same kernel with zero batch sums:
            # dx = gamma * invstd * dz, dgamma = sum dz * xhat, dbeta = sum dz (mean / invstd = the running statistics the forward
            # stored).  (Found by the reference's two-rank head fixture: this branch used to read an unwritten invstd.)
            sums = torch.zeros(2, x.shape[1], dtype=torch.float32, device=x.device)
            dx, dg, db = ops.bn1d_relu_bwd(dy, x, y, mean, invstd, None if weight is None else weight.detach(), ctx.relu, sums, 1,
                                           affine=weight is not None)
            return dx, dg, db, None, None, None
        if ctx.world > 1:
            sums = ops.bn1d_bwd_colsums(dy, x, y, mean, invstd, ctx.relu)
            dist.all_reduce(sums)
        dx, dg, db = ops.bn1d_relu_bwd(dy, x, y, mean, invstd, None if weight is None else weight.detach(), ctx.relu, sums, ctx.count,
                                       affine=weight is not None)
        return dx, dg, db, None, None, None


class NonLinearNeck(nn.Module):
    """fc0 -> BN(eps 1e-6) -> [ReLU -> fc_i (-> BN)]* (nonlinear_neck.py:35-102).  ``norm_cfg`` type 'SyncBN'
    becomes nn.SyncBatchNorm when a process group with more than one rank exists, else BatchNorm1d."""

    def __init__(self, in_channels, hid_channels, out_channels, num_layers=2, with_bias=False, with_last_bn=True,
                 with_last_bn_affine=True, with_last_bias=False, with_avg_pool=True, norm_cfg=dict(type='SyncBN', eps=1e-6),
                 init_cfg=None):
        super().__init__()
        self.with_avg_pool = with_avg_pool
        if with_avg_pool:
            self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.relu = nn.ReLU(inplace=True)
        eps = norm_cfg.get("eps", 1e-5)
        self._sync = norm_cfg.get("type", "BN") == "SyncBN"

        def norm(ch, affine=True):
            return nn.BatchNorm1d(ch, eps=eps, affine=affine)
        self.fc0 = nn.Linear(in_channels, hid_channels, bias=with_bias)
        self.bn0 = norm(hid_channels)
        self.fc_names, self.bn_names = [], []
        for i in range(1, num_layers):
            this = out_channels if i == num_layers - 1 else hid_channels
            if i != num_layers - 1:
                self.add_module(f'fc{i}', nn.Linear(hid_channels, this, bias=with_bias))
                self.add_module(f'bn{i}', norm(this))
                self.bn_names.append(f'bn{i}')
            else:
                self.add_module(f'fc{i}', nn.Linear(hid_channels, this, bias=with_last_bias))
                if with_last_bn:
                    self.add_module(f'bn{i}', norm(this, with_last_bn_affine))
                    self.bn_names.append(f'bn{i}')
                else:
                    self.bn_names.append(None)
            self.fc_names.append(f'fc{i}')
        for m in self.modules():          # init_cfg: Constant(1) on norm layers (nonlinear_neck.py:47-52)
            if isinstance(m, nn.BatchNorm1d) and m.affine:
                nn.init.constant_(m.weight, 1)

    def _bn(self, bn, x, relu):
        """BatchNorm1d over the rows (+ the ReLU that follows when ``relu``); eval mode uses the running statistics."""
        if not x.is_cuda:
            raise RuntimeError("NonLinearNeck: the HIP path needs CUDA/ROCm tensors (no CPU fallback)")
        return _BN1dFn.apply(x, bn.weight if bn.affine else None, bn.bias if bn.affine else None, bn, relu, self._sync)

    def forward(self, x):
        if self.with_avg_pool:
            x = self.avgpool(x)
        else:
            x = x[:, 0, :]
        x = x.reshape(x.size(0), -1)
        cdt = getattr(self, "compute_dtype", None)
        layers = [("fc0", "bn0")] + list(zip(self.fc_names, self.bn_names))
        for i, (fc_name, bn_name) in enumerate(layers):
            x = neck_linear(getattr(self, fc_name), x, cdt)
            last = i == len(layers) - 1
            if bn_name is not None:
                x = self._bn(getattr(self, bn_name), x, relu=not last)       # ReLU precedes every later fc (nonlinear_neck.py:96)
            elif not last:
                x = self.relu(x)
        return x.unsqueeze(dim=1)


class CMUNetPretrainHead(nn.Module):
    """Masked-reconstruction + contrastive losses (cmunet_head.py:25-91)."""

    def __init__(self, predictor, temperature=0.07, ct_weight=1.0, rc_weight=1.0):
        super().__init__()
        self.predictor = _build(predictor) if isinstance(predictor, dict) else predictor
        self.t = temperature
        self.ct_weight = ct_weight
        self.rc_weight = rc_weight

    def forward(self, x, pred_logits, mask_s, proj_s, proj_t, pred_channel=1):
        """``pred_logits`` is the pixel decoder's (B,2,H,W) output; the reference passes pred_pixel[:,1]
        (cmunet.py:133) -- the channel is selected inside the fused loss kernel instead."""
        loss_rc = masked_mse_loss(pred_logits, pred_channel, x, mask_s)
        pred_s = self.predictor(proj_s).squeeze(dim=1)
        pt = proj_t.squeeze(dim=1).detach().contiguous().float()
        ptn = torch.empty_like(pt)
        ops.l2_normalize_rows(pt, ptn)                                   # F.normalize(proj_t, dim=1)
        keys = concat_all_gather(ptn)
        rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
        loss_ct = _InfoNCEFn.apply(pred_s.float(), keys, rank, self.t, self.ct_weight)    # normalises pred_s inside
        return {'loss_ct': loss_ct, 'loss_rc': self.rc_weight * loss_rc}


_TYPES = {}


def _build(cfg):
    cfg = dict(cfg)
    return _TYPES[cfg.pop("type")](**cfg)


_SHARED_SKIPS = os.environ.get("CMU_SHARED_SKIPS", "1") != "0"     # A/B: "0" = the feature decoder copies the skips into its own concat buffers


class _CMUNetFn(torch.autograd.Function):
    """Fused conv part of CM_UNet.forward_train (cmunet.py:121-124): online encoder (masked), target encoder,
    pixel + feature decoders sharing the skips; boundary tensors are the two 2-channel logit maps and the
    target latent."""

    @staticmethod
    def forward(ctx, module, img, img_t, mask, reduce_w, reduce_b, names, *params):
        eng = module._engine(img.device)
        sd = _named_state(module)
        tr = module.training
        x = img.detach().float().contiguous()
        # the online encoder writes its skips straight into the pixel decoder's concat buffers (as the fused UNet does); the feature
        # decoder reads them from the same buffers through a (skip, up) view of channels [up_pixel | skip | up_feature] with its first
        # convs' input channels rotated to match (engine.decoder_alloc_shared; CMU_SHARED_SKIPS=0: its own buffers and four copies)
        shared = eng.decoder_alloc_shared(sd, x.shape[0], x.shape[1], x.shape[2], "pixel_decoder.", "feature_decoder.") if _SHARED_SKIPS else None
        fcats = None
        if shared is not None:
            pcats, fcats = shared
        else:
            pcats = eng.decoder_alloc(sd, x.shape[0], x.shape[1], x.shape[2], "pixel_decoder.")
        nd = eng.n_down(sd, "backbone.")
        fits = len(pcats) == nd and all(c["Cskip"] == sd[f"backbone.down_conv{i + 1}.double_conv.double_conv.0.weight"].shape[0]
                                        for i, c in enumerate(pcats))
        if not fits:
            fcats = None
        sdf = eng.rotated_weights(sd, "feature_decoder.", fcats) if fcats is not None else sd
        eng.prepack(sdf)               # every conv / conv-transpose weight pack of the four networks in one launch (the feature
                                       # decoder's first convs from their rotated copies)
        skip_out = [Act(c["buf"], c["Cup"], c["Cskip"]) for c in pcats] if fits else None
        skip_affine = [(c["scale"][c["Cup"]:], c["shift"][c["Cup"]:]) for c in pcats] if fits else None
        ectx = eng.encoder_forward(sd, x, tr, "backbone.", mask, not module.ref_compat, skip_out, skip_affine)
        tctx = eng.encoder_forward(sd, img_t.detach().float().contiguous(), tr, "target_backbone.", None)
        pctx = eng.decoder_forward(sd, ectx["latent"], ectx["skips"], tr, "pixel_decoder.", pcats if fits else None, True)
        fctx = eng.decoder_forward(sdf, ectx["latent"], ectx["skips"], tr, "feature_decoder.", fcats, True)
        fctx["shared_cats"] = fcats
        # cmunet.py:128-129: the per-call Conv2d(C, C/4, 1) on the target latent, straight from the raw NHWC latent and its
        # pending BatchNorm+ReLU into the NCHW fp32 tensor the reference re-views as an image (no gradient: target branch)
        latent_t = ops.conv1x1_nchw_fwd(tctx["latent"], reduce_w.detach().float(), None if reduce_b is None else reduce_b.detach().float())
        ctx.module, ctx.names, ctx.eng = module, names, eng
        ctx.saved = (ectx, pctx, fctx)
        if getattr(module, "keep_ctx", False):      # tests: the engine's saved state of this forward (raw conv outputs + pending transforms)
            module.last_ctx = (ectx, pctx, fctx)
        return pctx["logits"], fctx["logits"], latent_t

    @staticmethod
    def backward(ctx, d_pix, d_feat, _d_latent_t):
        eng = ctx.eng
        sd = _named_state(ctx.module)
        ectx, pctx, fctx = ctx.saved
        grads = {}
        # a data-parallel trainer is told as soon as a sub-network's gradients are final (pretrain.ArenaTrainer.notify_ready): their
        # all-reduce runs under the rest of this node instead of behind it
        ready = getattr(ctx.module, "_grads_ready", None)
        dl_p, ds_p = eng.decoder_backward(sd, pctx, d_pix.contiguous().float(), grads, True)
        if ready is not None:
            ready("pixel_decoder.", grads)
        fcats = fctx.get("shared_cats")
        if fcats is not None:
            dl_f, ds_f = eng.decoder_backward(eng.rotated_weights(sd, "feature_decoder.", fcats), fctx, d_feat.contiguous().float(), grads, True)
            eng.unrotate_grads(sd, "feature_decoder.", fcats, grads)
        else:
            dl_f, ds_f = eng.decoder_backward(sd, fctx, d_feat.contiguous().float(), grads, True)
        if ready is not None:
            ready("feature_decoder.", grads)
        d_latent = Act(dl_p.buf + dl_f.buf)

        d_skips = list(zip(ds_p, ds_f))          # summed inside the pool backward of each level (cmu_maxpool_bwd2): no add passes
        eng.encoder_backward(sd, ectx, d_latent, d_skips, grads,
                             after_bottleneck=(lambda: ready("backbone.double_conv.", grads)) if ready is not None else None)
        ctx.saved = None
        return (None, None, None, None, None, None, None, *[grads.get(n) for n in ctx.names])


class CM_UNet(_EngineOwner, nn.Module):
    """Contrastive + masked-reconstruction UNet (cmunet.py:7-135).

    ``backbone`` = dict(online=..., target=...), ``neck`` = dict(pixel=..., feature=..., projector=...),
    ``head`` = dict(type='CMUNetPretrainHead', ...) exactly as in ``cmunet_config.py:5-42``.
    forward(img, mode='loss', img_t=...) -> {'loss_ct', 'loss_rc'}.
    """

    def __init__(self, backbone, neck, head, base_momentum=0.996, init_cfg=None, target_cls=True, dtype="f16",
                 ref_compat=True, **kwargs):
        super().__init__()
        assert neck is not None and head is not None
        self.dtype = dtype
        self.ref_compat = ref_compat

        def with_dt(cfg):
            cfg = dict(cfg)
            if cfg.get("type") in ("UNet_encoder", "MUNetPretrainDecoder"):
                cfg.setdefault("dtype", dtype)
            return cfg
        self.backbone = _build(with_dt(backbone['online']))
        self.target_backbone = _build(with_dt(backbone['target']))
        self.pixel_decoder = _build(with_dt(neck['pixel']))
        self.feature_decoder = _build(with_dt(neck['feature']))
        self.projector = _build(neck['projector'])
        self.target_projector = _build(neck['projector'])
        # necks under the AMP configuration (cmunet_config.py:76-78): nn.Linear multiplies 16-bit operands into fp32; with fp32
        # storage (parity tests) the products are exact fp32
        self._neck_dtype = None if ops.dt_code(dtype) == ops.F32 else dtype
        self.target_cls = target_cls
        self.head = _build(head)
        for m in (self.projector, self.target_projector, getattr(self.head, "predictor", None)):
            if isinstance(m, NonLinearNeck):
                m.compute_dtype = self._neck_dtype
        self.base_momentum = base_momentum
        self.momentum = base_momentum
        for p in self.target_backbone.parameters():
            p.requires_grad = False
        for p in self.target_projector.parameters():
            p.requires_grad = False
        self._reduce_gen = None

    def init_weights(self):
        """cmunet.py:61-76: initialise, then copy online -> target for backbone and projector."""
        self.backbone.init_weights()
        self.pixel_decoder.init_weights()
        self.feature_decoder.init_weights()
        with torch.no_grad():
            for pb, pm in zip(self.backbone.parameters(), self.target_backbone.parameters()):
                pm.copy_(pb)
            for pb, pm in zip(self.projector.parameters(), self.target_projector.parameters()):
                pm.copy_(pb)

    @torch.no_grad()
    def momentum_update(self):
        """cmunet.py:78-92: p_t = p_t*m + p_o*(1-m) over backbone and projector parameters (EMA kernel)."""
        for src, dst in ((self.backbone, self.target_backbone), (self.projector, self.target_projector)):
            for pb, pm in zip(src.parameters(), dst.parameters()):
                ops.ema_update(pm.data.view(-1), pb.data.view(-1), self.momentum)

    def extract_feat(self, img):
        return self.backbone(img)

    def _reduce_weights(self, C, device, reduce_w=None, reduce_b=None):
        """cmunet.py:128-129: nn.Conv2d(C,256,1) constructed (default init) at every call, never trained (A-2); the weights
        are injectable for tests."""
        if reduce_w is None:
            # 256 for the reference geometry (1024 channels at /16): whatever makes Cr*(H/2^d)*(W/2^d) == H*W
            conv = nn.Conv2d(C, self.reduced_channels(), kernel_size=1).to(device)
            reduce_w, reduce_b = conv.weight, conv.bias
        return reduce_w, reduce_b

    def reduced_channels(self):
        n_down = sum(1 for n, _ in self.backbone.named_children() if n.startswith("down_conv"))
        return 4 ** n_down

    def forward_train(self, img, img_t=None, mask=None, reduce_w=None, reduce_b=None, **kwargs):
        _require_cuda(img, "CM_UNet")
        B, H, W = img.shape
        if mask is None:
            mask = self.backbone.make_mask(img)
        names, params = _param_args(self)
        C_lat = self.target_backbone.double_conv.double_conv[3].weight.shape[0]
        reduce_w, reduce_b = self._reduce_weights(C_lat, img.device, reduce_w, reduce_b)
        pred_pixel, pred_feature, lt = _CMUNetFn.apply(self, img, img_t, mask, reduce_w, reduce_b, names, *params)
        proj_s = self.projector(torch.mean(pred_feature, dim=1, keepdim=True))
        with torch.no_grad():
            lt = lt.reshape(B, -1).reshape(B, 1, H, W)          # 256*(H/16)*(W/16) == H*W (cmunet.py:130)
            proj_t = self.target_projector(torch.mean(lt, dim=1, keepdim=True))
        return self.head(img, pred_pixel, mask, proj_s, proj_t)

    def forward(self, img, mode='loss', **kwargs):
        """base.py:75-113: mode 'tensor' -> features, 'loss' -> dict of losses."""
        if mode == 'tensor':
            return self.extract_feat(img)
        if mode == 'loss':
            return self.forward_train(img, **kwargs)
        raise RuntimeError(f'Invalid mode "{mode}".')


def momentum_schedule(cur_iter, max_iter, base_momentum=0.996, end_momentum=0.996):
    """MomentumUpdateHook.before_train_iter (momentum_update_hook.py:29-40)."""
    return end_momentum - (end_momentum - base_momentum) * (math.cos(math.pi * cur_iter / float(max_iter)) + 1) / 2


_TYPES.update({"UNet_encoder": UNet_encoder, "MUNetPretrainDecoder": MUNetPretrainDecoder, "NonLinearNeck": NonLinearNeck,
               "CMUNetPretrainHead": CMUNetPretrainHead, "CM_UNet": CM_UNet})


def cmunet_config(img_size=224, dtype="f16", temperature=0.07, ct_weight=1.0, rc_weight=1.0, mask_ratio=0.65,
                  base_ch=64, depth=5):
    """The ``model`` dict of configs/cmunet_config.py:5-42 with the projector sized for ``img_size``
    (in_channels = H*W: 50176 at 224, 262144 at 512).  ``dtype`` defaults to 'f16': the reference trains this
    model under AmpOptimWrapper(loss_scale='dynamic') (cmunet_config.py:76-78), i.e. fp16 operands with fp32 accumulation --
    pair it with ``JointPretrainer(amp=True)`` (the device-side dynamic loss scaler); 'f32' / 'bf16' are opt-in."""
    neck = dict(type='NonLinearNeck', hid_channels=1536, out_channels=256, num_layers=2, with_bias=True, with_last_bn=False,
                with_avg_pool=False)
    enc = dict(type='UNet_encoder', patch_size=16, base_ch=base_ch, depth=depth)
    dec = dict(type='MUNetPretrainDecoder', base_ch=base_ch, depth=depth)
    return dict(
        type='CM_UNet', dtype=dtype,
        backbone=dict(online=dict(enc, mask_ratio=mask_ratio), target=dict(enc, mask_ratio=0.0)),
        neck=dict(pixel=dict(dec), feature=dict(dec), projector=dict(neck, in_channels=img_size * img_size)),
        head=dict(type='CMUNetPretrainHead', predictor=dict(neck, in_channels=256), temperature=temperature,
                  ct_weight=ct_weight, rc_weight=rc_weight))


def build_model(cfg):
    return _build(cfg)
