"""Fused CM-UNet masked-reconstruction pretraining step (BASELINE config 2).

What the reference spreads over mmengine's Runner (Pretraining/CM-UNet/training/train.py:51-94 ->
CM_UNet.forward_train cmunet.py:108-135 -> CMUNetPretrainHead.forward cmunet_head.py:47-70 ->
AmpOptimWrapper.update_params -> AdamW.step, cmunet_config.py:76-91) is ONE kernel schedule here:

    patch mask (UNet_encoder.py:106-158)  fused into the first conv's load (x * (1 - mask[0]))
    online encoder + pixel decoder        engine.unet_forward   (raw conv outputs, BN+ReLU applied on load)
    masked MSE on logits[:,1]             cmu_masked_mse_fwd_bwd (per-row normalised target, A-3)
    backward                              engine.unet_backward  (gradients written into the flat arena)
    gradient exchange                     one RCCL all-reduce over the arena (data parallel, C1)
    AdamW                                 cmu_adam_step over the arena (bias / norm parameters: no decay)

``ct_weight = 0`` drops the contrastive branch (target encoder, feature decoder, projector, predictor),
exactly the "masked-recon only" configuration SURVEY 8(d)-(2) names; the joint step lives in cmunet.py.
"""
import math

import numpy as np
import torch
import torch.distributed as dist

from . import _lib, ops
from .optim import FlatParams, FusedAdam, no_decay_bias_norm


def create_random_patch_mask(batch_size, img_size, patch_size=16, mask_ratio=0.65, rng=None):
    """Host mask generator with the reference's exact RNG consumption (UNet_encoder.py:106-139): per sample,
    shuffle the patch indices and mask patches until floor(mask_ratio*H*W) pixels are covered."""
    rng = np.random if rng is None else rng
    per_side = img_size // patch_size
    n_mask = int(mask_ratio * img_size * img_size) // (patch_size * patch_size)
    mask = np.zeros((batch_size, per_side * per_side), dtype=np.uint8)
    for i in range(batch_size):
        idx = np.arange(per_side * per_side)
        rng.shuffle(idx)
        mask[i, idx[:n_mask]] = 1
    mask = mask.reshape(batch_size, per_side, per_side)
    return np.repeat(np.repeat(mask, patch_size, axis=1), patch_size, axis=2)


def random_patch_mask_device(batch_size, H, W, patch_size=16, mask_ratio=0.65, generator=None, device="cuda", seed=None, offset=0):
    """Same distribution, generated on the device: a random permutation of the patches per
    sample, the first floor(ratio*H*W/patch^2) masked.  Returns uint8 (B,H,W), 1 = masked.
    ``seed`` given: one launch of the counter-based kernel (``ops.random_patch_mask``; the caller advances ``offset`` by
    B * patches per call); else torch's generator (rand + double argsort)."""
    if seed is not None:
        from . import ops
        return ops.random_patch_mask(batch_size, H, W, patch_size, mask_ratio, seed, offset, device)
    ph, pw = H // patch_size, W // patch_size
    n_mask = int(mask_ratio * H * W) // (patch_size * patch_size)
    r = torch.rand(batch_size, ph * pw, generator=generator, device=device)
    rank = r.argsort(dim=1).argsort(dim=1)
    m = (rank < n_mask).to(torch.uint8).view(batch_size, ph, pw)
    return m.repeat_interleave(patch_size, 1).repeat_interleave(patch_size, 2).contiguous()


class MaskedReconPretrainer:
    """One object = model + flat arenas + fused AdamW + (optional) data-parallel group."""

    def __init__(self, model, lr=1.5e-4, betas=(0.9, 0.95), weight_decay=0.05, eps=1e-8, rc_weight=1.0,
                 ref_compat=True, pred_channel=1, process_group=None, loss_scale=1.0):
        assert next(model.parameters()).is_cuda, "move the model to the GPU first"
        self.model = model.train()
        self.device = next(model.parameters()).device
        self.flat = FlatParams(model)
        self.opt = FusedAdam(self.flat, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, decoupled=True,
                             decay_filter=no_decay_bias_norm)
        self.engine = model._engine(self.device)
        self.sd = dict(model.named_parameters())
        self.sd.update(dict(model.named_buffers()))
        self.rc_weight, self.ref_compat, self.pred_channel = rc_weight, ref_compat, pred_channel
        self.group = process_group
        self.loss_scale = loss_scale
        self.loss = torch.zeros(1, dtype=torch.float32, device=self.device)
        self._dlogits = None
        self._ws = None
        # gradient exchange in two buckets: the decoder's parameters are the tail of the arena and their gradients are
        # complete half-way through the backward pass
        self._dec_off = self.flat.tail_offset(("up_conv", "conv_last"))
        self._pending = None

    def broadcast_parameters(self, src=0):
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            dist.broadcast(self.flat.arena, src=src, group=self.group)
            for n, b in self.model.named_buffers():
                if b.is_floating_point():
                    dist.broadcast(b, src=src, group=self.group)

    def forward_backward(self, img, mask):
        """img (B,H,W) fp32 cuda, mask (B,H,W) uint8 cuda (1 = masked).  Leaves gradients in the arena and
        returns the loss tensor (1,) on the device (no host sync)."""
        eng = self.engine
        B, H, W = img.shape
        eng.prepack(self.sd)      # all weight packs of the step in one launch
        logits, ctx = eng.unet_forward(self.sd, img, True, mask, mask_per_sample=not self.ref_compat)
        if self._dlogits is None or self._dlogits.shape != logits.shape:
            self._dlogits = torch.empty_like(logits)
            self._ws = torch.empty(_lib.lib().cmu_masked_mse_ws_bytes(B, H), dtype=torch.uint8, device=self.device)
        ops.masked_mse_fwd_bwd(logits, self.pred_channel, img, mask, self.loss, self._dlogits,
                               self.rc_weight * self.loss_scale, self._ws)
        eng.grad_target, eng.grad_prefix = self.flat.grad_views, ""
        self._pending = None

        def decoder_done():
            if self._dec_off is not None:
                self._pending = self.flat.all_reduce_range_async(self._dec_off, self.flat.grad.numel(), self.group)

        try:
            eng.unet_backward(self.sd, ctx, self._dlogits, after_decoder=decoder_done)
        finally:
            eng.grad_target = None
        return self.loss

    def exchange_gradients(self):
        """Finish the data-parallel gradient SUM (the decoder bucket may already be in flight) and return 1/world."""
        world = dist.get_world_size(self.group) if (dist.is_available() and dist.is_initialized()) else 1
        if world <= 1:
            return 1.0
        if self._pending is not None:
            rest = self.flat.all_reduce_range_async(0, self._dec_off, self.group)
            self._pending.wait()
            if rest is not None:
                rest.wait()
            self._pending = None
        else:
            self.flat.all_reduce_mean(self.group)
        return 1.0 / world

    def step(self, img, mask):
        loss = self.forward_backward(img, mask)
        scale = self.exchange_gradients() / self.loss_scale
        self.opt.step(grad_scale=scale)
        return loss


def cosine_warmup_lr(base_lr, it, warmup_iters, total_iters, start_factor=1e-4):
    """cmunet_config.py:94-109: LinearLR(start_factor 1e-4) for the warm-up, then CosineAnnealingLR to 0."""
    if it < warmup_iters:
        return base_lr * (start_factor + (1 - start_factor) * it / max(1, warmup_iters))
    t = (it - warmup_iters) / max(1, total_iters - warmup_iters)
    return 0.5 * base_lr * (1 + math.cos(math.pi * min(1.0, t)))
