"""Oracle: optimiser steps restated on CPU tensors.  TEST INFRASTRUCTURE ONLY.

  sgd_step    torch.optim.SGD as MoCo configures it (Pretraining/MoCo/moco2_module.py:339-344): L2 weight decay added to
              the gradient, momentum buffer = gradient on the first step, dampening, optional Nesterov.
  lamb_step   Pretraining/Spark/utils/lamb.py:67-159 (timm's LAMB): global gradient-norm clip, Adam moments with
              grad_averaging, bias correction, weight decay added to the update, per-tensor trust ratio for decayed
              tensors (or all with always_adapt), optional trust clipping.
Pinned by oracle/gen_golden.py, which runs the reference's LAMB class and torch.optim.SGD on the same tensors
(tests/golden/optim_traces.npz).
"""
import math

import torch


def sgd_step(params, grads, bufs, lr, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False, step=1, wd_flags=None):
    """In place on ``params`` / ``bufs`` (lists of tensors).  wd_flags[i] False: no weight decay on tensor i."""
    for i, (p, g) in enumerate(zip(params, grads)):
        d = g.clone()
        if weight_decay != 0 and (wd_flags is None or wd_flags[i]):
            d = d + weight_decay * p
        if momentum != 0:
            if step == 1:
                bufs[i].copy_(d)
            else:
                bufs[i].mul_(momentum).add_(d, alpha=1 - dampening)
            d = d + momentum * bufs[i] if nesterov else bufs[i]
        p.add_(d, alpha=-lr)


def lamb_step(params, grads, ms, vs, lr, wds, betas=(0.9, 0.999), eps=1e-6, bias_correction=True, grad_averaging=True,
              max_grad_norm=2.0, trust_clip=False, always_adapt=False, step=1):
    """In place on params / ms / vs.  wds[i]: weight decay of tensor i (0 = excluded)."""
    gn = math.sqrt(sum(float((g.double() ** 2).sum()) for g in grads))
    clip = 1.0 / (gn / max_grad_norm) if (max_grad_norm > 0 and gn > max_grad_norm) else 1.0
    b1, b2 = betas
    b3 = 1 - b1 if grad_averaging else 1.0
    bc1 = 1 - b1 ** step if bias_correction else 1.0
    bc2 = 1 - b2 ** step if bias_correction else 1.0
    for p, g, m, v, wd in zip(params, grads, ms, vs, wds):
        g = g * clip
        m.mul_(b1).add_(g, alpha=b3)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        upd = (m / bc1) / (v.sqrt() / math.sqrt(bc2) + eps)
        if wd != 0:
            upd = upd + wd * p
        if wd != 0 or always_adapt:
            wn, un = float(p.norm(2.0)), float(upd.norm(2.0))
            r = wn / un if (wn > 0 and un > 0) else 1.0
            if trust_clip:
                r = min(r, 1.0)
            upd = upd * r
        p.add_(upd, alpha=-lr)
    return gn
