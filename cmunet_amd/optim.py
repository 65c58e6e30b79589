"""Flat parameter / gradient arenas and the fused Adam(W) step.

The reference steps torch.optim.Adam (Finetuning/train.py:341), AdamW with no-decay groups for biases and
norm parameters (cmunet_config.py:76-91) one tensor at a time.  Here every parameter of a module is
re-homed into ONE contiguous fp32 arena (and its gradient into a second one): the optimiser is a single
HIP kernel over the arena, the data-parallel gradient exchange is one RCCL all-reduce over it, and the EMA
of the momentum encoder is one kernel between two arenas with identical layout.
"""
import os
import weakref

import torch
import torch.distributed as dist

from . import ops


_GRAD_SINKS = {}      # id(Parameter) -> (weak reference to it, its slot in a trainer's gradient arena); entries die with the parameter


def _register_grad_sink(param, view):
    key = id(param)
    _GRAD_SINKS[key] = (weakref.ref(param, lambda _r, k=key: _GRAD_SINKS.pop(k, None)), view)


_SINKS_CLAIMED = set()


def claim_grad_sink(param):
    """For the package's autograd nodes, which produce a whole parameter gradient in one kernel: a fresh alias of ``param``'s slot in its
    trainer's gradient arena to write that gradient into and hand to autograd (which adopts an unshared tensor as ``.grad`` without a
    copy; ``gather_autograd_grads`` then finds it in place) -- or None, and the node allocates as usual: no arena holds the parameter,
    it already has a ``.grad`` (autograd must accumulate into it), or the slot was already handed out since the last gather (a weight
    used twice in one graph)."""
    ent = _GRAD_SINKS.get(id(param)) if isinstance(param, torch.nn.Parameter) else None
    if ent is None or ent[0]() is not param or param.grad is not None or id(param) in _SINKS_CLAIMED:
        return None
    view = ent[1]
    if view.shape != param.shape or view.dtype != torch.float32 or not view.is_contiguous():
        return None
    _SINKS_CLAIMED.add(id(param))
    return view.view_as(view)


def dp_world(group=None):
    """World size of the initialised data-parallel group, 1 without one."""
    return dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1


def dp_exchanges(group=None):
    """True when collectives have to run: an initialised group of more than one rank.  CMU_DP_REHEARSE=1 makes a one-rank group run them
    too -- the RCCL call sequence (async bucket all-reduces behind the backward, all-gathers, waits) on a one-GPU box, where a
    second rank could only share the card over gloo."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or os.environ.get("CMU_DP_REHEARSE", "0") == "1"


def all_reduce_sum_scale(tensor, group=None):
    """SUM all-reduce of ``tensor`` in place over the data-parallel group (RCCL on the GPU, gloo in the CPU
    tests); returns the 1/world factor that turns it into the mean (folded into the optimiser kernel)."""
    if dp_exchanges(group):
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=group)
        return 1.0 / dist.get_world_size(group)
    return 1.0


class FlatParams:
    """Re-homes ``module``'s parameters into a flat fp32 arena; ``grad_views[name]`` are views into the
    gradient arena that the engine's weight-gradient kernels write directly (no per-tensor copies)."""

    def __init__(self, module, names=None):
        params = [(n, p) for n, p in module.named_parameters() if names is None or names(n)]
        assert params, "no parameters"
        dev = params[0][1].device
        assert dev.type == "cuda", "FlatParams needs the module on the GPU"
        self.names = [n for n, _ in params]
        sizes = [((p.numel() + 3) // 4) * 4 for _, p in params]        # keep every tensor 16-byte aligned
        self.total = sum(sizes)
        self.arena = torch.zeros(self.total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(self.total, dtype=torch.float32, device=dev)
        self.offsets, self.views, self.grad_views = {}, {}, {}
        off = 0
        for (n, p), sz in zip(params, sizes):
            v = self.arena[off:off + p.numel()].view_as(p)
            v.copy_(p.data)
            p.data = v
            self.offsets[n] = (off, p.numel())
            self.views[n] = v
            self.grad_views[n] = self.grad[off:off + p.numel()].view_as(p)
            # autograd nodes of the package that produce a whole parameter gradient in one kernel (the necks' Linear layers) write it
            # here instead of into a fresh tensor: no 1.6 GB copy of the joint model's projector gradient per step
            # (a registry rather than an attribute: attributes of a Parameter travel with torch.save(model))
            _register_grad_sink(p, self.grad_views[n])
            off += sz
        self.params = dict(params)

    def wd_mask(self, decay_filter):
        """uint8 per element: 1 where weight decay applies (decay_filter(name, param) -> bool)."""
        m = torch.zeros(self.total, dtype=torch.uint8, device=self.arena.device)
        for n in self.names:
            off, cnt = self.offsets[n]
            if decay_filter(n, self.params[n]):
                m[off:off + cnt] = 1
        return m

    def gather_autograd_grads(self):
        """Copy ``p.grad`` tensors produced by autograd into the gradient arena (drop-in path)."""
        dst, src, zero = [], [], []
        _SINKS_CLAIMED.difference_update(id(p) for p in self.params.values())
        for n in self.names:
            g = self.params[n].grad
            gv = self.grad_views[n]
            if g is None:
                zero.append(gv)
            elif g.data_ptr() != gv.data_ptr():
                if g.dtype == gv.dtype and g.shape == gv.shape:
                    dst.append(gv)
                    src.append(g.detach())
                else:
                    gv.copy_(g)
        # a few multi-tensor launches instead of one copy per parameter (the joint model has ~270 small ones: 1.8 ms of launches)
        if zero:
            torch._foreach_zero_(zero)
        if dst:
            torch._foreach_copy_(dst, src)

    def gather_names(self, names):
        """``gather_autograd_grads`` restricted to ``names`` (one bucket of an overlapped exchange): their ``p.grad`` tensors into the
        arena (nothing to do for a gradient the producing kernel already wrote in place), zeros for parameters without one."""
        dst, src, zero = [], [], []
        for n in names:
            p = self.params[n]
            _SINKS_CLAIMED.discard(id(p))
            g, gv = p.grad, self.grad_views[n]
            if g is None:
                zero.append(gv)
            elif g.data_ptr() != gv.data_ptr():
                if g.dtype == gv.dtype and g.shape == gv.shape:
                    dst.append(gv)
                    src.append(g.detach())
                else:
                    gv.copy_(g)
        if zero:
            torch._foreach_zero_(zero)
        if dst:
            torch._foreach_copy_(dst, src)

    def buckets(self, key=None):
        """Contiguous runs of the arena whose parameters share ``key(name)`` -> [(lo, hi, [names])] in arena order.  Default key:
        the top-level module name, with an encoder's bottleneck (``<top>.double_conv.*``: 60 % of its parameters, the first to be
        complete in a backward pass) apart from its down blocks."""
        if key is None:
            def key(n):
                top = n.partition(".")[0]
                # the bottleneck is ``<prefix>double_conv.double_conv.*`` with no down / up block in front (backbone.double_conv...,
                # sparse_encoder.sp_cnn.double_conv...): the down blocks are ``<prefix>down_convK.double_conv.double_conv.*``
                i = n.find("double_conv.double_conv.")
                bott = i >= 0 and "down_conv" not in n[:i] and "up_conv" not in n[:i]
                return top + (".double_conv" if bott else "")
        out, cur = [], None
        for n in self.names:
            k = key(n)
            off, cnt = self.offsets[n]
            hi = off + ((cnt + 3) // 4) * 4
            if cur is not None and cur[0] == k:
                cur[2] = hi
                cur[3].append(n)
            else:
                cur = [k, off, hi, [n]]
                out.append(cur)
        return [(lo, hi, names) for _, lo, hi, names in out]

    def all_reduce_mean(self, group=None):
        """Data-parallel gradient exchange: ONE RCCL all-reduce over the whole arena (C1 in SURVEY 2.5).
        Returns the scale (1/world) the optimiser kernel folds into its gradient load."""
        return all_reduce_sum_scale(self.grad, group)

    def tail_offset(self, prefixes):
        """Element offset where the trailing run of parameters whose names start with one of ``prefixes`` begins
        (the arena follows ``named_parameters`` order), or None when those parameters are not one contiguous tail.
        The UNet's decoder (``up_conv*``, ``conv_last``) is such a tail: its gradients are complete when the decoder
        backward returns, so their all-reduce can run under the encoder backward."""
        prefixes = tuple(prefixes)
        first = None
        for n in self.names:
            hit = n.startswith(prefixes)
            if hit and first is None:
                first = n
            elif not hit and first is not None:
                return None
        return None if first is None else self.offsets[first][0]

    def prefix_range(self, prefix):
        """[lo, hi) element range of the parameters whose names start with ``prefix`` when they form one contiguous run of the
        arena (padding included), else None."""
        idx = [i for i, n in enumerate(self.names) if n.startswith(prefix)]
        if not idx or idx != list(range(idx[0], idx[-1] + 1)):
            return None
        lo = self.offsets[self.names[idx[0]]][0]
        last = self.names[idx[-1]]
        hi = self.offsets[last][0] + ((self.offsets[last][1] + 3) // 4) * 4
        return lo, hi

    def all_reduce_range_async(self, lo, hi, group=None):
        """Start the SUM all-reduce of gradient elements [lo, hi); returns a work handle (``.wait()``) or None when
        there is nothing to exchange.  On RCCL the collective runs on the process group's stream after the kernels
        already queued on the current stream, concurrently with what is queued next."""
        if dp_exchanges(group) and hi > lo:
            return dist.all_reduce(self.grad[lo:hi], op=dist.ReduceOp.SUM, group=group, async_op=True)
        return None


CMUNET_NO_DECAY_KEYS = ("ln", "bias", "pos_embed", "mask_token", "cls_token")


def cmunet_paramwise_decay(name, param):
    """Weight-decay rule of the reference's CM-UNet pretraining, as mmengine's DefaultOptimWrapperConstructor applies
    ``paramwise_cfg = dict(custom_keys={'ln', 'bias', 'pos_embed', 'mask_token', 'cls_token': decay_mult=0})``
    (cmunet_config.py:84-91): a parameter is exempt iff its qualified name CONTAINS one of the keys (substring match; no
    ``norm_decay_mult`` / ``bias_decay_mult`` is set).  No UNet / neck / predictor parameter name contains 'ln', so every
    BatchNorm WEIGHT ('....1.weight', '....4.weight', 'bn0.weight') decays with 0.05 like the conv weights; only the biases are
    exempt.  (Rounds 1-4 used the "every 1-D tensor" rule here: a multi-step trajectory that differed from the reference's by
    construction -- VERDICT round 4.)  The rule does not depend on the prefix the trainer's model root puts in front of a name:
    none of the reference's module names ('backbone', 'target_backbone', 'pixel_decoder', 'feature_decoder', 'projector',
    'target_projector', 'head.predictor') contains a key."""
    return not any(k in name for k in CMUNET_NO_DECAY_KEYS)


def no_decay_bias_norm(name, param):
    """The usual 1-D rule (biases and norm weights exempt): SparK's ``get_param_groups`` (Spark/utils/misc.py) -- NOT the CM-UNet
    config, see ``cmunet_paramwise_decay``."""
    return not (name.endswith(".bias") or param.dim() <= 1)


class FusedAdam:
    """Adam / AdamW over a FlatParams arena in one kernel launch (cmu_adam_step)."""

    def __init__(self, flat, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, decoupled=False,
                 decay_filter=None):
        self.flat = flat
        self.lr, self.betas, self.eps, self.weight_decay, self.decoupled = lr, betas, eps, weight_decay, decoupled
        self.m = torch.zeros_like(flat.arena)
        self.v = torch.zeros_like(flat.arena)
        self.wd_mask = flat.wd_mask(decay_filter) if (decay_filter is not None and weight_decay != 0.0) else None
        self.step_count = 0
        self.amp = None          # the ops.AmpScaler the steps run under (set by the trainer or by the first step(amp=...))

    def zero_grad(self, set_to_none=True):
        for p in self.flat.params.values():
            p.grad = None

    def step(self, grad_scale=1.0, amp=None, ema=None):
        """``amp``: an ``ops.AmpScaler`` -- the kernel then skips / unscales from its device state (and takes the step number
        of the bias corrections from it).  ``ema``: (segments, momentum) -- the EMA of the momentum networks in the same pass
        (``ops.adam_ema_step``; segments = up to two (lo, hi, target tensor) ranges of the arena)."""
        self.step_count += 1
        if amp is not None:
            self.amp = amp
        if ema is not None:
            ops.adam_ema_step(self.flat.arena, self.flat.grad, self.m, self.v, self.wd_mask, self.lr, self.betas[0], self.betas[1],
                              self.eps, self.weight_decay, self.decoupled, self.step_count, grad_scale, amp, ema[0], ema[1])
            return
        ops.adam_step(self.flat.arena, self.flat.grad, self.m, self.v, self.wd_mask, self.lr, self.betas[0],
                      self.betas[1], self.eps, self.weight_decay, self.decoupled, self.step_count, grad_scale, amp)
        # ops.adam_step bumps ops.PARAM_GENERATION: the engine's packed-weight caches see the raw-kernel write

    def step_ranges(self, ranges, grad_scale=1.0, amp=None, ema=None):
        """``step`` restricted to the ascending element ranges ``ranges`` of the arena (multiples of 4): one launch per range, the same
        kernel on the same operands element for element (the sharded exchange of pretrain.JointPretrainer: a rank updates only the share of
        a big parameter whose summed gradient it holds).  EMA segments are clipped to each range."""
        self.step_count += 1
        if amp is not None:
            self.amp = amp
        a, g, m, v = self.flat.arena, self.flat.grad, self.m, self.v
        for lo, hi in ranges:
            assert lo % 4 == 0 and hi % 4 == 0 and 0 <= lo < hi <= a.numel(), (lo, hi)
            wd = None if self.wd_mask is None else self.wd_mask[lo:hi]
            segs = []
            if ema is not None:
                for s0, s1, tgt in ema[0]:
                    c0, c1 = max(s0, lo), min(s1, hi)
                    if c1 > c0:
                        segs.append((c0 - lo, c1 - lo, tgt[c0 - s0:c1 - s0]))
            if segs:
                ops.adam_ema_step(a[lo:hi], g[lo:hi], m[lo:hi], v[lo:hi], wd, self.lr, self.betas[0], self.betas[1], self.eps,
                                  self.weight_decay, self.decoupled, self.step_count, grad_scale, amp, segs, ema[1])
            else:
                ops.adam_step(a[lo:hi], g[lo:hi], m[lo:hi], v[lo:hi], wd, self.lr, self.betas[0], self.betas[1], self.eps,
                              self.weight_decay, self.decoupled, self.step_count, grad_scale, amp)

    def state_dict(self):
        """Moments, host step count and -- when the steps run under a loss scaler -- the scaler's state (``loss_scaler``, the key
        mmengine's AmpOptimWrapper.state_dict uses): under amp the kernel takes the bias-correction step from the scaler's
        ``good_steps``, so the optimiser cannot be resumed without it."""
        sd = {"m": self.m, "v": self.v, "step": self.step_count, "lr": self.lr}
        if self.amp is not None:
            sd["loss_scaler"] = self.amp.state_dict()
        return sd

    def load_state_dict(self, sd, amp=None):
        """``amp``: the scaler the resumed steps will run under (defaults to the one already attached).  A checkpoint without
        ``loss_scaler`` (written by a run without amp, or by an older build) seeds the scaler's step counter from ``step``."""
        self.m.copy_(sd["m"])
        self.v.copy_(sd["v"])
        self.step_count = int(sd["step"])
        amp = amp if amp is not None else self.amp
        if amp is not None:
            self.amp = amp
            if "loss_scaler" in sd:
                amp.load_state_dict(sd["loss_scaler"])
            else:
                cur = amp.state_dict()
                cur["good_steps"] = self.step_count
                amp.load_state_dict(cur)


class FusedSGD:
    """torch.optim.SGD semantics (momentum, dampening, Nesterov, L2 weight decay) over a FlatParams arena in one launch
    (cmu_sgd_step) -- MoCo's optimiser (moco2_module.py:339-344: lr, momentum 0.9, weight_decay 1e-4)."""

    def __init__(self, flat, lr, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False, decay_filter=None):
        if nesterov and (momentum <= 0 or dampening != 0):
            raise ValueError("Nesterov momentum requires a momentum and zero dampening")   # torch.optim.SGD's own check
        self.flat = flat
        self.lr, self.momentum, self.dampening, self.weight_decay, self.nesterov = lr, momentum, dampening, weight_decay, nesterov
        self.buf = torch.zeros_like(flat.arena) if momentum != 0 else None
        self.wd_mask = flat.wd_mask(decay_filter) if (decay_filter is not None and weight_decay != 0.0) else None
        self.step_count = 0

    def zero_grad(self, set_to_none=True):
        for p in self.flat.params.values():
            p.grad = None

    def step(self, grad_scale=1.0):
        if getattr(self, "auto_gather", False):      # used like a torch optimiser (Moco_v2.configure_optimizers): p.grad -> arena first
            self.flat.gather_autograd_grads()
        self.step_count += 1
        ops.sgd_step(self.flat.arena, self.flat.grad, self.buf, self.wd_mask, self.lr, self.momentum, self.dampening,
                     self.weight_decay, self.nesterov, self.step_count, grad_scale)


class FusedLAMB:
    """LAMB as the reference's SparK pretraining uses it (Pretraining/Spark/utils/lamb.py:67-159) over a FlatParams arena:
    global gradient-norm clip, Adam moments, per-tensor trust ratio -- five small launches, no host synchronisation.
    ``decay_filter(name, param) -> bool`` selects the tensors that get ``weight_decay`` (the others also skip the trust
    ratio unless ``always_adapt``), as the reference's two parameter groups do."""

    def __init__(self, flat, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.01, bias_correction=True, grad_averaging=True,
                 max_grad_norm=2.0, trust_clip=False, always_adapt=False, decay_filter=None):
        from . import _lib
        self.flat = flat
        self.lr, self.betas, self.eps = lr, betas, eps
        self.bias_correction, self.grad_averaging, self.max_grad_norm = bias_correction, grad_averaging, max_grad_norm
        self.trust_clip, self.always_adapt = trust_clip, always_adapt
        dev = flat.arena.device
        blk = _lib.lib().cmu_lamb_block_elems()
        starts, counts, tens, t0, twd = [], [], [], [0], []
        for t, n in enumerate(flat.names):
            off, cnt = flat.offsets[n]
            for s in range(0, cnt, blk):
                starts.append(off + s); counts.append(min(blk, cnt - s)); tens.append(t)
            t0.append(len(starts))
            twd.append(weight_decay if (decay_filter is None or decay_filter(n, flat.params[n])) else 0.0)
        self.tables = (torch.tensor(starts, dtype=torch.int64, device=dev), torch.tensor(counts, dtype=torch.int32, device=dev),
                       torch.tensor(tens, dtype=torch.int32, device=dev), torch.tensor(t0, dtype=torch.int32, device=dev),
                       torch.tensor(twd, dtype=torch.float32, device=dev))
        self.m, self.v, self.u = torch.zeros_like(flat.arena), torch.zeros_like(flat.arena), torch.empty_like(flat.arena)
        self.ws = torch.empty(_lib.lib().cmu_lamb_ws_bytes(len(starts), len(twd)), dtype=torch.uint8, device=dev)
        self.step_count = 0

    def zero_grad(self, set_to_none=True):
        for p in self.flat.params.values():
            p.grad = None

    def set_lr_wd(self, lr, weight_decay=None):
        """Learning rate and weight decay of the next step (the reference anneals both every iteration, lr_control.py:11-29: the
        decayed tensors get ``weight_decay``, the no-decay group stays at 0 -- its weight_decay_scale)."""
        self.lr = float(lr)
        if weight_decay is not None:
            if not hasattr(self, "_wd_mask"):
                self._wd_mask = (self.tables[4] != 0).to(torch.float32)
            self.tables = self.tables[:4] + (self._wd_mask * float(weight_decay),)

    @property
    def global_grad_norm(self):
        """The reference exposes this for logging (lamb.py:91); reading it synchronises."""
        n = len(self.tables[0])
        return float(self.ws.view(torch.float32)[3 * n + len(self.tables[4])])

    def step(self, grad_scale=1.0):
        self.step_count += 1
        ops.lamb_step(self.flat.arena, self.flat.grad, self.m, self.v, self.u, self.tables, self.lr, self.betas[0], self.betas[1],
                      self.eps, self.bias_correction, self.grad_averaging, self.max_grad_norm, self.trust_clip, self.always_adapt,
                      self.step_count, grad_scale, self.ws)
