"""Drop-in for the MoCo-v2 baseline on the UNet encoder (reference:
Pretraining/MoCo/pl_bolts/models/self_supervised/moco/moco2_module.py:80-309, moco_data_module.py:47-66).

PyTorch-Lightning's Trainer is not rebuilt; ``Moco_v2`` keeps the reference's constructor
arguments, buffers (``queue (emb_dim, K)``, ``queue_ptr (1,) int64``) and method names:

  forward(img_q, img_k, queue) -> (logits, labels, k, q)            moco2_module.py:224-270 (API-faithful; the
                                                                     (N, 1+K) logits are a plain library GEMM)
  training_step((x_q, x_k))   -> loss                                moco2_module.py:287-309, fused:
      EMA of the key encoder (cmu_ema_update, BEFORE the forward: A-8), both encoders on the HIP engine with
      the global average pool fused on the raw latent (cmu_gap_fwd), batch shuffle of the key images across
      ranks (shuffle-BN, moco2_module.py:177-222) when more than one rank runs, all-gather of the normalised keys,
      InfoNCE against the queue + ring-buffer enqueue in ONE kernel (cmu_moco_infonce_enqueue).
"""
import copy

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, ops
from .cmunet import UNet_encoder as _MaskEncoder, concat_all_gather
from .optim import dp_exchanges
from .model import _EngineOwner, _named_state, _param_args, _require_cuda


class _EncoderGapFn(torch.autograd.Function):
    """UNet down path + bottleneck + mean over (H,W) (moco_data_module.py:59-66) -> (B, C) fp32."""

    @staticmethod
    def forward(ctx, module, x, names, *params):
        eng = module._engine(x.device)
        sd = _named_state(module)
        eng.prepack(sd)
        B = x.shape[0]
        ectx = eng.encoder_forward(sd, x.detach().float().contiguous().view(B, x.shape[-2], x.shape[-1]), module.training, "")
        lat = ectx["latent"]
        out = torch.empty((B, lat.C), dtype=torch.float32, device=x.device)
        ops.gap_fwd(lat, out)
        ctx.module, ctx.ectx, ctx.names, ctx.eng = module, ectx, names, eng
        return out

    @staticmethod
    def backward(ctx, dout):
        eng, ectx = ctx.eng, ctx.ectx
        sd = _named_state(ctx.module)
        lat = ectx["latent"]
        dA = eng._new(lat.B, lat.H, lat.W, lat.C)
        ops.gap_bwd(dout.contiguous().float(), dA)
        grads = {}
        ready = getattr(ctx.module, "_grads_ready", None)      # pretrain.ArenaTrainer.notify_ready: the bottleneck's exchange starts early
        eng.encoder_backward(sd, ectx, dA, None, grads,
                             after_bottleneck=(lambda: ready("double_conv.", grads)) if ready is not None else None)
        ctx.ectx = None
        return (None, None, None, *[grads.get(n) for n in ctx.names])


class UNet_encoder(_MaskEncoder):
    """moco_data_module.py:47-66: the UNet encoder followed by torch.mean(x, dim=[2,3]); input (B,1,H,W)."""

    def __init__(self, out_classes=2, up_sample_mode='conv_transpose', base_ch=64, depth=5, dtype="f32"):
        super().__init__(out_classes, up_sample_mode, patch_size=16, mask_ratio=0.0, base_ch=base_ch, depth=depth, dtype=dtype)

    def forward(self, x):
        _require_cuda(x, "UNet_encoder")
        names, params = _param_args(self)
        return _EncoderGapFn.apply(self, x, names, *params)


class _QueueLogitsFn(torch.autograd.Function):
    """l_neg = q @ queue (moco2_module.py:262, ``einsum("nc,ck->nk")``) for <= 256 query rows on the weight-streaming skinny
    kernels: the queue (D, K) is the matrix whose rows are contiguous in k -- the "input gradient" shape forward, the "forward"
    shape backward (dq = dl @ queue^T); exact fp32 products."""

    @staticmethod
    def forward(ctx, q, queue):
        ctx.save_for_backward(queue)
        return ops.skinny_gemm_dgrad(q.detach().contiguous(), queue.detach())

    @staticmethod
    def backward(ctx, dl):
        (queue,) = ctx.saved_tensors
        return ops.skinny_gemm_fwd(dl.contiguous(), queue.detach()), None


class _L2NormRowsFn(torch.autograd.Function):
    """F.normalize(x, dim=1) (moco2_module.py:256, 259) on cmu_l2_normalize_rows / _bwd."""

    @staticmethod
    def forward(ctx, x):
        x = x.detach().float().contiguous()
        out = torch.empty_like(x)
        ops.l2_normalize_rows(x, out)
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dx = torch.empty_like(x)
        ops.l2_normalize_rows_bwd(x, dy.float().contiguous(), dx)
        return dx


class _MocoLogitsFn2(torch.autograd.Function):
    """logits = cat([einsum("nc,nc->n", q, k)[:, None], einsum("nc,ck->nk", q, queue)], 1) / T (moco2_module.py:258-267): the negatives on the
    skinny kernels (``_QueueLogitsFn``'s arithmetic), positives + concatenation + temperature in one assemble kernel; backward: one split
    kernel, the skinny kernel for dq = dl_neg @ queue^T, one kernel adding the positive term -- a single gradient for q (no accumulation
    of two autograd branches)."""

    @staticmethod
    def forward(ctx, q, k, queue, temperature):
        q, k, queue = q.detach().contiguous(), k.detach().contiguous(), queue.detach()
        lneg = ops.skinny_gemm_dgrad(q, queue)
        logits = torch.empty((q.shape[0], queue.shape[1] + 1), dtype=torch.float32, device=q.device)
        ops.moco_logits_assemble(q, k, lneg, logits, 1.0 / temperature)
        ctx.save_for_backward(k, queue)
        ctx.inv_t = 1.0 / temperature
        return logits

    @staticmethod
    def backward(ctx, dl):
        k, queue = ctx.saved_tensors
        dl = dl.float().contiguous()
        dlneg = torch.empty((dl.shape[0], dl.shape[1] - 1), dtype=torch.float32, device=dl.device)
        ops.moco_logits_split(dl, dlneg, ctx.inv_t)
        dq = ops.skinny_gemm_fwd(dlneg, queue)
        ops.moco_logits_addpos(dl, k, dq, ctx.inv_t)
        return dq, None, None, None


class _RowCrossEntropyFn(torch.autograd.Function):
    """F.cross_entropy(logits, target) with mean reduction (moco2_module.py:283, 324) on cmu_row_cross_entropy."""

    @staticmethod
    def forward(ctx, logits, target):
        loss, dl, _ = ops.row_cross_entropy(logits.detach().float().contiguous(), target, want_grad=logits.requires_grad)
        ctx.dl = dl
        return loss.reshape(())

    @staticmethod
    def backward(ctx, go):
        # (scaled into a fresh tensor: ctx.dl survives a backward with retain_graph=True, as F.cross_entropy's saved tensors do)
        dl = ctx.dl.clone()
        ops.scale_by_device_scalar(dl, go.float().contiguous())
        return dl, None


def row_cross_entropy(logits, target):
    return _RowCrossEntropyFn.apply(logits, target)


def queue_logits(q, queue):
    """``einsum("nc,ck->nk", [q, queue])`` of the non-fused ``Moco_v2.forward`` on the skinny kernels: up to
    ``ops.SKINNY_MAX_ROWS`` = 256 query rows per GPU (the reference's batch size, moco2_module.py:91); no library GEMM behind it."""
    if q.is_cuda and q.dtype == torch.float32 and queue.dtype == torch.float32 and 1 <= q.shape[0] <= ops.SKINNY_MAX_ROWS \
            and queue.is_contiguous() and queue.shape[1] % 8 == 0:
        return _QueueLogitsFn.apply(q, queue)
    raise RuntimeError(f"queue_logits: needs CUDA fp32 q (rows <= {ops.SKINNY_MAX_ROWS}) and a contiguous (D, K) queue with K % 8 == 0, "
                       f"got q {tuple(q.shape)} {q.dtype} on {q.device}, queue {tuple(queue.shape)} (no library / CPU fallback)")


class Moco_v2(nn.Module):
    def __init__(self, base_encoder=None, emb_dim=1024, num_negatives=65536, encoder_momentum=0.999,
                 softmax_temperature=0.07, learning_rate=0.03, momentum=0.9, weight_decay=1e-4, batch_size=256,
                 use_mlp=False, dtype="f32", base_ch=64, depth=5, shuffle_bn=True, **kwargs):
        super().__init__()
        self.hparams = dict(emb_dim=emb_dim, num_negatives=num_negatives, encoder_momentum=encoder_momentum,
                            softmax_temperature=softmax_temperature, learning_rate=learning_rate, momentum=momentum,
                            weight_decay=weight_decay, batch_size=batch_size, use_mlp=use_mlp)
        # batch shuffle of the key images across ranks (moco2_module.py:177-222, applied whenever the reference runs under
        # DDP): only acts when a process group with more than one rank exists
        self.shuffle_bn = bool(shuffle_bn)
        if use_mlp:
            raise NotImplementedError("use_mlp needs an fc attribute the reference's UNet_encoder does not have "
                                      "(moco2_module.py:115-118 would fail there too)")
        base_encoder = base_encoder if isinstance(base_encoder, nn.Module) else UNet_encoder(base_ch=base_ch, depth=depth, dtype=dtype)
        self.encoder_q, self.encoder_k = self.init_encoders(base_encoder)
        for pq, pk in zip(self.encoder_q.parameters(), self.encoder_k.parameters()):
            pk.data.copy_(pq.data)
            pk.requires_grad = False
        self.register_buffer("queue", F.normalize(torch.randn(emb_dim, num_negatives), dim=0))
        self.register_buffer("queue_ptr", torch.zeros(1, dtype=torch.long))
        self.register_buffer("val_queue", F.normalize(torch.randn(emb_dim, num_negatives), dim=0))
        self.register_buffer("val_queue_ptr", torch.zeros(1, dtype=torch.long))

    def init_encoders(self, base_encoder):
        return copy.deepcopy(base_encoder), copy.deepcopy(base_encoder)

    @torch.no_grad()
    def _momentum_update_key_encoder(self):
        em = self.hparams["encoder_momentum"]
        arenas = getattr(self, "_ema_arenas", None)      # set by pretrain.MocoPretrainer: both encoders live in flat arenas
        if arenas is not None:
            ops.ema_update(arenas[0], arenas[1], em)
            return
        for pq, pk in zip(self.encoder_q.parameters(), self.encoder_k.parameters()):
            ops.ema_update(pk.data.view(-1), pq.data.view(-1), em)

    @torch.no_grad()
    def _dequeue_and_enqueue(self, keys, queue_ptr, queue):
        """moco2_module.py:160-175 (API-faithful torch version; the fused step enqueues inside its kernel)."""
        keys = concat_all_gather(keys)
        bs = keys.shape[0]
        ptr = int(queue_ptr)
        assert self.hparams["num_negatives"] % bs == 0
        queue[:, ptr:ptr + bs] = keys.T
        queue_ptr[0] = (ptr + bs) % self.hparams["num_negatives"]

    @staticmethod
    def _world():
        return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1

    def _check_enqueue(self, n_keys):
        """The reference's enqueue asserts ``num_negatives % batch_size == 0`` and writes ``queue[:, ptr:ptr + bs]`` -- a pointer
        that is not a multiple of the (gathered) batch fails there with a shape error (moco2_module.py:169-172).  Same loud failure
        here, from a host-side shadow of the pointer (no device read per step; re-read from the buffer after a ``load_state_dict``
        or when the batch size changes)."""
        K = self.hparams["num_negatives"]
        if K % n_keys != 0:
            raise AssertionError(f"num_negatives={K} must be a multiple of the enqueued batch {n_keys} (moco2_module.py:169)")
        sh = self.__dict__.get("_ptr_shadow")
        # (the fused kernel advances the pointer through its raw address: no version bump; an in-place edit by torch -- a non-fused
        # _dequeue_and_enqueue on the training queue, queue_ptr.zero_() -- bumps it, and the shadow is re-read from the buffer)
        ver = self.queue_ptr._version
        if sh is None or sh[1] != n_keys or sh[2] != self.queue_ptr.data_ptr() or sh[3] != ver:
            sh = [int(self.queue_ptr.item()), n_keys, self.queue_ptr.data_ptr(), ver]
        if sh[0] % n_keys != 0:
            raise RuntimeError(f"queue pointer {sh[0]} is not a multiple of the enqueued batch {n_keys}: the reference's "
                               "queue[:, ptr:ptr + bs] = keys.T (moco2_module.py:172) fails here too")
        sh[0] = (sh[0] + n_keys) % K
        self.__dict__["_ptr_shadow"] = sh

    def _load_from_state_dict(self, *args, **kwargs):
        self.__dict__.pop("_ptr_shadow", None)
        return super()._load_from_state_dict(*args, **kwargs)

    @torch.no_grad()
    def _batch_shuffle_ddp(self, x):
        """moco2_module.py:177-201: gather the key images of all ranks, draw ONE permutation on rank 0 (global CPU generator,
        as the reference's ``torch.randperm(n).cuda()``), broadcast it, keep this rank's share -> (images, idx_unshuffle)."""
        n_this = x.shape[0]
        x_gather = concat_all_gather(x.contiguous())
        n_all = x_gather.shape[0]
        idx_shuffle = torch.randperm(n_all).to(x.device)
        dist.broadcast(idx_shuffle, src=0)
        idx_unshuffle = torch.argsort(idx_shuffle)
        idx_this = idx_shuffle.view(n_all // n_this, -1)[dist.get_rank()]
        return x_gather[idx_this], idx_unshuffle

    @torch.no_grad()
    def _batch_unshuffle_ddp(self, x, idx_unshuffle):
        """moco2_module.py:203-219."""
        n_this = x.shape[0]
        x_gather = concat_all_gather(x.contiguous())
        idx_this = idx_unshuffle.view(x_gather.shape[0] // n_this, -1)[dist.get_rank()]
        return x_gather[idx_this]

    def _encode_keys(self, img_k):
        """Key features with the batch shuffle around the key encoder when more than one rank runs (moco2_module.py:240-252)."""
        with torch.no_grad():
            if self.shuffle_bn and self._world() > 1:
                img_s, idx_unshuffle = self._batch_shuffle_ddp(img_k)
                return self._batch_unshuffle_ddp(self.encoder_k(img_s), idx_unshuffle)
            return self.encoder_k(img_k)

    def forward(self, img_q, img_k, queue):
        """moco2_module.py:224-270 on the library's kernels only (round 5: F.normalize / torch.cat / the elementwise positives are gone
        from this path): row normalisation, [q.k | q @ queue] / T."""
        q = _L2NormRowsFn.apply(self.encoder_q(img_q))
        with torch.no_grad():
            k = _L2NormRowsFn.apply(self._encode_keys(img_k))
        # moco2_module.py:262 multiplies by queue.clone().detach(): the reference's order is forward -> _dequeue_and_enqueue ->
        # loss.backward(), and the backward needs the PRE-enqueue queue.  The copy is only made where a backward can follow (grad
        # mode on and q in the graph); validation (no_grad) reads the live buffer.  The fused training_step needs neither.
        qd = queue.detach()
        if torch.is_grad_enabled() and q.requires_grad:
            qd = qd.clone()
        if not (q.is_cuda and 1 <= q.shape[0] <= ops.SKINNY_MAX_ROWS and qd.is_contiguous() and qd.shape[1] % 8 == 0 and qd.dtype == torch.float32):
            raise RuntimeError(f"Moco_v2.forward: needs CUDA fp32 features (rows <= {ops.SKINNY_MAX_ROWS}) and a contiguous (D, K) queue with K % 8 == 0, "
                               f"got q {tuple(q.shape)} on {q.device}, queue {tuple(qd.shape)} {qd.dtype} (no library / CPU fallback)")
        logits = _MocoLogitsFn2.apply(q, k, qd, float(self.hparams["softmax_temperature"]))
        labels = torch.zeros(logits.shape[0], dtype=torch.long, device=logits.device)
        return logits, labels, k, q

    def training_step(self, batch, batch_idx=0):
        """Fused step; ``batch`` = (img_q, img_k) or ((img_q, img_k), _) like the reference's loader output."""
        x = batch[0] if isinstance(batch[0], (tuple, list)) else batch
        img_q, img_k = x[0], x[1]
        self._check_enqueue(img_q.shape[0] * (self._world() if dp_exchanges() else 1))
        self._momentum_update_key_encoder()
        q_raw = self.encoder_q(img_q)
        k_raw = self._encode_keys(img_k)
        return _MocoLossFn.apply(q_raw, k_raw, self.queue, self.queue_ptr, self.hparams["softmax_temperature"])


    def _compute_l_s(self, output, target, keys, queue=None):
        """moco2_module.py:272-285 (the non-fused form of the loss, kept for callers of the reference's API): enqueue ``keys`` into the
        TRAINING queue -- the reference ignores its ``queue`` argument and always uses self.queue / self.queue_ptr -- then the
        InfoNCE cross entropy of ``output`` (the logits of ``forward``) against ``target``."""
        self._dequeue_and_enqueue(keys, queue=self.queue, queue_ptr=self.queue_ptr)
        self.__dict__.pop("_ptr_shadow", None)       # the pointer moved outside the fused step: its host-side shadow is re-read
        return row_cross_entropy(output, target)

    def configure_optimizers(self, max_epochs=None):
        """moco2_module.py:338-349: SGD(lr, momentum, weight_decay) over the trainable parameters + cosine annealing over
        ``max_epochs`` (the reference reads ``self.trainer.max_epochs``; without a Lightning trainer pass it here).  Returns
        ([optimizer], [scheduler]) like the reference -- the optimiser is the fused one-launch SGD over a flat arena
        (optim.FusedSGD, csrc/optim.hip), the scheduler an object with ``step()`` / ``get_last_lr()`` that follows
        torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, max_epochs) (closed form, pretrain.moco_cosine_lr)."""
        from .optim import FlatParams, FusedSGD
        from .pretrain import moco_cosine_lr
        if max_epochs is None:
            tr = getattr(self, "trainer", None)
            max_epochs = getattr(tr, "max_epochs", None)
        if max_epochs is None:
            raise ValueError("configure_optimizers: max_epochs is needed for the cosine schedule (moco2_module.py:346-348)")
        want = {n for n, p in self.named_parameters() if p.requires_grad}
        flat = FlatParams(self, names=lambda n: n in want)
        hp = self.hparams
        opt = FusedSGD(flat, lr=hp["learning_rate"], momentum=hp["momentum"], weight_decay=hp["weight_decay"])
        opt.auto_gather = True          # loss.backward(); optimizer.step() as with torch.optim.SGD

        class _Cosine:
            def __init__(sched):
                sched.base_lr, sched.epoch = hp["learning_rate"], 0

            def step(sched):
                sched.epoch += 1
                opt.lr = moco_cosine_lr(sched.base_lr, sched.epoch, max_epochs)

            def get_last_lr(sched):
                return [opt.lr]
        return [opt], [_Cosine()]

    @torch.no_grad()
    def validation_step(self, batch, batch_idx=0):
        """moco2_module.py:311-329: forward against ``val_queue``, enqueue the keys there, cross entropy and top-1 / top-5 precision.
        ``batch`` = ((img_1, img_2), labels) or (img_1, img_2); call it with the module in eval mode, as Lightning's loop does."""
        x = batch[0] if isinstance(batch[0], (tuple, list)) else batch
        output, target, keys, _ = self(img_q=x[0], img_k=x[1], queue=self.val_queue)
        self._dequeue_and_enqueue(keys, queue_ptr=self.val_queue_ptr, queue=self.val_queue)
        # cross entropy + the target's rank in one kernel (precision@k: hit iff fewer than k logits lie strictly above the target's)
        loss, _, rank = ops.row_cross_entropy(output.detach().float().contiguous(), target, want_grad=False, want_rank=True)
        acc1, acc5 = precision_from_rank(rank, (1, 5))
        return {"val_loss": loss.reshape(()), "val_acc1": acc1, "val_acc5": acc5}

    @staticmethod
    def validation_epoch_end(outputs):
        """moco2_module.py:331-337: the means of the per-batch results (returned instead of logged)."""
        return {k: torch.stack([o[k].reshape(()) for o in outputs]).mean() for k in ("val_loss", "val_acc1", "val_acc5")}


def precision_from_rank(rank, top_k=(1,)):
    """precision_at_k from the per-row rank of the target (cmu_row_cross_entropy): percentage of rows with rank < k, as (1,) tensors."""
    rows = rank.shape[0]
    return [(rank < k).sum(dtype=torch.float32).reshape(1) * (100.0 / rows) for k in top_k]


def precision_at_k(output, target, top_k=(1,)):
    """pl_bolts/metrics/aggregation.py:19-32: for each k the percentage of rows whose target is among the k largest logits, as (1,)
    tensors."""
    kmax, rows = max(top_k), target.shape[0]
    hit = output.topk(kmax, dim=1, largest=True, sorted=True).indices.eq(target.view(-1, 1))
    return [hit[:, :k].any(dim=1).sum(dtype=torch.float32).reshape(1) * (100.0 / rows) for k in top_k]


class _MocoLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q_raw, k_raw, queue, queue_ptr, temperature):
        B, D = q_raw.shape
        K = queue.shape[1]
        dev = q_raw.device
        keys_all = None
        if dp_exchanges():
            kn = torch.empty_like(k_raw)
            ops.l2_normalize_rows(k_raw.detach().contiguous(), kn)
            keys_all = concat_all_gather(kn)                       # moco2_module.py:163-164
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        dq = torch.empty_like(q_raw)
        ws = torch.empty(_lib.lib().cmu_moco_ws_bytes(B, D, K), dtype=torch.uint8, device=dev)
        ops.moco_infonce_enqueue(q_raw.detach().contiguous(), k_raw.detach().contiguous(), keys_all, queue, queue_ptr, loss, dq,
                                 None, temperature, ws)
        ctx.save_for_backward(dq)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dq,) = ctx.saved_tensors
        return dq * g, None, None, None, None
