"""Oracle: MoCo-v2 step on the UNet encoder, restated on CPU.  TEST INFRASTRUCTURE ONLY.

PINNED BY THE REFERENCE: oracle/gen_golden.py::gen_moco runs the reference's own Moco_v2 (moco2_module.py, UNet_encoder of
moco_data_module.py) in the build container behind a pytorch-lightning / torchvision plumbing stand-in, asserts this file
equal on two training steps and writes tests/golden/moco_ref.npz (one rank: the DDP-only shuffle-BN and key gather stay
restated and are covered by the two-rank tests).

Follows /root/reference/Pretraining/MoCo/pl_bolts/models/self_supervised/moco/:
  encoder (+ global average pool)   moco_data_module.py:47-66  (UNet down path + bottleneck, mean over H,W)
  init queue                        moco2_module.py:128-131    randn(emb, K), L2-normalised over dim 0
  momentum update                   moco2_module.py:153-158    BEFORE the forward (A-8)
  forward (logits / labels)         moco2_module.py:224-270
  dequeue_and_enqueue               moco2_module.py:160-175    queue[:, ptr:ptr+B] = keys.T ; ptr = (ptr+B) % K
  loss                              moco2_module.py:272-285    enqueue first, then CE on the pre-enqueue logits
  validation_step                   moco2_module.py:311-329    forward against val_queue (module in eval mode under Lightning),
                                                                enqueue into val_queue, CE, precision_at_k (metrics/aggregation.py:19-32);
                                                                pinned by tests/golden/moco_val_ref.npz (gen_golden.py::gen_moco_val)
"""
import torch
import torch.nn.functional as F

from . import unet as U


def encoder_gap(x_b1hw, sd, prefix, training=True):
    latent, _ = U.encoder_forward(x_b1hw, sd, prefix, training)
    return latent.mean([2, 3])


def init_queue(emb_dim, num_negatives, seed=0):
    g = torch.Generator().manual_seed(seed)
    return F.normalize(torch.randn(emb_dim, num_negatives, generator=g), dim=0)


def momentum_update(sd, m=0.999, src="encoder_q.", dst="encoder_k."):
    for k in list(sd.keys()):
        if k.startswith(src) and "running_" not in k and "num_batches" not in k:
            kt = dst + k[len(src):]
            sd[kt] = sd[kt] * m + sd[k].detach() * (1.0 - m)


def logits_from_embeddings(q_raw, k_raw, queue, temperature):
    """moco2_module.py:236-270 after the encoders: normalise, l_pos, l_neg, cat, /T, labels 0."""
    q = F.normalize(q_raw, dim=1)
    k = F.normalize(k_raw, dim=1).detach()
    l_pos = torch.einsum("nc,nc->n", q, k).unsqueeze(-1)
    l_neg = torch.einsum("nc,ck->nk", q, queue.clone().detach())
    logits = torch.cat([l_pos, l_neg], dim=1) / temperature
    labels = torch.zeros(logits.shape[0], dtype=torch.long)
    return logits, labels, k, q


def dequeue_and_enqueue(keys_all, queue, queue_ptr, num_negatives):
    """moco2_module.py:160-175; ``keys_all`` is already all-gathered (world*B, C)."""
    bs = keys_all.shape[0]
    ptr = int(queue_ptr)
    assert num_negatives % bs == 0
    queue[:, ptr:ptr + bs] = keys_all.T
    queue_ptr[0] = (ptr + bs) % num_negatives


def training_step(img_q, img_k, sd, queue, queue_ptr, temperature=0.07, m=0.999, gather=None, training=True):
    """moco2_module.py:287-309 (single global-crop pair).  Returns (loss, logits, k)."""
    momentum_update(sd, m)
    q_raw = encoder_gap(img_q, sd, "encoder_q.", training)
    with torch.no_grad():
        k_raw = encoder_gap(img_k, sd, "encoder_k.", training)
    logits, labels, k, q = logits_from_embeddings(q_raw, k_raw, queue, temperature)
    keys_all = k if gather is None else gather(k)
    dequeue_and_enqueue(keys_all, queue, queue_ptr, queue.shape[1])
    loss = F.cross_entropy(logits.float(), labels.long())
    return loss, logits, k


def precision_at_k(output, target, top_k=(1,)):
    """pl_bolts/metrics/aggregation.py:19-32: percentage of rows whose target is among the k largest logits, one (1,) tensor per k."""
    maxk, batch = max(top_k), target.size(0)
    pred = output.topk(maxk, 1, True, True)[1]
    hit = pred.eq(target.view(-1, 1))
    return [hit[:, :k].any(dim=1).float().sum().reshape(1) * (100.0 / batch) for k in top_k]


def validation_step(img_1, img_2, sd, val_queue, val_ptr, temperature=0.07, training=False):
    """moco2_module.py:311-329 on one rank; ``val_queue`` / ``val_ptr`` are updated in place.  Returns (loss, acc1, acc5)."""
    with torch.no_grad():
        q_raw = encoder_gap(img_1, sd, "encoder_q.", training)
        k_raw = encoder_gap(img_2, sd, "encoder_k.", training)
        logits, labels, k, _ = logits_from_embeddings(q_raw, k_raw, val_queue, temperature)
        dequeue_and_enqueue(k, val_queue, val_ptr, val_queue.shape[1])
        loss = F.cross_entropy(logits, labels)
        acc1, acc5 = precision_at_k(logits, labels, (1, 5))
    return loss, acc1, acc5


def make_moco_sd(seed, num_negatives=64, emb_dim=1024):
    """Seeded state of Moco_v2 in the reference's key names (encoder_q.* / encoder_k.* = the UNet encoder of
    moco_data_module.py:47-66, queue (emb, K), queue_ptr (1,)): what oracle/gen_golden.py::gen_moco loaded into the reference's own
    module for tests/golden/moco_ref.npz; regenerated from the seed by the tests."""
    sd = {}
    for prefix, s in (("encoder_q.", seed), ("encoder_k.", seed + 1)):
        part = U.make_state_dict(base_ch=64, depth=5, seed=s, encoder=True, decoder=False)
        sd.update({prefix + k: v for k, v in part.items()})
    sd["queue"] = init_queue(emb_dim, num_negatives, seed + 2)
    sd["queue_ptr"] = torch.zeros(1, dtype=torch.long)
    return sd


def moco_fixture_inputs(seed, B=4, S=64):
    g = torch.Generator().manual_seed(seed + 3)
    return tuple(torch.randn(B, 1, S, S, generator=g) for _ in range(4))      # (query, key) of step 1, (query, key) of step 2


def two_rank_step(sd0, inputs, perm, temperature, m):
    """One Moco_v2.training_step on TWO data-parallel ranks, emulated in one process (moco2_module.py:177-222, 224-285 with
    _use_ddp_or_ddp2 true): EMA first; the key images of both ranks gathered and permuted with rank 0's ``perm``; each rank's key
    encoder sees its shuffled share (its BatchNorm batch is a mixture of both ranks' images); the keys are un-shuffled back to their
    owners; every rank's logits use its own keys and the pre-enqueue queue; all 2B keys are enqueued on both ranks.
    ``inputs[r]`` = (query images, key images) of rank r.  Returns ([(loss_r, sd_r with .grad on the query parameters)], queue)."""
    B = inputs[0][0].shape[0]
    un = torch.argsort(perm)
    xk_all = torch.cat([inputs[0][1], inputs[1][1]])
    sds, ksh = [], []
    for r in range(2):
        sd = {k: v.clone() for k, v in sd0.items()}
        momentum_update(sd, m)
        with torch.no_grad():
            ksh.append(encoder_gap(xk_all[perm.view(2, -1)[r]], sd, "encoder_k.", True))
        sds.append(sd)
    k_all = F.normalize(torch.cat(ksh)[un], dim=1)
    out = []
    for r in range(2):
        sd = sds[r]
        for k in list(sd):
            if k.startswith("encoder_q.") and sd[k].is_floating_point() and "running" not in k:
                sd[k] = sd[k].clone().requires_grad_(True)
        q = F.normalize(encoder_gap(inputs[r][0], sd, "encoder_q.", True), dim=1)
        k_own = k_all[r * B:(r + 1) * B]
        logits = torch.cat([(q * k_own).sum(1, keepdim=True), q @ sd0["queue"]], 1) / temperature
        loss = F.cross_entropy(logits, torch.zeros(B, dtype=torch.long))
        loss.backward()
        out.append((loss.detach(), sd))
    queue = sd0["queue"].clone()
    queue[:, :2 * B] = k_all.t()
    return out, queue
