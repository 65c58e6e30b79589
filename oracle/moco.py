"""Oracle: MoCo-v2 step on the UNet encoder, restated on CPU.  TEST INFRASTRUCTURE ONLY.

The reference module needs pytorch-lightning / lightning-bolts (absent here: SURVEY 8c), so it is
restated.  PARITY UNPINNED BY THE REFERENCE beyond the shared conv blocks (oracle/unet.py) and
torch.nn.functional.cross_entropy.

Follows /root/reference/Pretraining/MoCo/pl_bolts/models/self_supervised/moco/:
  encoder (+ global average pool)   moco_data_module.py:47-66  (UNet down path + bottleneck, mean over H,W)
  init queue                        moco2_module.py:128-131    randn(emb, K), L2-normalised over dim 0
  momentum update                   moco2_module.py:153-158    BEFORE the forward (A-8)
  forward (logits / labels)         moco2_module.py:224-270
  dequeue_and_enqueue               moco2_module.py:160-175    queue[:, ptr:ptr+B] = keys.T ; ptr = (ptr+B) % K
  loss                              moco2_module.py:272-285    enqueue first, then CE on the pre-enqueue logits
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
