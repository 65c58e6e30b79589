"""Oracle: finetuning losses / metrics on logits.  TEST INFRASTRUCTURE ONLY.

Follows /root/reference/Finetuning/metrics.py:
  _threshold     metrics.py:128-133   (x > t).type(x.dtype)
  _take_channels metrics.py:111-125
  f_score        metrics.py:135-157   ((1+b^2)tp+eps)/((1+b^2)tp+b^2 fn+fp+eps)
  DiceLoss       metrics.py:160-180   1 - f_score(activation(y_pr), y_gt)
  iou / IoU      metrics.py:182-220   1 - (inter+eps)/(sum gt + sum pr - inter + eps)
  CrossEntropyLoss metrics.py:503     nn.CrossEntropyLoss with probability (one-hot float) targets
  SumOfLosses    metrics.py:53-62
Finetune loss used by the reference driver (train.py:455):
  DiceLoss(activation='softmax', threshold=0.5, ignore_channels=[0]) + CrossEntropyLoss()
nn.Softmax() with implicit dim resolves to dim=1 for 4-D input (SURVEY A-4).
"""
import torch
import torch.nn.functional as F


def _prep(logits, y, threshold, ignore_channels):
    pr = torch.softmax(logits, dim=1)
    if threshold is not None:
        pr = (pr > threshold).type(pr.dtype)
    if ignore_channels:
        keep = [c for c in range(pr.shape[1]) if c not in ignore_channels]
        idx = torch.tensor(keep)
        pr, y = pr.index_select(1, idx), y.index_select(1, idx)
    return pr, y


def dice_loss(logits, y, eps=1e-5, beta=1.0, threshold=0.5, ignore_channels=(0,)):
    pr, gt = _prep(logits, y, threshold, ignore_channels)
    tp = torch.sum(gt * pr)
    fp = torch.sum(pr) - tp
    fn = torch.sum(gt) - tp
    score = ((1 + beta ** 2) * tp + eps) / ((1 + beta ** 2) * tp + beta ** 2 * fn + fp + eps)
    return 1 - score


def iou_loss(logits, y, eps=1e-7, threshold=0.5, ignore_channels=(0,)):
    pr, gt = _prep(logits, y, threshold, ignore_channels)
    inter = torch.sum(gt * pr)
    union = torch.sum(gt) + torch.sum(pr) - inter + eps
    return 1 - (inter + eps) / union


def cross_entropy_prob(logits, y):
    """nn.CrossEntropyLoss()(logits, y) with y a probability tensor of logits' shape (mean over B*H*W)."""
    return F.cross_entropy(logits, y)


def dice_ce_loss(logits, y):
    """train.py:455 -- the finetuning criterion 'dice_loss + cross_entropy_loss'."""
    return dice_loss(logits, y) + cross_entropy_prob(logits, y)


def soft_skel(img, num_iter=10):
    """metrics.py:448-490 (SoftSkeletonize): iterated min/max pooling; img (B,C,H,W)."""
    import torch.nn.functional as F

    def erode(x):
        p1 = -F.max_pool2d(-x, (3, 1), (1, 1), (1, 0))
        p2 = -F.max_pool2d(-x, (1, 3), (1, 1), (0, 1))
        return torch.min(p1, p2)

    def opened(x):
        return F.max_pool2d(erode(x), (3, 3), (1, 1), (1, 1))

    skel = F.relu(img - opened(img))
    for _ in range(num_iter):
        img = erode(img)
        delta = F.relu(img - opened(img))
        skel = skel + F.relu(delta - skel * delta)
    return skel


def soft_cldice(logits, y_true, smooth=1.0, threshold=0.5, num_iter=10):
    """metrics.py:401-431 in the driver's configuration (train.py:464: activation 'softmax', threshold 0.5,
    ignore_channels [0]): binarised foreground vs the one-hot target's channel 1."""
    y_pred = (torch.softmax(logits, dim=1) > threshold).to(logits.dtype)[:, 1:2]
    yt = y_true[:, 1:2].to(logits.dtype)
    sp, st = soft_skel(y_pred, num_iter), soft_skel(yt, num_iter)
    tprec = ((sp * yt).sum() + smooth) / (sp.sum() + smooth)
    tsens = ((st * y_pred).sum() + smooth) / (st.sum() + smooth)
    return 1.0 - 2.0 * (tprec * tsens) / (tprec + tsens)
