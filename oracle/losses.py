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
