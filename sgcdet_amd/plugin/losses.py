"""Losses of the FCOS3D-style head, as plain differentiable torch functions.

``axis_aligned_iou_loss`` follows the vendored mmdet3d implementation (packages/mmdetection3d/mmdet3d/models/losses/
axis_aligned_iou_loss.py:10-29 + core/bbox/iou_calculators/iou3d_calculator.py ``axis_aligned_bbox_overlaps_3d``,
``is_aligned=True``) and is pinned to it by tests/golden/head_targets.npz.  ``sigmoid_focal_loss`` and
``sigmoid_bce_loss`` restate mmdet 2.25.1's ``FocalLoss(use_sigmoid=True)`` (= mmcv's ``sigmoid_focal_loss`` CUDA op:
integer targets, any target outside [0, C) -- the head's -1 background -- is negative for every class) and
``CrossEntropyLoss(use_sigmoid=True)``; mmdet / mmcv are pip dependencies of the reference that are not vendored, so
these two are UNPINNED against their originals (published formulas; checked against closed forms in the tests).
Reduction everywhere: ``sum(loss * weight) / avg_factor * loss_weight`` (mmdet ``weight_reduce_loss`` with
``reduction='mean'`` and an ``avg_factor``).  The ARKit config's ``RotatedIoU3DLoss`` (mmcv ``diff_iou_rotated_3d``)
is not built.
"""
import torch
import torch.nn.functional as F

_FLT_MIN = 1.1754943508222875e-38


def axis_aligned_iou(pred, target, eps=1e-6):
    """IoU of paired boxes [..., 6] (x1, y1, z1, x2, y2, z2)."""
    area1 = (pred[..., 3] - pred[..., 0]) * (pred[..., 4] - pred[..., 1]) * (pred[..., 5] - pred[..., 2])
    area2 = (target[..., 3] - target[..., 0]) * (target[..., 4] - target[..., 1]) * (target[..., 5] - target[..., 2])
    lt = torch.max(pred[..., :3], target[..., :3])
    rb = torch.min(pred[..., 3:], target[..., 3:])
    wh = (rb - lt).clamp(min=0)
    overlap = wh[..., 0] * wh[..., 1] * wh[..., 2]
    union = torch.max(area1 + area2 - overlap, overlap.new_tensor([eps]))
    return overlap / union


def _reduce(loss, weight, avg_factor, loss_weight):
    if weight is not None:
        loss = loss * weight
    loss = loss.sum() / avg_factor if avg_factor is not None else loss.mean()
    return loss_weight * loss


def axis_aligned_iou_loss(pred, target, weight=None, avg_factor=None, loss_weight=1.0):
    if weight is not None and not torch.any(weight > 0):
        return (pred * weight.reshape(-1, *([1] * (pred.dim() - 1)))).sum() * loss_weight
    return _reduce(1 - axis_aligned_iou(pred, target), weight, avg_factor, loss_weight)


def sigmoid_focal_loss(pred, target, gamma=2.0, alpha=0.25, weight=None, avg_factor=None, loss_weight=1.0):
    """pred [N, C] logits, target [N] int64 class index (anything outside [0, C) = background)."""
    C = pred.shape[1]
    p = pred.sigmoid()
    pos = target.reshape(-1, 1) == torch.arange(C, device=pred.device).reshape(1, C)
    term_pos = -alpha * (1 - p).pow(gamma) * torch.log(p.clamp(min=_FLT_MIN))
    term_neg = -(1 - alpha) * p.pow(gamma) * torch.log((1 - p).clamp(min=_FLT_MIN))
    loss = torch.where(pos, term_pos, term_neg)
    if weight is not None:
        weight = weight.reshape(-1, 1)
    return _reduce(loss, weight, avg_factor, loss_weight)


def sigmoid_bce_loss(pred, target, weight=None, avg_factor=None, loss_weight=1.0):
    """pred [N] logits, target [N] float in [0, 1] (the centerness targets)."""
    loss = F.binary_cross_entropy_with_logits(pred, target.float(), reduction="none")
    valid = (target >= 0).float()                      # mmdet's ignore mask; centerness targets are >= 0
    weight = valid if weight is None else weight * valid
    return _reduce(loss, weight, avg_factor, loss_weight)
