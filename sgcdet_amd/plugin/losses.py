"""Losses of the FCOS3D-style head, as plain differentiable torch functions.

``axis_aligned_iou_loss`` follows the vendored mmdet3d implementation (packages/mmdetection3d/mmdet3d/models/losses/
axis_aligned_iou_loss.py:10-29 + core/bbox/iou_calculators/iou3d_calculator.py ``axis_aligned_bbox_overlaps_3d``,
``is_aligned=True``) and is pinned to it by tests/golden/head_targets.npz.  ``sigmoid_focal_loss`` and
``sigmoid_bce_loss`` restate mmdet 2.25.1's ``FocalLoss(use_sigmoid=True)`` (= mmcv's ``sigmoid_focal_loss`` CUDA op:
integer targets, any target outside [0, C) -- the head's -1 background -- is negative for every class) and
``CrossEntropyLoss(use_sigmoid=True)``; mmdet / mmcv are pip dependencies of the reference that are not vendored, so
these two are UNPINNED against their originals (published formulas; checked against closed forms in the tests).
Reduction everywhere: ``sum(loss * weight) / avg_factor * loss_weight`` (mmdet ``weight_reduce_loss`` with
``reduction='mean'`` and an ``avg_factor``).  ``rotated_iou_3d_loss`` is the ARKit config's ``RotatedIoU3DLoss``
(vendored wrapper packages/mmdetection3d/mmdet3d/models/losses/rotated_iou_loss.py:10-26 around mmcv's
``diff_iou_rotated_3d``, not vendored): 1 - IoU of paired rotated boxes, differentiable through the exact intersection
polygon.  mmcv collects edge intersections + contained corners and sorts them with a CUDA op; here the rectangle of
the prediction is clipped against the four edges of the target (Sutherland-Hodgman on a fixed 8-slot polygon, all
pairs at once in torch), which yields the same polygon and the same gradients almost everywhere.  UNPINNED against
mmcv; its forward is checked against the float64 polygon clip of the NMS fixture and its gradients by gradcheck.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..mmcv_lite import LOSSES

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


def _reduce(loss, weight, avg_factor, loss_weight, reduction="mean"):
    """mmdet ``weight_reduce_loss``: element weights, then 'none' | 'sum' | 'mean' (``avg_factor`` replaces the count
    under 'mean' and is refused under 'sum')."""
    if weight is not None:
        loss = loss * weight
    if reduction == "none":
        return loss_weight * loss
    if reduction == "sum":
        if avg_factor is not None:
            raise ValueError('avg_factor can not be used with reduction="sum"')
        return loss_weight * loss.sum()
    if reduction != "mean":
        raise ValueError(f"unknown reduction {reduction!r}")
    loss = loss.sum() / avg_factor if avg_factor is not None else loss.mean()
    return loss_weight * loss


def axis_aligned_iou_loss(pred, target, weight=None, avg_factor=None, loss_weight=1.0, reduction="mean"):
    if weight is not None and not torch.any(weight > 0) and reduction != "none":
        return (pred * weight.reshape(-1, *([1] * (pred.dim() - 1)))).sum() * loss_weight
    return _reduce(1 - axis_aligned_iou(pred, target), weight, avg_factor, loss_weight, reduction)


def sigmoid_focal_loss(pred, target, gamma=2.0, alpha=0.25, weight=None, avg_factor=None, loss_weight=1.0,
                       reduction="mean"):
    """pred [N, C] logits, target [N] int64 class index (anything outside [0, C) = background)."""
    C = pred.shape[1]
    p = pred.sigmoid()
    pos = target.reshape(-1, 1) == torch.arange(C, device=pred.device).reshape(1, C)
    term_pos = -alpha * (1 - p).pow(gamma) * torch.log(p.clamp(min=_FLT_MIN))
    term_neg = -(1 - alpha) * p.pow(gamma) * torch.log((1 - p).clamp(min=_FLT_MIN))
    loss = torch.where(pos, term_pos, term_neg)
    if weight is not None:
        weight = weight.reshape(-1, 1)
    return _reduce(loss, weight, avg_factor, loss_weight, reduction)


def sigmoid_bce_loss(pred, target, weight=None, avg_factor=None, loss_weight=1.0, reduction="mean"):
    """pred [N] logits, target [N] float in [0, 1] (the centerness targets)."""
    loss = F.binary_cross_entropy_with_logits(pred, target.float(), reduction="none")
    valid = (target >= 0).float()                      # mmdet's ignore mask; centerness targets are >= 0
    weight = valid if weight is None else weight * valid
    return _reduce(loss, weight, avg_factor, loss_weight, reduction)


def _rect_corners(x, y, w, h, a):
    """[n] each -> [n,4,2] counter-clockwise corners of rectangles centred (x, y), size (w, h), rotated by a."""
    c, s_ = torch.cos(a), torch.sin(a)
    dx = torch.stack((-w, w, w, -w), -1) * 0.5
    dy = torch.stack((-h, -h, h, h), -1) * 0.5
    return torch.stack((x[:, None] + dx * c[:, None] - dy * s_[:, None], y[:, None] + dx * s_[:, None] + dy * c[:, None]), -1)


def _clip_convex(poly, valid, a, b):
    """One Sutherland-Hodgman step for n polygons at once.  poly [n,K,2] with the first ``count`` slots live
    (valid [n,K] bool, a prefix), clipped against the half plane left of a -> b ([n,2] each); returns a polygon in
    2K slots' worth of candidates packed back into K slots (a convex polygon cut by a line gains at most one vertex;
    K = 8 holds a rectangle clipped four times)."""
    n, K, _ = poly.shape
    count = valid.sum(1, keepdim=True)                                       # [n,1]
    idx = torch.arange(K, device=poly.device)[None]
    nxt = torch.where(idx + 1 < count, idx + 1, torch.zeros_like(idx))       # successor inside the live prefix
    q = torch.gather(poly, 1, nxt[..., None].expand(-1, -1, 2))
    e = (b - a)[:, None]                                                      # [n,1,2]
    sp = e[..., 0] * (poly[..., 1] - a[:, None, 1]) - e[..., 1] * (poly[..., 0] - a[:, None, 0])
    sq = e[..., 0] * (q[..., 1] - a[:, None, 1]) - e[..., 1] * (q[..., 0] - a[:, None, 0])
    p_in, q_in = sp >= 0, sq >= 0
    denom = sp - sq
    t = sp / torch.where(denom == 0, torch.ones_like(denom), denom)
    cross = poly + t[..., None] * (q - poly)
    keep_p = valid & p_in                                                     # emit p
    keep_x = valid & (p_in != q_in)                                           # emit the crossing after p
    cand = torch.stack((poly, cross), 2).reshape(n, 2 * K, 2)                # p0, x0, p1, x1, ...
    cmask = torch.stack((keep_p, keep_x), 2).reshape(n, 2 * K)
    order = torch.sort((~cmask).to(torch.int8), dim=1, stable=True)[1][:, :K]     # live candidates first, in order
    out = torch.gather(cand, 1, order[..., None].expand(-1, -1, 2))
    return out, torch.gather(cmask, 1, order)


def rotated_bev_intersection(b1, b2):
    """[n,5] (x, y, w, h, angle) x2 -> [n] area of the intersection of paired rotated rectangles (differentiable)."""
    n = b1.shape[0]
    c1 = _rect_corners(b1[:, 0], b1[:, 1], b1[:, 2], b1[:, 3], b1[:, 4])
    c2 = _rect_corners(b2[:, 0], b2[:, 1], b2[:, 2], b2[:, 3], b2[:, 4])
    poly = torch.cat((c1, c1.new_zeros(n, 4, 2)), 1)
    valid = torch.arange(8, device=b1.device)[None].expand(n, 8) < 4
    for k in range(4):
        poly, valid = _clip_convex(poly, valid, c2[:, k], c2[:, (k + 1) % 4])
    count = valid.sum(1, keepdim=True)
    idx = torch.arange(8, device=b1.device)[None]
    nxt = torch.where(idx + 1 < count, idx + 1, torch.zeros_like(idx))
    q = torch.gather(poly, 1, nxt[..., None].expand(-1, -1, 2))
    cross = (poly[..., 0] * q[..., 1] - poly[..., 1] * q[..., 0]) * valid
    return 0.5 * cross.sum(1).abs()


def rotated_iou_3d(pred, target):
    """paired boxes [n,7] (x, y, z, w, l, h, alpha), gravity centre -> [n] IoU (mmcv diff_iou_rotated_3d)."""
    inter2d = rotated_bev_intersection(pred[:, [0, 1, 3, 4, 6]], target[:, [0, 1, 3, 4, 6]])
    zmax = torch.min(pred[:, 2] + pred[:, 5] / 2, target[:, 2] + target[:, 5] / 2)
    zmin = torch.max(pred[:, 2] - pred[:, 5] / 2, target[:, 2] - target[:, 5] / 2)
    inter = inter2d * (zmax - zmin).clamp(min=0)
    vol = pred[:, 3] * pred[:, 4] * pred[:, 5] + target[:, 3] * target[:, 4] * target[:, 5]
    return inter / (vol - inter)


def rotated_iou_3d_loss(pred, target, weight=None, avg_factor=None, loss_weight=1.0, reduction="mean"):
    if weight is not None and not torch.any(weight > 0) and reduction != "none":
        return pred.sum() * 0
    return _reduce(1 - rotated_iou_3d(pred, target), weight, avg_factor, loss_weight, reduction)


# ---- the LOSSES registry (config ``loss_*=dict(type=...)`` entries resolve here) -----------------
class _RegisteredLoss(nn.Module):
    """Shared shell of the registered losses: mmdet's calling convention
    ``loss(pred, target, weight=None, avg_factor=None, reduction_override=None)`` over one of the functions above."""
    fn = None

    def __init__(self, reduction="mean", loss_weight=1.0, **fn_kwargs):
        super().__init__()
        if reduction not in ("none", "sum", "mean"):
            raise ValueError(f"reduction must be 'none', 'sum' or 'mean', got {reduction!r}")
        self.reduction = reduction
        self.loss_weight = loss_weight
        self.fn_kwargs = fn_kwargs

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        if reduction_override not in (None, "none", "sum", "mean"):
            raise ValueError(f"reduction_override must be None, 'none', 'sum' or 'mean', got {reduction_override!r}")
        return type(self).fn(pred, target, weight=weight, avg_factor=avg_factor, loss_weight=self.loss_weight,
                             reduction=reduction_override or self.reduction, **self.fn_kwargs)

    def extra_repr(self):
        return f"reduction={self.reduction!r}, loss_weight={self.loss_weight}"


@LOSSES.register_module()
class AxisAlignedIoULoss(_RegisteredLoss):
    """``dict(type='AxisAlignedIoULoss', loss_weight=1.0)`` of the ScanNet configs (configs/SGCDet_ScanNet.py:111;
    mmdet3d/models/losses/axis_aligned_iou_loss.py:30-80): boxes [..., 6] as (x1, y1, z1, x2, y2, z2)."""
    fn = staticmethod(axis_aligned_iou_loss)


@LOSSES.register_module()
class RotatedIoU3DLoss(_RegisteredLoss):
    """``dict(type='RotatedIoU3DLoss', loss_weight=1.0)`` of the ARKit configs (configs/SGCDet_ARKit.py:114;
    mmdet3d/models/losses/rotated_iou_loss.py:29-84): boxes [n, 7] as (x, y, z, w, l, h, alpha); a per-coordinate
    weight [n, 7] is averaged over its last axis first, as the reference wrapper does (:75-76)."""
    fn = staticmethod(rotated_iou_3d_loss)

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        if weight is not None and weight.dim() > 1:
            weight = weight.mean(-1)
        return super().forward(pred, target, weight, avg_factor, reduction_override)


@LOSSES.register_module()
class FocalLoss(_RegisteredLoss):
    """The head's default ``loss_cls`` (imvoxel_head_v2.py:52-57): mmdet ``FocalLoss(use_sigmoid=True)`` on integer
    targets.  Only the sigmoid form exists (mmdet refuses the other too)."""
    fn = staticmethod(sigmoid_focal_loss)

    def __init__(self, use_sigmoid=True, gamma=2.0, alpha=0.25, reduction="mean", loss_weight=1.0, activated=False):
        if not use_sigmoid or activated:
            raise NotImplementedError("FocalLoss: only use_sigmoid=True on logits (activated=False) is implemented")
        super().__init__(reduction, loss_weight, gamma=gamma, alpha=alpha)


@LOSSES.register_module()
class CrossEntropyLoss(_RegisteredLoss):
    """The head's default ``loss_centerness`` (imvoxel_head_v2.py:46-50): mmdet ``CrossEntropyLoss(use_sigmoid=True)``
    = binary cross-entropy on logits with soft targets.  The softmax / mask forms are not on the path."""
    fn = staticmethod(sigmoid_bce_loss)

    def __init__(self, use_sigmoid=False, use_mask=False, reduction="mean", class_weight=None, loss_weight=1.0):
        if not use_sigmoid or use_mask or class_weight is not None:
            raise NotImplementedError("CrossEntropyLoss: only use_sigmoid=True without mask / class weights is implemented")
        super().__init__(reduction, loss_weight)
