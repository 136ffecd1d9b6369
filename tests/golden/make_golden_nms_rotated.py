#!/usr/bin/env python3
"""Generates tests/golden/nms_rotated_multiclass.npz by running the reference's own ``box3d_multiclass_nms`` and
``nms_bev`` (/root/reference/packages/mmdetection3d/mmdet3d/core/post_processing/box3d_nms.py:8-128, :231-268) on
seeded ARKit-like candidates, called exactly as ``SunRgbdImVoxelHeadV2._nms`` does
(/root/reference/mmdet3d_plugin/models/dense_heads/imvoxel_head_v2.py:565-584: dummy background column, BEV
rectangles, ``max_num = nms_pre``).  Build-container only (needs /root/reference).

What this pins and what it cannot: ``mmcv.ops.nms_rotated`` is a compiled CUDA op of mmcv-full 1.5.3, a pip
dependency that is not vendored in the reference and not installed here.  It is substituted by a plain greedy NMS
over an exact float64 rotated IoU (convex polygon clipping, below) -- so the fixture pins the reference's *glue*
(class loop, score threshold, BEV conversion, ordering, concatenation, max_num cut) bit for bit and the rotated IoU
only up to the candidates whose IoU lies within 1e-4 of the threshold (none in these cases: checked on generation).
Nothing of the reference is copied: the committed fixture holds inputs and outputs."""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/packages/mmdetection3d/mmdet3d/core/post_processing/box3d_nms.py"


# ---- exact rotated IoU in float64: Sutherland-Hodgman clip of rectangle A by rectangle B ------------------------
def rect_corners(b):
    x, y, w, h, a = (float(v) for v in b)
    c, s = np.cos(a), np.sin(a)
    loc = np.array([[-w / 2, -h / 2], [w / 2, -h / 2], [w / 2, h / 2], [-w / 2, h / 2]])
    rot = np.array([[c, -s], [s, c]])
    return loc @ rot.T + np.array([x, y])            # counter-clockwise


def clip(subject, a, b):
    """keep the part of polygon ``subject`` on the left of the directed line a -> b"""
    out = []
    n = len(subject)
    for i in range(n):
        p, q = subject[i], subject[(i + 1) % n]
        sp = (b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0])
        sq = (b[0] - a[0]) * (q[1] - a[1]) - (b[1] - a[1]) * (q[0] - a[0])
        if sp >= 0:
            out.append(p)
        if (sp >= 0) != (sq >= 0):
            t = sp / (sp - sq)
            out.append(p + t * (q - p))
    return out


def area(poly):
    if len(poly) < 3:
        return 0.0
    p = np.array(poly)
    return 0.5 * abs(np.dot(p[:, 0], np.roll(p[:, 1], -1)) - np.dot(p[:, 1], np.roll(p[:, 0], -1)))


def iou_rotated_f64(b1, b2):
    a1, a2 = float(b1[2]) * float(b1[3]), float(b2[2]) * float(b2[3])
    if a1 <= 0 or a2 <= 0:
        return 0.0
    d = np.hypot(float(b1[0]) - float(b2[0]), float(b1[1]) - float(b2[1]))
    if d > 0.5 * (np.hypot(b1[2], b1[3]) + np.hypot(b2[2], b2[3])):
        return 0.0
    poly = list(rect_corners(b1))
    cb = rect_corners(b2)
    for i in range(4):
        poly = clip(poly, cb[i], cb[(i + 1) % 4])
        if not poly:
            return 0.0
    inter = area(poly)
    return inter / (a1 + a2 - inter)


NEAR = []      # |IoU - thr| of every decision the stand-in took: the fixture must not hinge on the last digits


def nms_rotated_standin(dets, scores, iou_threshold, labels=None):
    """mmcv.ops.nms_rotated(dets [n,5] xywhr, scores, thr) -> (dets+score, keep indices by descending score)"""
    order = scores.sort(0, descending=True)[1]
    b = dets.double().numpy()
    removed = np.zeros(len(b), bool)
    keep = []
    for pi, p in enumerate(order.tolist()):
        if removed[p]:
            continue
        keep.append(p)
        for q in order.tolist()[pi + 1:]:
            if removed[q]:
                continue
            v = iou_rotated_f64(b[p], b[q])
            NEAR.append(abs(v - iou_threshold))
            if v > iou_threshold:
                removed[q] = True
    keep = torch.tensor(keep, dtype=torch.long)
    return torch.cat((dets[keep], scores[keep].reshape(-1, 1)), dim=1), keep


def load_reference():
    numba = types.ModuleType("numba")
    numba.jit = lambda *a, **k: (lambda fn: fn)
    sys.modules.setdefault("numba", numba)
    mmcv = sys.modules.setdefault("mmcv", types.ModuleType("mmcv"))
    ops = types.ModuleType("mmcv.ops")
    ops.nms = None
    ops.nms_rotated = nms_rotated_standin
    mmcv.ops = ops
    sys.modules["mmcv.ops"] = ops
    spec = importlib.util.spec_from_file_location("_ref_box3d_nms_rot", REF)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.box3d_multiclass_nms


def candidates(n, n_cls, seed, n_obj):
    """detections piled around a few oriented objects, sparse class scores with exact zeros (the head multiplies by
    the valid mask), distinct positive scores"""
    g = torch.Generator().manual_seed(seed)
    ctr = (torch.rand(n_obj, 3, generator=g) - 0.5) * torch.tensor([5.0, 5.0, 1.5])
    size = 0.3 + torch.rand(n_obj, 3, generator=g) * 1.4
    yaw = (torch.rand(n_obj, generator=g) - 0.5) * 6.0
    obj = torch.randint(0, n_obj, (n,), generator=g)
    c = ctr[obj] + torch.randn(n, 3, generator=g) * 0.10
    s = size[obj] * (1.0 + torch.randn(n, 3, generator=g) * 0.12).clamp(0.5, 1.5)
    a = yaw[obj] + torch.randn(n, generator=g) * 0.15
    boxes = torch.cat([c, s, a[:, None]], dim=1).float()
    scores = torch.rand(n, n_cls, generator=g).float() ** 3
    scores = scores * (torch.rand(n, n_cls, generator=g) < 0.45)          # many exact zeros
    scores = scores * (torch.rand(n, 1, generator=g) < 0.9)               # rows masked out by `valid`
    return boxes, scores


def main():
    multiclass = load_reference()
    out = {}
    cases = [  # n, classes, seed, objects, score_thr, nms_thr, max_num
        (160, 5, 11, 9, 0.0, 0.15, 1000),     # the ARKit test_cfg
        (220, 3, 12, 6, 0.05, 0.15, 10),      # survivors cut to max_num
        (90, 17, 13, 12, 0.0, 0.30, 1000),    # 17 ARKit classes
        (12, 4, 14, 3, 0.60, 0.15, 1000),     # classes with no candidate at all
        (8, 3, 15, 2, 0.999, 0.15, 1000),     # nothing passes the score threshold
    ]
    for ci, (n, n_cls, seed, n_obj, score_thr, nms_thr, max_num) in enumerate(cases):
        boxes, scores = candidates(n, n_cls, seed, n_obj)
        padded = torch.cat([scores, scores.new_zeros(n, 1)], dim=1)
        bev = torch.stack((boxes[:, 0] - boxes[:, 3] / 2, boxes[:, 1] - boxes[:, 4] / 2,
                           boxes[:, 0] + boxes[:, 3] / 2, boxes[:, 1] + boxes[:, 4] / 2, boxes[:, 6]), dim=1)
        cfg = types.SimpleNamespace(use_rotate_nms=True, nms_thr=nms_thr)
        NEAR.clear()
        res = multiclass(mlvl_bboxes=boxes, mlvl_bboxes_for_nms=bev, mlvl_scores=padded, score_thr=score_thr,
                         max_num=max_num, cfg=cfg)
        margin = min(NEAR) if NEAR else 1.0
        assert margin > 1e-4, f"case {ci}: an IoU within {margin} of the threshold; change the seed"
        out[f"c{ci}_boxes"], out[f"c{ci}_scores"] = boxes.numpy(), scores.numpy()
        out[f"c{ci}_cfg"] = np.array([score_thr, nms_thr, max_num], dtype=np.float64)
        out[f"c{ci}_out_boxes"], out[f"c{ci}_out_scores"] = res[0].numpy(), res[1].numpy()
        out[f"c{ci}_out_labels"] = res[2].numpy()
        print(f"case {ci}: {n} x {n_cls} -> {len(res[1])} kept, closest IoU to the threshold {margin:.2e}")
    # a table of pairwise IoUs for the IoU entry point itself (float64 clip as the independent value)
    g = torch.Generator().manual_seed(21)
    a = torch.cat([(torch.rand(40, 2, generator=g) - 0.5) * 3, 0.2 + torch.rand(40, 2, generator=g) * 2,
                   (torch.rand(40, 1, generator=g) - 0.5) * 7], dim=1).float()
    b = a[torch.randperm(40, generator=g)] + torch.randn(40, 5, generator=g).float() * 0.1
    b[:, 2:4] = b[:, 2:4].abs() + 0.05
    b[:5] = a[:5]                                      # identical boxes
    b[5:8, :2] = a[5:8, :2]; b[5:8, 2:4] = a[5:8, 2:4] * 0.5; b[5:8, 4] = a[5:8, 4]     # contained
    b[8, :] = torch.tensor([50.0, 50.0, 1.0, 1.0, 0.3])                                # far away
    a[9, 4] = 0.0; b[9, 4] = 0.0                                                        # axis aligned pair
    iou = np.array([[iou_rotated_f64(x, y) for y in b.double().numpy()] for x in a.double().numpy()])
    out["iou_a"], out["iou_b"], out["iou_f64"] = a.numpy(), b.numpy(), iou
    np.savez_compressed(os.path.join(HERE, "nms_rotated_multiclass.npz"), **out)


if __name__ == "__main__":
    main()
