"""Indoor detection metrics: AP / recall per class and IoU threshold as the
reference's ``indoor_eval`` reports them (packages/mmdetection3d/mmdet3d/core/evaluation/indoor_eval.py:8-309:
``average_precision`` 'area' mode, the VOC-style greedy matching of ``eval_det_cls``, the result keys
``<class>_AP_<thr>``, ``mAP_<thr>``, ``<class>_rec_<thr>``, ``mAR_<thr>``), consuming what
``SGCDet.simple_test_from_features(..., as_results=True)`` returns (``bbox3d2result`` dicts).

Boxes are plain tensors here -- rows ``(cx, cy, cz, dx, dy, dz[, yaw])`` with the GRAVITY centre, i.e. what the heads
emit and what the datasets store as ``gt_boxes_upright_depth`` -- instead of mmdet3d box structures; objects with a
``gravity_center`` / ``tensor`` pair (mmdet3d's structures) are converted.  The 3-D IoU follows
``BaseInstance3DBoxes.overlaps`` (core/bbox/structures/base_box3d.py:424-487): height overlap x BEV overlap with the
BEV IoU of ``box_iou_rotated``; upright boxes use the closed form on the host, rotated ones the library's
``sgc_box_iou_rotated`` on the GPU.  Pinned by tests/golden/indoor_eval.npz, produced by running the reference's own
``indoor_eval`` (tests/golden/make_golden_eval.py).
"""
import numpy as np
import torch


def _rows(boxes):
    """-> float32 [n, 7] (gravity centre, dims, yaw)."""
    if not torch.is_tensor(boxes) and hasattr(boxes, "gravity_center"):
        boxes = torch.cat((boxes.gravity_center, boxes.tensor[:, 3:]), dim=1)
    if not torch.is_tensor(boxes):
        boxes = torch.from_numpy(np.asarray(boxes, dtype=np.float32))
    boxes = boxes.float()
    if boxes.numel() == 0:
        return boxes.new_zeros((0, 7))
    boxes = boxes.reshape(-1, boxes.shape[-1])
    if boxes.shape[1] == 6:
        boxes = torch.cat((boxes, boxes.new_zeros(boxes.shape[0], 1)), dim=1)
    return boxes


def iou3d_pairwise(a, b):
    """[n,7] x [m,7] -> [n,m] 3-D IoU (base_box3d.py:424-487)."""
    a, b = _rows(a), _rows(b)
    n, m = a.shape[0], b.shape[0]
    if n * m == 0:
        return a.new_zeros((n, m))
    top = torch.min((a[:, 2] + a[:, 5] / 2)[:, None], (b[:, 2] + b[:, 5] / 2)[None])
    bot = torch.max((a[:, 2] - a[:, 5] / 2)[:, None], (b[:, 2] - b[:, 5] / 2)[None])
    overlaps_h = (top - bot).clamp(min=0)
    area_a, area_b = (a[:, 3] * a[:, 4])[:, None], (b[:, 3] * b[:, 4])[None]
    if bool((a[:, 6] == 0).all()) and bool((b[:, 6] == 0).all()):
        w = (torch.min((a[:, 0] + a[:, 3] / 2)[:, None], (b[:, 0] + b[:, 3] / 2)[None])
             - torch.max((a[:, 0] - a[:, 3] / 2)[:, None], (b[:, 0] - b[:, 3] / 2)[None])).clamp(min=0)
        h = (torch.min((a[:, 1] + a[:, 4] / 2)[:, None], (b[:, 1] + b[:, 4] / 2)[None])
             - torch.max((a[:, 1] - a[:, 4] / 2)[:, None], (b[:, 1] - b[:, 4] / 2)[None])).clamp(min=0)
        overlaps_bev = w * h
    else:
        from . import ext
        dev = torch.device("cuda", torch.cuda.current_device())
        iou2d = ext.ops().box_iou_rotated(a[:, [0, 1, 3, 4, 6]].contiguous().to(dev),
                                          b[:, [0, 1, 3, 4, 6]].contiguous().to(dev)).to(a.device)
        overlaps_bev = iou2d * (area_a + area_b) / (1 + iou2d)
    overlaps_3d = overlaps_bev * overlaps_h
    vol_a, vol_b = (a[:, 3] * a[:, 4] * a[:, 5])[:, None], (b[:, 3] * b[:, 4] * b[:, 5])[None]
    return overlaps_3d / torch.clamp(vol_a + vol_b - overlaps_3d, min=1e-8)


def average_precision(recalls, precisions, mode="area"):
    """indoor_eval.py:8-53.  1-D inputs, mode 'area' (what ``indoor_eval`` uses): area under the monotone envelope of the
    precision-recall curve, returned as a float32 scalar.  2-D inputs [num_scales, num_dets] give one value per scale as in
    the reference; mode '11points' is restated with the reference's arithmetic, including its ``ap /= 11`` INSIDE the loop
    over scales (scale i is divided num_scales - i times) -- the value its own known-answer test pins
    (tests/test_metrics/test_indoor_eval.py:184-189: 8 / 121 for the first of two scales)."""
    recalls, precisions = np.asarray(recalls), np.asarray(precisions)
    one_d = recalls.ndim == 1
    if one_d:
        recalls, precisions = recalls[None], precisions[None]
    assert recalls.shape == precisions.shape and recalls.ndim == 2
    ap = np.zeros(recalls.shape[0], dtype=np.float32)
    if mode == "area":
        for i in range(recalls.shape[0]):
            mrec = np.concatenate(([0.0], recalls[i], [1.0]))
            mpre = np.concatenate(([0.0], precisions[i], [0.0]))
            mpre = np.maximum.accumulate(mpre[::-1])[::-1]
            step = np.where(mrec[1:] != mrec[:-1])[0]
            ap[i] = np.sum((mrec[step + 1] - mrec[step]) * mpre[step + 1])
    elif mode == "11points":
        for i in range(recalls.shape[0]):
            for thr in np.arange(0, 1 + 1e-3, 0.1):
                precs = precisions[i, recalls[i] >= thr]
                ap[i] += precs.max() if precs.size > 0 else 0
            ap /= 11
    else:
        raise ValueError('Unrecognized mode, only "area" and "11points" are supported')
    return np.float32(ap[0]) if one_d else ap


def eval_class(dets, gts, iou_thrs):
    """One class.  dets: {scene: (boxes [k,7], scores [k])}, gts: {scene: boxes [g,7]} ->
    [(recall, precision, ap)] per threshold (indoor_eval.py:56-161): detections in descending score over all scenes,
    each matched to the ground truth of its scene with the highest IoU; a ground truth counts once per threshold."""
    npos = sum(len(g) for g in gts.values())
    scene_of, score, best_iou, best_gt = [], [], [], []
    for sid, (boxes, scores) in dets.items():
        k = len(scores)
        if k == 0:
            continue
        g = gts.get(sid)
        if g is not None and len(g) > 0:
            iou = iou3d_pairwise(boxes, g).cpu().numpy()
            bi = iou.argmax(axis=1)                          # first maximum, as the reference's `>` scan
            bv = iou[np.arange(k), bi]
        else:                                                # the reference appends a single zero IoU
            bi, bv = np.zeros(k, dtype=np.int64), np.full(k, -np.inf)
        scene_of += [sid] * k
        score += [float(s) for s in scores]
        best_iou += list(bv)
        best_gt += list(bi)
    order = np.argsort(-np.asarray(score, dtype=np.float64))
    out = []
    for thr in iou_thrs:
        taken = {sid: np.zeros(len(g), dtype=bool) for sid, g in gts.items()}
        tp = np.zeros(len(order))
        fp = np.zeros(len(order))
        for rank, d in enumerate(order):
            sid = scene_of[d]
            if best_iou[d] > thr and not taken[sid][best_gt[d]]:
                tp[rank] = 1.0
                taken[sid][best_gt[d]] = True
            else:
                fp[rank] = 1.0
        ctp, cfp = np.cumsum(tp), np.cumsum(fp)
        recall = ctp / float(npos)
        precision = ctp / np.maximum(ctp + cfp, np.finfo(np.float64).eps)
        out.append((recall, precision, average_precision(recall, precision)))
    return out


def indoor_eval(gt_annos, dt_annos, metric, label2cat, logger=None, box_type_3d=None, box_mode_3d=None):
    """``indoor_eval(gt_annos, dt_annos, metric, label2cat)`` (indoor_eval.py:203-309).

    gt_annos[i]: dict(gt_num, gt_boxes_upright_depth [g,6|7] gravity-centre rows, class [g]);
    dt_annos[i]: dict(boxes_3d [k,6|7] rows or an mmdet3d box structure, scores_3d [k], labels_3d [k]).
    ``box_type_3d`` / ``box_mode_3d`` are accepted for signature compatibility (boxes are already in depth
    coordinates).  Returns the reference's result dict; a table is printed through ``logger`` when it is callable."""
    assert len(dt_annos) == len(gt_annos)
    pred, gt = {}, {}
    for sid, (det, ann) in enumerate(zip(dt_annos, gt_annos)):
        labels = np.asarray(det["labels_3d"].cpu() if torch.is_tensor(det["labels_3d"]) else det["labels_3d"]).astype(np.int64)
        boxes, scores = _rows(det["boxes_3d"]).cpu(), torch.as_tensor(det["scores_3d"]).cpu().float()
        for c in np.unique(labels):
            sel = torch.from_numpy(labels == c)
            pred.setdefault(int(c), {})[sid] = (boxes[sel], scores[sel])
            gt.setdefault(int(c), {}).setdefault(sid, boxes.new_zeros((0, 7)))   # the reference registers an empty list
        if ann["gt_num"] != 0:
            gb = _rows(torch.as_tensor(np.asarray(ann["gt_boxes_upright_depth"], dtype=np.float32)))
            gl = np.asarray(ann["class"]).astype(np.int64)
            for c in np.unique(gl):
                gt.setdefault(int(c), {})[sid] = gb[torch.from_numpy(gl == c)]
    # classes in the reference's insertion order: first appearance while walking detections, then ground truths
    order = []
    for sid, (det, ann) in enumerate(zip(dt_annos, gt_annos)):
        for c in np.asarray(det["labels_3d"].cpu() if torch.is_tensor(det["labels_3d"]) else det["labels_3d"]).astype(np.int64):
            if int(c) not in order:
                order.append(int(c))
        if ann["gt_num"] != 0:
            for c in np.asarray(ann["class"]).astype(np.int64):
                if int(c) not in order:
                    order.append(int(c))
    res = {c: eval_class(pred[c], gt[c], metric) if c in pred else None for c in order}
    ret = {}
    for i, thr in enumerate(metric):
        aps, recs = [], []
        for c in order:
            if res[c] is None:                               # ground truth only: AP 0, recall 0
                ap, rec = 0.0, 0.0
            else:
                recall, _, ap = res[c][i]
                rec = float(recall[-1]) if len(recall) else 0.0
            ret[f"{label2cat[c]}_AP_{thr:.2f}"] = float(ap)
            aps.append(float(ap))
        ret[f"mAP_{thr:.2f}"] = float(np.mean(aps))
        for c in order:
            rec = 0.0 if res[c] is None else (float(res[c][i][0][-1]) if len(res[c][i][0]) else 0.0)
            ret[f"{label2cat[c]}_rec_{thr:.2f}"] = rec
            recs.append(rec)
        ret[f"mAR_{thr:.2f}"] = float(np.mean(recs))
    if callable(logger):
        rows = [f"{label2cat[c]:>16s} " + " ".join(f"{ret[f'{label2cat[c]}_AP_{t:.2f}']:.4f} {ret[f'{label2cat[c]}_rec_{t:.2f}']:.4f}" for t in metric)
                for c in order]
        logger("\n".join(rows + ["Overall".rjust(16) + " " + " ".join(f"{ret[f'mAP_{t:.2f}']:.4f} {ret[f'mAR_{t:.2f}']:.4f}" for t in metric)]))
    return ret
