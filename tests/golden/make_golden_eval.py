#!/usr/bin/env python3
"""Generates tests/golden/indoor_eval.npz by running the reference's own ``indoor_eval``
(/root/reference/packages/mmdetection3d/mmdet3d/core/evaluation/indoor_eval.py:203-309, with ``eval_det_cls`` /
``eval_map_recall`` / ``average_precision`` above it) on seeded detections and ground truths.  Build-container only.
mmdet3d's box structures (and the mmcv ``box_iou_rotated`` op under their ``overlaps``) are not importable here, so the
generator hands the reference a small stand-in box class with the members indoor_eval touches (``convert_to``,
``__getitem__``, ``tensor``, ``new_box``, ``__len__``, ``overlaps``) whose ``overlaps`` evaluates mmdet3d's formula
(base_box3d.py:424-487: height overlap x BEV overlap) with an exact float64 polygon clip for the BEV IoU (from
make_golden_nms_rotated.py).  ``mmcv.utils.print_log`` and ``terminaltables.AsciiTable`` are stubbed.  The fixture
pins the matching / AP / recall / result-key logic; nothing of the reference is copied."""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden_nms_rotated import iou_rotated_f64  # noqa: E402

REF = "/root/reference/packages/mmdetection3d/mmdet3d/core/evaluation/indoor_eval.py"


class Box:
    """rows (x, y, z_bottom, dx, dy, dz, yaw), the storage of mmdet3d's DepthInstance3DBoxes"""

    def __init__(self, tensor, box_dim=7, origin=(0.5, 0.5, 0)):
        t = torch.as_tensor(np.asarray(tensor, dtype=np.float32)) if not torch.is_tensor(tensor) else tensor.clone().float()
        if t.numel() == 0:
            t = t.reshape(0, 7)
        if t.dim() == 1:
            t = t[None]
        if t.shape[1] == 6:
            t = torch.cat((t, t.new_zeros(t.shape[0], 1)), 1)
        if tuple(origin) != (0.5, 0.5, 0):
            t = t.clone()
            t[:, :3] += t[:, 3:6] * (t.new_tensor((0.5, 0.5, 0)) - t.new_tensor(origin))
        self.tensor = t

    def convert_to(self, mode):
        return self

    def __len__(self):
        return self.tensor.shape[0]

    def __getitem__(self, i):
        b = Box.__new__(Box)
        b.tensor = self.tensor[i]
        return b

    def new_box(self, data):
        b = Box.__new__(Box)
        b.tensor = data
        return b

    @classmethod
    def overlaps(cls, b1, b2):
        a, b = b1.tensor.double().numpy(), b2.tensor.double().numpy()
        out = np.zeros((len(a), len(b)))
        for i, p in enumerate(a):
            for j, q in enumerate(b):
                h = max(min(p[2] + p[5], q[2] + q[5]) - max(p[2], q[2]), 0.0)
                iou2d = iou_rotated_f64(p[[0, 1, 3, 4, 6]], q[[0, 1, 3, 4, 6]])
                bev = iou2d * (p[3] * p[4] + q[3] * q[4]) / (1 + iou2d)
                o3 = bev * h
                out[i, j] = o3 / max(p[3] * p[4] * p[5] + q[3] * q[4] * q[5] - o3, 1e-8)
        return torch.from_numpy(out).float()


def load_reference():
    mmcv = sys.modules.setdefault("mmcv", types.ModuleType("mmcv"))
    utils = types.ModuleType("mmcv.utils")
    utils.print_log = lambda *a, **k: None
    mmcv.utils = utils
    sys.modules["mmcv.utils"] = utils
    tt = types.ModuleType("terminaltables")

    class AsciiTable:
        def __init__(self, data):
            self.table = ""
    tt.AsciiTable = AsciiTable
    sys.modules["terminaltables"] = tt
    spec = importlib.util.spec_from_file_location("_ref_indoor_eval", REF)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.indoor_eval


def scenes(n_scenes, n_cls, seed, with_yaw):
    g = torch.Generator().manual_seed(seed)
    gts, dts = [], []
    for s in range(n_scenes):
        n_gt = int(torch.randint(0 if s == 1 else 2, 9, (1,), generator=g))
        ctr = (torch.rand(n_gt, 3, generator=g) - 0.5) * torch.tensor([5.0, 5.0, 1.2]) + torch.tensor([0, 0, 0.8])
        size = 0.3 + torch.rand(n_gt, 3, generator=g) * 1.3
        yaw = (torch.rand(n_gt, 1, generator=g) - 0.5) * 3.0 if with_yaw else torch.zeros(n_gt, 0)
        gt_boxes = torch.cat([ctr, size, yaw], 1).float()
        gt_cls = torch.randint(0, n_cls, (n_gt,), generator=g)
        gts.append(dict(gt_num=n_gt, gt_boxes_upright_depth=gt_boxes.numpy(), **{"class": gt_cls.numpy()}))
        # detections: jittered copies of the ground truths (some duplicated, some with the wrong class) + clutter
        rep = torch.randint(0, 3, (n_gt,), generator=g)
        idx = torch.repeat_interleave(torch.arange(n_gt), rep)
        d = gt_boxes[idx].clone()
        d[:, :3] += torch.randn(len(idx), 3, generator=g) * 0.12
        d[:, 3:6] *= (1 + torch.randn(len(idx), 3, generator=g) * 0.12).clamp(0.6, 1.4)
        if with_yaw:
            d[:, 6] += torch.randn(len(idx), generator=g) * 0.15
        dl = gt_cls[idx].clone()
        flip = torch.rand(len(idx), generator=g) < 0.15
        dl[flip] = torch.randint(0, n_cls, (int(flip.sum()),), generator=g)
        n_cl = int(torch.randint(0, 6, (1,), generator=g))
        cl = torch.cat([(torch.rand(n_cl, 3, generator=g) - 0.5) * 5, 0.3 + torch.rand(n_cl, 3, generator=g),
                        torch.zeros(n_cl, gt_boxes.shape[1] - 6)], 1).float()
        boxes = torch.cat([d, cl])
        labels = torch.cat([dl, torch.randint(0, n_cls, (n_cl,), generator=g)])
        scores = torch.rand(len(boxes), generator=g)
        dts.append(dict(boxes_gravity=boxes, labels_3d=labels, scores_3d=scores))
    # every class that is predicted has a ground truth somewhere (otherwise the reference divides 0 by 0)
    have = set(int(c) for a in gts for c in a["class"])
    for dt in dts:
        keep = torch.tensor([int(l) in have for l in dt["labels_3d"]], dtype=torch.bool)
        for k in ("boxes_gravity", "labels_3d", "scores_3d"):
            dt[k] = dt[k][keep]
    return gts, dts


def main():
    indoor_eval = load_reference()
    out = {}
    metric = [0.25, 0.5]
    for ci, (n_scenes, n_cls, seed, with_yaw) in enumerate(((6, 5, 1, False), (5, 4, 2, True), (3, 18, 3, False))):
        gts, dts = scenes(n_scenes, n_cls, seed, with_yaw)
        label2cat = {i: f"c{i}" for i in range(n_cls)}
        ref_dts = [dict(boxes_3d=Box(d["boxes_gravity"], origin=(0.5, 0.5, 0.5)), labels_3d=d["labels_3d"], scores_3d=d["scores_3d"])
                   for d in dts]
        res = indoor_eval(gts, ref_dts, metric, label2cat, box_type_3d=Box, box_mode_3d=None)
        k = f"case{ci}_"
        out[k + "n_scenes"] = np.array(n_scenes)
        out[k + "n_cls"] = np.array(n_cls)
        for s in range(n_scenes):
            out[k + f"gt_boxes{s}"] = gts[s]["gt_boxes_upright_depth"]
            out[k + f"gt_cls{s}"] = gts[s]["class"]
            out[k + f"dt_boxes{s}"] = dts[s]["boxes_gravity"].numpy()
            out[k + f"dt_labels{s}"] = dts[s]["labels_3d"].numpy()
            out[k + f"dt_scores{s}"] = dts[s]["scores_3d"].numpy()
        out[k + "keys"] = np.array(list(res.keys()))
        out[k + "values"] = np.array([res[x] for x in res.keys()], dtype=np.float64)
        print(ci, {x: round(res[x], 4) for x in res if x.startswith("mA")})
    out["metric"] = np.array(metric)
    np.savez_compressed(os.path.join(HERE, "indoor_eval.npz"), **out)


if __name__ == "__main__":
    main()
