#!/usr/bin/env python3
"""Generates tests/golden/head_targets.npz: outputs of the reference's own ``get_targets``
(/root/reference/mmdet3d_plugin/models/dense_heads/imvoxel_head_v2.py:361-435 ScanNetImVoxelHeadV2, :485-561
SunRgbdImVoxelHeadV2, ``get_points`` :237-247) on seeded ground-truth boxes, and of the vendored
``axis_aligned_bbox_overlaps_3d`` (packages/mmdetection3d/mmdet3d/core/bbox/iou_calculators/iou3d_calculator.py)
that AxisAlignedIoULoss thresholds.  Build-container only; the stub machinery of make_golden.py is reused.  The
ground-truth container is a 10-line stand-in for mmdet3d's DepthInstance3DBoxes exposing what get_targets reads
(``volume``, ``gravity_center``, ``tensor``, ``device``, ``len``) with mmdet3d's definitions (bottom-centre z +
half height, w*l*h).  Nothing of the reference is copied: the fixture holds inputs and outputs."""
import importlib
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402


class Boxes:
    """tensor [n,7] = (x, y, z_bottom, dx, dy, dz, yaw) as DepthInstance3DBoxes stores them"""

    def __init__(self, tensor):
        self.tensor = tensor

    def __len__(self):
        return self.tensor.shape[0]

    @property
    def device(self):
        return self.tensor.device

    @property
    def volume(self):
        return self.tensor[:, 3] * self.tensor[:, 4] * self.tensor[:, 5]

    @property
    def gravity_center(self):
        bc = self.tensor[:, :3]
        gc = torch.zeros_like(bc)
        gc[:, :2] = bc[:, :2]
        gc[:, 2] = bc[:, 2] + self.tensor[:, 5] * 0.5
        return gc


def scene_boxes(n, seed, with_yaw):
    g = torch.Generator().manual_seed(seed)
    ctr = (torch.rand(n, 3, generator=g) - 0.5) * torch.tensor([5.0, 5.0, 1.0]) + torch.tensor([0.0, 0.0, 0.3])
    size = 0.25 + torch.rand(n, 3, generator=g) ** 2 * torch.tensor([2.5, 2.5, 1.6])
    if n >= 3:
        size[0] = torch.tensor([0.12, 0.12, 0.12])         # smaller than a voxel: may own no point at all
        ctr[1], size[1] = ctr[2] + 0.05, size[2] * 0.6       # nested boxes: minimal volume wins
    yaw = (torch.rand(n, 1, generator=g) - 0.5) * 6.0 if with_yaw else torch.zeros(n, 1)
    return torch.cat([ctr, size, yaw], dim=1).float()


def main():
    ml = mg.install_stubs()
    head_mod = importlib.import_module("mmdet3d_plugin.models.dense_heads.imvoxel_head_v2")
    out = {}
    sizes = [(40, 40, 16), (20, 20, 8), (10, 10, 4)]
    origin = (0.0, 0.0, 0.5)
    for tag, cls_name, n_reg, with_yaw in (("scannet", "ScanNetImVoxelHeadV2", 6, False), ("sunrgbd", "SunRgbdImVoxelHeadV2", 7, True)):
        bh = getattr(head_mod, cls_name)(n_classes=18, n_channels=8, n_reg_outs=n_reg, n_scales=3, limit=27,
                                         centerness_topk=18)
        bh.voxel_size = (.16, .16, .2)
        points = bh.get_points(sizes, origin, torch.device("cpu"))
        for case, (n, seed) in enumerate(((14, 3), (1, 4), (40, 5))):
            b = scene_boxes(n, seed, with_yaw)
            gc = b.clone()
            stored = b.clone()
            stored[:, 2] = gc[:, 2] - gc[:, 5] / 2                      # bottom-centre storage
            g = torch.Generator().manual_seed(seed + 100)
            labels = torch.randint(0, 18, (n,), generator=g)
            ct, bt, lb, occ = bh.get_targets(points, Boxes(stored), labels)
            k = f"{tag}{case}_"
            out[k + "boxes_gravity"] = torch.cat((Boxes(stored).gravity_center, stored[:, 3:]), 1).numpy()
            out[k + "gt_labels"] = labels.numpy()
            out[k + "centerness"], out[k + "bbox"], out[k + "labels"], out[k + "occ"] = ct.numpy(), bt.numpy(), lb.numpy(), occ.numpy()
            print(tag, case, "boxes", n, "positives", int((lb >= 0).sum()), "inside any", int(occ.sum()))
    out["points"] = torch.cat(points).numpy()
    out["scales"] = torch.cat([torch.full((len(p),), i, dtype=torch.int32) for i, p in enumerate(points)]).numpy()
    out["cfg"] = np.array([3, 27, 18])
    # vendored axis-aligned IoU (is_aligned=True), the quantity of AxisAlignedIoULoss
    spec = importlib.util.spec_from_file_location(
        "_ref_iou3d", os.path.join(mg.REF, "packages/mmdetection3d/mmdet3d/core/bbox/iou_calculators/iou3d_calculator.py"))
    src = open(spec.origin).read()
    mod = {}
    start = src.index("def axis_aligned_bbox_overlaps_3d")
    exec(compile("import torch\n" + src[start:], spec.origin, "exec"), mod)           # the function only (no mmdet imports)
    g = torch.Generator().manual_seed(77)
    c = (torch.rand(64, 3, generator=g) - 0.5) * 2
    s = 0.2 + torch.rand(64, 3, generator=g)
    a = torch.cat([c - s / 2, c + s / 2], 1)
    c2 = c + torch.randn(64, 3, generator=g) * 0.3
    s2 = s * (0.6 + torch.rand(64, 3, generator=g))
    b2 = torch.cat([c2 - s2 / 2, c2 + s2 / 2], 1)
    out["iou_a"], out["iou_b"] = a.numpy(), b2.numpy()
    out["iou_aligned"] = mod["axis_aligned_bbox_overlaps_3d"](a, b2, is_aligned=True).numpy()
    np.savez_compressed(os.path.join(HERE, "head_targets.npz"), **out)


if __name__ == "__main__":
    if not os.path.isdir(mg.REF):
        sys.exit("needs /root/reference (build container only)")
    torch.set_num_threads(1)
    main()
