#!/usr/bin/env python3
"""Generates tests/golden/nms_aligned.npz by running the reference's own ``aligned_3d_nms``
(/root/reference/packages/mmdetection3d/mmdet3d/core/post_processing/box3d_nms.py:131-178, pure torch) on seeded
box sets.  Build-container only (needs /root/reference); its imports of numba / mmcv.ops are stubbed -- the function
under test uses neither.  Nothing of the reference is copied: the committed fixture holds inputs and kept indices."""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/packages/mmdetection3d/mmdet3d/core/post_processing/box3d_nms.py"


def load_reference():
    numba = types.ModuleType("numba")
    numba.jit = lambda *a, **k: (lambda fn: fn)
    sys.modules.setdefault("numba", numba)
    mmcv = sys.modules.setdefault("mmcv", types.ModuleType("mmcv"))
    ops = types.ModuleType("mmcv.ops")
    ops.nms = ops.nms_rotated = None
    mmcv.ops = ops
    sys.modules["mmcv.ops"] = ops
    spec = importlib.util.spec_from_file_location("_ref_box3d_nms", REF)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.aligned_3d_nms


def boxes_case(n, n_cls, seed, cluster):
    g = torch.Generator().manual_seed(seed)
    if cluster:   # detections pile up around a few objects, like a head's top-k candidates
        n_obj = max(1, n // 12)
        ctr = (torch.rand(n_obj, 3, generator=g) - 0.5) * torch.tensor([6.0, 6.0, 2.0])
        size = 0.3 + torch.rand(n_obj, 3, generator=g) * 1.2
        obj = torch.randint(0, n_obj, (n,), generator=g)
        c = ctr[obj] + torch.randn(n, 3, generator=g) * 0.08
        s = size[obj] * (1 + torch.randn(n, 3, generator=g) * 0.1).clamp(0.5, 1.5)
        labels = (obj % n_cls + (torch.rand(n, generator=g) < 0.15).long()) % n_cls
    else:
        c = (torch.rand(n, 3, generator=g) - 0.5) * 4.0
        s = 0.2 + torch.rand(n, 3, generator=g) * 1.5
        labels = torch.randint(0, n_cls, (n,), generator=g)
    boxes = torch.cat([c - s / 2, c + s / 2], 1).float()
    scores = torch.rand(n, generator=g).float()
    return boxes, scores, labels.long()


def main():
    nms = load_reference()
    out = {}
    cases = [(1, 3, 0, False, 0.25), (2, 1, 1, True, 0.25), (37, 4, 2, False, 0.25), (64, 2, 3, True, 0.25),
             (65, 18, 4, True, 0.25), (300, 18, 5, True, 0.25), (1000, 18, 6, True, 0.5), (700, 189, 7, True, 0.25),
             (129, 1, 8, True, 0.1)]
    for k, (n, n_cls, seed, cluster, thr) in enumerate(cases):
        boxes, scores, labels = boxes_case(n, n_cls, seed, cluster)
        if k == 4:      # degenerate boxes: zero volume (0/0 -> NaN IoU against themselves' twins) and inverted corners
            boxes[3, 3:] = boxes[3, :3]
            boxes[7] = boxes[3]
            boxes[11, 3] = boxes[11, 0] - 0.1
        keep = nms(boxes, scores, labels, thr)
        out[f"boxes{k}"], out[f"scores{k}"], out[f"labels{k}"] = boxes.numpy(), scores.numpy(), labels.numpy()
        out[f"thr{k}"], out[f"keep{k}"] = np.float32(thr), keep.numpy().astype(np.int64)
        print(f"case {k}: n={n} classes={n_cls} thr={thr} kept {len(keep)}")
    out["n_cases"] = np.int64(len(cases))
    np.savez_compressed(os.path.join(HERE, "nms_aligned.npz"), **out)


if __name__ == "__main__":
    main()
