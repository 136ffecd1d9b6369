"""Checks of the rotated BEV NMS shared by the CPU (oracle) and GPU (HIP library) test modules: the same
assertions run against either TensorOps."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "nms_rotated_multiclass.npz")


def check_multiclass_golden(ops, device):
    """tests/golden/nms_rotated_multiclass.npz: outputs of the reference's own box3d_multiclass_nms / nms_bev called
    as SunRgbdImVoxelHeadV2._nms does (make_golden_nms_rotated.py; mmcv's compiled nms_rotated substituted by an
    exact float64 greedy NMS, no IoU within 1e-3 of the threshold)."""
    from sgcdet_amd.plugin.bbox_head import box3d_multiclass_nms_rotated
    d = np.load(GOLDEN)
    n_cases = len([k for k in d.files if k.endswith("_cfg")])
    assert n_cases >= 5
    for c in range(n_cases):
        boxes = torch.from_numpy(d[f"c{c}_boxes"]).to(device)
        scores = torch.from_numpy(d[f"c{c}_scores"]).to(device)
        score_thr, nms_thr, max_num = d[f"c{c}_cfg"]
        ob, os_, ol = box3d_multiclass_nms_rotated(ops, boxes, scores, float(score_thr), int(max_num), float(nms_thr))
        assert torch.equal(ol.cpu(), torch.from_numpy(d[f"c{c}_out_labels"])), c
        assert torch.equal(os_.cpu(), torch.from_numpy(d[f"c{c}_out_scores"])), c
        assert torch.equal(ob.cpu(), torch.from_numpy(d[f"c{c}_out_boxes"])), c
        assert ol.dtype == torch.int64 and ob.shape[1] == 7


def check_iou_against_float64_clip(ops, device):
    """the fp32 restatement of mmcv's intersection-points + Graham-scan IoU against an independent float64
    Sutherland-Hodgman clip of the same rectangles (identical, contained, far, axis-aligned pairs included)"""
    d = np.load(GOLDEN)
    a, b = torch.from_numpy(d["iou_a"]).to(device), torch.from_numpy(d["iou_b"]).to(device)
    got = ops.box_iou_rotated(a, b).cpu().double().numpy()
    want = d["iou_f64"]
    assert got.shape == want.shape == (40, 40)
    assert np.abs(got - want).max() < 2e-5, np.abs(got - want).max()
    assert (want > 0.05).sum() > 100 and (want == 0).sum() > 100           # the table is not trivial
    assert np.abs(np.diag(got)[:5] - 1.0).max() < 1e-5                     # identical boxes
    assert np.all(got[8 * 0 + np.arange(40), 8] == 0.0)                    # column 8 is 50 m away: exact zeros


def arkit_like(n, n_cls, seed):
    """the ARKit head's candidates: 3 x nms_pre boxes around a handful of objects, every class scored, rows the
    valid mask zeroed"""
    g = torch.Generator().manual_seed(seed)
    n_obj = max(2, n // 60)
    ctr = (torch.rand(n_obj, 3, generator=g) - 0.5) * torch.tensor([6.0, 6.0, 2.0])
    size = 0.3 + torch.rand(n_obj, 3, generator=g) * 1.5
    yaw = (torch.rand(n_obj, generator=g) - 0.5) * 6.3
    obj = torch.randint(0, n_obj, (n,), generator=g)
    c = ctr[obj] + torch.randn(n, 3, generator=g) * 0.12
    s = size[obj] * (1.0 + torch.randn(n, 3, generator=g) * 0.1).clamp(0.5, 1.5)
    boxes = torch.cat([c, s, (yaw[obj] + torch.randn(n, generator=g) * 0.2)[:, None]], dim=1).float()
    scores = (torch.rand(n, n_cls, generator=g) ** 2).float() * (torch.rand(n, 1, generator=g) < 0.8)
    return boxes.contiguous(), scores.contiguous()


def bev_of(boxes):
    return torch.stack((boxes[:, 0] - boxes[:, 3] / 2, boxes[:, 1] - boxes[:, 4] / 2,
                        boxes[:, 0] + boxes[:, 3] / 2, boxes[:, 1] + boxes[:, 4] / 2, boxes[:, 6]), dim=1).contiguous()


def greedy_from_iou(ops, bev, scores_c, cand, thr):
    """plain greedy NMS of one class driven by the ops' own pairwise IoU -- the mask/sweep decomposition must
    give the same survivors as the textbook loop"""
    idx = torch.nonzero(cand)[:, 0]
    order = idx[scores_c[idx].sort(descending=True, stable=True)[1]]
    b = bev[order]
    xywhr = torch.stack(((b[:, 0] + b[:, 2]) / 2, (b[:, 1] + b[:, 3]) / 2, b[:, 2] - b[:, 0], b[:, 3] - b[:, 1], b[:, 4]), 1)
    iou = ops.box_iou_rotated(xywhr.contiguous(), xywhr.contiguous()).cpu()
    removed = torch.zeros(len(order), dtype=torch.bool)
    keep = []
    for p in range(len(order)):
        if removed[p]:
            continue
        keep.append(int(order[p]))
        removed[p + 1:] |= iou[p, p + 1:] > thr
    return keep


def check_mask_sweep_equals_textbook_loop(ops, device, n=300, n_cls=4):
    boxes, scores = arkit_like(n, n_cls, seed=5)
    boxes, scores = boxes.to(device), scores.to(device)
    bev = bev_of(boxes)
    keep, n_keep = ops.nms_rotated_bev(bev, scores, 0.0, 0.15)
    for c in range(n_cls):
        want = greedy_from_iou(ops, bev, scores[:, c], scores[:, c] > 0.0, 0.15)
        assert keep[c, :int(n_keep[c])].tolist() == want, c
        assert 0 < len(want) < int((scores[:, c] > 0).sum())
