"""Checks of the head's target assignment shared by the CPU (oracle) and GPU (HIP library) test modules."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "head_targets.npz")


def check_targets_golden(ops, device):
    """tests/golden/head_targets.npz: outputs of the reference's own get_targets (make_golden_targets.py) for both
    heads: labels, occupancy and the assigned boxes bit-exact everywhere; centerness to one ulp on the axis-aligned
    head (see below), 1e-5 on the rotated one (its rotation is an einsum in the reference)."""
    d = np.load(GOLDEN)
    pts = torch.from_numpy(d["points"]).to(device).contiguous()
    scales = torch.from_numpy(d["scales"]).to(device).contiguous()
    n_scales, limit, topk = (int(v) for v in d["cfg"])
    for tag, rotated in (("scannet", False), ("sunrgbd", True)):
        for case in range(3):
            k = f"{tag}{case}_"
            boxes = torch.from_numpy(d[k + "boxes_gravity"]).to(device).contiguous()
            gl = torch.from_numpy(d[k + "gt_labels"]).to(device)
            ct, bt, lb, occ = ops.assign_targets(pts, scales, boxes, gl, rotated, n_scales, limit, topk)
            want_l = torch.from_numpy(d[k + "labels"])
            assert torch.equal(occ.cpu(), torch.from_numpy(d[k + "occ"])), k
            assert torch.equal(lb.cpu(), want_l), (k, int((lb.cpu() != want_l).sum()))
            pos = want_l >= 0
            assert int(pos.sum()) > 0
            want_c, want_b = torch.from_numpy(d[k + "centerness"]), torch.from_numpy(d[k + "bbox"])
            if rotated:
                assert torch.equal(bt.cpu(), want_b), k                                  # a copy of the assigned gt row
                assert (ct.cpu()[pos] - want_c[pos]).abs().max() < 1e-5, k
            else:
                # centerness to one ulp: the fixture was made with torch's CPU sqrt, which in this build (AVX-512
                # path) is not correctly rounded -- e.g. sqrt(0.010009498) comes out one ulp below sqrtf's and
                # CUDA's result, in about 1 % of the positives
                assert torch.isclose(ct.cpu()[pos], want_c[pos], rtol=2.4e-7, atol=0).all(), k
                assert (ct.cpu()[pos] != want_c[pos]).float().mean() < 0.05, k
                assert torch.equal(bt.cpu()[pos], want_b[pos]), k
                # background points too: the reference reads box 0 for them (argmin of an all-1e8 row)
                same = torch.isclose(ct.cpu(), want_c, rtol=2.4e-7, atol=0, equal_nan=True)
                assert bool(same.all()), (k, int((~same).sum()))
                assert torch.equal(bt.cpu(), want_b), k


def random_boxes(n, seed, with_yaw):
    g = torch.Generator().manual_seed(seed)
    ctr = (torch.rand(n, 3, generator=g) - 0.5) * torch.tensor([5.6, 5.6, 2.0]) + torch.tensor([0.0, 0.0, 0.5])
    size = 0.2 + torch.rand(n, 3, generator=g) ** 2 * torch.tensor([2.6, 2.6, 1.8])
    yaw = (torch.rand(n, 1, generator=g) - 0.5) * 6.2 if with_yaw else torch.zeros(n, 1)
    return torch.cat([ctr, size, yaw], 1).float().contiguous(), torch.randint(0, 18, (n,), generator=g)


def check_against_oracle(ops, oracle_ops, device):
    """the config-2 point set (29 200 points, 3 scales), 1 / 7 / 64 boxes, both heads: labels, occupancy and box
    targets identical to the oracle, centerness bit-exact on the axis-aligned head"""
    d = np.load(GOLDEN)
    pts, scales = torch.from_numpy(d["points"]).contiguous(), torch.from_numpy(d["scales"]).contiguous()
    for rotated in (False, True):
        for n, seed in ((1, 1), (7, 2), (64, 3)):
            boxes, gl = random_boxes(n, seed, rotated)
            want = oracle_ops.assign_targets(pts, scales, boxes, gl, rotated, 3, 27, 18)
            got = ops.assign_targets(pts.to(device), scales.to(device), boxes.to(device), gl.to(device), rotated, 3, 27, 18)
            ct, bt, lb, occ = (t.cpu() for t in got)
            if rotated:      # sinf / cosf of the device vs libm: a point within 1e-6 of a face may flip
                assert (lb != want[2]).sum() <= 2 and (occ != want[3]).sum() <= 2, (n, int((lb != want[2]).sum()))
                same = (lb == want[2]) & (lb >= 0)           # background rows carry box 0's (possibly NaN) value
                assert (ct[same] - want[0][same]).abs().max() < 1e-5
                assert torch.equal(bt[same], want[1][same])
            else:
                assert torch.equal(lb, want[2]) and torch.equal(occ, want[3]), n
                assert torch.equal(bt, want[1]), n
                assert torch.equal(ct.view(torch.int32), want[0].view(torch.int32)), n       # NaNs of background points included
            assert n == 1 or int((lb >= 0).sum()) > 20
