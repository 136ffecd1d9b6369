#!/usr/bin/env python3
"""tests/golden/box_iou_rotated.npz: rotated-box IoU by the REFERENCE's own code.

The DFA3D package under /root/reference vendors mmcv's ``box_iou_rotated_utils.hpp`` (plain C++); oracle/Makefile's
``_ref`` target compiles it with g++ from where it lies (oracle/_ref/libref_box_iou.so, build container only).  This
script evaluates ``single_box_iou_rotated`` (mode 0) on seeded boxes -- overlapping clusters as the ARKit head's NMS
sees them, plus the edge cases: identical boxes, shared edges, containment, tiny and degenerate boxes, large angles --
in fp32 (the arithmetic mmcv's CUDA kernel runs) and in fp64 (a yardstick for the quoted tolerance).

The header sorts the hull points in one of two ways: ``#ifdef __CUDACC__`` an exchange sort -- the branch of the CUDA
kernels the reference actually runs (nms_rotated / box_iou_rotated are GPU ops there) -- otherwise std::sort.  Both are
built (oracle/ref_build/ref_box_iou_rotated.cpp); ``iou`` / ``iou_f64`` come from the CUDA branch, ``iou_cpu_branch``
from the std::sort branch.  They differ on 1 of the 11 236 pairs here, where the std::sort branch is off by 0.026
against the exact polygon clip (its comparator is not a strict weak ordering for near-collinear points) -- recorded,
not used as the yardstick.
Usage:  python tests/golden/make_golden_iou.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle  # noqa: E402


def main():
    rng = np.random.RandomState(20260)
    n = 96
    ctr = rng.uniform(-1.5, 1.5, (n, 2))
    wh = rng.uniform(0.2, 2.0, (n, 2))
    ang = rng.uniform(-np.pi, np.pi, (n, 1))
    a = np.concatenate([ctr, wh, ang], 1)
    b = a[rng.permutation(n)] + np.concatenate([rng.normal(0, 0.15, (n, 2)), rng.normal(0, 0.1, (n, 2)),
                                                rng.normal(0, 0.2, (n, 1))], 1)
    b[:, 2:4] = np.abs(b[:, 2:4]) + 0.05
    special = np.array([
        [0, 0, 1, 1, 0], [0, 0, 1, 1, 0],                 # identical
        [1, 0, 1, 1, 0],                                  # shares an edge with the first
        [0, 0, 0.5, 0.5, 0.7],                            # contained, rotated
        [0, 0, 1, 1, np.pi / 2], [0, 0, 1, 1, np.pi / 4], # same square turned
        [0, 0, 1e-8, 1e-8, 0.1],                          # area below the 1e-14 cut
        [5, 5, 1, 2, 7.0],                                # far away, angle beyond 2 pi
        [0, 0, 3, 0.2, -3.0], [0, 0, 0.2, 3, 0.1],        # thin crossing bars
    ], dtype=np.float64)
    a = np.concatenate([a, special]).astype(np.float32)
    b = np.concatenate([b, special[::-1]]).astype(np.float32)
    iou32 = oracle.ref_box_iou_rotated(a, b, variant="cuda")
    if iou32 is None:
        sys.exit("oracle/_ref is not available (needs /root/reference)")
    iou64 = oracle.ref_box_iou_rotated(a.astype(np.float64), b.astype(np.float64), f64=True, variant="cuda")
    iou_cpu = oracle.ref_box_iou_rotated(a, b, variant="cpu")
    path = os.path.join(HERE, "box_iou_rotated.npz")
    np.savez_compressed(path, a=a, b=b, iou=iou32, iou_f64=iou64, iou_cpu_branch=iou_cpu)
    print(f"wrote {path}: {a.shape[0]} x {b.shape[0]} pairs, {(iou32 > 0).mean():.2f} overlapping, "
          f"max |fp32 - fp64| = {np.abs(iou32 - iou64).max():.2e}; std::sort branch differs on "
          f"{int((np.abs(iou_cpu - iou32) > 1e-5).sum())} pairs (max {np.abs(iou_cpu - iou32).max():.3f})")


if __name__ == "__main__":
    main()
