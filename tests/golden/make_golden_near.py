#!/usr/bin/env python3
"""Golden for the NEAR-PLANE rule of the visibility mask: ``point_sampling_near.npz``.

``VoxFormerEncoder_DFA3D.point_sampling`` (mmdet3d_plugin/models/im2voxel/transformer_utils/encoder.py:179-223)
takes ``points_d = reference_points_cam[..., 2:3]`` -- a VIEW -- and overwrites that slice in place with the
normalised depth ``(z - d_near) / (d_far - d_near)`` (:211) before ``volume_mask = points_d > eps`` (:213) runs.  The
depth test therefore drops every point closer than d_near (+ eps * range), not only points behind the camera.
``point_sampling.npz`` (inward-looking ring at 2.2 m) has no in-image voxel in the 0 .. 0.2 m band, so it cannot
tell the two rules apart; this fixture puts the cameras INSIDE the voxel grid, looking along rows of voxel centres.

Runs only in the build container (imports the reference's own class through make_golden.install_stubs).
Usage:  python tests/golden/make_golden_near.py
"""
import importlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402


def look_at(eye, target):
    """World -> camera 4x4 (x right, y down, z forward), float32."""
    eye, target = np.asarray(eye, np.float64), np.asarray(target, np.float64)
    fwd = target - eye
    fwd /= np.linalg.norm(fwd)
    up = np.array([0.0, 0.0, 1.0])
    right = np.cross(fwd, up)
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    R = np.stack([right, down, fwd])
    E = np.eye(4)
    E[:3, :3] = R
    E[:3, 3] = -R @ eye
    return E.astype(np.float32)


def main():
    ml = mg.install_stubs()
    tu = "mmdet3d_plugin.models.im2voxel.transformer_utils."
    importlib.import_module(tu + "encoder")
    importlib.import_module(tu + "transformer")
    importlib.import_module("mmdet3d_plugin.models.im2voxel.DenseHead")
    C = 32
    grid, size = (16, 16, 8), (.16, .16, .2)
    xf = mg.voxel_head_cfg(C, [grid], [size], [])["base_head_configs"][0]
    dh = ml.build_head(xf).eval()
    origin = np.array([0.0, 0.0, 0.5], dtype=np.float32)
    # cameras a few centimetres behind voxel corners inside the grid, looking along +x / +y / a diagonal / down:
    # whole rows of voxels project into the image at depths 0.01 .. 0.2 m (kept by `z > eps`, dropped by `zn > eps`)
    eyes = [(-0.335, 0.005, 0.505), (0.005, -0.49, 0.305), (0.163, 0.162, 0.83), (-0.70, -0.70, 0.31)]
    targets = [(2.0, 0.02, 0.5), (0.0, 2.0, 0.32), (0.165, 0.16, -1.0), (1.0, 1.0, 0.45)]
    ext = [look_at(e, t) for e, t in zip(eyes, targets)]
    K = np.eye(4, dtype=np.float32)
    K[:3, :3] = np.array([[70.0, 0, 160.0], [0, 70.0, 120.0], [0, 0, 1]], dtype=np.float32)
    # wide-angle intrinsics (f = 17 px after the 240 -> 59 resize) so that off-axis voxels 0.05 .. 0.2 m away stay in the image
    meta = dict(img_shape=(59, 80, 3), ori_shape=(240, 320, 3),
                lidar2img=dict(extrinsic=ext, intrinsic=K, origin=origin))
    enc = dh.cross_transformer.encoder
    with torch.no_grad():
        ref_cam, mask = enc.point_sampling(dh.ref_3d[None, None], img_meta=meta)
    # how many in-image points sit in the band the two rules disagree on (for the test's sanity assert)
    pts = (dh.ref_3d + torch.from_numpy(origin)).numpy()
    band = 0
    for i, E in enumerate(ext):
        cam = (E[:3, :3] @ pts.T).T + E[:3, 3]
        z = cam[:, 2]
        u, v = ref_cam[i, 0, :, 0, 0].numpy(), ref_cam[i, 0, :, 0, 1].numpy()
        inside = (u > 1e-5) & (u < 1 - 1e-5) & (v > 1e-5) & (v < 1 - 1e-5)
        band += int(((z > 1e-5) & (z <= 0.2) & inside).sum())
    print(f"in-image points with 0 < z <= d_near: {band}; visible: {int(mask.sum())} of {mask.numel()}")
    assert band >= 8, "fixture must exercise the near-plane rule"
    mg.save("point_sampling_near", ref_3d=dh.ref_3d, ref_cam=ref_cam, mask=mask.to(torch.uint8),
            dbound=np.array([0.2, 5.0]), n_band=np.array(band), **mg.meta_arrays(meta))


if __name__ == "__main__":
    main()
