"""Plane-sweep cost volume of ``DepthNet_Fusion``: the part of the depth head that
is the same kernel family as the hot path.

Reference: mmdet3d_plugin/models/im2voxel/depth_utils/depth_est_fusion.py -- ``get_closest_frame_ids`` (:53-64),
``collect_proj`` (:67-84), ``homo_warping`` (:87-126) and the cost-volume loop of ``DepthNet_Fusion.forward``
(:203-240).  The 2-D CNNs around it (``ResNetFPN``, ``SimpleUnet2D``) are dense library convolutions and are not
built here.  ``plane_sweep_correlation`` takes what ``forward`` has at :222 (matching features + ``img_meta``) and
returns what it has at :240 (``correlation``), computed by ``sgc_plane_sweep_corr`` without materialising the warped
neighbour features [N, C, D, H, W].
"""
import numpy as np
import torch

from .. import ext


def closest_frame_ids(num_cams, num_select):
    """Time-adjacent neighbour views, K/2 past and K/2 future, shifted inwards at both ends of the sequence
    (``get_closest_frame_ids``, :53-64)."""
    assert num_select % 2 == 0
    half = num_select // 2
    main = torch.arange(num_cams).unsqueeze(1)
    offsets = torch.cat([torch.arange(-half, 0), torch.arange(1, half + 1)]).unsqueeze(0)
    ids = main + offsets
    ids[0:half, :] = ids[0:half, :] + half + 1
    ids[num_cams - half:num_cams, :] = ids[num_cams - half:num_cams, :] - half - 1
    return ids


def relative_projections(w2c, intrinsic, neighbor_ids):
    """[N, K, 3, 4] rows of ``nei_proj @ inverse(ref_proj)`` with proj = K' @ w2c (``collect_proj`` :67-84 and
    ``homo_warping`` :97-99); torch's own matmul / inverse, as in the reference."""
    if intrinsic.dim() == 2:
        intrinsic = intrinsic.unsqueeze(0).repeat(w2c.shape[0], 1, 1)
    proj = torch.matmul(intrinsic, w2c)
    inv_ref = torch.inverse(proj)
    rel = [torch.matmul(proj[neighbor_ids[:, k]], inv_ref)[:, :3, :4] for k in range(neighbor_ids.shape[1])]
    return torch.stack(rel, 1)


def plane_sweep_correlation(f_mvs, img_meta, stride, depth_values, neighbor_img_num=2):
    """f_mvs [N, C, H, W] matching features of the N views (any memory format); ``stride`` = image / feature
    resolution ratio used for the intrinsics (:209-213); depth_values [D] plane depths (:179).
    Returns correlation [N, D, H, W] (:240)."""
    N, C, H, W = f_mvs.shape
    dev = f_mvs.device
    ops = ext.ops() if f_mvs.is_cuda else None
    if ops is None:
        raise RuntimeError("plane_sweep_correlation: the product path runs on the GPU (no CPU fallback)")
    w2c = torch.tensor(np.array(img_meta["lidar2img"]["extrinsic"]), dtype=torch.float32)
    intr = torch.tensor(np.array(img_meta["lidar2img"]["intrinsic"]), dtype=torch.float32).clone()
    ratio = img_meta["ori_shape"][0] / (img_meta["img_shape"][0] / stride)
    if intr.dim() == 2:
        intr[:2] /= ratio
    else:
        intr[:, :2] /= ratio
    k = min(neighbor_img_num, N - 1)
    nbr = closest_frame_ids(N, k)
    rt = relative_projections(w2c, intr, nbr).reshape(N, k, 12).contiguous().to(dev)
    # channels-last rows: zero-copy when the extractor already writes channels-last, one transpose launch otherwise
    if f_mvs.is_contiguous(memory_format=torch.channels_last) and f_mvs.dtype == torch.float32:
        rows = f_mvs.permute(0, 2, 3, 1).reshape(N, H * W, C)
    else:
        rows = ops.nchw_to_nhwc_crop(f_mvs.float().contiguous(), H, W)
    depth = torch.as_tensor(depth_values, dtype=torch.float32).to(dev).contiguous()
    return ops.plane_sweep_corr(rows, nbr.to(torch.int32).to(dev).contiguous(), rt, depth, H, W)
