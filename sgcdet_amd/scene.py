"""Seeded synthetic scenes shaped like what the SGCDet 2D stage hands to the hot path.

There is no dataset or checkpoint in the build/bench environment, so timing and parity use
synthetic inputs of the real shapes: an inward-looking camera ring
with ScanNet- or ARKit-like intrinsics, random FPN maps and softmax depth distributions.
``img_meta`` follows the reference's schema
(mmdet3d_plugin/datasets/scannet_multiview_dataset.py:33-38, arkit_dataset.py:39-43):

    img_meta = {'img_shape': (h, w, 3), 'ori_shape': (H, W, 3),
                'lidar2img': {'extrinsic': [N x 4x4 float32 world->camera],
                              'intrinsic': 4x4 float32, 'origin': float32[3]}}
"""
import math

import numpy as np
import torch

SCANNET = dict(K=[[1169.6, 0.0, 646.3], [0.0, 1167.1, 489.9], [0.0, 0.0, 1.0]],
               ori_shape=(968, 1296, 3), img_shape=(239, 320, 3), pad_shape=(240, 320), origin="fixed")
ARKIT = dict(K=[[212.0, 0.0, 128.0], [0.0, 212.0, 96.0], [0.0, 0.0, 1.0]],
             ori_shape=(192, 256, 3), img_shape=(240, 320, 3), pad_shape=(240, 320), origin="mean_cam")


def camera_ring(n_views, rng, radius=2.2, height=1.4):
    """World->camera 4x4 matrices (x right, y down, z forward) of cameras on a ring looking
    at a jittered point near the scene centre; returns (extrinsics list, camera positions)."""
    ext, pos = [], []
    for i in range(n_views):
        a = 2.0 * math.pi * i / n_views + rng.uniform(-0.05, 0.05)
        p = np.array([radius * math.cos(a), radius * math.sin(a), height])
        target = np.array([0.0, 0.0, 0.6]) + rng.uniform(-0.3, 0.3, size=3)
        fwd = target - p
        fwd /= np.linalg.norm(fwd)
        right = np.cross(fwd, np.array([0.0, 0.0, 1.0]))
        right /= np.linalg.norm(right)
        down = np.cross(fwd, right)
        R = np.stack([right, down, fwd])
        E = np.eye(4)
        E[:3, :3] = R
        E[:3, 3] = -R @ p
        ext.append(E.astype(np.float32))
        pos.append(p)
    return ext, np.stack(pos)


def make_img_meta(n_views, kind="scannet", seed=0, img_hw=None):
    """``img_hw``: override the resized image size (the unpadded ``img_shape``); BASELINE.json's north star quotes
    256x320 inputs where the reference's ScanNet config resizes to 239x320 (padded 240x320)."""
    spec = SCANNET if kind == "scannet" else ARKIT
    if img_hw is not None:
        spec = dict(spec, img_shape=(int(img_hw[0]), int(img_hw[1]), 3))
    rng = np.random.RandomState(seed)
    ext, pos = camera_ring(n_views, rng)
    K = np.eye(4, dtype=np.float32)
    K[:3, :3] = np.array(spec["K"], dtype=np.float32)
    if spec["origin"] == "fixed":
        origin = np.array([0.0, 0.0, 0.5], dtype=np.float32)
    else:
        origin = pos.mean(0).astype(np.float32)
        origin[2] = 0.5
    return dict(img_shape=spec["img_shape"], ori_shape=spec["ori_shape"],
                lidar2img=dict(extrinsic=ext, intrinsic=K, origin=origin))


def make_scene(n_views, channels, kind="scannet", n_depth=12, seed=0, device="cpu", n_levels=3,
               dtype=torch.float32, pad_shape=None, img_hw=None):
    """Returns (mlvl_feats, dpt_dist, img_meta).

    mlvl_feats[l]: [1, N, C, Hp/(4*2^l), Wp/(4*2^l)] standard normal (n_levels + 1 maps like
    the FPN's 4 outputs; the last is unused by the path); dpt_dist: [1, N, D, Hp/4, Wp/4]
    softmax(2 * randn) over D."""
    spec = SCANNET if kind == "scannet" else ARKIT
    hp, wp = pad_shape or spec["pad_shape"]
    if img_hw is not None and pad_shape is None:          # pad to the FPN's size divisor like the reference's pipeline
        hp, wp = (-(-int(img_hw[0]) // 32)) * 32, (-(-int(img_hw[1]) // 32)) * 32
    g = torch.Generator().manual_seed(seed)
    feats = []
    for l in range(n_levels + 1):
        ds = 4 * 2 ** l
        h, w = math.ceil(hp / ds), math.ceil(wp / ds)
        feats.append(torch.randn(1, n_views, channels, h, w, generator=g, dtype=torch.float32).to(device=device, dtype=dtype))
    dpt = torch.randn(1, n_views, n_depth, hp // 4, wp // 4, generator=g).mul(2).softmax(2).to(device=device, dtype=dtype)
    return feats, dpt, make_img_meta(n_views, kind, seed, img_hw=img_hw)


def clustered_occupancy(n_voxels_list, seed=0, device="cpu"):
    """A controlled, SURFACE-CLUSTERED occupancy for the refined levels (SURVEY.md 8d allows a controlled mask in place of the
    data-dependent one; with random weights the learned occupancy scatters the top 25 % uniformly, which no sparse kernel can
    exploit -- real occupancy sits on surfaces).  Returns one score tensor [X*Y*Z] float32 per refined level (levels 1 ...),
    for ``AdaptiveSparseHead.occupancy_override``: score = -(distance of the voxel centre to the nearest surface of a small
    seeded "room") + a seeded tie-breaker, so the level's top-k picks the k voxels nearest to the surfaces -- a floor, two
    walls and the shell of a box (a piece of furniture), the same world surfaces at every level.  Coordinates are the unit
    cube; k itself stays the config's ``topk_list`` (25 % of the level)."""
    g = torch.Generator().manual_seed(1000 + seed)
    r = torch.rand(8, generator=g)
    floor_z = 0.06 + 0.06 * float(r[0])
    wall_x = 0.05 + 0.08 * float(r[1])
    wall_y = 0.95 - 0.08 * float(r[2])
    lo = torch.tensor([0.30 + 0.2 * float(r[3]), 0.25 + 0.2 * float(r[4]), floor_z])
    hi = lo + torch.tensor([0.25 + 0.1 * float(r[5]), 0.30 + 0.1 * float(r[6]), 0.30 + 0.2 * float(r[7])])
    out = []
    for nx, ny, nz in n_voxels_list[1:]:
        ax = [(torch.arange(n, dtype=torch.float32) + 0.5) / n for n in (nx, ny, nz)]
        x, y, z = torch.meshgrid(*ax, indexing="ij")
        p = torch.stack([x, y, z], -1)
        d_planes = torch.minimum(torch.minimum((z - floor_z).abs(), (x - wall_x).abs()), (y - wall_y).abs())
        q = torch.maximum(lo - p, p - hi)                              # signed box distance, per axis
        outside = q.clamp(min=0).norm(dim=-1)
        inside = q.max(dim=-1).values.clamp(max=0)
        d = torch.minimum(d_planes, (outside + inside).abs())
        noise = torch.rand(d.shape, generator=g) * 1e-4                # breaks the exact ties of a symmetric grid
        out.append((-(d + noise)).reshape(-1).contiguous().to(device))
    return out


# BASELINE.json configs as concrete hot-path shapes
def _lvl(finest, finest_size):
    grids = [tuple(v // 4 for v in finest), tuple(v // 2 for v in finest), tuple(finest)]
    sizes = [tuple(round(s * 4, 6) for s in finest_size), tuple(round(s * 2, 6) for s in finest_size), tuple(finest_size)]
    return grids, sizes


def workload(name):
    """name -> dict(n_views, embed_dims, n_voxels_list, voxel_size_list, topk_list, kind, head, n_classes, n_reg_outs)."""
    table = {
        "cfg1_plumbing": dict(n_views=2, embed_dims=256, finest=(20, 20, 8), size=(.32, .32, .4), kind="scannet",
                              head="ScanNetImVoxelHeadV2", n_classes=18, n_reg_outs=6),
        "cfg2_scannet": dict(n_views=40, embed_dims=256, finest=(40, 40, 16), size=(.16, .16, .2), kind="scannet",
                             head="ScanNetImVoxelHeadV2", n_classes=18, n_reg_outs=6),
        "cfg2_scannet_100v": dict(n_views=100, embed_dims=256, finest=(40, 40, 16), size=(.16, .16, .2), kind="scannet",
                                  head="ScanNetImVoxelHeadV2", n_classes=18, n_reg_outs=6),
        "cfg3_arkit": dict(n_views=60, embed_dims=256, finest=(48, 48, 16), size=(.16, .16, .2), kind="arkit",
                           head="SunRgbdImVoxelHeadV2", n_classes=17, n_reg_outs=7),
        "cfg4_scannet200_large": dict(n_views=50, embed_dims=128, finest=(80, 80, 32), size=(.08, .08, .1),
                                      kind="scannet", head="ScanNetImVoxelHeadV2", n_classes=189, n_reg_outs=6),
        "cfg5_arkit_large": dict(n_views=100, embed_dims=128, finest=(96, 96, 32), size=(.08, .08, .1), kind="arkit",
                                 head="SunRgbdImVoxelHeadV2", n_classes=17, n_reg_outs=7),
    }
    w = dict(table[name])
    grids, sizes = _lvl(w.pop("finest"), w.pop("size"))
    w["n_voxels_list"], w["voxel_size_list"] = grids, sizes
    # sparse-volume ratio 0.25: refine the top 25 % of the voxels of levels 1, 2
    w["topk_list"] = [grids[1][0] * grids[1][1] * grids[1][2] // 4, grids[2][0] * grids[2][1] * grids[2][2] // 4]
    w["name"] = name
    return w


def model_config(w, dbound=(0.2, 5, 0.4), neck_out=128, nms_pre=1000):
    """An mmcv-style ``model`` dict for workload ``w`` with the reference's ``type=`` names."""
    C = w["embed_dims"]
    xf = dict(type="PerceptionTransformer_DFA3D", embed_dims=C, encoder=dict(
        type="VoxFormerEncoder_DFA3D", num_layers=1, return_intermediate=False, dbound=list(dbound),
        transformerlayers=dict(
            type="VoxFormerLayer",
            attn_cfgs=[dict(type="DeformCrossAttention_DFA3D", embed_dims=C, inter_view_aggregation="attn", dropout=0,
                            deformable_attention=dict(type="MSDeformableAttention3D_DFA3D", embed_dims=C, num_heads=8,
                                                      num_points=4, num_levels=1, im2col_step=128))],
            ffn_cfgs=dict(type="FFN", embed_dims=C, feedforward_channels=C * 2, num_fcs=2, ffn_drop=0.1,
                          act_cfg=dict(type="ReLU", inplace=True)),
            operation_order=("cross_attn", "norm", "ffn", "norm"))))
    heads = [dict(type="DenseHead", voxel_size=vs, n_voxels=nv, embed_dims=C, cross_transformer=xf)
             for vs, nv in zip(w["voxel_size_list"], w["n_voxels_list"])]
    test_cfg = dict(nms_pre=nms_pre, iou_thr=.25, score_thr=.01) if w["head"].startswith("ScanNet") else \
        dict(nms_pre=nms_pre, nms_thr=.15, use_rotate_nms=True, score_thr=.0)
    return dict(
        type="SGCDet",
        voxel_head=dict(type="AdaptiveSparseHead", embed_dims=C, topk_list=w["topk_list"],
                        voxel_size_list=w["voxel_size_list"], n_voxels_list=w["n_voxels_list"],
                        base_head_configs=heads),
        neck_3d=dict(type="FastIndoorImVoxelNeck", in_channels=C, out_channels=neck_out, n_blocks=[1, 1, 1]),
        bbox_head=dict(type=w["head"], n_classes=w["n_classes"], n_channels=neck_out, n_reg_outs=w["n_reg_outs"],
                       n_scales=3, limit=27, centerness_topk=18),
        voxel_size=w["voxel_size_list"][-1], n_voxels=w["n_voxels_list"][-1], test_cfg=test_cfg, train_cfg=dict())
