#!/usr/bin/env python3
"""Generates tests/golden/plane_sweep.npz with the reference's own plane-sweep code
(/root/reference/mmdet3d_plugin/models/im2voxel/depth_utils/depth_est_fusion.py: get_closest_frame_ids :53-64,
collect_proj :67-84, homo_warping :87-126, and the cost-volume loop of DepthNet_Fusion.forward :222-238 replayed
with those functions).  Build-container only; mmcv / mmdet imports are stubbed, the module's sibling
``extractor_matching`` is not needed for these functions and is stubbed as well.  The fixture holds inputs
(features, poses, intrinsics, depth planes) and the reference's outputs (neighbour ids, relative projections,
correlation volume) -- no reference source."""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/mmdet3d_plugin/models/im2voxel/depth_utils/depth_est_fusion.py"


def load_reference():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Reg:
        def register_module(self, *a, **k):
            return lambda cls: cls

    mod("mmdet"); mod("mmdet.models", HEADS=_Reg())
    mod("mmcv"); mod("mmcv.runner", force_fp32=lambda *a, **k: (lambda f: f), auto_fp16=lambda *a, **k: (lambda f: f))
    pkg = mod("_ref_depth_utils"); pkg.__path__ = []
    mod("_ref_depth_utils.extractor_matching", ResNetFPN=object)
    spec = importlib.util.spec_from_file_location("_ref_depth_utils.depth_est_fusion", REF)
    m = importlib.util.module_from_spec(spec)
    m.__package__ = "_ref_depth_utils"
    spec.loader.exec_module(m)
    return m


def main():
    ref = load_reference()
    from sgcdet_amd.scene import make_img_meta
    out = {}
    cases = [(6, 32, 12, 16, 2, 0), (7, 64, 9, 12, 2, 1), (7, 128, 8, 10, 4, 2)]
    for k, (N, C, H, W, K, seed) in enumerate(cases):
        g = torch.Generator().manual_seed(seed)
        meta = make_img_meta(N, "scannet", seed)
        f_mvs = torch.randn(N, C, H, W, generator=g)
        w2c = torch.tensor(np.array(meta["lidar2img"]["extrinsic"]))
        intr = torch.tensor(np.array(meta["lidar2img"]["intrinsic"])).clone()
        stride = 320 // W                                   # feature stride of this toy resolution
        ratio = meta["ori_shape"][0] / (meta["img_shape"][0] / stride)
        intr[:2] /= ratio                                   # depth_est_fusion.py:209-213
        dbound = (0.2, 5.0, 0.4)
        depth_values = torch.tensor(np.arange(dbound[0], dbound[1], dbound[2], dtype=np.float32) + dbound[2] / 2)
        D = depth_values.numel()
        kk = min(K, N - 1)
        nbr = ref.get_closest_frame_ids(N, kk)                                    # :222
        nei_features = torch.unbind(f_mvs[nbr.view(-1)].view(N, kk, C, H, W), dim=1)
        ref_proj, nei_projs = ref.collect_proj(w2c, intr, nbr)                     # :228
        dv = depth_values.unsqueeze(0).repeat(N, 1)
        corr = torch.zeros((N, D, H, W))
        rel = []
        for nei_fea, nei_proj in zip(nei_features, nei_projs):                     # :233-240
            warped = ref.homo_warping(nei_fea, nei_proj, ref_proj, dv)
            corr += (warped * f_mvs.unsqueeze(2)).sum(dim=1) / torch.sqrt(torch.tensor(C).float())
            rel.append(torch.matmul(nei_proj, torch.inverse(ref_proj))[:, :3, :4])
        corr = corr / kk
        out[f"f_mvs{k}"], out[f"w2c{k}"], out[f"intr{k}"] = f_mvs.numpy(), w2c.numpy(), intr.numpy()
        out[f"depth{k}"], out[f"nbr{k}"], out[f"corr{k}"] = depth_values.numpy(), nbr.numpy().astype(np.int64), corr.numpy()
        out[f"rel{k}"] = torch.stack(rel, 1).numpy()        # [N, K, 3, 4]
        print(f"case {k}: N={N} C={C} {H}x{W} K={kk} D={D} |corr| max {float(corr.abs().max()):.3f} "
              f"nonzero {float((corr != 0).float().mean()):.2f}")
    out["n_cases"] = np.int64(len(cases))
    np.savez_compressed(os.path.join(HERE, "plane_sweep.npz"), **out)


if __name__ == "__main__":
    main()
