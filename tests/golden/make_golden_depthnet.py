#!/usr/bin/env python3
"""tests/golden/depth_net.npz: outputs of the REFERENCE's own ``DepthNet_Fusion`` (depth_est_fusion.py:166-329 with its
``ResNetFPN`` extractor, extractor_matching.py / layer_matching.py) on a small seeded scene, in eval mode: the depth
distribution [1, N, D, H, W], the down-sampled one-hot depth labels and the depth loss.

The weights are NOT stored: both sides fill every state-dict tensor from a generator seeded by its key
(tests/golden_util.py::fill_by_name), which also pins the state-dict key set.  Runs only in the build container (imports
the reference through stub modules for mmdet / mmcv); nothing of the reference is copied.
Usage:  python tests/golden/make_golden_depthnet.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF_DIR = "/root/reference/mmdet3d_plugin/models/im2voxel/depth_utils"


def load_reference():
    def mod(name, **kw):
        m = types.ModuleType(name)
        m.__dict__.update(kw)
        sys.modules[name] = m
        return m

    class _Reg:
        def register_module(self, *a, **k):
            return lambda cls: cls

    mod("mmdet"); mod("mmdet.models", HEADS=_Reg())
    mod("mmcv"); mod("mmcv.runner", force_fp32=lambda *a, **k: (lambda f: f), auto_fp16=lambda *a, **k: (lambda f: f))
    pkg = mod("_ref_depth_utils"); pkg.__path__ = [REF_DIR]
    out = {}
    for name in ("layer_matching", "extractor_matching", "depth_est_fusion"):
        spec = importlib.util.spec_from_file_location("_ref_depth_utils." + name, os.path.join(REF_DIR, name + ".py"))
        m = importlib.util.module_from_spec(spec)
        m.__package__ = "_ref_depth_utils"
        sys.modules["_ref_depth_utils." + name] = m
        spec.loader.exec_module(m)
        out[name] = m
    return out["depth_est_fusion"]


def main():
    from golden_util import fill_by_name
    from sgcdet_amd.scene import make_img_meta
    ref = load_reference()
    N, mono_c, Hf, Wf, stride = 5, 32, 12, 16, 4
    dbound = [0.2, 5.0, 0.4]
    net = ref.DepthNet_Fusion(neighbor_img_num=2, downsample_factor=stride, dbound=dbound, mono_channels=mono_c,
                              loss_weight=0.5, max_tol=0, init_weight="none").eval()
    fill_by_name(net, base_seed=7, scale=0.15)
    g = torch.Generator().manual_seed(11)
    meta = make_img_meta(N, "scannet", 3)
    meta["img_shape"] = (Hf * stride - 1, Wf * stride, 3)          # as the ScanNet pipeline: one row short of the padded size
    xs = torch.randn(1, N, mono_c, Hf, Wf, generator=g)
    imgs = torch.randn(1, N, 3, Hf * stride, Wf * stride, generator=g)
    with torch.no_grad():
        pred = net(xs, imgs, [meta], stride)
    depth_maps = torch.rand(1, N, Hf * stride, Wf * stride, generator=g) * 6.0
    depth_maps[torch.rand(depth_maps.shape, generator=g) < 0.3] = 0.0           # missing measurements
    labels = net.get_downsampled_gt_depth(depth_maps)
    loss = net.loss(depth_maps, pred)["loss_dpt"]
    net_tol = ref.DepthNet_Fusion(neighbor_img_num=2, downsample_factor=stride, dbound=dbound, mono_channels=mono_c,
                                  max_tol=1, init_weight="none")
    labels_tol = net_tol.get_downsampled_gt_depth(depth_maps)
    keys = sorted(net.state_dict().keys())
    np.savez_compressed(os.path.join(HERE, "depth_net.npz"), xs=xs.numpy(), imgs=imgs.numpy(), pred=pred.numpy(),
                        depth_maps=depth_maps.numpy(), labels=labels.numpy(), labels_tol=labels_tol.numpy(),
                        loss=np.float32(loss), keys=np.array(keys), stride=np.int64(stride), dbound=np.array(dbound),
                        meta_extrinsic=np.stack(meta["lidar2img"]["extrinsic"]), meta_intrinsic=meta["lidar2img"]["intrinsic"],
                        meta_origin=meta["lidar2img"]["origin"], meta_img_shape=np.array(meta["img_shape"]),
                        meta_ori_shape=np.array(meta["ori_shape"]))
    print(f"wrote depth_net.npz: pred {tuple(pred.shape)} sum over D {float(pred.sum(2).mean()):.4f} peak {float(pred.max()):.3f} "
          f"loss {float(loss):.4f} labelled pixels {int((labels.sum(1) > 0).sum())} of {labels.shape[0]}, {len(keys)} state-dict keys")


if __name__ == "__main__":
    main()
