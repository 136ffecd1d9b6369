"""Helpers to read the committed golden fixtures (tests/golden/*.npz, made by
tests/golden/make_golden.py from the reference's own Python in the build container)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    data, sd = {}, {}
    for k in z.files:
        t = torch.from_numpy(z[k])
        if k.startswith("sd::"):
            sd[k[4:]] = t
        else:
            data[k] = t
    return data, sd


def img_meta(d):
    return dict(img_shape=tuple(int(v) for v in d["meta_img_shape"]), ori_shape=tuple(int(v) for v in d["meta_ori_shape"]),
                lidar2img=dict(extrinsic=[e for e in d["meta_extrinsic"].numpy()], intrinsic=d["meta_intrinsic"].numpy(),
                               origin=d["meta_origin"].numpy()))


def depth_pyramid(dpt):
    import torch.nn.functional as F
    return [dpt, F.interpolate(dpt, scale_factor=(1, 0.5, 0.5), mode="nearest"),
            F.interpolate(dpt, scale_factor=(1, 0.25, 0.25), mode="nearest")]


def max_err(a, b):
    return (a.detach().cpu().double() - b.detach().cpu().double()).abs().max().item()
