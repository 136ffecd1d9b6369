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
        if z[k].dtype.kind in "US":          # string arrays (e.g. state-dict key lists): read with numpy where needed
            continue
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


def fill_by_name(module, base_seed=0, scale=0.2):
    """Deterministic weights from the state-dict KEYS (no fixture bytes): every tensor is drawn from a generator seeded with
    base_seed + crc32(key).  Two modules with the same keys and shapes -- the reference's class in the golden generator,
    the product's class in the test -- end up with identical parameters and buffers."""
    import zlib
    sd = module.state_dict()
    with torch.no_grad():
        for k in sorted(sd):
            t = sd[k]
            if not t.is_floating_point():
                continue
            g = torch.Generator().manual_seed(base_seed + zlib.crc32(k.encode()))
            if k.endswith("running_var"):
                v = 0.5 + torch.rand(t.shape, generator=g)
            elif k.endswith("running_mean"):
                v = 0.1 * torch.randn(t.shape, generator=g)
            elif t.dim() == 1 and k.endswith("weight"):                      # norm scales
                v = 1.0 + 0.2 * torch.randn(t.shape, generator=g)
            elif t.dim() >= 2:                                               # conv / linear weights: keep activations O(1)
                fan_in = t[0].numel()
                v = torch.randn(t.shape, generator=g) * (scale * 5.0 / max(fan_in, 1) ** 0.5)
            else:
                v = scale * torch.randn(t.shape, generator=g)
            t.copy_(v.to(t.dtype))
    return module
