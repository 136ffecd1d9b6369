"""Where do the small device-to-device copies / fills of a scene come from?  One eager config-2 scene under
torch.profiler with Python stacks; prints aten::copy_ / fill_ / zero_ call sites by count."""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sgcdet_amd.plugin  # noqa: F401
from sgcdet_amd.mmcv_lite import build_detector
from sgcdet_amd.scene import make_scene, model_config, workload
w = workload("cfg2_scannet")
det = build_detector(model_config(w)).cuda().eval()
det.scene_graph, det.use_graph = False, False
feats, dpt, meta = make_scene(w["n_views"], w["embed_dims"], kind="scannet", seed=1, device="cuda", img_hw=(256, 320))
meta["_sgc_static"] = True
import traceback
sites = collections.Counter()
def site():
    for f in reversed(traceback.extract_stack()[:-2]):
        if "/sgcdet_amd/" in f.filename:
            return f"{f.filename.split('/sgcdet_amd/')[-1]}:{f.lineno}"
    return "?"
def wrap(obj, name):
    orig = getattr(obj, name)
    def w(*a, **k):
        sites[(name, site())] += 1
        return orig(*a, **k)
    setattr(obj, name, w)
with torch.no_grad():
    for _ in range(2): det.forward_features(feats, [meta], dpt)
    torch.cuda.synchronize()
    for n in ("copy_", "zero_", "fill_", "clone", "contiguous", "float", "to"): wrap(torch.Tensor, n)
    for n in ("zeros", "cat", "full", "zeros_like", "stack"): wrap(torch, n)
    det.forward_features(feats, [meta], dpt)
    torch.cuda.synchronize()
for (name, st), n in sites.most_common(70):
    print(f"{n:4d} {name:12s} {st}")
