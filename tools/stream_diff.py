"""Does the same scene give bit-identical voxel features on the default stream and on a side stream?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sgcdet_amd.scene import make_scene, workload
from sgcdet_amd import ext
w = workload("cfg2_scannet")
dev = torch.device("cuda", 0)
det = bench.build_path(w, dev)
det.use_graph = False
feats, dpt, meta = make_scene(w["n_views"], w["embed_dims"], kind=w["kind"], seed=0, device=dev)
ops = ext.ops()
trace = []
def wrap(name):
    fn = getattr(ops, name)
    def f(*a, **k):
        out = fn(*a, **k)
        t = out[0] if isinstance(out, tuple) else out
        if torch.is_tensor(t) and t.is_floating_point():
            trace.append((name, t.detach().clone()))
        return out
    setattr(ops, name, f)
for n in ("project_points", "nchw_to_nhwc_crop", "pairs_geometry_sample", "conv3d_cl_bf16x3", "depth_pairs",
          "pairs_deform_gather", "view_mean", "view_attend", "upsample2x_occ"):
    wrap(n)
import torch.nn.functional as F
_lin = F.linear
def lin(x, w_, b=None):
    y = _lin(x, w_, b); trace.append(("F.linear", y.detach().clone())); return y
F.linear = lin
def run(stream):
    trace.clear()
    with torch.no_grad():
        if stream is None:
            r = det.forward_features(feats, [meta], dpt)
        else:
            with torch.cuda.stream(stream):
                r = det.forward_features(feats, [meta], dpt)
    torch.cuda.synchronize()
    return r["volume"].clone(), list(trace)
run(None)
v0, t0 = run(None)
v0b, t0b = run(None)
s = torch.cuda.Stream()
v1, t1 = run(s)
v1b, t1b = run(s)
print("default vs default:", float((v0 - v0b).abs().max()), " side vs side:", float((v1 - v1b).abs().max()),
      " default vs side:", float((v0 - v1).abs().max()))
for (n0, a), (n1, b) in zip(t0, t1):
    if a.shape != b.shape or not torch.equal(a, b):
        print("first differing stage default vs side:", n0, tuple(a.shape), tuple(b.shape),
              float((a - b).abs().max()) if a.shape == b.shape else None)
        break
else:
    print("all traced stages identical")
