"""Which torch (aten) operators still launch kernels inside a scene's launch sequence?  Runs the static-count scene body of the
whole-scene hipGraph eagerly under a TorchDispatchMode and prints every aten op that touches a CUDA tensor, with the source
line of the product code that issued it (views / metadata ops are listed separately: they launch nothing).
python tools/graph_ops.py [workload]"""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from sgcdet_amd.scene import make_scene, workload  # noqa: E402
from sgcdet_amd.plugin.voxformer import scene_constants_host  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2_scannet"
w = workload(name)
det = bench.build_path(w, "cuda")
det.use_graph = det.scene_graph = False
feats, dpt, meta = make_scene(w["n_views"], w["embed_dims"], kind=w["kind"], seed=0, device="cuda",
                              img_hw=(256, 320) if name.startswith("cfg2") else None)
meta = dict(meta)
meta["_sgc_scene_const"] = scene_constants_host(meta).cuda()
meta["_sgc_static"] = True

VIEW_OPS = ("view", "reshape", "permute", "transpose", "unsqueeze", "squeeze", "slice", "select", "expand", "as_strided", "alias",
            "detach", "_unsafe_view", "t.default", "unbind", "split", "narrow", "size", "stride", "is_", "_local_scalar", "sym_", "lift_fresh",
            "empty", "_to_copy")


class Spy(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.kern = collections.Counter()
        self.views = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        n = str(func)
        flat = [a for a in list(args) + list((kwargs or {}).values()) if torch.is_tensor(a)]
        if torch.is_tensor(out):
            flat.append(out)
        if not any(t.is_cuda for t in flat):
            return out
        where = "?"
        for fr in reversed(traceback.extract_stack()):
            if "/sgcdet_amd/" in fr.filename and "tensor_api" not in fr.filename and "_abi" not in fr.filename:
                where = f"{os.path.relpath(fr.filename, ROOT)}:{fr.lineno}"
                break
            if "/sgcdet_amd/tensor_api" in fr.filename:
                where = f"{os.path.relpath(fr.filename, ROOT)}:{fr.lineno}"
        shape = tuple(out.shape) if torch.is_tensor(out) else ""
        if any(v in n for v in VIEW_OPS) and "copy" not in n.replace("_to_copy", ""):
            if "empty" in n or "_to_copy" in n:
                self.views[(n, where)] += 1
            return out
        self.kern[(n, where, str(shape))] += 1
        return out


with torch.no_grad():
    det.forward_features(feats, [meta], dpt)          # warm-up: plans, caches
    torch.cuda.synchronize()
    spy = Spy()
    with spy:
        volume, valid, occ = det.build_volume_from_features(feats, [meta], dpt)
        outs = det._neck_head_eager(volume, valid)
    torch.cuda.synchronize()
print(f"{name}: aten ops on CUDA tensors inside one scene's launch sequence (kernel-launching candidates): {sum(spy.kern.values())}")
for (n, where, shape), c in sorted(spy.kern.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    print(f"  {c:3d} x {n:42s} {shape:28s} {where}")
print("allocations / dtype copies (no kernel unless a conversion):")
for (n, where), c in sorted(spy.views.items(), key=lambda kv: -kv[1])[:12]:
    print(f"  {c:3d} x {n:42s} {where}")
