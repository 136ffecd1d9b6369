import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sgcdet_amd.scene import make_scene, workload
w = workload("cfg1_plumbing")
dev = torch.device("cuda", 0)
det = bench.build_path(w, dev)
feats, dpt, meta = make_scene(4, w["embed_dims"], kind=w["kind"], seed=3, device=dev)
stash = {}
orig = det._neck_head_eager
def nh(volume):
    f3 = det.extract_feat(volume)
    outs = det.bbox_head(f3)
    stash["vol"] = volume; stash["neck"] = list(f3)
    return tuple(list(o) for o in outs)
det._neck_head_eager = nh
det.scene_graph, det.use_graph = False, False
with torch.no_grad():
    r = det.forward_features(feats, [meta], dpt)
torch.cuda.synchronize()
e_neck = [t.clone() for t in stash["neck"]]; e_vol = stash["vol"].clone()
det.scene_graph = True
for k in range(3):
    with torch.no_grad():
        r = det.forward_features(feats, [meta], dpt)
    torch.cuda.synchronize()
    print("replay", k, "vol diff", float((stash["vol"] - e_vol).abs().max()), "neck diffs",
          [f"{float((a - b).abs().max()):.2e}" for a, b in zip(stash["neck"], e_neck)],
          "vol ptr", stash["vol"].data_ptr() % 100000, "contig/strides", stash["vol"].stride())
