import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sgcdet_amd.scene import make_scene, workload
w = workload(os.environ.get("WL", "cfg1_plumbing"))
dev = torch.device("cuda", 0)
det = bench.build_path(w, dev)
nv = int(os.environ.get("NV", "4"))
feats, dpt, meta = make_scene(nv, w["embed_dims"], kind=w["kind"], seed=3, device=dev)
def run(sg, ug):
    det.scene_graph, det.use_graph = sg, ug
    with torch.no_grad():
        r = det.forward_features(feats, [meta], dpt)
    torch.cuda.synchronize()
    return {k: ([t.clone() for t in v] if isinstance(v, (list, tuple)) else v.clone()) for k, v in r.items()}
e = run(False, False)
for name, (sg, ug) in (("tail", (False, True)), ("scene", (True, True)), ("scene again", (True, True))):
    r = run(sg, ug)
    msg = []
    for k in ("volume", "valid", "occ"):
        msg.append(f"{k} {float((r[k].float() - e[k].float()).abs().max()):.2e}")
    for k in ("centerness", "bbox_pred", "cls_score"):
        msg.append(k + " " + " ".join(f"{float((a - b).abs().max()):.2e}" for a, b in zip(r[k], e[k])))
    print(name, "|", " | ".join(msg))
