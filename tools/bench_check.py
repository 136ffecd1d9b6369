"""bench.py's stepping pattern (3 scenes over 2 streams, no clones, no syncs) checked against serial eager results."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sgcdet_amd.scene import make_scene, workload
w = workload(os.environ.get("WL", "cfg2_scannet"))
dev = torch.device("cuda", 0)
det = bench.build_path(w, dev)
n_scenes = int(os.environ.get("NSC", "3"))
scenes = []
for s in range(n_scenes):
    feats, dpt, meta = make_scene(w["n_views"], w["embed_dims"], kind=w["kind"], seed=s, device=dev)
    scenes.append((feats, dpt, [meta]))
det.use_graph = False
serial = []
with torch.no_grad():
    for feats, dpt, metas in scenes:
        r = det.forward_features(feats, metas, dpt)
        serial.append((r["volume"].clone(), r["occ"].clone()))
torch.cuda.synchronize()
streams = [torch.cuda.Stream() for _ in range(2)]
def trial(name, graph, clone, sync, n=30):
    det.use_graph = graph
    runs = []
    with torch.no_grad():
        for i in range(n):
            feats, dpt, metas = scenes[i % n_scenes]
            with torch.cuda.stream(streams[i % 2]):
                r = det.forward_features(feats, metas, dpt)
                runs.append((i, (r["volume"].clone(), r["occ"].clone()) if clone else (r["volume"], r["occ"]), r))
            if sync:
                torch.cuda.synchronize()
    torch.cuda.synchronize()
    bad = [i for i, (v, o), _ in runs if not (torch.equal(v, serial[i % n_scenes][0]) and torch.equal(o, serial[i % n_scenes][1]))]
    print(f"{name}: graph={graph} clone={clone} sync={sync}: {len(bad)} of {n} wrong {bad[:12]}", flush=True)
for graph in (True, False):
    trial("A", graph, clone=False, sync=False)
    trial("B", graph, clone=True, sync=False)
    trial("C", graph, clone=False, sync=True)
