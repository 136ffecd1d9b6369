"""Per-step host time of the scene-graph stepping pattern (1 vs 2 streams)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sgcdet_amd.scene import make_scene, workload
w = workload("cfg2_scannet")
dev = torch.device("cuda", 0)
det = bench.build_path(w, dev)
det.scene_graph = True
n_scenes = int(os.environ.get("NSC", "3"))
scenes = []
for s in range(n_scenes):
    feats, dpt, meta = make_scene(w["n_views"], w["embed_dims"], kind=w["kind"], seed=s, device=dev)
    scenes.append((feats, dpt, [meta]))
for n_streams in (1, 2):
    streams = [torch.cuda.Stream() for _ in range(n_streams)]
    def step(i):
        feats, dpt, metas = scenes[i % n_scenes]
        with torch.no_grad(), torch.cuda.stream(streams[i % n_streams]):
            return det.forward_features(feats, metas, dpt)
    for i in range(12): step(i)
    torch.cuda.synchronize()
    ts = []
    t0 = time.perf_counter()
    for i in range(24):
        t = time.perf_counter(); step(i); ts.append((time.perf_counter() - t) * 1e3)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"streams {n_streams}: wall {t_all / 24 * 1e3:.2f} ms/scene, host issue {t_issue / 24 * 1e3:.2f} ms/scene; per-step host ms:",
          " ".join(f"{v:.2f}" for v in ts))
