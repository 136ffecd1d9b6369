"""How much of a pipelined scene is host work?  Times the 2-stream stepping loop and the part of it the host
spends blocked in the per-level pair-count read-back (Tensor.tolist)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sgcdet_amd.scene import make_scene, workload
w = workload(os.environ.get("WL", "cfg2_scannet"))
dev = torch.device("cuda", 0)
det = bench.build_path(w, dev)
scenes = []
for s in range(3):
    feats, dpt, meta = make_scene(w["n_views"], w["embed_dims"], kind=w["kind"], seed=s, device=dev)
    scenes.append((feats, dpt, [meta]))
blocked = [0.0]
_tolist = torch.Tensor.tolist
def tolist(self):
    t = time.perf_counter(); r = _tolist(self); blocked[0] += time.perf_counter() - t; return r
torch.Tensor.tolist = tolist
for n_streams in (1, 2, 3):
    streams = [torch.cuda.Stream() for _ in range(n_streams)]
    def step(i):
        feats, dpt, metas = scenes[i % 3]
        with torch.no_grad(), torch.cuda.stream(streams[i % n_streams]):
            return det.forward_features(feats, metas, dpt)
    for i in range(8): step(i)
    torch.cuda.synchronize()
    blocked[0] = 0.0
    n = 60
    t = time.perf_counter()
    for i in range(n): step(i)
    t_issue = time.perf_counter() - t
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t
    print(f"streams {n_streams}: wall {t_all / n * 1e3:.2f} ms/scene; host loop {t_issue / n * 1e3:.2f} ms/scene of which blocked in "
          f"read-backs {blocked[0] / n * 1e3:.2f} -> host busy {(t_issue - blocked[0]) / n * 1e3:.2f} ms/scene")
import cProfile, pstats
torch.Tensor.tolist = _tolist
streams = [torch.cuda.Stream() for _ in range(2)]
def step2(i):
    feats, dpt, metas = scenes[i % 3]
    with torch.no_grad(), torch.cuda.stream(streams[i % 2]):
        return det.forward_features(feats, metas, dpt)
for i in range(6): step2(i)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for i in range(30): step2(i)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
