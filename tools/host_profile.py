"""Host-side cost of one scene: cProfile of the Python orchestration + issue-only timing."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sgcdet_amd.scene import make_scene, workload

w = workload("cfg2_scannet")
det = bench.build_path(w, torch.device("cuda"))
feats, dpt, meta = make_scene(40, 256, seed=0, device="cuda")


def step():
    with torch.no_grad():
        return det.forward_features(feats, [meta], dpt)


for _ in range(5):
    step()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(20):
    step()
t_issue = (time.perf_counter() - t) / 20
torch.cuda.synchronize()
t_all = (time.perf_counter() - t) / 20
print(f"issue {t_issue * 1e3:.2f} ms/scene, total {t_all * 1e3:.2f} ms/scene")
# per-stage wall with syncs
for name, fn in [("voxel_head", lambda: det.build_volume_from_features(feats, [meta], dpt)),]:
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10):
        with torch.no_grad():
            vol, valid, occ = fn()
    torch.cuda.synchronize(); print(name, (time.perf_counter() - t) / 10 * 1e3, "ms")
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(10):
    with torch.no_grad():
        f3 = det.extract_feat(vol)
        outs = det.bbox_head(f3)
torch.cuda.synchronize(); print("neck+head", (time.perf_counter() - t) / 10 * 1e3, "ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
