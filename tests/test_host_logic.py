"""Host-side logic that must agree with the reference's own torch calls bit for bit."""
import numpy as np
import torch

from sgcdet_amd.plugin.voxformer import compute_projection, compute_projection_loop
from sgcdet_amd.scene import make_img_meta, workload, model_config


def test_single_mm_projection_equals_reference_loop():
    rng = np.random.RandomState(0)
    for trial in range(50):
        n = int(rng.randint(1, 101))
        meta = make_img_meta(n, "scannet" if trial % 2 else "arkit", seed=trial)
        meta["lidar2img"]["extrinsic"] = [(e + rng.randn(4, 4).astype(np.float32) * 0.3) for e in meta["lidar2img"]["extrinsic"]]
        assert torch.equal(compute_projection(meta), compute_projection_loop(meta))


def test_workloads_match_baseline_configs():
    w = workload("cfg2_scannet")
    assert w["n_voxels_list"] == [(10, 10, 4), (20, 20, 8), (40, 40, 16)] and w["topk_list"] == [800, 6400]
    assert workload("cfg1_plumbing")["topk_list"] == [100, 800]
    assert workload("cfg5_arkit_large")["topk_list"] == [9216, 73728]
    cfg = model_config(w)
    assert cfg["voxel_head"]["base_head_configs"][2]["n_voxels"] == (40, 40, 16)
