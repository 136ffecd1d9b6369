"""Near-tie flips of the top-k voxel selection against the oracle over seeded scenes of configs 1 / 2 / 5 (views reduced so
that the CPU oracle finishes in seconds): flips of the finest selected set, flips of the coarse set, max |occupancy
difference|.  (Round 4 used it with a measurement knob to price the IEEE forms of the gather's phase 1:
profiles/r04_gather_ieee.txt; they are the product's forms now.)  Usage: python tools/tie_flips.py [seeds]"""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sgcdet_amd.plugin  # noqa: F401,E402
from sgcdet_amd import ext  # noqa: E402
from sgcdet_amd.mmcv_lite import build_detector  # noqa: E402
from sgcdet_amd.scene import make_scene, model_config, workload  # noqa: E402
from oracle.ref_path import RefPath  # noqa: E402

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ops = ext.ops()
rows = []
for name, n_views in (("cfg1_plumbing", 2), ("cfg2_scannet", 6), ("cfg5_arkit_large", 3)):
    w = workload(name)
    torch.manual_seed(7)
    det = build_detector(model_config(w)).eval()
    gen = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for _, p in det.voxel_head.named_parameters():
            p.add_(torch.randn(p.shape, generator=gen) * 0.05)
    rp = RefPath(det.voxel_head.state_dict(), dict(embed_dims=w["embed_dims"], n_voxels_list=w["n_voxels_list"],
                                                   voxel_size_list=w["voxel_size_list"], topk_list=w["topk_list"],
                                                   dbound=(0.2, 5.0), num_heads=8, num_points=4), omp=True)
    det = det.cuda()
    n_fin = w["n_voxels_list"][-1][0] * w["n_voxels_list"][-1][1] * w["n_voxels_list"][-1][2]
    tot = {m: dict(fine=0, coarse=0, occ=0.0) for m in range(1)}
    for seed in range(n_seeds):
        feats, dpt, meta = make_scene(n_views, w["embed_dims"], kind=w["kind"], seed=100 + seed)
        dpts = [dpt, F.interpolate(dpt, scale_factor=(1, 0.5, 0.5), mode="nearest"),
                F.interpolate(dpt, scale_factor=(1, 0.25, 0.25), mode="nearest")]
        vol_c, valid_c, occ_c = rp.adaptive_sparse_head(feats, meta, dpts)
        set1_c = set(torch.topk(occ_c[0, n_fin:], w["topk_list"][0]).indices.tolist())
        for mode in range(1):
            with torch.no_grad():
                r = det.forward_features([f.cuda() for f in feats], [meta], dpt.cuda())
            occ_g = r["occ"].cpu()
            set1_g = set(torch.topk(occ_g[0, n_fin:], w["topk_list"][0]).indices.tolist())
            tot[mode]["fine"] += int((r["valid"].cpu() != valid_c).sum())
            tot[mode]["coarse"] += len(set1_c ^ set1_g) // 2
            tot[mode]["occ"] = max(tot[mode]["occ"], float((occ_g - occ_c).abs().max()))
    for mode in range(1):
        rows.append(dict(config=name, views=n_views, seeds=n_seeds, fine_flips=tot[mode]["fine"],
                         coarse_flips=tot[mode]["coarse"], max_occ_diff=tot[mode]["occ"]))
        print(json.dumps(rows[-1]), flush=True)
    del det
