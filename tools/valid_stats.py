"""How sparse is the decoder tail really?  For bench-like scenes: fraction of finest-level voxels in valid, in its
3x3x3 dilations (what out_block_0 / up_block_1's 3x3x3 conv must produce for the head to be exact on valid), and the
fraction of 256-voxel bricks (4x4x16 / 4x8x8 / 8x8x4) holding at least one such voxel.
python tools/valid_stats.py [workload] [predicted|clustered]   (clustered: scene.clustered_occupancy as the selection's scores)"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2_scannet"
from sgcdet_amd.scene import make_scene, workload  # noqa: E402
w = workload(name)
det = bench.build_path(w, "cuda")
if len(sys.argv) > 2 and sys.argv[2] == "clustered":
    from sgcdet_amd.scene import clustered_occupancy  # noqa: E402
    det.voxel_head.occupancy_override = clustered_occupancy(w["n_voxels_list"], seed=0, device="cuda")
    print("occupancy: clustered override")
for seed in range(3):
    feats, dpt, meta = make_scene(w["n_views"], w["embed_dims"], kind=w["kind"], seed=seed, device="cuda",
                                  img_hw=(256, 320) if name.startswith("cfg2") else None)
    with torch.no_grad():
        r = det.forward_features(feats, [meta], dpt)
    v = r["valid"].float()                      # [1,1,X,Y,Z]
    line = [f"seed {seed}: valid {v.mean():.3f}"]
    d = v
    for k in (1, 2, 3):
        d = F.max_pool3d(d, 3, 1, 1)
        line.append(f"dilate{k} {d.mean():.3f}")
        for bx, by, bz in ((4, 4, 16), (4, 8, 8), (8, 8, 4)):
            if k <= 2:
                b = F.max_pool3d(d, (bx, by, bz), (bx, by, bz), ceil_mode=True)
                line.append(f"bricks{bx}x{by}x{bz} {b.mean():.3f}")
    for s in (1, 2):
        vs = torch.nn.Upsample(size=tuple(x // (2 ** s) for x in v.shape[-3:]), mode="trilinear")(v).round()
        line.append(f"valid@1/{2 ** s} {vs.mean():.3f} dilate1 {F.max_pool3d(vs, 3, 1, 1).mean():.3f}")
    print("  ".join(line))
