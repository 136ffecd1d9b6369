"""Timings of the widened rows on the GPU, written as one JSON object (profiles/r01_widened.json)."""
import json, os, sys, time
import numpy as np
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgcdet_amd import ext
from sgcdet_amd.scene import make_img_meta
from sgcdet_amd.plugin.plane_sweep import plane_sweep_correlation, closest_frame_ids, relative_projections
ops = ext.ops()
def timeit(fn, n=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n, r
out = {}
# ---- f-2: plane-sweep cost volume at DepthNet_Fusion's config-2 shapes ----
N, C, H, W, K = 40, 128, 60, 80, 2
meta = make_img_meta(N, "scannet", 0)
f = torch.randn(N, C, H, W, device="cuda")
depth = np.arange(0.2, 5.0, 0.4, dtype=np.float32) + 0.2
D = len(depth)
w2c = torch.tensor(np.array(meta["lidar2img"]["extrinsic"])); intr = torch.tensor(np.array(meta["lidar2img"]["intrinsic"])).clone()
intr[:2] /= meta["ori_shape"][0] / (meta["img_shape"][0] / 4)
nbr = closest_frame_ids(N, K); rel = relative_projections(w2c, intr, nbr).cuda(); dv = torch.from_numpy(depth).cuda()
def torch_formulation():
    y, x = torch.meshgrid(torch.arange(H, dtype=torch.float32, device="cuda"), torch.arange(W, dtype=torch.float32, device="cuda"), indexing="ij")
    xyz = torch.stack((x.reshape(-1), y.reshape(-1), torch.ones(H * W, device="cuda")))[None].repeat(N, 1, 1)
    corr = torch.zeros(N, D, H, W, device="cuda")
    for k in range(K):
        rot, trans = rel[:, k, :, :3], rel[:, k, :, 3:4]
        p = (rot @ xyz).unsqueeze(2) * dv.view(1, 1, D, 1) + trans.view(N, 3, 1, 1)
        xy = p[:, :2] / p[:, 2:3]
        grid = torch.stack((xy[:, 0] / ((W - 1) / 2) - 1, xy[:, 1] / ((H - 1) / 2) - 1), dim=3)
        warped = F.grid_sample(f[nbr[:, k].cuda()], grid.view(N, D * H, W, 2), mode="bilinear", padding_mode="zeros",
                               align_corners=False).view(N, C, D, H, W)
        corr += (warped * f.unsqueeze(2)).sum(dim=1) / (C ** 0.5)
    return corr / K
t_f, a = timeit(lambda: plane_sweep_correlation(f, meta, 4, depth, neighbor_img_num=K))
t_t, b = timeit(torch_formulation, 3, 1)
comp = N * H * W * C * 4 + N * D * H * W * 4
issued = N * H * W * K * D * 4 * C * 4
out["plane_sweep"] = dict(shape=f"{N} views x {C} ch x {H}x{W}, {D} planes, {K} neighbours", fused_ms=round(t_f * 1e3, 3),
                          reference_formulation_in_torch_ms=round(t_t * 1e3, 2), max_abs_diff=float((a - b).abs().max()),
                          compulsory_bytes=comp, issued_corner_row_bytes=issued,
                          issued_rate_TBps=round(issued / t_f / 1e12, 2), compulsory_rate_GBps=round(comp / t_f / 1e9, 1),
                          note="bound by the issued corner rows (L2 -> CU), not by HBM; the reference formulation also writes and "
                               "re-reads a [N,C,D,H,W] warped tensor per neighbour (1.18 GB each)")
# ---- f-4: aligned 3D NMS on the head's candidate count ----
g = torch.Generator().manual_seed(0)
n = 3000
c = (torch.rand(n, 3, generator=g) - 0.5) * torch.tensor([6.4, 6.4, 2.5])
c = c[torch.randint(0, 150, (n,), generator=g)] + torch.randn(n, 3, generator=g) * 0.05
s = 0.4 + torch.rand(n, 3, generator=g)
boxes = torch.cat([c - s / 2, c + s / 2], 1).cuda(); scores = torch.rand(n, generator=g).cuda(); labels = torch.randint(0, 18, (n,), generator=g).cuda()
t_n, keep = timeit(lambda: ops.aligned_nms3d(boxes, scores, labels, 0.25), 20)
def reference_loop():      # the reference's algorithm, torch ops on the GPU (one nonzero host sync per kept box)
    x1, y1, z1, x2, y2, z2 = boxes.unbind(1)
    area = (x2 - x1) * (y2 - y1) * (z2 - z1)
    zero = boxes.new_zeros(1)
    order = torch.argsort(scores); pick = []
    while order.shape[0] != 0:
        last = order.shape[0]; i = order[-1]; pick.append(i); o = order[:last - 1]
        inter = (torch.max(zero, torch.min(x2[i], x2[o]) - torch.max(x1[i], x1[o])) * torch.max(zero, torch.min(y2[i], y2[o]) - torch.max(y1[i], y1[o]))
                 * torch.max(zero, torch.min(z2[i], z2[o]) - torch.max(z1[i], z1[o])))
        iou = inter / (area[i] + area[o] - inter) * (labels[i] == labels[o]).float()
        order = o[torch.nonzero(iou <= 0.25, as_tuple=False).flatten()]
    return torch.stack(pick)
t_r, keep_r = timeit(reference_loop, 2, 1)
out["aligned_nms3d"] = dict(candidates=n, kept=int(keep.numel()), hip_ms=round(t_n * 1e3, 3), reference_loop_in_torch_ms=round(t_r * 1e3, 1),
                            identical_indices=bool(torch.equal(keep, keep_r)))
# ---- f-4: rotated BEV NMS of the ARKit head: 3 x nms_pre candidates, 17 classes, score_thr 0 ----
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from nms_rotated_contract import arkit_like, bev_of
from sgcdet_amd.plugin.bbox_head import box3d_multiclass_nms_rotated
rb, rs = arkit_like(3000, 17, seed=9)
rb, rs = rb.cuda(), rs.cuda()
bev = bev_of(rb)
t_k, (keep, nk) = timeit(lambda: ops.nms_rotated_bev(bev, rs, 0.0, 0.15), 10)
t_g, res = timeit(lambda: box3d_multiclass_nms_rotated(ops, rb, rs, 0.0, 1000, 0.15), 10)
cand = int((rs > 0).sum())
pairs = int(((rs > 0).sum(0).double() ** 2 / 2).sum())
out["nms_rotated_bev"] = dict(candidates_per_class=3000, classes=17, candidates_total=cand, pairs=pairs, kept=int(nk.sum()),
                              sort_mask_sweep_ms=round(t_k * 1e3, 3), with_glue_and_readback_ms=round(t_g * 1e3, 3),
                              note="reference: 17 x (sort + mmcv nms_rotated mask kernel + bit-matrix copy to the host + CPU sweep)")
# ---- f-3: target assignment of the head at the config-2 point set (29 200 points x 3 scales), 40 boxes ----
from targets_contract import random_boxes
d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "head_targets.npz"))
pts, scl = torch.from_numpy(d["points"]).cuda().contiguous(), torch.from_numpy(d["scales"]).cuda().contiguous()
out["assign_targets"] = {}
for rotated in (False, True):
    tb, tl = random_boxes(40, 3, rotated)
    tb, tl = tb.cuda(), tl.cuda()
    t_a, res = timeit(lambda: ops.assign_targets(pts, scl, tb, tl, rotated, 3, 27, 18), 20)
    out["assign_targets"]["rotated" if rotated else "axis_aligned"] = dict(points=int(pts.shape[0]), boxes=40, hip_ms=round(t_a * 1e3, 3),
                                                                           positives=int((res[2] >= 0).sum()))
out["assign_targets"]["note"] = "reference: ~40 torch launches over [n_points, n_boxes(, 6)] tensors (29 200 x 40 x 6 floats = 28 MB each)"
print(json.dumps(out))
