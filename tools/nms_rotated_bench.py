"""Rotated multi-class BEV NMS at the ARKit head's size (3 x nms_pre candidates, 17 classes, score_thr 0): time of
the HIP path and, under `rocprofv3 --kernel-trace --stats`, its kernel breakdown."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from sgcdet_amd import ext
from nms_rotated_contract import arkit_like, bev_of
ops = ext.ops()
rb, rs = arkit_like(3000, 17, seed=9)
rb, rs = rb.cuda(), rs.cuda()
bev = bev_of(rb)
for _ in range(3): ops.nms_rotated_bev(bev, rs, 0.0, 0.15)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(20): keep, nk = ops.nms_rotated_bev(bev, rs, 0.0, 0.15)
torch.cuda.synchronize()
print(json.dumps(dict(ms=round((time.perf_counter() - t) / 20 * 1e3, 3), kept=int(nk.sum()))))
