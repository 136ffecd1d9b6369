"""A/B of the deformable-gather kernel variants on the config-2 finest-level shape (one process,
interleaved rounds, HIP events).  Usage: python tools/gather_bench.py [n_views] [C]"""
import ctypes
import sys

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgcdet_amd import ext
from sgcdet_amd.scene import make_img_meta
from sgcdet_amd.plugin.voxformer import compute_projection

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
C = int(sys.argv[2]) if len(sys.argv) > 2 else 256
H, W, D, M, P = 59, 80, 12, 8, 4
ops = ext.ops()
if os.environ.get("SGC_DIAG_LIB"):      # diagnostic builds (tools/diag): timing only, results are garbage
    from sgcdet_amd._abi import Library
    from sgcdet_amd.tensor_api import TensorOps
    ops = TensorOps(Library(os.environ["SGC_DIAG_LIB"]), "cuda")
dev = "cuda"
meta = make_img_meta(N, "scannet", 0)
proj = compute_projection(meta).float().to(dev).contiguous()
origin = torch.tensor(meta["lidar2img"]["origin"]).to(dev)
# 6400 random voxels of the 40x40x16 grid (top-k 25 %)
g = torch.Generator().manual_seed(0)
nx, ny, nz = 40, 40, 16
idx = torch.randperm(nx * ny * nz, generator=g)[:6400].sort().values
xs = torch.stack([idx // (ny * nz), (idx // nz) % ny, idx % nz], 1).float()
ref3d = (xs * torch.tensor([.16, .16, .2]) - torch.tensor([nx, ny, nz]) / 2 * torch.tensor([.16, .16, .2])).to(dev).contiguous()
ref_cam, mask = ops.project_points(ref3d, origin, proj, 320, 239, 0.2, 5.0)
pc = ops.compact_pairs(mask)
n_pairs = int(pc["totals"][0])
vbuf = torch.randn(N * H * W + 1, C, device=dev)
vbuf[-1].zero_()
value = vbuf[:N * H * W].view(N, H * W, M, C // M)
ZR = os.environ.get("SGC_ZR", "1") == "1"
dist = torch.randn(N, H * W, D, device=dev).mul(2).softmax(-1).contiguous()
raw = torch.randn(n_pairs, M * P * 4, device=dev)
raw[:, :M * P * 2] *= float(os.environ.get("SGC_OFFSET_SCALE", "2.0"))
if os.environ.get("SGC_OFFSETS") == "ring":   # the reference's init (deformable_cross_attention.py:194-212) + noise, as bench.py's weights
    import math
    th = torch.arange(M, dtype=torch.float32) * (2 * math.pi / M)
    ring = torch.stack([th.cos(), th.sin()], -1)
    ring = ring / ring.abs().max(-1, keepdim=True)[0]
    steps = torch.arange(1, P + 1, dtype=torch.float32)
    uvb = (ring.view(M, 1, 2) * steps.view(1, P, 1)).reshape(-1)
    dzb = (((th.cos() + th.sin()) / 2).view(M, 1) * steps.view(1, P)).reshape(-1)
    noise = float(os.environ.get("SGC_RING_NOISE", "0.3"))
    raw[:, :M * P * 2] = uvb.to(dev) + torch.randn(n_pairs, M * P * 2, device=dev) * noise
    raw[:, M * P * 2:M * P * 3] = dzb.to(dev) + torch.randn(n_pairs, M * P, device=dev) * noise
if os.environ.get("SGC_ZERO_DEPTH_OFF"):
    raw[:, M * P * 2:M * P * 3] = 0
feat = value.view(N, H * W, C)
alg = N * H * W * C * 4 + N * H * W * D * 4 + n_pairs * 512 + n_pairs * C * 4
alg_geo = N * H * W * C * 4 + N * H * W * D * 4 + n_pairs * 12 + n_pairs * C * 4
print(f"pairs {n_pairs}  algorithmic bytes deform {alg / 1e6:.1f} MB  geom {alg_geo / 1e6:.1f} MB")


dp = ops.depth_pairs(dist, H, W) if os.environ.get("SGC_DP", "1") == "1" else None
if os.environ.get("SGC_PAIR_REF"):      # diagnostic build -DSGC_DIAG_PAIR_REF: per-pair (x, y, z, camera) precomputed
    pcam, pq = pc["pair_cam"][:n_pairs].long(), pc["pair_q"][:n_pairs].long()
    pref = torch.cat([ref_cam[pcam, pq], pc["pair_cam"][:n_pairs].view(torch.float32)[:, None]], 1).contiguous()
    ops.lib._dll.sgc_debug_buffer.argtypes = [ctypes.c_void_p]
    ops.lib._dll.sgc_debug_buffer(ctypes.c_void_p(pref.data_ptr()))


def run(kind):
    if kind == "deform":
        return ops.pairs_deform_gather(value, dist, ref_cam, raw, pc["pair_cam"], pc["pair_q"], n_pairs, H, W, M, P,
                                       dist_pairs=dp, zero_row=ZR)
    return ops.pairs_geometry_sample(feat, dist, ref_cam, pc["pair_cam"], pc["pair_q"], n_pairs, H, W)


variants = [int(v) for v in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["0", "1"])]
outs = {}
times = {(v, k): [] for v in variants for k in ("deform", "geom")}
for rnd in range(12):
    for v in variants:
        ops.lib.call("sgc_set_tuning", b"fwd_variant", 1 if v >= 1 else 0)
        ops.lib.call("sgc_set_tuning", b"fwd_spl", v if v > 1 else 1)
        for k in ("deform", "geom"):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            o = run(k)
            e1.record()
            torch.cuda.synchronize()
            if rnd >= 2:
                times[(v, k)].append(e0.elapsed_time(e1) * 1e3)
            outs[(v, k)] = o
for (v, k), t in times.items():
    t = sorted(t)
    b = alg if k == "deform" else alg_geo
    print(f"variant {v} {k:7s} median {t[len(t) // 2]:7.1f} us  min {t[0]:7.1f} us  -> {b / t[len(t) // 2] / 1e3:7.1f} GB/s "
          f"({b / t[len(t) // 2] / 1e3 / 8000:.3f} of 8 TB/s)")
for k in ("deform", "geom"):
    ref = outs[(variants[0], k)]
    for v in variants[1:]:
        print(k, "variant", v, "max |diff| vs variant", variants[0], (outs[(v, k)] - ref).abs().max().item())
