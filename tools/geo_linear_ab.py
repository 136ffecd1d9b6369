"""sgc_pairs_geometry_linear_bf16x3 of two builds of the library against each other (tools/diag/libsgc_old.so = the other build) on the
finest-level shapes of config 2 (C = 256) and config 5 (C = 128): alternated rounds, results compared bit for bit, and against
geometry sample + Linear of this build.  Usage: python tools/geo_linear_ab.py [other.so]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgcdet_amd._abi import Library
from sgcdet_amd.tensor_api import TensorOps
from sgcdet_amd import ext
from sgcdet_amd.scene import make_img_meta
from sgcdet_amd.plugin.voxformer import compute_projection
libs = {"this": ext.ops()}
other = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tools/diag/libsgc_old.so")
if os.path.exists(other):
    libs["other"] = TensorOps(Library(other), "cuda")
def timed(fn, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, N, C, grid, vox, topk in [("cfg2", 40, 256, (40, 40, 16), (.16, .16, .2), 6400), ("cfg5", 100, 128, (96, 96, 32), (.08, .08, .1), 73728)]:
    ops = libs["this"]
    H, W, D = 60, 80, 12
    meta = make_img_meta(N, "scannet", 0)
    proj = compute_projection(meta).float().cuda().contiguous()
    origin = torch.tensor(meta["lidar2img"]["origin"]).cuda()
    g = torch.Generator().manual_seed(0)
    nx, ny, nz = grid
    idx = torch.randperm(nx * ny * nz, generator=g)[:topk].sort().values
    xs = torch.stack([idx // (ny * nz), (idx // nz) % ny, idx % nz], 1).float()
    ref3d = (xs * torch.tensor(vox) - torch.tensor([nx, ny, nz]) / 2 * torch.tensor(vox)).cuda().contiguous()
    ref_cam, mask = ops.project_points(ref3d, origin, proj, 320, H * 4, 0.2, 5.0)
    pc = ops.compact_pairs(mask)
    pc = ops.bin_pairs(ref_cam, pc, H, W, 16, 22)
    n = int(pc["totals"][0])
    feat = torch.randn(N, H * W, C, device="cuda")
    dist = torch.randn(N, H * W, D, device="cuda").mul(2).softmax(-1).contiguous()
    w = torch.randn(1, 128, C, device="cuda") * 0.1
    hi, lo = ops.split_bf16(w)
    two = ops.linear_rows_bf16x3(ops.pairs_geometry_sample(feat, dist, ref_cam, pc["pair_cam"], pc["pair_q"], n, H, W), hi, lo)
    t_geo = timed(lambda: ops.pairs_geometry_sample(feat, dist, ref_cam, pc["pair_cam"], pc["pair_q"], n, H, W))
    geo = ops.pairs_geometry_sample(feat, dist, ref_cam, pc["pair_cam"], pc["pair_q"], n, H, W)
    t_lin = timed(lambda: ops.linear_rows_bf16x3(geo, hi, lo))
    line = [f"{name}: {n} pairs, two launches {t_geo:.1f} + {t_lin:.1f} us"]
    for rnd in range(3):
        for nm, o in libs.items():
            t = timed(lambda: o.pairs_geometry_linear(feat, dist, ref_cam, pc["pair_cam"], pc["pair_q"], n, H, W, hi, lo))
            y = o.pairs_geometry_linear(feat, dist, ref_cam, pc["pair_cam"], pc["pair_q"], n, H, W, hi, lo)
            assert torch.equal(y, two), (name, nm)
            line.append(f"{nm} {t:.1f}")
    print(" | ".join(line), flush=True)
