"""The LDS-tiled DFA3D backward (sgc_dfa3d_backward_binned) against the item kernel (sgc_dfa3d_backward_items) on the finest-level
shapes of config 2 (C = 256, 40 views, 6400 voxels) and config 4 (C = 128, 50 views, 51200 voxels): the deformable call (8 heads x 4
points) and the geometry sample (one shared sample, 8 channel groups).  One process, interleaved rounds, HIP events.
Usage: python tools/bwd_tile_bench.py cfg2|cfg4 [HxW]      env SGC_BWD_CONFIGS="bw,bh,hx,hy,nw[,diag];..." (diag needs SGC_DIAG=1)"""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgcdet_amd import ext
from sgcdet_amd.scene import make_img_meta
from sgcdet_amd.plugin.voxformer import compute_projection

which = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
H, W = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "64x80").split("x"))
D, M, P = 12, 8, 4
N, C, grid, vox, topk = (40, 256, (40, 40, 16), (.16, .16, .2), 6400) if which == "cfg2" else (50, 128, (80, 80, 32), (.08, .08, .1), 51200)
ops = ext.ops()
dev = "cuda"
meta = make_img_meta(N, "scannet", 0)
proj = compute_projection(meta).float().to(dev).contiguous()
origin = torch.tensor(meta["lidar2img"]["origin"]).to(dev)
g = torch.Generator().manual_seed(0)
nx, ny, nz = grid
idx = torch.randperm(nx * ny * nz, generator=g)[:topk].sort().values
xs = torch.stack([idx // (ny * nz), (idx // nz) % ny, idx % nz], 1).float()
ref3d = (xs * torch.tensor(vox) - torch.tensor([nx, ny, nz]) / 2 * torch.tensor(vox)).to(dev).contiguous()
ref_cam, mask = ops.project_points(ref3d, origin, proj, 320, H * 4, 0.2, 5.0)
pc0 = ops.compact_pairs(mask)
n = int(pc0["totals"][0])
Cm = C // M
S = H * W
value = torch.randn(N, S, M, Cm, device=dev)
dist = torch.randn(N, S, 1, D, device=dev).mul(2).softmax(-1).contiguous()
shapes3 = torch.tensor([[H, W, D]], dtype=torch.int64, device=dev)
lsi = torch.zeros(1, dtype=torch.int64, device=dev)
# the reference's initial sampling offsets (a ring per head, steps 1..P) + noise, in pixels
th = torch.arange(M, dtype=torch.float32) * (2 * math.pi / M)
ring = torch.stack([th.cos(), th.sin()], -1)
ring = ring / ring.abs().max(-1, keepdim=True)[0]
steps = torch.arange(1, P + 1, dtype=torch.float32)
off = torch.cat([ring.view(M, 1, 2) * steps.view(1, P, 1), (((th.cos() + th.sin()) / 2).view(M, 1) * steps.view(1, P)).unsqueeze(-1)], -1).to(dev)
norm = torch.tensor([W, H, D], dtype=torch.float32, device=dev)
head_shift = torch.round(ring * steps.mean()).to(torch.int32).to(dev).contiguous()     # what the module derives from the offsets' bias
use_shift = os.environ.get("SGC_BWD_SHIFT", "1") != "0" 
print(f"{which} {H}x{W} C={C} pairs {n}")


def timed(fn, rounds=6):
    ts = []
    for r in range(rounds + 2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); o = fn(); e1.record(); torch.cuda.synchronize()
        if r >= 2:
            ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], o


def inputs(pc):
    cam, q = pc["pair_cam"][:n].long(), pc["pair_q"][:n].long()
    ref = ref_cam[cam, q]
    gen = torch.Generator(device=dev).manual_seed(1)
    loc = (ref.view(n, 1, 1, 1, 3) + (off.view(1, M, 1, P, 3) + torch.randn(n, M, 1, P, 3, device=dev, generator=gen) * 0.3) / norm).contiguous()
    attn = torch.rand(n, M, 1, P, device=dev, generator=gen).softmax(-1).contiguous()
    go = torch.randn(n, C, device=dev, generator=gen)
    return cam.to(torch.int32), ref.view(n, 1, 1, 1, 3).contiguous(), loc, attn, go


item, loc1, loc, attn, go = inputs(pc0)
ones = torch.ones(n, 1, 1, 1, device=dev)
t_d, ref_d = timed(lambda: ops.dfa3d_backward_items(value, dist, shapes3, lsi, loc, attn, item, go))
t_g, ref_g = timed(lambda: ops.dfa3d_backward_items(value.view(N, S, 1, C), dist, shapes3, lsi, loc1, ones, item, go))
print(f"item kernel (pairs in ascending-voxel order): deformable call {t_d:8.1f} us   geometry sample {t_g:8.1f} us   (incl. the zero fills of the two gradients)")
configs = [tuple(int(v) for v in spec.split(",")) for spec in os.environ.get("SGC_BWD_CONFIGS", "16,22,3,3,8").split(";") if spec]
for cfg in configs:
    bw, bh, hx, hy, nw = cfg[:5]
    diag = cfg[5] if len(cfg) > 5 else 0
    if not ops.dfa3d_backward_binned_fits(H, W, Cm if Cm in (16, 32) else 32, D, bw, bh, (hx, hy)):
        print("skip (LDS)", cfg); continue
    ops.lib.call("sgc_set_tuning", b"bwd_tile_diag", diag)
    pc = ops.bin_pairs(ref_cam, dict(pc0, slot=pc0["slot"].clone()), H, W, bw, bh)
    item_b, loc1_b, loc_b, attn_b, go_b = inputs(pc)
    t_i, _ = timed(lambda: ops.dfa3d_backward_items(value, dist, shapes3, lsi, loc_b, attn_b, item_b, go_b))
    t1, o1 = timed(lambda: ops.dfa3d_backward_binned(value, dist, loc_b, attn_b, pc["bin_offset"], go_b, H, W, bw, bh, (hx, hy),
                                                     head_shift=head_shift if use_shift else None))
    t2, o2 = timed(lambda: ops.dfa3d_backward_binned(value, dist, loc1_b, None, pc["bin_offset"], go_b, H, W, bw, bh, (hx, hy),
                                                      want_grad_loc=False, want_grad_attn=False))
    cnt = (pc["bin_offset"][1:] - pc["bin_offset"][:-1]).float()
    print(f"bins {bw:2d}x{bh:2d} halo {hx},{hy} nw {nw:2d} diag {diag:2d}: deformable {t1:8.1f} us  geometry {t2:8.1f} us   item kernel on the binned order {t_i:8.1f} us   "
          f"pairs per (camera, bin): mean {cnt.mean().item():.0f} max {cnt.max().item():.0f} empty {(cnt == 0).float().mean().item():.2f}", flush=True)
ops.lib.call("sgc_set_tuning", b"bwd_tile_diag", 0)
