"""Gather determinism next to hipGraph replays (same stream / other stream / both)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgcdet_amd import ext
from sgcdet_amd.scene import make_img_meta
from sgcdet_amd.plugin.voxformer import compute_projection
ops = ext.ops()
dev = "cuda"
N, C, H, W, D, M, P = 4, 256, 3, 5, 12, 8, 4
meta = make_img_meta(N, "scannet", 0)
proj = compute_projection(meta).float().to(dev).contiguous()
origin = torch.tensor(meta["lidar2img"]["origin"]).to(dev)
nx, ny, nz = 5, 5, 2
idx = torch.arange(nx * ny * nz)
xs = torch.stack([idx // (ny * nz), (idx // nz) % ny, idx % nz], 1).float()
ref3d = (xs * torch.tensor([.64, .64, .8]) - torch.tensor([nx, ny, nz]) / 2 * torch.tensor([.64, .64, .8])).to(dev).contiguous()
ref_cam, mask = ops.project_points(ref3d, origin, proj, 320, 239, 0.2, 5.0)
pc = ops.compact_pairs(mask)
n_pairs = int(pc["totals"][0])
g = torch.Generator().manual_seed(0)
vbuf = torch.randn(N * H * W + 1, C, generator=g).to(dev); vbuf[-1].zero_()
value = vbuf[:N * H * W].view(N, H * W, M, C // M)
dist = torch.randn(N, H * W, D, generator=g).mul(2).softmax(-1).contiguous().to(dev)
raw = torch.randn(n_pairs, M * P * 4, generator=g).to(dev)
dp = ops.depth_pairs(dist, H, W)
print("pairs", n_pairs)
xb = torch.randn(25600, 256, device=dev); wb = torch.randn(27, 256, 256, device=dev) * 0.01
xs_ = torch.randn(1600, 256, device=dev)
wh, wl = ops.split_bf16(wb)
def heavy():
    y = ops.conv3d_cl_bf16x3(xb, wh, wl, (40, 40, 16), 3, 1, False)
    z = ops.conv3d_cl_bf16x3(xs_, wh, wl, (10, 10, 16), 3, 1, False)     # small grid: split-K + memset node
    return y, z
def make_graph():
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        heavy()
    torch.cuda.current_stream().wait_stream(side)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        outs = heavy()
    return gr, outs
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
with torch.cuda.stream(s1):
    g1 = make_graph()
with torch.cuda.stream(s2):
    g2 = make_graph()
torch.cuda.synchronize()
def gather():
    return ops.pairs_deform_gather(value, dist, ref_cam, raw, pc["pair_cam"], pc["pair_q"], n_pairs, H, W, M, P,
                                   dist_pairs=dp, zero_row=True)
ref = gather().clone()
torch.cuda.synchronize()
def run(mode, iters=500):
    bad = 0
    for it in range(iters):
        with torch.cuda.stream(s1):
            if mode in ("same", "both"):
                g1[0].replay()
            outs = [gather() for _ in range(3)]
        with torch.cuda.stream(s2):
            if mode in ("other", "both"):
                g2[0].replay()
            if mode == "both":
                outs += [gather() for _ in range(3)]
        if mode == "eager_same":
            with torch.cuda.stream(s1):
                heavy(); outs += [gather() for _ in range(3)]
        torch.cuda.synchronize()
        bad += sum(int(not torch.equal(o, ref)) for o in outs)
    print(f"mode {mode}: {bad} mismatching gathers in {iters} iterations", flush=True)
for mode in ("same", "other", "both", "eager_same"):
    run(mode)
