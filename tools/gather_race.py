"""Does the gather kernel stay deterministic while another stream keeps the chip busy?"""
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
# heavy work for the other stream
xb = torch.randn(25600, 256, device=dev); wb = torch.randn(27, 256, 256, device=dev) * 0.01
wh, wl = ops.split_bf16(wb)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def gather(variant, zr, use_dp):
    ops.lib.call("sgc_set_tuning", b"fwd_variant", variant)
    return ops.pairs_deform_gather(value, dist, ref_cam, raw, pc["pair_cam"], pc["pair_q"], n_pairs, H, W, M, P,
                                   dist_pairs=dp if use_dp else None, zero_row=zr)
for variant, zr, use_dp in [(1, True, True), (1, False, True), (1, False, False), (0, False, False)]:
    ref = gather(variant, zr, use_dp).clone()
    torch.cuda.synchronize()
    bad = 0
    for it in range(600):
        with torch.cuda.stream(s2):
            ops.conv3d_cl_bf16x3(xb, wh, wl, (40, 40, 16), 3, 1, False)
        with torch.cuda.stream(s1):
            outs = [gather(variant, zr, use_dp) for _ in range(4)]
        torch.cuda.synchronize()
        bad += sum(int(not torch.equal(o, ref)) for o in outs)
    print(f"variant {variant} zero_row {zr} dist_pairs {use_dp}: {bad} / 2400 mismatching launches")
