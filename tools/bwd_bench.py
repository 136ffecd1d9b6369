"""The DFA3D backward (row a11b) at the config-2 finest-level shape in the reference's padded-rebatch layout
(B = cameras, Q = max_len): HIP events around sgc_dfa3d_backward.  SGC_DIAG_LIB=<.so> times a diagnostic build.
Usage: python tools/bwd_bench.py [n_views]"""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgcdet_amd import ext
from sgcdet_amd.scene import make_img_meta
from sgcdet_amd.plugin.voxformer import compute_projection

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
C, H, W, D, M, P = 256, 64, 80, 12, 8, 4
ops = ext.ops()
if os.environ.get("SGC_DIAG_LIB"):
    from sgcdet_amd._abi import Library
    from sgcdet_amd.tensor_api import TensorOps
    ops = TensorOps(Library(os.environ["SGC_DIAG_LIB"]), "cuda")
dev = "cuda"
meta = make_img_meta(N, "scannet", 0, img_hw=(256, 320))
proj = compute_projection(meta).float().to(dev).contiguous()
origin = torch.tensor(meta["lidar2img"]["origin"]).to(dev)
g = torch.Generator().manual_seed(0)
nx, ny, nz = 40, 40, 16
idx = torch.randperm(nx * ny * nz, generator=g)[:6400].sort().values
xs = torch.stack([idx // (ny * nz), (idx // nz) % ny, idx % nz], 1).float()
ref3d = (xs * torch.tensor([.16, .16, .2]) - torch.tensor([nx, ny, nz]) / 2 * torch.tensor([.16, .16, .2])).to(dev).contiguous()
ref_cam, mask = ops.project_points(ref3d, origin, proj, 320, 256, 0.2, 5.0)
mask = mask.bool()
counts = mask.sum(1)
Q = int(counts.max())
order = torch.argsort((~mask).to(torch.uint8), dim=1, stable=True)[:, :Q]
live = torch.arange(Q, device=dev)[None] < counts[:, None]
ref = torch.gather(ref_cam, 1, order[..., None].expand(-1, -1, 3)) * live[..., None]
off = torch.randn(N, Q, M, 1, P, 3, device=dev) * torch.tensor([2.0 / W, 2.0 / H, 1.0 / D], device=dev)
loc = (ref.view(N, Q, 1, 1, 1, 3) + off).contiguous()
attn = torch.randn(N, Q, M, 1, P, device=dev).softmax(-1).contiguous()
value = torch.randn(N, H * W, M, C // M, device=dev)
dist = torch.randn(N, H * W, 1, D, device=dev).mul(2).softmax(-1).contiguous()
shapes3 = torch.tensor([[H, W, D]], device=dev); lsi = torch.zeros(1, dtype=torch.int64, device=dev)
gout = torch.randn(N, Q, C, device=dev) * live[..., None]
print(f"items {N * Q} (live {int(counts.sum())}), samples {N * Q * M * P}")
for _ in range(2): ops.dfa3d_backward(value, dist, shapes3, lsi, loc, attn, gout)
ops.event_log = []; ops.event_names = {"sgc_dfa3d_backward"}
for _ in range(5): r = ops.dfa3d_backward(value, dist, shapes3, lsi, loc, attn, gout)
torch.cuda.synchronize()
ts = [e0.elapsed_time(e1) for _, _m, e0, e1 in ops.event_log]
fw = ops.dfa3d_forward(value, dist, shapes3, lsi, loc, attn)[0]
print(json.dumps(dict(bwd_ms=round(sum(ts) / len(ts), 3), grad_value_abs_sum=float(r[0].abs().sum()), grad_loc_abs_sum=float(r[2].abs().sum()))))
