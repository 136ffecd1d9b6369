"""Sweep of the LDS-tiled deformable gather (dfa3d_tile.hip) against the wave kernel on the finest-level shapes of
config 2 (C = 256, 40 views, 40x40x16 voxels, top-k 6400) and config 4 (C = 128, 50 views, 80x80x32, top-k 51200).
One process, interleaved rounds, HIP events.  Usage: python tools/tile_bench.py cfg2|cfg4 [HxW] [offsets=ring|rand]"""
import os
import sys
import math

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgcdet_amd import ext
from sgcdet_amd.scene import make_img_meta
from sgcdet_amd.plugin.voxformer import compute_projection
from tests.tile_contract import raw_to_headmajor, value_to_headmajor

which = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
HW = sys.argv[2] if len(sys.argv) > 2 else "64x80"
offsets = sys.argv[3] if len(sys.argv) > 3 else "ring"
H, W = (int(v) for v in HW.split("x"))
D, M, P = 12, 8, 4
if which == "cfg2":
    N, C, grid, vox, topk = 40, 256, (40, 40, 16), (.16, .16, .2), 6400
else:
    N, C, grid, vox, topk = 50, 128, (80, 80, 32), (.08, .08, .1), 51200
ops = ext.ops()
if os.environ.get("SGC_DIAG_LIB"):      # another build of the library (A/B of two builds on one box)
    from sgcdet_amd._abi import Library
    from sgcdet_amd.tensor_api import TensorOps
    ops = TensorOps(Library(os.environ["SGC_DIAG_LIB"]), "cuda")
dev = "cuda"
meta = make_img_meta(N, "scannet", 0)
proj = compute_projection(meta).float().to(dev).contiguous()
origin = torch.tensor(meta["lidar2img"]["origin"]).to(dev)
g = torch.Generator().manual_seed(0)
nx, ny, nz = grid
idx = torch.randperm(nx * ny * nz, generator=g)[:topk].sort().values
xs = torch.stack([idx // (ny * nz), (idx // nz) % ny, idx % nz], 1).float()
ref3d = (xs * torch.tensor(vox) - torch.tensor([nx, ny, nz]) / 2 * torch.tensor(vox)).to(dev).contiguous()
img_h = 239 if H in (59, 60) else H * 4
ref_cam, mask = ops.project_points(ref3d, origin, proj, 320, img_h, 0.2, 5.0)
pc = ops.compact_pairs(mask)
n_pairs = int(pc["totals"][0])
Cm = C // M
value = torch.randn(N, H * W, M, Cm, device=dev)
dist = torch.randn(N, H * W, D, device=dev).mul(2).softmax(-1).contiguous()
raw = torch.randn(n_pairs, M * P * 4, device=dev)
if offsets == "ring":        # the reference's init (deformable_cross_attention.py:194-212) + noise, as bench.py's weights
    th = torch.arange(M, dtype=torch.float32) * (2 * math.pi / M)
    ring = torch.stack([th.cos(), th.sin()], -1)
    ring = ring / ring.abs().max(-1, keepdim=True)[0]
    steps = torch.arange(1, P + 1, dtype=torch.float32)
    uvb = (ring.view(M, 1, 2) * steps.view(1, P, 1)).reshape(-1)
    dzb = (((th.cos() + th.sin()) / 2).view(M, 1) * steps.view(1, P)).reshape(-1)
    noise = float(os.environ.get("SGC_RING_NOISE", "0.3"))
    raw[:, :M * P * 2] = uvb.to(dev) + torch.randn(n_pairs, M * P * 2, device=dev) * noise
    raw[:, M * P * 2:M * P * 3] = dzb.to(dev) + torch.randn(n_pairs, M * P, device=dev) * noise
    head_shift = torch.round(ring * steps.mean()).to(torch.int32).to(dev).contiguous()
else:
    raw[:, :M * P * 2] *= float(os.environ.get("SGC_OFFSET_SCALE", "2.0"))
    head_shift = None
vhm = value_to_headmajor(value)
alg = N * H * W * C * 4 + N * H * W * D * 4 + n_pairs * 512 + n_pairs * C * 4
print(f"{which} {H}x{W} C={C} pairs {n_pairs} ({n_pairs / N:.0f}/camera) algorithmic bytes {alg / 1e6:.1f} MB  offsets={offsets}")
dp = ops.depth_pairs(dist, H, W)
vbuf = torch.cat([value.reshape(N * H * W, C), torch.zeros(1, C, device=dev)])
vzr = vbuf[:N * H * W].view(N, H * W, M, Cm)
slot0 = pc["slot"].clone()


def run_wave():
    return ops.pairs_deform_gather(vzr, dist, ref_cam, raw, pc["pair_cam"], pc["pair_q"], n_pairs, H, W, M, P,
                                   dist_pairs=dp, zero_row=True)


def timed(fn, rounds=8):
    ts = []
    for r in range(rounds + 2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        o = fn()
        e1.record()
        torch.cuda.synchronize()
        if r >= 2:
            ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], o


t, ref = timed(run_wave)
print(f"wave kernel                         {t:8.1f} us  {alg / t / 1e3 / 8000:.3f} of 8 TB/s")
if os.environ.get("SGC_WAVE_SORTED"):      # the wave kernel on spatially sorted pair lists (L1 / L2 locality only)
    for bw, bh in [(80, 64), (16, 16), (8, 8), (4, 4), (2, 2), (8, 2), (2, 8)]:
        b = ops.bin_pairs(ref_cam, dict(pc, slot=slot0.clone()), H, W, bw, bh)
        old = slot0.long()[pc["pair_cam"][:n_pairs].long(), b["pair_q"][:n_pairs].long()]
        raw_s = torch.zeros_like(raw)
        raw_s[:n_pairs] = raw[old]
        tt, o = timed(lambda: ops.pairs_deform_gather(vzr, dist, ref_cam, raw_s, pc["pair_cam"], b["pair_q"], n_pairs, H, W, M, P,
                                                      dist_pairs=dp, zero_row=True))
        print(f"wave kernel, pairs sorted by {bw}x{bh} bins   {tt:8.1f} us  {alg / tt / 1e3 / 8000:.3f}   err {(o[:n_pairs] - ref[old]).abs().max().item():.1e}")
# config = bw, bh, hx, hy, depth_lds, nw, shift, nbuf, hg
configs = []
for spec in os.environ.get("SGC_TILE_CONFIGS", "").split(";"):
    if spec:
        configs.append(tuple(int(v) for v in spec.split(",")))
max_shift = (int(head_shift[:, 0].abs().max()), int(head_shift[:, 1].abs().max())) if head_shift is not None else (0, 0)
res = []
for cfg in configs:
    bw, bh, hx, hy, dl, nw, sh, nbuf, hg = cfg[:9]
    xcd = cfg[9] if len(cfg) > 9 else int(os.environ.get("SGC_TILE_XCD", "0"))        # optional 10th field: tile_xcd
    ds = cfg[10] if len(cfg) > 10 else 1                                               # optional 11th field: tile_ds (one window test for value and depth)
    for key, val in (("tile_nw", nw), ("tile_nbuf", nbuf), ("tile_hg", hg), ("tile_xcd", xcd), ("tile_ds", ds)):
        ops.lib.call("sgc_set_tuning", key.encode(), val)
    hs = head_shift if sh else None
    ms = max_shift if sh else (0, 0)
    geo = ops.tile_window(H, W, Cm, D, bw, bh, hx, hy, ms, depth_in_lds=bool(dl))
    if geo["lds_bytes"] > 160 * 1024:
        print("skip (LDS)", (bw, bh, hx, hy, dl, nw, sh, nbuf, hg), geo)
        continue

    def do_bin():
        return ops.bin_pairs(ref_cam, dict(pc, slot=slot0.clone()), H, W, bw, bh)
    tb, binned = timed(do_bin, rounds=3)
    old = slot0.long()[pc["pair_cam"][:n_pairs].long(), binned["pair_q"][:n_pairs].long()]
    raw_new = torch.zeros_like(raw)
    raw_new[:n_pairs] = raw[old]
    rhm = raw_to_headmajor(raw_new, M, P)
    call = lambda: ops.pairs_deform_gather_tiled(vhm, dist, binned["pair_ref"], binned["bin_offset"], rhm, H, W, P, bw, bh,  # noqa: E731
                                                 hx, hy, head_shift=hs, max_shift=ms, depth_in_lds=bool(dl))
    try:
        t, o = timed(call)
    except Exception as e:      # noqa: BLE001
        print("failed", (bw, bh, hx, hy, dl, nw, sh, nbuf, hg), e)
        continue
    err = (o[:n_pairs] - ref[old]).abs().max().item()
    line = (f"tile bin {bw:2d}x{bh:2d} halo {hx},{hy} dl {dl}->{geo['depth_in_lds']} nw {nw:2d} shift {sh} nbuf {nbuf}->{geo['nbuf']} hg {hg} xcd {xcd} ds {ds} "
            f"win {geo['tw']}x{geo['th']} ({geo['lds_bytes'] / 1024:5.1f} KB) {t:8.1f} us  {alg / t / 1e3 / 8000:.3f}  (bin {tb - 0:5.1f} us incl. slot clone)  "
            f"err {err:.1e}")
    if os.environ.get("SGC_TILE_DIAG"):
        extra = []
        for dg in (1, 2):
            ops.lib.call("sgc_set_tuning", b"tile_diag", dg)
            td, _ = timed(call)
            extra.append(td)
        ops.lib.call("sgc_set_tuning", b"tile_diag", 0)
        line += f"  fill-only {extra[0]:6.1f}  compute-only {extra[1]:6.1f}"
    print(line, flush=True)
    res.append((t, bw, bh, hx, hy, dl, nw, sh, nbuf, hg))
res.sort()
print("best:")
for r in res[:6]:
    print("  ", r)
