"""The image FPN of plugin/fpn.py (row f-1) on the HIP kernels: 40 views, ResNet-50 strides 4 .. 32 at 256 x 320, channels-last in and
out.  Alternates the 2-D form of the halo kernel (`halo_2d` 1) with the tile kernel (0) for the 3 x 3 output convolutions."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sgcdet_amd.plugin  # noqa: F401
from sgcdet_amd import ext
from sgcdet_amd.mmcv_lite import NECKS
ops = ext.ops()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
fpn = NECKS.build(dict(type="FPN", in_channels=[256, 512, 1024, 2048], out_channels=256, num_outs=4)).cuda().eval()
g = torch.Generator().manual_seed(0)
feats = [torch.randn(N, c, 256 // s, 320 // s, generator=g).cuda().contiguous(memory_format=torch.channels_last)
         for c, s in zip((256, 512, 1024, 2048), (4, 8, 16, 32))]
def run():
    with torch.no_grad():
        return fpn(feats)
def timed(n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    run(); torch.cuda.synchronize(); e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
outs = {}
for rnd in range(3):
    line = []
    for m in (1, 0):
        ops.lib.call("sgc_set_tuning", b"halo_2d", m)
        line.append(f"halo_2d={m} {timed():6.3f} ms")
        outs[m] = [o.float().clone() for o in run()]
    print(f"round {rnd}: " + " | ".join(line), flush=True)
ops.lib.call("sgc_set_tuning", b"halo_2d", 1)
print("max |halo - tile| / scale per level:", [f"{float((a - b).abs().max() / b.abs().max()):.1e}" for a, b in zip(outs[1], outs[0])])
