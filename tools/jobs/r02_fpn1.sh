#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_conv3d.py -x -q -k "conv2d" 2>&1 | tail -8
timeout 600 python -m pytest tests/test_gpu_modules.py -x -q -k "fpn_on_hip" 2>&1 | tail -15
timeout 300 python - <<'PY'
import torch, time, sys
sys.path.insert(0, ".")
from sgcdet_amd.plugin.fpn import FPN
fpn = FPN([256, 512, 1024, 2048], 256, 4).eval().cuda(); fpn.init_weights()
N = 40
feats = [torch.randn(N, c, h, w, device="cuda") for c, (h, w) in zip([256, 512, 1024, 2048], [(64, 80), (32, 40), (16, 20), (8, 10)])]
feats_cl = [f.contiguous(memory_format=torch.channels_last) for f in feats]
def timed(f, n=5):
    for _ in range(2): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
with torch.no_grad():
    print("FPN 40 views: HIP (NCHW inputs) %.2f ms | HIP (channels-last inputs) %.2f ms | torch/MIOpen NCHW %.2f ms | torch channels_last %.2f ms" % (
        timed(lambda: fpn(feats)), timed(lambda: fpn(feats_cl)), timed(lambda: fpn._forward_torch(feats)), timed(lambda: fpn._forward_torch(feats_cl))))
PY
