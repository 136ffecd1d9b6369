#!/bin/bash
python - <<'PY'
import sys, time, torch
sys.path.insert(0, ".")
from sgcdet_amd import ext
ops = ext.ops()
for name, cin, cout, g in [("128->28 @40x40x16", 128, 28, (40, 40, 16)), ("128->28 @20x20x8", 128, 28, (20, 20, 8)), ("128->32 @40x40x16", 128, 32, (40, 40, 16)), ("256->64 @40x40x16", 256, 64, (40, 40, 16))]:
    V = g[0] * g[1] * g[2]
    x = torch.randn(V, cin, device="cuda"); wt = torch.randn(27, cout, cin, device="cuda") * 0.01
    sc = torch.rand(cout, device="cuda") + 0.5; sh = torch.randn(cout, device="cuda")
    wh, wl = ops.split_bf16(wt)
    res, outs = {0: [], 1: []}, {}
    for r in range(6):
        for v in (0, 1):
            ops.lib.call("sgc_set_tuning", b"halo_narrow", v)
            for _ in range(3): y = ops.conv3d_cl_bf16x3(x, wh, wl, g, 3, 1, False, sc, sh, None, True)[0]
            outs[v] = y.clone()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(40): ops.conv3d_cl_bf16x3(x, wh, wl, g, 3, 1, False, sc, sh, None, True)
            torch.cuda.synchronize(); res[v].append((time.perf_counter() - t0) / 40 * 1e6)
    print(f"{name:22s} 128-col tiles {sorted(res[0])[3]:6.1f} us | 64-col tiles {sorted(res[1])[3]:6.1f} us | identical {torch.equal(outs[0], outs[1])}", flush=True)
PY
timeout 600 python -m pytest tests/test_gpu_conv3d.py -x -q 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_modules.py -x -q -k "hot_path or masked or get_bboxes" 2>&1 | tail -2
