#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "topk" 2>&1 | tail -5
timeout 200 python - <<'PY'
import torch, time, sys
sys.path.insert(0, ".")
from sgcdet_amd import ext
ops = ext.ops()
for n, k in [(3200, 800), (25600, 6400), (204800, 51200), (294912, 73728)]:
    s = torch.sigmoid(torch.randn(n, device="cuda"))
    out = []
    for mm in (1 << 30, 1):
        ops.lib.call("sgc_set_tuning", b"topk_multi_min", mm)
        for _ in range(3): ops.topk_select(s, k, want_valid=True, want_mask=True)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(20): ops.topk_select(s, k, want_valid=True, want_mask=True)
        torch.cuda.synchronize(); out.append((time.perf_counter() - t) / 20 * 1e6)
    print(f"n {n:7d} k {k:6d}: one workgroup {out[0]:7.1f} us | many {out[1]:7.1f} us")
PY
