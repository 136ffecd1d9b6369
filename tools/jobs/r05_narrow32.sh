#!/bin/bash
# round 5: 32-column tiles (8 x 1 waves) for the head's 28-column convolution -- bit-identity, then the layer alone and the bench A/B
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_conv3d.py -x -q -k "32_column or head_activation or staggered or masked" 2>&1 | tail -2
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_narrow32_ab.txt
import torch, sys, os
sys.path.insert(0, os.getcwd())
from sgcdet_amd import ext
ops = ext.ops()
def timed(fn, n=30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("head convolution (fused centerness | reg | cls), halo kernel: 64-column tiles (halo_narrow 64) vs 32-column tiles (1), alternated")
for name, Cin, Cout, g in (("128->28 @40x40x16", 128, 28, (40, 40, 16)), ("128->28 @20x20x8", 128, 28, (20, 20, 8)), ("128->28 @96x96x32", 128, 28, (96, 96, 32)),
                           ("128->200 @40x40x16 (unaffected)", 128, 200, (40, 40, 16))):
    V = g[0] * g[1] * g[2]
    x = torch.randn(V, Cin, device="cuda"); wt = torch.randn(27, Cout, Cin, device="cuda") * 0.01
    wh, wl = ops.split_bf16(wt)
    line = []
    for rnd in range(4):
        for form in (64, 1):
            ops.lib.call("sgc_set_tuning", b"halo_narrow", form)
            t = timed(lambda: ops.conv3d_cl_bf16x3(x, wh, wl, g, 3, 1, False, None, None, None, 0))
            if rnd: line.append(f"{form}: {t:6.1f}")
    print(f"{name:34s} " + " | ".join(line))
ops.lib.call("sgc_set_tuning", b"halo_narrow", 1)
PY
for f in 64 1; do
SGC_TUNE="halo_narrow=$f" timeout 600 python bench.py --no-cpu-baseline --no-strict-fp32 --steps 60 --warmup 15 > gpurun_out/r05_narrow${f}_cfg2.json 2>/dev/null
done
python - <<'PY'
import json
for f in (64, 1):
    d = json.loads(open(f"gpurun_out/r05_narrow{f}_cfg2.json").readline())
    print("cfg2 halo_narrow", f, d["value"], "sustained", d["sustained"]["value"], "self_check", d["self_check"]["mismatching"])
PY
