import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgcdet_amd import ext
ops = ext.ops()
if os.environ.get("SGC_DIAG_LIB"):      # diagnostic builds (tools/diag): timing only
    from sgcdet_amd._abi import Library
    from sgcdet_amd.tensor_api import TensorOps
    ops = TensorOps(Library(os.environ["SGC_DIAG_LIB"]), "cuda")
for rows, cin, cout in [(188800, 256, 256), (77000, 256, 128), (77000, 256, 512), (6400, 256, 256), (6400, 512, 256)]:
    x = torch.randn(rows, cin, device="cuda"); wt = torch.randn(1, cout, cin, device="cuda") * 0.05
    sh = torch.randn(cout, device="cuda"); wh, wl = ops.split_bf16(wt)
    for _ in range(3): y = ops.linear_rows_bf16x3(x, wh, wl, sh)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): y = ops.linear_rows_bf16x3(x, wh, wl, sh)
    torch.cuda.synchronize(); print(f"{rows:7d} x {cin} -> {cout}: {(time.perf_counter()-t)/20*1e6:7.1f} us")
