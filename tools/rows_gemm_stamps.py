"""In-kernel cycle stamps of the staggered row GEMM (diagnostic build: bash tools/diag_build.sh rgstamps rows_gemm.hip -DSGC_RG_STAMPS)."""
import ctypes, os, sys, torch
os.environ["SGC_DIAG"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgcdet_amd._abi import Library
from sgcdet_amd.tensor_api import TensorOps
lib = Library(os.path.join(ROOT, "tools/diag/libsgc_rgstamps.so"))
ops = TensorOps(lib, "cuda")
raw = ctypes.CDLL(os.path.join(ROOT, "tools/diag/libsgc_rgstamps.so"))
buf = torch.zeros(8 * 2 * 32 * 8, dtype=torch.int64, device="cuda")
raw.sgc_diag_rows_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
rows, cin, cout = 204800, 256, 256
x = torch.randn(rows, cin, device="cuda"); wt = torch.randn(1, cout, cin, device="cuda") * 0.05
sh = torch.randn(cout, device="cuda"); wh, wl = ops.split_bf16(wt); y = torch.empty(rows, cout, device="cuda")
for d, name in [(0, "full (setprio)"), (8, "full, no setprio"), (3, "no loads+stores (setprio)"), (11, "no loads+stores, no setprio")]:
    lib.call("sgc_set_tuning", b"rows_diag", d)
    for _ in range(5): ops.linear_rows_bf16x3(x, wh, wl, sh, out=y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.linear_rows_bf16x3(x, wh, wl, sh, out=y)
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch (stamped build)")
    s = buf.view(8, 2, 32, 8).cpu()
    print(f"--- {name}: cycles per phase (median over iterations 3..22, workgroups 0..7)")
    for late in (0, 1):
        seg = s[:, late, 3:23]                        # [wg, it, 8]
        if late == 0:
            names = [("multiply", 0, 1), ("barrier 1", 1, 4), ("split(+wait loads)", 4, 5), ("issue loads+stores", 5, 6), ("barrier 2", 6, 7), ("period", 0, 7)]
        else:
            names = [("store", 0, 1), ("split(+wait loads)", 1, 2), ("issue loads", 2, 3), ("barrier 1", 3, 4), ("multiply", 4, 5), ("barrier 2", 5, 7), ("period", 0, 7)]
        out = []
        for nm, a, b in names:
            d_ = (seg[:, :, b] - seg[:, :, a]).flatten().float()
            out.append(f"{nm} {int(d_.median())}")
        print(("  late waves : " if late else "  early waves: ") + " | ".join(out))
