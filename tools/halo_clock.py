"""Clock the chip holds while the halo convolution runs (diagnostic build: bash tools/diag_build.sh halostamps conv3d.hip
-DSGC_HALO_STAMPS): per workgroup, shader cycles (s_memtime) over real time (s_memrealtime, 100 MHz) around the tap loop,
after >= 2 s of back-to-back launches on random data; plus the same for interleaved launches of two layers."""
import ctypes, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgcdet_amd._abi import Library
from sgcdet_amd.tensor_api import TensorOps
so = os.path.join(ROOT, "tools/diag/libsgc_halostamps.so")
ops = TensorOps(Library(so), "cuda")
raw = ctypes.CDLL(so)
buf = torch.zeros(4096 * 4, dtype=torch.int64, device="cuda")
raw.sgc_diag_halo_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
for name, Cin, Cout, g in [("256->256 @40x40x16", 256, 256, (40, 40, 16)), ("512->512 @20x20x8", 512, 512, (20, 20, 8)),
                           ("256->128 @40x40x16", 256, 128, (40, 40, 16))]:
    V = g[0] * g[1] * g[2]
    x = torch.randn(V, Cin, device="cuda"); wt = torch.randn(27, Cout, Cin, device="cuda") * 0.01
    sc = torch.ones(Cout, device="cuda"); sh = torch.zeros(Cout, device="cuda")
    wh, wl = ops.split_bf16(wt)
    f = lambda: ops.conv3d_cl_bf16x3(x, wh, wl, g, 3, 1, False, sc, sh, None, True)
    for secs in (0.0, 2.0):
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < secs:
            for _ in range(50): f()
            torch.cuda.synchronize()
        buf.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        s = buf.view(-1, 4).cpu()
        s = s[s[:, 0] != 0]
        cyc = (s[:, 2] - s[:, 0]).double(); rt = (s[:, 3] - s[:, 1]).double()
        clk = (cyc / rt * 100.0)           # MHz
        fl = 2.0 * Cin * Cout * V * 27
        print(f"{name}: after {secs:.0f} s of launches: {us:7.1f} us/launch ({fl * 3 / us / 1e6 / 1e3:6.1f} TF issued), {len(s)} workgroups, "
              f"tap loop {cyc.median():.0f} cycles / {rt.median() / 100:.1f} us, clock median {clk.median():.0f} MHz "
              f"(min {clk.min():.0f}, max {clk.max():.0f}); MFMA-only floor of the loop at that clock: "
              f"{27 * (Cin // 32) * 24 * 32 * 2 / clk.median():.1f} us", flush=True)
