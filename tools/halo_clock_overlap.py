"""Clock and tap-loop cycles of the halo convolution (diagnostic build with SGC_HALO_STAMPS) when ONE stream runs it (200 of 256
CUs busy) against TWO streams (all 256 CUs busy): is the 14 % longer workgroup under full occupancy a lower clock or more cycles?"""
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
Cin = Cout = 256; g = (40, 40, 16); V = g[0] * g[1] * g[2]
xs = [torch.randn(V, Cin, device="cuda") for _ in range(2)]
wt = torch.randn(27, Cout, Cin, device="cuda") * 0.01
sc = torch.ones(Cout, device="cuda"); sh = torch.zeros(Cout, device="cuda")
wh, wl = ops.split_bf16(wt)
streams = [torch.cuda.Stream() for _ in range(2)]
def burst(ns, n):
    for i in range(n):
        with torch.cuda.stream(streams[i % ns]):
            ops.conv3d_cl_bf16x3(xs[i % 2], wh, wl, g, 3, 1, False, sc, sh, None, True)
    torch.cuda.synchronize()
for rnd in range(3):
    for ns in (1, 2):
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 1.5: burst(ns, 40)
        buf.zero_()
        t1 = time.perf_counter(); burst(ns, 40); dt = (time.perf_counter() - t1) / 40 * 1e6
        s = buf.view(-1, 4).cpu(); s = s[s[:, 0] != 0]          # the stamps of the last launch that used each slot
        cyc = (s[:, 2] - s[:, 0]).double(); rt = (s[:, 3] - s[:, 1]).double()
        clk = cyc / rt * 100.0
        print(f"round {rnd}, {ns} stream(s): {dt:6.1f} us/launch; tap loop {cyc.median():.0f} cycles, {rt.median() / 100:.1f} us, clock {clk.median():.0f} MHz", flush=True)
