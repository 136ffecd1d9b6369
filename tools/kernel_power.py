"""Power, clock and energy per launch of the path's heavy kernels, each looped alone for a few seconds with rocm-smi sampled
beside it (the throughput regime sits at the package power limit: what a kernel costs there is joules, not microseconds).
Timing builds of the halo convolution (tools/diag/libsgc_skip*.so, SGC_HALO_SKIP) show what its non-MFMA parts draw.
Usage: python tools/kernel_power.py [seconds per kernel]"""
import glob, json, os, re, subprocess, sys, threading, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgcdet_amd._abi import Library
from sgcdet_amd.tensor_api import TensorOps
from sgcdet_amd import ext

SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
ops = ext.ops()
libs = {"halo": ops}
for f in sorted(glob.glob(os.path.join(ROOT, "tools/diag/libsgc_skip*.so")), key=lambda f: int(re.findall(r"skip(\d+)", f)[0])):
    libs["halo_" + re.findall(r"(skip\d+)", f)[0]] = TensorOps(Library(f), "cuda")


def smi():
    try:
        out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5).stdout
        c = json.loads(out)["card0"]
        return int(re.findall(r"\d+", c["sclk clock speed:"])[0]), float(c["Current Socket Graphics Package Power (W)"])
    except Exception:
        return None


def measure(name, fn, flop=None, nbytes=None):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    samples, stop = [], threading.Event()

    def sampler():
        time.sleep(0.8)                          # let clocks and the power average settle
        while not stop.is_set():
            s = smi()
            if s:
                samples.append(s)
            time.sleep(0.15)
    th = threading.Thread(target=sampler)
    th.start()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < SECS:
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        n += 50
    el = time.perf_counter() - t0
    stop.set()
    th.join()
    us = el / n * 1e6
    mhz = sum(s[0] for s in samples) / max(1, len(samples))
    watt = sum(s[1] for s in samples) / max(1, len(samples))
    row = dict(kernel=name, us=round(us, 1), sclk_mhz=round(mhz), watt=round(watt), mj_per_launch=round(watt * us * 1e-3, 2),
               mj_dynamic=round((watt - 240.0) * us * 1e-3, 2), smi_samples=len(samples))
    if flop:
        row["tflops_issued"] = round(flop / us / 1e6, 1)
        row["pj_per_flop_dynamic"] = round((watt - 240.0) * us * 1e-6 / flop * 1e12, 3)
    if nbytes:
        row["gb_s"] = round(nbytes / us / 1e3, 1)
    print(json.dumps(row), flush=True)


g = (40, 40, 16)
V = g[0] * g[1] * g[2]
x = torch.randn(V, 256, device="cuda").relu_()
wt = torch.randn(27, 256, 256, device="cuda") * 0.01
sc, sh = torch.ones(256, device="cuda"), torch.zeros(256, device="cuda")
wh, wl = ops.split_bf16(wt)
flop = 2.0 * V * 256 * 256 * 27 * 3
for nm, o in libs.items():
    o.lib.call("sgc_set_tuning", b"halo_stagger", 1)
    measure(nm + " 256->256 40x40x16", lambda o=o: o.conv3d_cl_bf16x3(x, wh, wl, g, 3, 1, False, sc, sh, None, True), flop=flop)
ops.lib.call("sgc_set_tuning", b"halo_stagger", 0)
measure("halo lockstep form", lambda: ops.conv3d_cl_bf16x3(x, wh, wl, g, 3, 1, False, sc, sh, None, True), flop=flop)
ops.lib.call("sgc_set_tuning", b"halo_stagger", 1)
# 1x1x1 layer on the tile kernel, a row GEMM, the layout transpose, a plain copy
w1 = torch.randn(1, 256, 256, device="cuda") * 0.05
w1h, w1l = ops.split_bf16(w1)
measure("igemm 1x1x1 256->256", lambda: ops.conv3d_cl_bf16x3(x, w1h, w1l, g, 1, 1, False, sc, sh, None, True), flop=2.0 * V * 256 * 256 * 3)
fm = torch.randn(40, 256, 64, 80, device="cuda")
measure("nchw_to_nhwc 40x256x64x80", lambda: ops.nchw_to_nhwc_crop(fm, 64, 80), nbytes=2.0 * fm.numel() * 4)
dst = torch.empty_like(fm)
measure("torch copy 210 MB", lambda: dst.copy_(fm), nbytes=2.0 * fm.numel() * 4)
