"""The tile kernel's split policy on the layers config 2 runs on it (stride-2, transposed, 400-voxel, 1x1 stride-2): groups of whole
taps (`split_free` 0: 3 / 9 / 27 splits, the rounds 2 - 5 rule) against splits of whole K steps (`split_free` 1, round 6), in the
latency (`split_target` 512) and the throughput geometry (256).  Alternated rounds, HIP events, median; every variant's result is
compared with a float64 torch convolution:  python tools/split_free_ab.py [min_steps,max ...] [targets=512,256]"""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgcdet_amd import ext
ops = ext.ops()
LAYERS = [("256->512 s2 @40x40x16", 256, 512, (40, 40, 16), 3, 2, False), ("512->1024 s2 @20x20x8", 512, 1024, (20, 20, 8), 3, 2, False),
          ("1024->128 @10x10x4", 1024, 128, (10, 10, 4), 3, 1, False), ("1024->512 T @10x10x4", 1024, 512, (10, 10, 4), 2, 2, True),
          ("512->256 T @20x20x8", 512, 256, (20, 20, 8), 2, 2, True), ("1x1 s2 512->1024 @20x20x8", 512, 1024, (20, 20, 8), 1, 2, False),
          ("1x1 s2 256->512 @40x40x16", 256, 512, (40, 40, 16), 1, 2, False)]
args = [a for a in sys.argv[1:] if not a.startswith("targets=")]
targets = [int(v) for a in sys.argv[1:] if a.startswith("targets=") for v in a[8:].split(",")] or [512, 256]
variants = [(0, 8, 32)] + [(1,) + tuple(int(v) for v in a.split(",")) for a in (args or ["8,32"])]


def timed(fn, n=30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def reference(x, wt, g, k, s, transposed, sc, sh):
    X = x.double().view(1, *g, -1).permute(0, 4, 1, 2, 3)
    if transposed:      # wt [8 parities][Cout][Cin] -> ConvTranspose3d weight [Cin, Cout, 2, 2, 2]
        W = wt.double().view(2, 2, 2, wt.shape[1], wt.shape[2]).permute(4, 3, 0, 1, 2)
        Y = F.conv_transpose3d(X, W, stride=2)
    else:
        W = wt.double().view(k, k, k, wt.shape[1], wt.shape[2]).permute(3, 4, 0, 1, 2)
        Y = F.conv3d(X, W, stride=s, padding=k // 2)
    Y = Y.permute(0, 2, 3, 4, 1).reshape(-1, wt.shape[1])
    return torch.relu(Y * sc.double() + sh.double())


for target in targets:
    ops.lib.call("sgc_set_tuning", b"split_target", target)
    print(f"split_target {target}" + {512: " (latency geometry)", 256: " (throughput geometry)"}.get(target, ""), flush=True)
    for name, Cin, Cout, g, k, s, tr in LAYERS:
        V = g[0] * g[1] * g[2]
        x = torch.randn(V, Cin, device="cuda")
        wt = torch.randn(8 if tr else k ** 3, Cout, Cin, device="cuda") * (1.0 / ((8 if tr else k ** 3) * Cin) ** 0.5)
        sc = torch.rand(Cout, device="cuda") + 0.5; sh = torch.randn(Cout, device="cuda") * 0.1
        wh, wl = ops.split_bf16(wt)
        ref = reference(x, wt, g, k, s, tr, sc, sh)
        scale = float(ref.abs().max())
        f = lambda: ops.conv3d_cl_bf16x3(x, wh, wl, g, k, s, tr, sc, sh, None, True)
        ts = {v: [] for v in variants}; err = {}
        for rnd in range(5):
            for v in variants:
                for key, val in zip((b"split_free", b"split_min_steps", b"split_max"), v):
                    ops.lib.call("sgc_set_tuning", key, val)
                err[v] = float((f()[0].double() - ref).abs().max()) / scale
                t = timed(f)
                if rnd: ts[v].append(t)
        print(f"  {name:28s} " + " | ".join(f"{'taps' if v[0] == 0 else 'steps>=%d,<=%d' % v[1:]}: {sorted(ts[v])[len(ts[v]) // 2]:6.1f} us ({err[v]:.1e})"
                                          for v in variants), flush=True)
for key, val in ((b"split_free", 1), (b"split_min_steps", 8), (b"split_max", 32), (b"split_target", 512)):
    ops.lib.call("sgc_set_tuning", key, val)
