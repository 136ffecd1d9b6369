"""Forward + backward of the path (voxel head -> neck -> head convolutions, a dummy quadratic loss on every head
tensor) at a BASELINE workload, training mode: ms per step and, under `rocprofv3 --kernel-trace --stats`, the kernel
breakdown (row a11b: what the DFA3D backward costs at scale next to the library convolutions)."""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sgcdet_amd.plugin  # noqa: F401
from sgcdet_amd.mmcv_lite import build_detector
from sgcdet_amd.scene import make_scene, model_config, workload

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="cfg2_scannet")
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--no-neck", action="store_true", help="stop at the volume (view transform only)")
ap.add_argument("--profile", action="store_true", help="print the 25 kernels with the most GPU time (torch.profiler)")
ap.add_argument("--input-layout", default="nchw", choices=["nchw", "nhwc"],
                help="memory layout of the feature / depth maps handed to the path (nhwc = what plugin/fpn.py produces)")
ap.add_argument("--wgrad-layers", action="store_true", help="list every weight-gradient call of one step: shapes, kernel form, time")
ap.add_argument("--cprofile", action="store_true", help="host side: the 45 python functions with the most own time over 5 steps (the step is launch-bound)")
ap.add_argument("--sites", action="store_true", help="torch ops of one step by the line of this package that issues them (a TorchDispatchMode: forward and backward)")
ap.add_argument("--glue", action="store_true", help="attribute the torch glue ops (copy / add / fill / sum / mul ...) to source lines of this package")
args = ap.parse_args()
w = workload(args.workload)
torch.manual_seed(0)
det = build_detector(model_config(w)).cuda().train()
feats, dpt, meta = make_scene(w["n_views"], w["embed_dims"], kind=w["kind"], seed=1, device="cuda", img_hw=(256, 320))
if args.input_layout == "nhwc":         # same logical [1, N, C, H, W] tensors, channels-last in memory
    cl = lambda t: t[0].contiguous(memory_format=torch.channels_last).unsqueeze(0)   # noqa: E731
    feats, dpt = [cl(f) for f in feats], cl(dpt)
feats = [f.requires_grad_(True) for f in feats]
dpt = dpt.requires_grad_(True)
params = [p for p in det.parameters() if p.requires_grad]

def step():
    for p in params:
        p.grad = None
    if args.no_neck:
        import torch.nn.functional as F
        dpts = [dpt, F.interpolate(dpt, scale_factor=(1, 0.5, 0.5), mode="nearest"), F.interpolate(dpt, scale_factor=(1, 0.25, 0.25), mode="nearest")]
        vol, valid, occ = det.voxel_head(feats, meta, dpts)
        loss = (vol ** 2).mean() + occ.mean()
    else:
        r = det.forward_features(feats, [meta], dpt)
        loss = sum((t ** 2).mean() for k in ("centerness", "bbox_pred", "cls_score") for t in r[k]) + r["occ"].mean()
    loss.backward()
    return loss.detach()          # no read-back per step: the host issues the next forward while the GPU finishes this backward

for _ in range(2):
    step()
# blocks of <= 10 steps, a synchronize between blocks; the reported figure is the MEDIAN block (the host of a shared box stalls for a
# millisecond now and then and the step is launch-bound in its forward half: a mean over 30 steps moved by +-1 ms between runs)
blocks = []
left = args.steps
while left > 0:
    n = min(10, left); left -= n
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        l = step()
    torch.cuda.synchronize()
    blocks.append((time.perf_counter() - t) / n * 1e3)
ms_step = round(sorted(blocks)[len(blocks) // 2], 2)
ms_min = round(min(blocks), 2)
l = float(l)
n_grads = sum(int(p.grad is not None) for p in params)
if args.profile:
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(2):
            step()
        torch.cuda.synchronize()
    allrows = prof.key_averages()
    print(f"device time of all kernels: {sum(e.device_time_total for e in allrows) / 2e3:.2f} ms/step over "
          f"{sum(e.count for e in allrows) // 2} launches (wall {ms_step} ms/step)", file=sys.stderr)
    rows = sorted(allrows, key=lambda e: -e.device_time_total)[:25]
    for e in rows:
        print(f"{e.device_time_total / 2e3:9.3f} ms/step  x{e.count // 2:<5d} {e.key[:110]}", file=sys.stderr)
    print("by launch count (the step is launch-bound on the host):", file=sys.stderr)
    for e in sorted(allrows, key=lambda e: -e.count)[:25]:
        print(f"    x{e.count // 2:<5d} {e.device_time_total / 2e3:9.3f} ms/step  {e.key[:110]}", file=sys.stderr)
if args.sites:
    import collections, traceback
    from torch.utils._python_dispatch import TorchDispatchMode
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    seen = collections.Counter()

    class Sites(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, a=(), kw=None):
            out = func(*a, **(kw or {}))
            name = func.__name__ if hasattr(func, "__name__") else str(func)
            if any(k in name for k in ("fill", "zero", "copy", "add", "cat", "sum", "mul", "div", "index", "clone", "contiguous", "to_copy")):
                fr = [f for f in traceback.extract_stack() if f.filename.startswith(root) and "/tools/" not in f.filename]
                where = f"{os.path.relpath(fr[-1].filename, root)}:{fr[-1].lineno}" if fr else "(no frame of this package)"
                shape = tuple(out.shape) if isinstance(out, torch.Tensor) else ()
                seen[(name, where, shape)] += 1
            return out

    with Sites():
        step()
    torch.cuda.synchronize()
    print("torch ops of one step by issuing line (count, op, line, result shape):", file=sys.stderr)
    for (name, where, shape), n in sorted(seen.items(), key=lambda kv: (-kv[1], kv[0][1])):
        print(f"  x{n:<4d} {name:28s} {where:46s} {shape}", file=sys.stderr)
if args.cprofile:
    import cProfile, pstats, io
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    pr.enable()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    pr.disable()
    for key in ("tottime", "cumulative"):
        buf = io.StringIO()
        pstats.Stats(pr, stream=buf).sort_stats(key).print_stats(45)
        print(buf.getvalue()[:9000], file=sys.stderr)
if args.wgrad_layers:
    from sgcdet_amd import ext
    ops = ext.ops()
    calls, orig = [], ops.conv3d_wgrad_bf16x3

    def spy(x, dy, grid, ksize, stride=1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = orig(x, dy, grid, ksize, stride)
        e1.record()
        calls.append((x.shape[1], dy.shape[1], tuple(grid), ksize, stride, e0, e1))
        return out
    ops.conv3d_wgrad_bf16x3 = spy
    step()
    torch.cuda.synchronize()
    ops.conv3d_wgrad_bf16x3 = orig
    tot = 0.0
    for cin, cout, grid, k, st, e0, e1 in calls:
        us = e0.elapsed_time(e1) * 1e3
        tot += us
        gf = 2.0 * grid[0] * grid[1] * grid[2] / st ** 3 * cin * cout * k ** 3 * (1 if k != 2 else 1) / 1e9
        print(f"{us:8.1f} us  Cin {cin:5d} Cout {cout:5d} grid {str(grid):16s} k {k} s {st}  {gf:6.2f} GF", file=sys.stderr)
    print(f"{len(calls)} weight-gradient calls, {tot / 1e3:.2f} ms", file=sys.stderr)
if args.glue:
    import collections
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        step()
        torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0.0, 0])
    for ev in prof.events():
        if not ev.name.startswith("aten::") or ev.device_time_total <= 0 or ev.cpu_children and any(c.name.startswith("aten::") and c.device_time_total > 0 for c in ev.cpu_children):
            continue                                    # leaf aten ops that own device time
        if any(k in ev.name for k in ("convolution", "mm", "native_batch_norm")):
            continue
        site = next((f for f in (ev.stack or []) if "/sgcdet_amd/" in f or "/tools/" in f), "(autograd engine / no python frame)")
        shp = str([list(x) for x in (ev.input_shapes or []) if x][:2])
        key = (ev.name, shp[:70] + " " + site.split("/repo/")[-1][:60])
        agg[key][0] += ev.device_time_total; agg[key][1] += 1
    tot = sum(v[0] for v in agg.values())
    print(f"torch glue ops with device time: {tot / 1e3:.2f} ms per step", file=sys.stderr)
    for (name, site), (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:40]:
        print(f"{t / 1e3:8.3f} ms  x{n:<4d} {name:28s} {site}", file=sys.stderr)
print(json.dumps(dict(workload=args.workload, ms_per_step=ms_step, ms_per_step_best_block=ms_min, loss=l, params_with_grad=f"{n_grads}/{len(params)}",
                      peak_mem_gb=round(torch.cuda.max_memory_allocated() / 2**30, 2), no_neck=args.no_neck,
                      input_layout=args.input_layout)))
