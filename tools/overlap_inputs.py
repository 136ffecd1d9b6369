"""Overlapping cfg2 scenes: repeat the level-0 deformable gather with one input at a time moved to a fresh
address, to find which input the wrong launches read wrongly."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sgcdet_amd.scene import make_scene, workload
from sgcdet_amd import ext
w = workload("cfg2_scannet")
dev = torch.device("cuda", 0)
det = bench.build_path(w, dev)
det.use_graph = False
scenes = []
for s in range(3):
    feats, dpt, meta = make_scene(w["n_views"], w["embed_dims"], kind=w["kind"], seed=s, device=dev)
    scenes.append((feats, dpt, [meta]))
ops = ext.ops()
if os.environ.get("NONECK"):
    det._neck_head = lambda volume: ([volume], [volume], [volume])
if os.environ.get("NECKSUB"):
    kind = os.environ["NECKSUB"]
    def mk(grid, cin, cout, k):
        x = torch.randn(grid[0] * grid[1] * grid[2], cin, device=dev)
        wt = torch.randn(k ** 3 if k != 2 else 8, cout, cin, device=dev) * 0.02
        wh, wl = ops.split_bf16(wt)
        return x, wh, wl
    if kind == "halo":
        X = mk((40, 40, 16), 256, 256, 3); call = lambda: ops.conv3d_cl_bf16x3(X[0], X[1], X[2], (40, 40, 16), 3, 1, False)
    elif kind == "s2":       # stride-2 igemm 256 -> 512
        X = mk((40, 40, 16), 256, 512, 3); call = lambda: ops.conv3d_cl_bf16x3(X[0], X[1], X[2], (40, 40, 16), 3, 2, False)
    elif kind == "splitk":   # small grid, 27 taps -> split-K with atomics + memset + epilogue
        X = mk((10, 10, 4), 1024, 1024, 3); call = lambda: ops.conv3d_cl_bf16x3(X[0], X[1], X[2], (10, 10, 4), 3, 1, False)
    elif kind == "mid":      # 20x20x8 x 512
        X = mk((20, 20, 8), 512, 512, 3); call = lambda: ops.conv3d_cl_bf16x3(X[0], X[1], X[2], (20, 20, 8), 3, 1, False)
    elif kind == "tr":       # transposed 2x2x2
        X = mk((20, 20, 8), 512, 256, 2); call = lambda: ops.conv3d_cl_bf16x3(X[0], X[1], X[2], (20, 20, 8), 2, 2, True)
    elif kind == "k1":
        X = mk((25600, 1, 1), 256, 256, 1); call = lambda: ops.conv3d_cl_bf16x3(X[0], X[1], X[2], (25600, 1, 1), 1, 1, False)
    reps = int(os.environ.get("REPS", "12"))
    def sub(volume):
        for _ in range(reps):
            call()
        return ([volume], [volume], [volume])
    det._neck_head = sub
if os.environ.get("CONVHALO"):
    ops.lib.call("sgc_set_tuning", b"conv_halo", int(os.environ["CONVHALO"]))
if os.environ.get("CONVWAVES"):
    ops.lib.call("sgc_set_tuning", b"conv_waves", int(os.environ["CONVWAVES"]))
if os.environ.get("CONVMODE"):
    from sgcdet_amd.plugin.conv_plan import set_conv_mode
    set_conv_mode(os.environ["CONVMODE"])
if os.environ.get("FWDV"):
    ops.lib.call("sgc_set_tuning", b"fwd_variant", int(os.environ["FWDV"]))
cur = []
_pdg = ops.pairs_deform_gather
def pdg(value, dist, ref_cam, raw, pair_cam, pair_q, n_pairs, H, W, M, P, totals=None, dist_pairs=None, zero_row=False):
    kw = dict(totals=totals, dist_pairs=dist_pairs, zero_row=zero_row)
    mode = os.environ.get("MODE", "")
    if mode == "sync":
        torch.cuda.current_stream().synchronize()
    elif mode == "event":
        e = torch.cuda.Event(); e.record(); torch.cuda.current_stream().wait_event(e)
    elif mode == "tiny":
        torch.zeros(1, device=raw.device)
    elif mode == "freshraw":
        raw = raw + 0.0
    out = _pdg(value, dist, ref_cam, raw, pair_cam, pair_q, n_pairs, H, W, M, P, **kw)
    if len(cur) == 0:           # level 0 only
        again = _pdg(value, dist, ref_cam, raw, pair_cam, pair_q, n_pairs, H, W, M, P, **kw)
        raw2 = _pdg(value, dist, ref_cam, raw.clone(), pair_cam, pair_q, n_pairs, H, W, M, P, **kw)
        dp2 = _pdg(value, dist, ref_cam, raw, pair_cam, pair_q, n_pairs, H, W, M, P, totals=totals, dist_pairs=dist_pairs.clone(), zero_row=zero_row)
        val2 = _pdg(value.clone(), dist, ref_cam, raw, pair_cam, pair_q, n_pairs, H, W, M, P, totals=totals, dist_pairs=dist_pairs, zero_row=False)
        ref2 = _pdg(value, dist, ref_cam.clone(), raw, pair_cam.clone(), pair_q.clone(), n_pairs, H, W, M, P, **kw)
        cur.append(dict(out=out.clone(), again=again, raw2=raw2, dp2=dp2, val2=val2, ref2=ref2, raw=raw.clone()))
    else:
        cur.append(None)
    return out
ops.pairs_deform_gather = pdg
def run(i, stream):
    global cur
    cur = []
    feats, dpt, metas = scenes[i]
    with torch.no_grad(), torch.cuda.stream(stream):
        det.forward_features(feats, metas, dpt)
    return cur
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
base = []
for i in range(3):
    t = run(i, s0); torch.cuda.synchronize(); base.append(t[0]["out"])
    assert all(torch.equal(t[0][k], t[0]["out"]) for k in ("again", "raw2", "dp2", "val2", "ref2"))
tot = dict(out=0, again=0, raw2=0, dp2=0, val2=0, ref2=0)
for trial in range(10):
    got = [run(i, (s0, s1)[i % 2]) for i in range(3)]
    torch.cuda.synchronize()
    for i in range(3):
        for k in tot:
            tot[k] += int(not torch.equal(got[i][0][k], base[i]))
print("wrong level-0 gathers out of 30:", tot)
