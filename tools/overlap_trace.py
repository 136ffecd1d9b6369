"""Two cfg2 scenes overlapping on two streams: first traced stage that differs from the serial run."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sgcdet_amd.scene import make_scene, workload
from sgcdet_amd import ext
import torch.nn.functional as F
w = workload("cfg2_scannet")
dev = torch.device("cuda", 0)
det = bench.build_path(w, dev)
det.use_graph = False
scenes = []
for s in range(3):
    feats, dpt, meta = make_scene(w["n_views"], w["embed_dims"], kind=w["kind"], seed=s, device=dev)
    scenes.append((feats, dpt, [meta]))
ops = ext.ops()
cur = []
def rec(name, t):
    cur.append((name, t.detach().clone()))
def wrap(name):
    fn = getattr(ops, name)
    def f(*a, **k):
        out = fn(*a, **k)
        if isinstance(out, dict):
            rec(name + ".totals", out["totals"].float())
            return out
        t = out[0] if isinstance(out, tuple) else out
        if torch.is_tensor(t):
            rec(name, t.float())
        return out
    setattr(ops, name, f)
for n in ("project_points", "compact_pairs", "nchw_to_nhwc_crop", "pairs_geometry_sample", "conv3d_cl_bf16x3", "depth_pairs",
          "pairs_deform_gather", "view_mean", "view_attend", "upsample2x_occ", "scatter_add_rows", "scatter_rows"):
    wrap(n)
if os.environ.get("RAWSHIFT"):
    _g = ops.pairs_deform_gather
    sh = int(os.environ["RAWSHIFT"])          # floats
    def g(value, dist, ref_cam, raw, *a, **k):
        buf = torch.empty(raw.numel() + sh, dtype=raw.dtype, device=raw.device)
        r2 = buf[sh:].view(raw.shape); r2.copy_(raw)
        return _g(value, dist, ref_cam, r2, *a, **k)
    ops.pairs_deform_gather = g
_lin = F.linear
def lin(x, w_, b=None):
    y = _lin(x, w_, b); rec("F.linear", y); return y
F.linear = lin
_topk = torch.topk
def topk(*a, **k):
    r = _topk(*a, **k); rec("topk.idx", r[1].float()); return r
torch.topk = topk
def run(i, stream):
    global cur
    cur = []
    feats, dpt, metas = scenes[i]
    with torch.no_grad(), torch.cuda.stream(stream):
        r = det.forward_features(feats, metas, dpt)
    return cur, r
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
base = []
for i in range(3):
    t, r = run(i, s0); torch.cuda.synchronize(); base.append(t)
for trial in range(4):
    got = []
    for i in range(3):
        got.append(run(i, (s0, s1)[i % 2]))
    torch.cuda.synchronize()
    for i in range(3):
        t = got[i][0]
        msg = "identical"
        for k, ((n0, a), (n1, b)) in enumerate(zip(base[i], t)):
            if n0 != n1 or a.shape != b.shape or not torch.equal(a, b):
                msg = f"first differing stage #{k} {n0}/{n1} {tuple(a.shape)} {tuple(b.shape)} maxdiff {float((a - b).abs().max()) if a.shape == b.shape else None}"
                if n0 == "pairs_deform_gather" and a.shape == b.shape:
                    d = a != b
                    rows = d.any(1).nonzero().view(-1); cols = d.any(0).nonzero().view(-1)
                    print(f"   wrong rows {rows.numel()} of {a.shape[0]} first {rows[:12].tolist()} last {rows[-4:].tolist()}; wrong cols {cols.numel()} first {cols[:4].tolist()} last {cols[-4:].tolist()}")
                    # per row: which 32-col head groups are wrong
                    hg = d.view(a.shape[0], 8, 32).any(2)
                    print("   wrong head-groups histogram", hg.sum(0).tolist(), " rows parity (even, odd)", int((rows % 2 == 0).sum()), int((rows % 2 == 1).sum()))
                    r0 = int(rows[0]); c0 = int(hg[r0].nonzero()[0]) * 32
                    chunk = b[r0, c0:c0 + 32]
                    print("   wrong chunk", chunk[:4].tolist(), " right", a[r0, c0:c0 + 4].tolist())
                    for jj in range(3):
                        for kk, (nn, tt) in enumerate(got[jj][0]):
                            if tt.numel() % 32 or tt.numel() < 32:
                                continue
                            m_ = (tt.reshape(-1, 32) == chunk).all(1)
                            if bool(m_.any()):
                                print(f"   chunk found in scene {jj} stage #{kk} {nn} {tuple(tt.shape)} at flat row {int(m_.nonzero()[0])}")
                break
        print(f"trial {trial} scene {i} (stream {i % 2}): {msg}", flush=True)
