import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sgcdet_amd.plugin
from sgcdet_amd.mmcv_lite import build_detector
from sgcdet_amd.pipeline import ScenePipeline
from sgcdet_amd.scene import make_scene, model_config, workload
w = workload("cfg1_plumbing")
torch.manual_seed(11)
det = build_detector(model_config(w)).eval().cuda()
gen = torch.Generator().manual_seed(2)
with torch.no_grad():
    for _, p in det.voxel_head.named_parameters():
        p.add_(torch.randn(p.shape, generator=gen).to(p.device) * 0.05)
scenes = []
for s in range(5):
    feats, dpt, meta = make_scene(4, w["embed_dims"], kind=w["kind"], seed=40 + s, device="cuda")
    scenes.append((feats, [meta], dpt))
NS, NP = int(os.environ.get("NS", "3")), int(os.environ.get("NP", "40"))
det.use_graph = os.environ.get("GRAPH", "1") == "1"
if os.environ.get("CONVMODE"):
    from sgcdet_amd.plugin.conv_plan import set_conv_mode
    set_conv_mode(os.environ["CONVMODE"])
print("graph", det.use_graph, "convmode", os.environ.get("CONVMODE", "bf16x3"))
def diff(a, b):
    return int((a["valid"] != b["valid"]).sum()), float((a["occ"] - b["occ"]).abs().max()), float((a["volume"] - b["volume"]).abs().max())


if "FWDV" in os.environ:
    from sgcdet_amd import ext as _e
    _e.ops().lib.call("sgc_set_tuning", b"fwd_variant", int(os.environ["FWDV"]))
    print("fwd_variant", os.environ["FWDV"])
if os.environ.get("TRACE", "0") != "1":
    det.use_graph = os.environ.get("GRAPH", "1") == "1"
    p1, p2 = ScenePipeline(det, n_streams=1), ScenePipeline(det, n_streams=2)
    base = p1.run(scenes)
    torch.cuda.synchronize()
    bad = 0
    for i in range(NP):
        r = p2.run(scenes)
        torch.cuda.synchronize()
        for j in range(len(scenes)):
            if not (torch.equal(base[j]["occ"], r[j]["occ"]) and torch.equal(base[j]["volume"], r[j]["volume"])):
                bad += 1
    print("graph", det.use_graph, "scene-runs with wrong voxel features:", bad, "of", NP * len(scenes))
    sys.exit(0)
# capture intermediates: seed rows of every level + the stages inside the cross attention of the coarsest level
trace = []
from sgcdet_amd import ext as _ext
_ops = _ext.ops()
_orig = {}
def _wrap(name):
    fn = getattr(_ops, name)
    _orig[name] = fn
    def w(*a, **k):
        out = fn(*a, **k)
        t = out[0] if isinstance(out, tuple) else out
        if torch.is_tensor(t) and t.is_floating_point():
            trace[-1].append((name, t.detach().clone()))
        elif isinstance(out, dict):
            trace[-1].append((name + ".totals", out["totals"].clone().float()))
        return out
    setattr(_ops, name, w)
for n in ("project_points", "compact_pairs", "nchw_to_nhwc_crop", "pairs_geometry_sample", "conv3d_cl_bf16x3",
          "depth_pairs", "pairs_deform_gather", "view_mean", "view_attend", "upsample2x_occ"):
    _wrap(n)
_pdg = _ops.pairs_deform_gather
def pdg(value, dist, ref_cam, raw, pair_cam, pair_q, n_pairs, H, W, M, P, totals=None, dist_pairs=None, zero_row=False):
    for nm, t in (("in.value", value), ("in.dist", dist), ("in.ref_cam", ref_cam), ("in.raw", raw),
                  ("in.pair_cam", pair_cam[:n_pairs].float()), ("in.pair_q", pair_q[:n_pairs].float()), ("in.dp", dist_pairs)):
        trace[-1].append((nm, t.detach().clone()))
    if zero_row:
        zr = torch.as_strided(value, (value.shape[-1] * value.shape[-2],), (1,), value.storage_offset() + value.numel())
        trace[-1].append(("in.zero_row", zr.clone()))
    return _pdg(value, dist, ref_cam, raw, pair_cam, pair_q, n_pairs, H, W, M, P, totals=totals, dist_pairs=dist_pairs, zero_row=zero_row)
_ops.pairs_deform_gather = pdg
_ff = det.forward_features
def ff(*a, **k):
    trace.append([])
    return _ff(*a, **k)
det.forward_features = ff
det.use_graph = os.environ.get("GRAPH", "1") == "1"

p1, p2 = ScenePipeline(det, n_streams=1), ScenePipeline(det, n_streams=2)
base = p1.run(scenes)
torch.cuda.synchronize()
base_trace = [[(n, t.cpu()) for n, t in tr] for tr in trace]
trace.clear()
for i in range(NP):
    trace.clear()
    r = p2.run(scenes)
    torch.cuda.synchronize()
    for j in range(len(scenes)):
        for (n0, t0), (n1, t1) in zip(base_trace[j], trace[j]):
            t1 = t1.cpu()
            if t0.shape != t1.shape or float((t0 - t1).abs().max()) > 1e-6:
                print(f"run {i} scene {j}: first differing stage = {n1} shape {tuple(t1.shape)} vs {tuple(t0.shape)} maxdiff {float((t0 - t1).abs().max()) if t0.shape == t1.shape else -1:.3e}")
                if t0.shape == t1.shape and t1.dim() == 2:
                    d = (t0 != t1)
                    rows = d.any(1).nonzero().view(-1).tolist(); cols = d.any(0).nonzero().view(-1).tolist()
                    print("   rows", rows[:40], "n", len(rows), " cols", cols[:16], "...", cols[-4:], "n", len(cols))
                    bad = t1[d]; print("   wrong values: zeros", int((bad == 0).sum()), "of", bad.numel(), " nan", int(bad.isnan().sum()),
                                       " sample wrong", bad[:6].tolist(), " right", t0[d][:6].tolist())
                    pc_ = [t for n_, t in trace[j] if n_ == "in.pair_cam"][0].cpu().long(); pq_ = [t for n_, t in trace[j] if n_ == "in.pair_q"][0].cpu().long()
                    print("   pair_cam of wrong rows", pc_[rows[:40]].tolist(), " pair_q", pq_[rows[:40]].tolist())
                break
print("trace done")
sys.exit(0)
base = p1.run(scenes)
torch.cuda.synchronize()
base_cpu = [{k: (v.cpu() if torch.is_tensor(v) else v) for k, v in r.items() if k in ("valid", "occ", "volume")} for r in base]
bad_now, bad_later = 0, 0
kept = []
for i in range(NP):
    r = p2.run(scenes)
    torch.cuda.synchronize()
    now = [{k: r_[k].cpu() for k in ("valid", "occ", "volume")} for r_ in r]
    for j in range(len(scenes)):
        d = diff(base_cpu[j], now[j])
        if any(d):
            bad_now += 1
            print(f"immediate: run {i} scene {j}: {d}")
    kept.append((r, now))
torch.cuda.synchronize()
for i, (r, now) in enumerate(kept):      # re-read the GPU copies at the very end
    for j in range(len(scenes)):
        later = {k: r[j][k].cpu() for k in ("valid", "occ", "volume")}
        d = diff(now[j], later)
        if any(d):
            bad_later += 1
            print(f"changed after the fact: run {i} scene {j}: {d}")
print("bad immediately", bad_now, "changed later", bad_later)
sys.exit(0)
if os.environ.get("REUSE", "0") == "1":      # one pipeline object (fixed streams, 2 graphs) reused for every run
    p1, p2 = ScenePipeline(det, n_streams=1), ScenePipeline(det, n_streams=2)
    p1.host_sync = p2.host_sync = os.environ.get("HOSTSYNC", "0") == "1"
    runs = [p1.run(scenes) for _ in range(NS)] + [p2.run(scenes) for _ in range(NP)]
else:
    runs = [ScenePipeline(det, n_streams=1).run(scenes) for _ in range(NS)] + [ScenePipeline(det, n_streams=2).run(scenes) for _ in range(NP)]
torch.cuda.synchronize()
for i, r in enumerate(runs[1:], 1):
    for j, (a, b) in enumerate(zip(runs[0], r)):
        dv = int((a["valid"] != b["valid"]).sum()); do = float((a["occ"] - b["occ"]).abs().max()); dvol = float((a["volume"] - b["volume"]).abs().max())
        if dv or do or dvol:
            occ2 = a["occ"][0, :3200]; k = w["topk_list"][1]; srt = occ2.sort(descending=True).values
            print(f"run {i} ({'serial' if i < NS else 'piped'}) scene {j}: valid diff {dv} occ diff {do:.2e} vol diff {dvol:.2e}  cut gap {float(srt[k-1]-srt[k]):.2e} ties at cut {int((occ2 == srt[k-1]).sum())}")
print("done")
