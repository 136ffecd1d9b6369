"""Diagnostic build (-DSGC_DIAG_DESC): dump the descriptors phase 1 produced and phase 2 consumed for the
level-0 gather while another scene's neck runs on the other stream; say where the wrong rows go wrong."""
import os, sys, ctypes, torch
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
dll = ops.lib._dll
dll.sgc_debug_buffer.argtypes = [ctypes.c_void_p]
cur = []
_pdg = ops.pairs_deform_gather
def pdg(value, dist, ref_cam, raw, pair_cam, pair_q, n_pairs, H, W, M, P, totals=None, dist_pairs=None, zero_row=False):
    if len(cur) == 0:
        dbg = torch.full((2, n_pairs, M * P, 12), float("nan"), device=raw.device)
        dll.sgc_debug_buffer(ctypes.c_void_p(dbg.data_ptr()))
        out = _pdg(value, dist, ref_cam, raw, pair_cam, pair_q, n_pairs, H, W, M, P, totals=totals, dist_pairs=dist_pairs, zero_row=zero_row)
        dll.sgc_debug_buffer(None)
        cur.append((out.clone(), dbg, raw.clone()))
        return out
    cur.append(None)
    return _pdg(value, dist, ref_cam, raw, pair_cam, pair_q, n_pairs, H, W, M, P, totals=totals, dist_pairs=dist_pairs, zero_row=zero_row)
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
    t = run(i, s0); torch.cuda.synchronize(); base.append(t[0])
    prod, cons = t[0][1][0], t[0][1][1]
    assert torch.equal(prod[..., :8].view(torch.int32), cons[..., :8].view(torch.int32)), "serial: consumed != produced"
names = ["w0", "w1", "w2", "w3", "o0", "o1", "o2", "o3", "x", "y", "z", "aw"]
shown = 0
for trial in range(10):
    got = [run(i, (s0, s1)[i % 2]) for i in range(3)]
    torch.cuda.synchronize()
    for i in range(3):
        out, dbg, raw = got[i][0]
        bout, bdbg, braw = base[i]
        if torch.equal(out, bout):
            continue
        rows = (out != bout).any(1).nonzero().view(-1)
        prod, cons = dbg[0].view(torch.int32), dbg[1].view(torch.int32)
        bprod = bdbg[0].view(torch.int32)
        prod_bad = (prod != bprod).any(-1)            # [pairs, 32] samples whose PRODUCED descriptor differs from the serial run
        cons_bad = (cons[..., :8] != prod[..., :8]).any(-1)   # consumed differs from what this launch produced
        print(f"trial {trial} scene {i}: wrong rows {rows.numel()}; rows with wrongly PRODUCED descriptors {int(prod_bad.any(1).sum())}, "
              f"rows where CONSUMED != PRODUCED {int(cons_bad.any(1).sum())}, raw identical {torch.equal(raw, braw)}")
        if shown < 3 and prod_bad.any():
            shown += 1
            r = int(prod_bad.any(1).nonzero()[0])
            smp = prod_bad[r].nonzero().view(-1).tolist()
            print("   row", r, "bad samples", smp)
            for sidx in smp[:3]:
                a = dbg[0][r, sidx].tolist(); b = bdbg[0][r, sidx].tolist()
                fields = [n for n, u, v in zip(names, prod[r, sidx].tolist(), bprod[r, sidx].tolist()) if u != v]
                print("   sample", sidx, "differing fields", fields, " got x,y,z,aw", [round(v, 5) for v in a[8:]], " expected", [round(v, 5) for v in b[8:]])
                print("      o got", prod[r, sidx, 4:8].tolist(), "expected", bprod[r, sidx, 4:8].tolist(), " w got", [round(v, 5) for v in a[:4]], "expected", [round(v, 5) for v in b[:4]])
                print("      raw uv/dz/lg got", raw[r, sidx * 2:sidx * 2 + 2].tolist(), float(raw[r, 64 + sidx]), float(raw[r, 96 + sidx]))
