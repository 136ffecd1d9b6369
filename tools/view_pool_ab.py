"""view_mean / view_attend: group kernels (one slot load per lane, visible cameras only, next row prefetched) vs the per-camera
loops, on the pair lists of a config-2 scene: bit equality and interleaved timing."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgcdet_amd import ext
ops = ext.ops()
def timed(fn, n=30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
torch.manual_seed(0)
for N, Nq, C, vis in [(40, 6400, 256, 0.3), (40, 800, 256, 0.3), (40, 400, 256, 0.3), (100, 73728, 128, 0.3)]:
    mask = (torch.rand(N, Nq, device="cuda") < vis).to(torch.uint8)
    pc = ops.compact_pairs(mask)
    n_pairs, n_valid = int(pc["totals"][0]), int(pc["totals"][1])
    feat = torch.randn(n_pairs, C, device="cuda"); q = torch.randn(n_valid, C, device="cuda"); kv = torch.randn(n_pairs, 2 * C, device="cuda")
    res = {}
    for g in (0, 1):
        ops.lib.call("sgc_set_tuning", b"view_group", g)
        res[g] = (ops.view_mean(feat, pc["slot"], pc["valid_index"], n_valid), ops.view_attend(q, kv, pc["slot"], pc["valid_index"], 8))
    same = torch.equal(res[0][0], res[1][0]), torch.equal(res[0][1], res[1][1])
    line = []
    for rnd in range(3):
        for g in (0, 1):
            ops.lib.call("sgc_set_tuning", b"view_group", g)
            line.append(f"{'group' if g else 'loop '} mean {timed(lambda: ops.view_mean(feat, pc['slot'], pc['valid_index'], n_valid)):5.1f} attend {timed(lambda: ops.view_attend(q, kv, pc['slot'], pc['valid_index'], 8)):5.1f}")
    print(f"N={N} Nq={Nq} C={C} pairs={n_pairs}: identical {same} | " + " | ".join(line), flush=True)
ops.lib.call("sgc_set_tuning", b"view_group", 1)
