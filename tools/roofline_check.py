"""Every bench line committed under profiles/ must follow from its own fields: roofline.frac = achieved / peak with
achieved = algorithmic bytes / average launch time (HBM-bound kernel) or issued FLOPs / average launch time (MFMA kernel),
path_roofline.frac = floor / measured time per scene, value = scenes / second of the timed region.
Usage: python tools/roofline_check.py [files...]   (default: profiles/r04_bench_*.json, r05_bench_*.json and r06_bench_*.json); exits non-zero on a mismatch.
tests/test_host_logic.py runs it over the committed files."""
import glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def check_line(d, name="line"):
    """Returns the list of inconsistencies of one bench line (empty = consistent)."""
    bad = []
    def near(a, b, rel, what):
        if not (abs(a - b) <= rel * max(abs(a), abs(b), 1e-12)):
            bad.append(f"{name}: {what}: {a} vs {b}")
    spp = d.get("config", {}).get("scenes_per_step_per_gpu", 1)          # scenes in the batch one step processes
    near(d["value"], d["n_gpus"] * spp * 1e3 / d["ms_per_step"], 2e-3, "value vs n_gpus * scenes_per_step / ms_per_step")
    if "ms_per_scene" in d:                                              # round 5: the per-scene unit beside the per-batch one
        near(d["value"], 1e3 / d["ms_per_scene"], 2e-3, "value vs 1 / ms_per_scene")
    r = d.get("roofline")
    if r:
        near(r["frac"], r["achieved"] / r["peak"], 2e-3, "roofline.frac vs achieved / peak")
        if r["bound"] == "hbm":
            near(r["achieved"], r["algorithmic_bytes"] / (r["avg_launch_us"] * 1e-6) / 1e9, 5e-3, "roofline.achieved vs algorithmic_bytes / avg_launch_us")
    m = d.get("roofline_mfma")
    if m:
        near(m["frac"], m["achieved"] / m["peak"], 2e-3, "roofline_mfma.frac vs achieved / peak")
        nprod = 3 if m["peak"] > 1000 and "three bf16 products" in m.get("note", "") else 1
        near(m["achieved"], m["algorithmic_gflop"] * m.get("mac_frac_issued", 1.0) * nprod / (m["avg_launch_us"] * 1e-6) / 1e3, 1e-2,
             "roofline_mfma.achieved vs issued GFLOP / avg_launch_us")
    p = d.get("path_roofline")
    if p:
        near(p["frac"], p["floor_ms_per_scene"] / (d["ms_per_step"] / spp) * d["n_gpus"], 5e-3, "path_roofline.frac vs floor / ms per scene")
    s = d.get("sustained")
    if s:
        near(s["value"], d["n_gpus"] * 1e3 / s.get("ms_per_scene", s.get("ms_per_step")), 2e-3, "sustained.value vs 1 / ms per scene")
    return bad


if __name__ == "__main__":
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "profiles", "r04_bench_*.json")) + glob.glob(os.path.join(ROOT, "profiles", "r05_bench_*.json")) + glob.glob(os.path.join(ROOT, "profiles", "r06_bench_*.json")))
    bad = []
    for f in files:
        line = open(f).readline()
        if not line.strip().startswith("{"):
            continue
        bad += check_line(json.loads(line), os.path.basename(f))
    print("\n".join(bad) if bad else f"{len(files)} bench line(s) consistent")
    sys.exit(1 if bad else 0)
