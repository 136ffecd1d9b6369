"""The bench table of DESIGN.md section 5 from the committed lines: python tools/bench_table.py [round prefix, default r06]"""
import json, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pre = sys.argv[1] if len(sys.argv) > 1 else "r06"
rows = [("cfg2", "**cfg2 (headline)**"), ("cfg2_driver_cmd", "cfg2, the driver's command (`--steps 20 --warmup 5`)"),
        ("cfg2_direct", "cfg2 with the direct form only (`--winograd off`; same job)"), ("cfg2_nhwc", "cfg2, channels-last hand-over (`--input-layout nhwc`)"),
        ("cfg2_100v", "**cfg2 at the reference's 100 test views** (`cfg2_scannet_100v`)"), ("cfg3", "cfg3 (60 views, 48×48×16, ARKit head)"),
        ("cfg4", "cfg4 (50 views, 80×80×32, C = 128, 189 classes)"), ("cfg5", "cfg5 (100 views, 96×96×32, C = 128)"),
        ("cfg2_f32", "cfg2 strict fp32 (`--conv-mode f32`)"), ("cfg2_fp16_bf16maps", "cfg2 opt-in fp16 products + bf16 maps (N2; never the headline)")]
print("| line (`profiles/%s_bench_*.json`, one job, one box) | scenes/s | sustained | gather µs / frac | largest conv µs / frac of 2.5 PF (issued) | `path_roofline` (of the power-limited floor) |" % pre)
print("|---|---|---|---|---|---|")
for key, label in rows:
    d = json.loads(open(os.path.join(R, "profiles", f"{pre}_bench_{key}.json")).readline())
    rf, rm, pr = d["roofline"], d.get("roofline_mfma") or {}, d.get("path_roofline") or {}
    print(f"| {label} | {d['value']:.1f} | {(d.get('sustained') or {}).get('value', float('nan')):.1f} | {rf['avg_launch_us']:.1f} / {rf['frac']:.3f} | "
          f"{rm.get('avg_launch_us', float('nan')):.1f} / {rm.get('frac', float('nan')):.4f} | {pr.get('frac', float('nan')):.3f} ({pr.get('frac_of_power_limited_floor', float('nan')):.3f}) |")
    sc = d.get("self_check") or {}
    if key in ("cfg2", "cfg4", "cfg5"):
        print(f"<!-- {key}: winograd_vs_direct {sc.get('winograd_vs_direct_head_max_rel_diff')}, cpu_baseline {(d.get('cpu_baseline') or {}).get('value')}, strict {(d.get('strict_fp32') or {}).get('value')} -->", file=sys.stderr)
