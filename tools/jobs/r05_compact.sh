#!/bin/bash
# round 5: two-launch segment form of sgc_compact_pairs -- parity, then A/B of the bench lines (SGC_TUNE=compact2=0|1)
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "compaction" 2>&1 | tail -2
timeout 1500 python -m pytest tests/test_gpu_modules.py -x -q -k "hot_path_against_oracle or scene_graph or golden" 2>&1 | tail -2
for wl in cfg2_scannet cfg5_arkit_large; do
n=${wl%%_*}
for c in 0 1; do
SGC_TUNE="compact2=$c" timeout 600 python bench.py --workload $wl --no-cpu-baseline --no-strict-fp32 --steps 40 --warmup 10 > gpurun_out/r05_compact${c}_${n}.json 2>/dev/null; echo $wl compact2=$c rc $?
done
done
python - <<'PY'
import json
for n in ("cfg2", "cfg5"):
    for c in (0, 1):
        d = json.loads(open(f"gpurun_out/r05_compact{c}_{n}.json").readline())
        print(n, "compact2", c, d["value"], "sustained", d["sustained"]["value"], "self_check", d["self_check"]["mismatching"])
PY
