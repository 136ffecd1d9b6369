#!/bin/bash
# round 5, review item 5: the occupancy-masked decoder tail (row N1) on a workload that can show a saving -- the seeded
# surface-clustered occupancy override -- with and without the mask on the same scenes, plus the parity tests that go with it
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_gpu_modules.py -x -q -k "masked_decoder_tail or fp16_mode_with_gradients or reduced_precision" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "bf16_storage" 2>&1 | tail -2
for wl in cfg2_scannet cfg5_arkit_large; do
n=${wl%%_*}
timeout 600 python bench.py --workload $wl --occupancy clustered --no-cpu-baseline --no-strict-fp32 > gpurun_out/r05_bench_${n}_clustered_dense.json 2>/dev/null; echo dense rc $?
timeout 600 python bench.py --workload $wl --occupancy clustered --masked-tail --no-cpu-baseline --no-strict-fp32 > gpurun_out/r05_bench_${n}_masked_clustered.json 2>/dev/null; echo masked rc $?
done
timeout 300 python tools/valid_stats.py cfg2_scannet clustered 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r05_valid_stats_clustered.txt
timeout 300 python tools/valid_stats.py cfg2_scannet 2>&1 | grep -v "amdgpu.ids" >> gpurun_out/r05_valid_stats_clustered.txt
python - <<'PY'
import json
for n in ("cfg2_clustered_dense", "cfg2_masked_clustered", "cfg5_clustered_dense", "cfg5_masked_clustered"):
    try:
        d = json.loads(open(f"gpurun_out/r05_bench_{n}.json").readline())
        print(n, d["value"], "sustained", (d.get("sustained") or {}).get("value"), "self_check", d["self_check"]["mismatching"], d["config"].get("masked_tail_stats"))
    except Exception as e:
        print(n, "failed", e)
PY
cat gpurun_out/r05_valid_stats_clustered.txt
