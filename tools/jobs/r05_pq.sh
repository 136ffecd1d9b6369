#!/bin/bash
# round 5: projected-query inter-view attention (sgc_view_attend_pq) -- parity tests, then A/B of the bench lines with the form
# forced off / on (auto = on from 24 views), per workload; plus the torch-op census of a scene and the fill / compute split of the
# tiled gather at config 2
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "projected_query" 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_gpu_modules.py -x -q -k "hot_path_against_oracle or full_size_config2 or full_view_count" 2>&1 | tail -3
for wl in cfg2_scannet cfg5_arkit_large cfg4_scannet200_large cfg3_arkit; do
n=${wl%%_*}
for pq in 0 1; do
SGC_PROJECTED_QUERY=$pq timeout 600 python bench.py --workload $wl --no-cpu-baseline --no-strict-fp32 --steps 40 --warmup 10 > gpurun_out/r05_pq${pq}_${n}.json 2>/dev/null; echo $wl pq$pq rc $?
done
done
timeout 300 python tools/graph_ops.py cfg2_scannet 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r05_graph_ops_cfg2.txt
SGC_DIAG=1 SGC_TILE_DIAG=1 SGC_TILE_CONFIGS="16,22,3,3,0,8,1,1,1,1;16,22,3,3,1,16,1,1,1,1" timeout 300 python tools/tile_bench.py cfg2 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r05_tile_diag_cfg2.txt
python - <<'PY'
import json
for n in ("cfg2", "cfg5", "cfg4", "cfg3"):
    for pq in (0, 1):
        try:
            d = json.loads(open(f"gpurun_out/r05_pq{pq}_{n}.json").readline())
            print(n, "pq", pq, d["value"], "sustained", (d.get("sustained") or {}).get("value"), "path", d["path_roofline"]["frac"], d["path_roofline"]["gemm_gflop_algorithmic"],
                  "self_check", d["self_check"]["mismatching"], "calls", d["config"]["library_calls_per_scene"])
        except Exception as e:
            print(n, pq, "failed", e)
PY
cat gpurun_out/r05_graph_ops_cfg2.txt gpurun_out/r05_tile_diag_cfg2.txt
