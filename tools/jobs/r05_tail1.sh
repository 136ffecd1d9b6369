#!/bin/bash
# round 5, launch tail step 1: exp(scale(reg)) in the head convolution's epilogue, the zero row written by the value GEMM, occupancy
# scores written into their slice, contiguous ref_3d -- no torch kernel left inside a scene's launch sequence.  Tests + census + bench.
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_conv3d.py -x -q -k "head_activation" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "zero_row_behind or projected_query" 2>&1 | tail -2
timeout 2400 python -m pytest tests/test_gpu_modules.py -x -q 2>&1 | tail -3
timeout 300 python tools/graph_ops.py cfg2_scannet 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r05_graph_ops_cfg2_after.txt
timeout 600 python bench.py --no-cpu-baseline --no-strict-fp32 > gpurun_out/r05_tail1_cfg2.json 2>/dev/null; echo bench rc $?
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_tail1_cfg2.json").readline())
print("cfg2", d["value"], "sustained", d["sustained"]["value"], "gather", d["roofline"]["frac"], "self_check", d["self_check"]["mismatching"], "calls", d["config"]["library_calls_per_scene"])
PY
head -20 gpurun_out/r05_graph_ops_cfg2_after.txt
