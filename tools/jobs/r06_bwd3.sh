#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "binned_backward or training_level or item_list" 2>&1 | tail -3
for spec in 0 "8,22,2,2" "8,16,2,2" "12,11,2,2" 0 "8,22,2,2"; do
  echo "SGC_TRAIN_BWD=$spec: $(SGC_TRAIN_BWD=$spec timeout 600 python tools/train_step_bench.py --steps 40 2>/dev/null | tail -1)"
done 2>&1 | tee gpurun_out/r06_train_bwd_ab.txt
timeout 600 python tools/train_step_bench.py --steps 10 --profile 2>&1 | grep -v "amdgpu.ids\|warn\|Warning" | cut -c1-200 | head -12
