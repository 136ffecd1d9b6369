#!/bin/bash
# round 5: training step -- 16-byte loads in the pack kernel, weight gradient delivered in the parameter layout by one call
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_conv3d.py tests/test_gpu_modules.py -x -q -k "pack or train or grad or backward or function" 2>&1 | tail -3
python tools/pack_bench.py 2>&1 | grep -v amdgpu.ids
for i in 1 2 3; do
echo "step" $(python tools/train_step_bench.py --steps 40 2>/dev/null | tail -1)
done
python tools/train_step_bench.py --steps 10 --profile 2>&1 | grep -v amdgpu.ids | tail -28
