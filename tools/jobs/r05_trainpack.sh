#!/bin/bash
# round 5: training step -- weight planes repacked in one launch per step (functions.TrainWeightPlanes), the level's host syncs back to
# back behind the value projection: tests, then A/B of the pack form, then the kernel table
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_conv3d.py tests/test_gpu_modules.py -x -q -k "pack or train or grad or backward" 2>&1 | tail -3
for i in 1 2 3; do
echo "pack per use  " $(SGC_TRAIN_PACK_BATCH=0 python tools/train_step_bench.py --steps 40 2>/dev/null | tail -1)
echo "one batch/step" $(python tools/train_step_bench.py --steps 40 2>/dev/null | tail -1)
done
python tools/train_step_bench.py --steps 10 --profile 2>&1 | grep -v amdgpu.ids | tail -28
