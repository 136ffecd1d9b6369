#!/bin/bash
# round 5: `+ identity` / ReLU behind the neck's BatchNorms inside the normalisation passes (sgc_bn_rows_act_*): tests, then A/B
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py tests/test_gpu_conv3d.py -x -q -k "batch_norm or train or grad or backward or function or neck" 2>&1 | tail -3
for i in 1 2 3; do
echo "torch tail " $(SGC_BN_FUSE_TAIL=0 python tools/train_step_bench.py --steps 40 2>/dev/null | tail -1)
echo "fused tail " $(python tools/train_step_bench.py --steps 40 2>/dev/null | tail -1)
done
python tools/train_step_bench.py --steps 10 --profile 2>&1 | grep -v amdgpu.ids | tail -28 | head -4
