#!/bin/bash
python -m pytest tests/test_gpu_conv3d.py -x -q -k "wgrad or k2_s2 or channels_last" 2>&1 | tail -15
python -m pytest tests/test_gpu_modules.py -x -q -k "autograd_on_hip or training" 2>&1 | tail -15
for m in hip library; do
  SGC_TRAIN_CONV=$m python tools/train_step_bench.py --steps 5 2>&1 | grep -v amdgpu.ids | tail -1
done
