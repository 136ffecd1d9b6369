#!/bin/bash
# round 4, item 2: staggered schedule of the halo convolution -- bit-identity test, conv suite, alternated A/B against lockstep
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_conv3d.py -x -q 2>&1 | tail -4
timeout 600 python tools/halo_knob_ab.py halo_stagger 0,1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_halo_stagger_ab.log
